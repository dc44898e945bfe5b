"""CPU-side checks of the boundary: the library builds for gfx950, loads, and
exports every symbol `include/dns_amd.h` declares; host-only argument checks."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def capi():
    from dolfin_navier_scipy_amd.build import build_library
    build_library(verbose=False)
    from dolfin_navier_scipy_amd import _capi
    _capi.load_library()
    return _capi


def test_header_symbols_all_exported(capi):
    hdr = open(os.path.join(ROOT, 'include', 'dns_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(dns_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(capi.SIGNATURES.keys())
    lib = capi.load_library()
    for name in declared:
        assert hasattr(lib, name), name


def test_status_strings_and_defaults(capi):
    lib = capi.load_library()
    assert lib.dns_version() >= 100
    assert lib.dns_status_string(0) == b'ok'
    assert lib.dns_status_string(1) == b'not converged'
    import ctypes as ct
    o = capi.dns_solve_opts()
    lib.dns_default_solve_opts(ct.byref(o))
    assert o.restart <= 64 and o.rtol > 0 and o.maxiter > 0
    p = capi.dns_precond_opts()
    lib.dns_default_precond_opts(ct.byref(p))
    assert p.cheb_degree >= 1


def test_struct_layout_matches_header(capi):
    import ctypes as ct
    assert ct.sizeof(capi.dns_csr) == 40
    assert ct.sizeof(capi.dns_precond_opts) == 64
    assert ct.sizeof(capi.dns_solve_opts) == 40
    assert ct.sizeof(capi.dns_solve_stats) == 48
    assert ct.sizeof(capi.dns_imex_coeffs) == 48


def test_null_and_bad_arguments_fail_cleanly(capi):
    import ctypes as ct
    lib = capi.load_library()
    assert lib.dns_saddle_solve(None, None, None, None, None, None, None) \
        == capi.DNS_ERR_BAD_ARGUMENT
    assert b'null' in lib.dns_last_error()
    assert lib.dns_saddle_setup_precond(None, None) == capi.DNS_ERR_BAD_ARGUMENT
    out = ct.c_void_p()
    assert lib.dns_saddle_create(0, None, None, None, ct.byref(out)) \
        == capi.DNS_ERR_BAD_ARGUMENT
    # malformed CSR is rejected on the host before any device work
    A = sps.identity(4, format='csr')
    view = capi.CsrView(A)
    view.indices[2] = 9
    x = np.ones(4)
    y = np.zeros(4)
    st = lib.dns_spmv(0, view.byref(), capi.dptr(x), capi.dptr(y), 1.0, 0.0, 0)
    assert st == capi.DNS_ERR_BAD_ARGUMENT


def test_no_device_is_loud(capi):
    """without a GPU the product path must raise, never fall back"""
    if capi.device_count() > 0:
        pytest.skip('a GPU is present')
    from dolfin_navier_scipy_amd import saddle
    A = sps.identity(4, format='csr')
    J = sps.csr_matrix(np.ones((1, 4)))
    with pytest.raises(capi.DnsError):
        saddle.SaddleSystem(A, J)
    from dolfin_navier_scipy_amd import lin_alg_utils as lau
    with pytest.raises(capi.DnsError):
        lau.solve_sadpnt_smw(amat=A, jmat=J, rhsv=np.ones((4, 1)))


def test_install_as_lau():
    import sys
    import dolfin_navier_scipy_amd as pkg
    mod = pkg.install_as_lau()
    import sadptprj_riclyap_adi.lin_alg_utils as lau
    assert lau is mod
    for name in ('solve_sadpnt_smw', 'app_prj_via_sadpnt',
                 'SpslaKrylovCounter', 'apply_massinv'):
        assert hasattr(lau, name)
    sys.modules.pop('sadptprj_riclyap_adi.lin_alg_utils')
    sys.modules.pop('sadptprj_riclyap_adi')


def test_product_does_not_import_oracle():
    pkgdir = os.path.join(ROOT, 'dolfin_navier_scipy_amd')
    for dirpath, _, files in os.walk(pkgdir):
        for fn in files:
            if fn.endswith('.py'):
                src = open(os.path.join(dirpath, fn)).read()
                assert 'import oracle' not in src and 'from oracle' not in src
                assert 'krylov_model' not in src


def test_pattern_helpers_of_the_newton_picard_module():
    """host logic of `newton_picard`: union pattern with explicit zeros kept,
    values laid out in a super-pattern"""
    import scipy.sparse as sps
    from dolfin_navier_scipy_amd import newton_picard as dnp
    rng = np.random.default_rng(0)
    A = sps.random(40, 40, density=0.1, format='csr', random_state=rng)
    B = sps.random(40, 40, density=0.1, format='csr', random_state=rng)
    B.data[::3] = 0.                      # explicit zeros must stay
    P = dnp.union_pattern(A, B)
    assert P.has_canonical_format and np.all(P.data == 1.)
    want = set(zip(*A.nonzero())) | set(
        zip(np.repeat(np.arange(40), np.diff(B.indptr)), B.indices))
    got = set(zip(np.repeat(np.arange(40), np.diff(P.indptr)), P.indices))
    assert got == want
    va, vb = dnp.values_in_pattern(A, P), dnp.values_in_pattern(B, P)
    rebuilt = sps.csr_matrix((va + 2*vb, P.indices, P.indptr), shape=P.shape)
    assert abs(rebuilt - (A + 2*B)).max() <= 1e-15
    with pytest.raises(ValueError):
        dnp.values_in_pattern(sps.identity(40, format='csr') + A,
                              dnp.union_pattern(A))


def test_pressure_prolongations_of_a_refined_mesh():
    """input of the multigrid Schur block: P1 prolongations between nested
    pressure spaces in the spaces' own (RCM) dof numbering"""
    from dolfin_navier_scipy_amd.fem import (
        channel_cylinder_mesh, refine_uniform, TaylorHood,
        pressure_prolongations)
    coarse = channel_cylinder_mesh()
    mid, par1 = refine_uniform(coarse)
    fine, par2 = refine_uniform(mid)
    spaces = [TaylorHood(fine), TaylorHood(mid), TaylorHood(coarse)]
    prols = pressure_prolongations(spaces, [par2, par1, None])
    assert [p.shape for p in prols] == [(fine.nverts, mid.nverts),
                                        (mid.nverts, coarse.nverts)]
    for l, P in enumerate(prols):
        assert np.abs(np.asarray(P.sum(axis=1)).ravel() - 1.).max() < 1e-15
        assert P.data.min() >= 0.5 - 1e-15 and P.data.max() <= 1. + 1e-15
        # affine functions are reproduced (straight edges: exactly)
        fs, cs = spaces[l], spaces[l + 1]
        xc = np.empty(cs.pdim)
        xc[cs.vert_pdof] = 2*cs.mesh.verts[:, 0] - cs.mesh.verts[:, 1] + 3
        xf = np.empty(fs.pdim)
        xf[fs.vert_pdof] = 2*fs.mesh.verts[:, 0] - fs.mesh.verts[:, 1] + 3
        assert np.abs(P @ xc - xf).max() < 1e-12


def test_build_dependencies_cover_every_include():
    """`needs_build()` must see every file the translation unit includes
    (a hand-kept list once missed trap.hpp / trap_capi.inc)"""
    from dolfin_navier_scipy_amd import build
    deps = set(os.path.realpath(p) for p in build.dependencies())
    seen, todo = set(), [os.path.join(build.CSRC, s) for s in build.SOURCES]
    while todo:
        path = os.path.realpath(todo.pop())
        if path in seen:
            continue
        seen.add(path)
        for inc in re.findall(r'#include\s+"([^"]+)"', open(path).read()):
            todo.append(os.path.join(os.path.dirname(path), inc))
    assert len(seen) > 10
    missing = seen - deps
    assert not missing, missing


def test_spawn_retry_only_for_rendezvous_errors():
    import spawn_util
    assert spawn_util.is_rendezvous_error(
        RuntimeError('The server socket has failed to bind to [::]:29500 '
                     '(errno: 98 - Address already in use)'))
    assert not spawn_util.is_rendezvous_error(
        RuntimeError('process 1 terminated with signal SIGABRT'))
    assert not spawn_util.is_rendezvous_error(AssertionError('1e-3 > 1e-9'))
