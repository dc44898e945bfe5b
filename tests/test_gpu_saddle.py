"""Parity of the GPU saddle-point solve with the CPU oracle (direct solve) and
with the NumPy model of the device algorithm (iteration counts)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sps

import krylov_model as km
from oracle import saddle_oracle

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope='module')
def sad():
    from dolfin_navier_scipy_amd import saddle, _capi
    assert _capi.device_count() > 0, 'HIP device required for -m gpu tests'
    return saddle


def _capi_mod():
    from dolfin_navier_scipy_amd import _capi
    return _capi


@pytest.fixture(scope='module')
def small(toy_prob):
    M, A, J = (toy_prob['smc'][k] for k in 'MAJ')
    dt = 5e-3
    F = (M + .5*dt*A).tocsr()
    rng = np.random.default_rng(11)
    NP, NV = J.shape
    rhsv = M @ rng.standard_normal(NV)
    rhsp = 1e-3*rng.standard_normal(NP)
    ref = saddle_oracle.solve_sadpnt_smw(amat=F, jmat=J, rhsv=rhsv,
                                         rhsp=rhsp).reshape(-1)
    return dict(F=F, J=J, M=M, A=A, rhsv=rhsv, rhsp=rhsp, ref=ref, dt=dt)


def test_apply_K_and_bounds(sad, small):
    F, J = small['F'], small['J']
    system = sad.SaddleSystem(F, J)
    K = km.saddle(F, J)
    x = np.random.default_rng(0).standard_normal(K.shape[0])
    y = system.apply(x)
    assert np.abs(y - K @ x).max() <= 1e-13*np.abs(K @ x).max()
    system.setup_precond(cheb_degree=3, schur='dense')
    lo, hi = system.cheb_bounds()
    mlo, mhi = km.power_bounds(F, 1/F.diagonal())
    assert abs(hi - 1.05*mhi) <= 1e-6*mhi
    assert abs(lo - 0.9*mlo) <= 1e-3*mlo
    system.close()


@pytest.mark.parametrize('fhat', ['cheb', 'explicit'])
@pytest.mark.parametrize('degree', [1, 2, 4])
def test_precond_matches_model(sad, small, degree, fhat):
    F, J = small['F'], small['J']
    system = sad.SaddleSystem(F, J)
    system.setup_precond(cheb_degree=degree, schur='dense', fhat=fhat,
                         fp32_store=False, drop_tol=0.)
    lo, hi = system.cheb_bounds()
    cheb = km.ChebJacobi(F, degree=degree, lmin=lo, lmax=hi)
    P = km.BlockTriPrecond(F, J, cheb=cheb)
    r = np.random.default_rng(3).standard_normal(system.n)
    z = system.apply_precond(r)
    zm = P.apply(r)
    assert np.linalg.norm(z - zm) <= 1e-8*np.linalg.norm(zm)
    system.close()


@pytest.mark.parametrize('method,graph', [('gmres', False), ('gmres', True),
                                          ('bicgstab', False)])
@pytest.mark.parametrize('schur', ['dense', 'jacobi'])
def test_solve_matches_oracle(sad, small, method, graph, schur):
    system = sad.SaddleSystem(small['F'], small['J'])
    system.setup_precond(cheb_degree=4, schur=schur)
    x = system.solve(small['rhsv'], small['rhsp'], method=method, rtol=1e-12,
                     maxiter=3000, use_graph=graph)
    if graph:   # replay of the cached graphs must give the same answer
        x2 = system.solve(small['rhsv'], small['rhsp'], method=method,
                          rtol=1e-12, maxiter=3000, use_graph=True)
        assert np.linalg.norm(x2 - x) <= 1e-10*np.linalg.norm(x)
    st = system.last_stats
    assert st['status'] == 0
    assert st['true_relres'] <= 5e-12
    NV = small['F'].shape[0]
    ref = small['ref']
    assert np.linalg.norm(x[:NV] - ref[:NV]) <= 1e-9*np.linalg.norm(ref[:NV])
    assert np.linalg.norm(x[NV:] - ref[NV:]) <= 1e-7*np.linalg.norm(ref[NV:])
    hist = system.residual_history()
    assert hist.size == st['iters'] + 1
    assert hist[-1] <= 1e-12*st['bnorm']*1.0001
    system.close()


def test_gmres_iteration_count_matches_model(sad, small):
    F, J = small['F'], small['J']
    system = sad.SaddleSystem(F, J)
    system.setup_precond(cheb_degree=3, schur='dense', fp32_store=False,
                         drop_tol=0.)
    lo, hi = system.cheb_bounds()
    b = np.concatenate([small['rhsv'], small['rhsp']])
    x = system.solve(small['rhsv'], small['rhsp'], rtol=1e-10)
    hist = system.residual_history()
    P = km.BlockTriPrecond(F, J, cheb=km.ChebJacobi(F, degree=3, lmin=lo,
                                                    lmax=hi))
    xm, hm, its = km.gmres(km.saddle(F, J), b, P, rtol=1e-10)
    assert abs(system.last_stats['iters'] - its) <= 1
    k = min(hist.size, hm.size) - 2
    assert np.allclose(hist[:k], hm[:k], rtol=1e-3)
    assert np.linalg.norm(x - xm) <= 1e-8*np.linalg.norm(xm)
    system.close()


@pytest.mark.parametrize('schur', ['dense', 'jacobi'])
def test_full_block_factorisation(sad, small, schur):
    """`factorization='full'` (block LDU): the apply matches the NumPy model,
    every Krylov variant converges to the oracle's answer, and with the dense
    Schur block it needs far fewer steps than the triangular form"""
    F, J = small['F'], small['J']
    NV = F.shape[0]
    ref = small['ref']
    system = sad.SaddleSystem(F, J)
    system.setup_precond(cheb_degree=4, schur=schur, fhat='explicit',
                         fp32_store=False, drop_tol=0.,
                         factorization='full')
    lo, hi = system.cheb_bounds()
    cheb = km.ChebJacobi(F, degree=4, lmin=lo, lmax=hi)
    if schur == 'dense':
        P = km.BlockFullPrecond(F, J, cheb=cheb)
    else:
        sd = 1.0/np.asarray((J.multiply(J) @ (1.0/F.diagonal()))).reshape(-1)
        P = km.BlockFullPrecond(F, J, cheb=cheb, schur_inv=np.diag(sd))
    r = np.random.default_rng(3).standard_normal(system.n)
    z, zm = system.apply_precond(r), P.apply(r)
    assert np.linalg.norm(z - zm) <= 1e-8*np.linalg.norm(zm)
    its = {}
    for method, reorth, graph in (('gmres', 1, True), ('gmres', 0, True),
                                  ('gmres', 2, True), ('gmres', 1, False),
                                  ('bicgstab', 1, False)):
        x = system.solve(small['rhsv'], small['rhsp'], method=method,
                         rtol=1e-12, maxiter=3000, reorth=reorth,
                         use_graph=graph)
        st = system.last_stats
        assert st['status'] == 0 and st['true_relres'] <= 5e-12, (method, st)
        assert np.linalg.norm(x[:NV] - ref[:NV]) <= \
            1e-9*np.linalg.norm(ref[:NV])
        assert np.linalg.norm(x[NV:] - ref[NV:]) <= \
            1e-7*np.linalg.norm(ref[NV:])
        its[(method, reorth, graph)] = st['iters']
    xm, hm, im = km.gmres(km.saddle(F, J),
                          np.concatenate([small['rhsv'], small['rhsp']]), P,
                          rtol=1e-12)
    if im < 100:        # (long restarted runs drift apart)
        assert abs(its[('gmres', 1, True)] - im) <= 1
    if schur == 'dense':
        system.setup_precond(cheb_degree=4, schur=schur, fhat='explicit',
                             fp32_store=False, drop_tol=0.)
        system.solve(small['rhsv'], small['rhsp'], rtol=1e-12, reorth=1)
        assert its[('gmres', 1, True)] < system.last_stats['iters']
    # the recurrence form of Fh^-1 cannot carry the explicit J Fh^-1
    from dolfin_navier_scipy_amd import _capi
    with pytest.raises(_capi.DnsError):
        system.setup_precond(cheb_degree=4, schur=schur, fhat='cheb',
                             factorization='full')
    system.close()


@pytest.mark.parametrize('schur', ['dense', 'jacobi'])
@pytest.mark.parametrize('fp32', [False, True])
def test_gram_schmidt_modes_agree(sad, small, schur, fp32):
    """reorth 0 (CGS), 1 (CGS2), 2 (CGS folded into the next head kernel, norm
    by Pythagoras): same residual history while the basis is well conditioned,
    same answer; restart 7 exercises the fused tail and the restarts"""
    system = sad.SaddleSystem(small['F'], small['J'])
    system.setup_precond(cheb_degree=4, schur=schur, fp32_store=fp32)
    NV = small['F'].shape[0]
    ref = small['ref']
    hists = {}
    # (GMRES(7) with the weak Jacobi Schur block needs thousands of steps)
    restarts = (60, 7) if schur == 'dense' else (60,)
    for mode in (1, 0, 2):
        for restart in restarts:
            x = system.solve(small['rhsv'], small['rhsp'], rtol=1e-12,
                             maxiter=3000, restart=restart, reorth=mode,
                             use_graph=True)
            st = system.last_stats
            assert st['status'] == 0 and st['true_relres'] <= 5e-12, (mode, st)
            assert np.linalg.norm(x[:NV] - ref[:NV]) <= \
                1e-9*np.linalg.norm(ref[:NV])
            assert np.linalg.norm(x[NV:] - ref[NV:]) <= \
                1e-7*np.linalg.norm(ref[NV:])
            hists[(mode, restart)] = (system.residual_history(), st['iters'])
    for restart in restarts:
        h1, i1 = hists[(1, restart)]
        for mode in (0, 2):
            hm, im = hists[(mode, restart)]
            # (long restarted runs with the weak Jacobi Schur block drift apart:
            # compare counts only where the solve is short)
            if i1 < 300:
                assert abs(im - i1) <= max(2, 0.1*i1), (mode, restart, im, i1)
            k = min(h1.size, hm.size, 8)
            assert np.allclose(hm[:k], h1[:k], rtol=1e-6), (mode, restart)
    system.close()


@pytest.mark.parametrize('fp32,drop', [(True, 3e-3), (True, 0.), (False, 1e-2)])
def test_inexact_preconditioner_same_answer(sad, small, fp32, drop):
    """fp32 storage / dropped entries only change the PRECONDITIONER: the
    fp64 Krylov iteration still converges to the oracle's solution"""
    system = sad.SaddleSystem(small['F'], small['J'])
    system.setup_precond(cheb_degree=4, schur='dense', fhat='explicit',
                         fp32_store=fp32, drop_tol=drop)
    x = system.solve(small['rhsv'], small['rhsp'], rtol=1e-12, use_graph=True)
    st = system.last_stats
    assert st['status'] == 0 and st['true_relres'] <= 5e-12
    assert st['iters'] <= 40
    NV = small['F'].shape[0]
    ref = small['ref']
    assert np.linalg.norm(x[:NV] - ref[:NV]) <= 1e-9*np.linalg.norm(ref[:NV])
    assert np.linalg.norm(x[NV:] - ref[NV:]) <= 1e-7*np.linalg.norm(ref[NV:])
    system.close()


def test_restart_and_x0(sad, small):
    system = sad.SaddleSystem(small['F'], small['J'])
    system.setup_precond(cheb_degree=2, schur='jacobi')
    x = system.solve(small['rhsv'], small['rhsp'], rtol=1e-11, restart=20,
                     maxiter=4000)
    st = system.last_stats
    assert st['restarts'] >= 1 and st['true_relres'] <= 5e-11
    ref = small['ref']
    assert np.linalg.norm(x - ref) <= 1e-6*np.linalg.norm(ref)
    # exact initial guess: zero iterations
    x2 = system.solve(small['rhsv'], small['rhsp'], x0=x, rtol=1e-10)
    assert system.last_stats['iters'] == 0
    assert np.array_equal(x2, x)
    system.close()


def test_multi_column_solve_is_the_column_by_column_solve(sad, small):
    """`dns_saddle_solve_multi`: `(NV, k)` blocks in one call of the boundary
    (`lau.solve_sadpnt_smw` takes them: `umat`/`vmat`, `apply_massinv`) -- the
    same answers, iteration counts and residual histories as k single calls"""
    F, J = small['F'], small['J']
    NP, NV = J.shape
    rng = np.random.default_rng(21)
    RV = small['M'] @ rng.standard_normal((NV, 3))
    RP = 1e-3*rng.standard_normal((NP, 3))
    system = sad.SaddleSystem(F, J)
    system.setup_precond(cheb_degree=4, schur='dense')
    X = system.solve_multi(RV, RP, rtol=1e-11, use_graph=True)
    assert X.shape == (NV + NP, 3) and len(system.last_stats_cols) == 3
    for c in range(3):
        xc = system.solve(RV[:, c], RP[:, c], rtol=1e-11, use_graph=True)
        st = system.last_stats
        assert np.array_equal(xc, X[:, c])
        assert st['iters'] == system.last_stats_cols[c]['iters']
        hist = system.residual_history()
        assert np.array_equal(hist, system.residual_history_col(c))
        ref = saddle_oracle.solve_sadpnt_smw(amat=F, jmat=J, rhsv=RV[:, c],
                                             rhsp=RP[:, c]).reshape(-1)
        assert np.linalg.norm(xc - ref) <= 1e-8*np.linalg.norm(ref)
    # no pressure block handed over = zeros; ONE start vector for all columns,
    # or one each (the exact solutions: no iteration)
    X0 = system.solve_multi(RV, None, rtol=1e-11)
    for c in range(3):
        assert np.array_equal(X0[:, c], system.solve(RV[:, c], None, rtol=1e-11))
    Xs = system.solve_multi(RV, RP, x0=X, rtol=1e-9)
    assert [st['iters'] for st in system.last_stats_cols] == [0, 0, 0]
    assert np.array_equal(Xs, X)
    Xo = system.solve_multi(RV, RP, x0=X[:, :1], rtol=1e-9)
    assert system.last_stats_cols[0]['iters'] == 0
    assert system.last_stats_cols[1]['iters'] > 0
    assert np.linalg.norm(Xo - X) <= 1e-7*np.linalg.norm(X)
    # a column that cannot converge is reported, not swallowed
    with pytest.raises(_capi_mod().NotConverged):
        system.solve_multi(RV, RP, rtol=1e-14, maxiter=2)
    with pytest.raises(ValueError):
        system.solve_multi(RV, RP, x0=X[:, :2])
    system.close()


def test_multi_column_solve_many_calls_of_changing_width(sad, small):
    """blocks of 1..7 columns, 120 calls on one resident system, host arrays
    that die right behind the call (the block copies of the boundary: a pitched
    2-D copy out of pageable memory once faulted the GPU here)"""
    F, J = small['F'], small['J']
    NP, NV = J.shape
    rng = np.random.default_rng(5)
    system = sad.SaddleSystem(F, J)
    system.setup_precond(cheb_degree=4, schur='dense')
    ref = {}
    for call in range(120):
        k = 1 + call % 7
        RV = small['M'] @ rng.standard_normal((NV, k))
        RP = 1e-3*rng.standard_normal((NP, k)) if call % 3 else None
        X = system.solve_multi(RV, RP, rtol=1e-10, use_graph=True)
        assert X.shape == (NV + NP, k)
        if call % 20 == 0:
            for c in range(k):
                rp = None if RP is None else RP[:, c]
                ref = system.solve(RV[:, c], rp, rtol=1e-10, use_graph=True)
                assert np.array_equal(ref, X[:, c])
        del RV, RP, X
    system.close()


def test_zero_rhs_and_no_rhsp(sad, small):
    system = sad.SaddleSystem(small['F'], small['J'])
    system.setup_precond(cheb_degree=3, schur='dense')
    x = system.solve(np.zeros(system.NV))
    assert np.array_equal(x, np.zeros(system.n))
    x = system.solve(small['rhsv'], rtol=1e-12)      # rhsp omitted -> 0
    ref = saddle_oracle.solve_sadpnt_smw(amat=small['F'], jmat=small['J'],
                                         rhsv=small['rhsv']).reshape(-1)
    assert np.linalg.norm(x - ref) <= 1e-8*np.linalg.norm(ref)
    system.close()


def test_not_converged_raises_and_not_ready(sad, small):
    from dolfin_navier_scipy_amd import _capi
    system = sad.SaddleSystem(small['F'], small['J'])
    with pytest.raises(_capi.DnsError):
        system.solve(small['rhsv'], small['rhsp'])
    system.setup_precond(cheb_degree=1, schur='jacobi')
    with pytest.raises(_capi.NotConverged):
        system.solve(small['rhsv'], small['rhsp'], rtol=1e-14, maxiter=3)
    system.close()


def test_update_values_same_pattern(sad, small):
    """Newton/Picard: F <- M + dt/2 (A + N), same pattern (snu:1034)"""
    M, A, J = small['M'], small['A'], small['J']
    dt = small['dt']
    system = sad.SaddleSystem(small['F'], J)
    system.setup_precond(cheb_degree=4, schur='dense')
    rng = np.random.default_rng(4)
    pert = A.copy()
    pert.data = pert.data*(1 + 0.3*rng.standard_normal(pert.nnz))
    F2 = (M + .5*dt*pert).tocsr()
    assert np.array_equal(F2.indices, small['F'].indices)
    system.update_values(F2.data)
    x = system.solve(small['rhsv'], small['rhsp'], rtol=1e-12)
    ref = saddle_oracle.solve_sadpnt_smw(amat=F2, jmat=J, rhsv=small['rhsv'],
                                         rhsp=small['rhsp']).reshape(-1)
    assert np.linalg.norm(x - ref) <= 1e-8*np.linalg.norm(ref)
    system.close()


def test_bad_arguments(sad, small):
    from dolfin_navier_scipy_amd import _capi
    F, J = small['F'], small['J']
    with pytest.raises(_capi.DnsError):
        sad.SaddleSystem(F, J[:, :-1])
    bad = F.copy()
    bad.indices = bad.indices.copy()
    bad.indices[0] = F.shape[1] + 5
    with pytest.raises(_capi.DnsError):
        sad.SaddleSystem(bad, J)


def test_lau_surface(small):
    from dolfin_navier_scipy_amd import lin_alg_utils as lau
    M, A, J = small['M'], small['A'], small['J']
    NP, NV = J.shape
    rng = np.random.default_rng(8)
    rhsv = rng.standard_normal((NV, 2))
    rhsp = 1e-2*rng.standard_normal((NP, 2))
    amat = (M + 0.01*A).tocsr()
    vp = lau.solve_sadpnt_smw(amat=amat, jmat=J, jmatT=J.T, rhsv=rhsv,
                              rhsp=rhsp)
    ref = saddle_oracle.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv,
                                         rhsp=rhsp)
    assert vp.shape == (NV+NP, 2)
    assert np.linalg.norm(vp - ref) <= 1e-8*np.linalg.norm(ref)
    # the tolerance that stood in for the direct solve is on record, and a
    # caller that knows it penalised rows (Robin control) can say so instead
    # of leaving it to the look at the diagonal
    assert lau.LAST_SOLVE == dict(rtol=lau.DEFAULTS['direct_tol'],
                                  rtol_chosen_by='diagonal heuristic')
    lau.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv[:, :1],
                         krplsprms=dict(penalised=True))
    assert lau.LAST_SOLVE['rtol'] == lau.DEFAULTS['direct_tol_penalised']
    lau.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv[:, :1],
                         krplsprms=dict(direct_tol=1e-9))
    assert lau.LAST_SOLVE == dict(rtol=1e-9, rtol_chosen_by='direct_tol given')
    # krylov kwargs, residual history sink, warm start
    stats = []
    vpk = lau.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv[:, :1],
                               rhsp=rhsp[:, :1], krylov='Gmres',
                               krpslvprms={'tol': 1e-6, 'maxiter': 200,
                                           'convstatsl': stats,
                                           'x0': ref[:, :1]})
    assert len(stats) == 1 and len(stats[0]) == 1   # x0 exact: no iteration
    assert np.linalg.norm(vpk - ref[:, :1]) <= 1e-5*np.linalg.norm(ref[:, 0])
    # low-rank update (SMW)
    U = 1e-3*rng.standard_normal((NV, 2))
    V = rng.standard_normal((2, NV))
    vps = lau.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv, rhsp=rhsp,
                               umat=U, vmat=V)
    refs = saddle_oracle.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv,
                                          rhsp=rhsp, umat=U, vmat=V)
    assert np.linalg.norm(vps - refs) <= 1e-7*np.linalg.norm(refs)
    # return_alu
    sol, fn = lau.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=0*rhsv[:, :1],
                                   return_alu=True)
    assert np.array_equal(sol, np.zeros((NV+NP, 1)))
    got = fn(np.vstack([rhsv[:, :1], rhsp[:, :1]]))
    assert got.shape == (NV+NP, 1)
    assert np.linalg.norm(got - ref[:, :1]) <= 1e-8*np.linalg.norm(ref[:, 0])
    # projector
    f = rng.standard_normal((NV, 1))
    pf = lau.app_prj_via_sadpnt(amat=M, jmat=J, rhsv=f)
    assert np.linalg.norm(J @ pf) <= 1e-8*np.linalg.norm(f)
    ptf = lau.app_prj_via_sadpnt(amat=M, jmat=J, rhsv=f, transposedprj=True)
    pto = saddle_oracle.app_prj_via_sadpnt(amat=M, jmat=J, rhsv=f,
                                           transposedprj=True)
    assert np.linalg.norm(ptf - pto) <= 1e-7*np.linalg.norm(pto)
    # the decoupled variant (`amat` omitted, snu:1622): with solve_A = identity
    # the system is [[I, J^T], [J, 0]]
    dec = lau.solve_sadpnt_smw(jmat=J, rhsv=rhsv[:, :1], decouplevp=True,
                               symmetric=True, solve_A=lambda x: x,
                               cgtol=1e-12)
    refd = saddle_oracle.solve_sadpnt_smw(amat=sps.identity(NV, format='csr'),
                                          jmat=J, rhsv=rhsv[:, :1])
    assert np.linalg.norm(dec - refd) <= 1e-8*np.linalg.norm(refd)
    lau.clear_cache()


def test_replay_of_the_reference_integrators_boundary_calls(golden_dir):
    """the `(amat, jmat, rhsv, rhsp) -> vp` pairs recorded at the `lau` stub
    while the reference's own `time_int_utils` ran (tests/golden/lau_calls.npz,
    SURVEY.md 8c), replayed through the drop-in `lin_alg_utils` on the GPU:
    Heun predictor `M + dt A` / corrector `M` (tiu:402,466) of six scenarios
    and the `return_alu` call of `semi_implicit_euler` (tiu:605)"""
    from dolfin_navier_scipy_amd import lin_alg_utils as lau
    from test_oracle_golden import load_lau_calls
    J, calls = load_lau_calls(golden_dir)
    NP, NV = J.shape
    for tag, amat, rhsv, rhsp, vp in calls:
        got = lau.solve_sadpnt_smw(amat=amat, jmat=J, jmatT=J.T, rhsv=rhsv,
                                   rhsp=rhsp)
        assert got.shape == vp.shape, tag
        if not np.any(vp):       # tiu:605 solves with a zero right-hand side
            assert not np.any(got), tag
            continue
        ev = np.linalg.norm(got[:NV] - vp[:NV])/np.linalg.norm(vp[:NV])
        ep = np.linalg.norm(got[NV:] - vp[NV:])/np.linalg.norm(vp[NV:])
        assert ev <= 1e-9 and ep <= 1e-7, (tag, ev, ep)
    # the same resident system served all calls with that pattern (two
    # patterns: `M + dt A` and `M`)
    assert len(lau._cache) <= 2
    lau.clear_cache()


def test_same_pattern_different_J_is_a_different_system(small):
    """the cache of resident systems must not serve a stale `J`"""
    from dolfin_navier_scipy_amd import lin_alg_utils as lau
    M, A, J = small['M'], small['A'], small['J']
    amat = (M + 0.01*A).tocsr()
    rhsv, rhsp = small['rhsv'], small['rhsp']
    lau.clear_cache()
    v1 = lau.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv, rhsp=rhsp)
    J2 = (2.0*J).tocsr()                       # same pattern, other values
    v2 = lau.solve_sadpnt_smw(amat=amat, jmat=J2, rhsv=rhsv, rhsp=rhsp)
    r1 = saddle_oracle.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv,
                                        rhsp=rhsp)
    r2 = saddle_oracle.solve_sadpnt_smw(amat=amat, jmat=J2, rhsv=rhsv,
                                        rhsp=rhsp)
    assert np.linalg.norm(v1 - r1) <= 1e-8*np.linalg.norm(r1)
    assert np.linalg.norm(v2 - r2) <= 1e-8*np.linalg.norm(r2)
    assert np.linalg.norm(r2 - r1) > 1e-3*np.linalg.norm(r1)
    assert len(lau._cache) == 2
    lau.clear_cache()


def test_multigrid_schur_block_on_a_refined_mesh(sad):
    """`schur='mg'` (one V-cycle on the sparse Schur complement, nested
    pressure spaces of a red-refined mesh, dense inverse on the coarsest):
    the solve of the once-refined cylinder problem (NV=38 018, NP=4 991)
    matches the oracle's direct solve, needs only a few more Krylov steps than
    the dense Schur inverse, and the single-level hierarchy IS the dense one"""
    from dolfin_navier_scipy_amd.fem import (
        get_sysmats, cylinder_mesh_hierarchy, pressure_prolongations,
        TaylorHood)
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=1, Re=100)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    hier = cylinder_mesh_hierarchy(N=2, refine=1)
    spaces = [TaylorHood(m) for m, _ in hier][::-1]
    prols = pressure_prolongations(spaces, [p for _, p in hier][::-1])
    assert prols[0].shape == (NP, 1289)
    F = (M + A/1024.).tocsr()
    rng = np.random.default_rng(1)
    rhsv = M @ rng.standard_normal(NV)
    rhsp = 1e-3*(J @ rng.standard_normal(NV))
    ref = saddle_oracle.solve_sadpnt_smw(amat=F, jmat=J, rhsv=rhsv,
                                         rhsp=rhsp).reshape(-1)
    its = {}
    for name, schur, fact, fhat, pr in (
            ('dense', 'dense', 'full', 'explicit', None),
            ('mg full', 'mg', 'full', 'explicit', prols),
            ('mg half', 'mg', 'full', 'explicit', prols),
            ('mg tri', 'mg', 'triangular', 'explicit', prols),
            ('mg cheb', 'mg', 'triangular', 'cheb', prols),
            ('mg 1 level', 'mg', 'full', 'explicit', [])):
        system = sad.SaddleSystem(F, J)
        if pr is not None:
            system.set_schur_mg(pr)
        if name == 'mg half':
            # the 1289-row level as a dense inverse in HALF precision
            # (`mg_dense_half_max`, csrc/dense_half.hpp)
            system.set_option('mg_dense_max', 1000)
            system.set_option('mg_dense_half_max', 2000)
        system.setup_precond(cheb_degree=6, schur=schur, fhat=fhat,
                             drop_tol=1e-3, factorization=fact)
        for reorth in (1, 2):
            x = system.solve(rhsv, rhsp, rtol=1e-11, maxiter=400,
                             reorth=reorth, use_graph=True)
            st = system.last_stats
            assert st['status'] == 0 and st['true_relres'] <= 2e-11, (name, st)
            assert np.linalg.norm(x[:NV] - ref[:NV]) <= \
                1e-9*np.linalg.norm(ref[:NV]), name
            assert np.linalg.norm(x[NV:] - ref[NV:]) <= \
                1e-6*np.linalg.norm(ref[NV:]), name
            if reorth == 1:
                # (the fused Gram-Schmidt may end a cold-start cycle early and
                # go on with the explicit kernel: its count is not comparable)
                its[name] = st['iters']
        system.close()
    assert abs(its['mg 1 level'] - its['dense']) <= 1    # (fp64 vs fp32 inverse)
    assert its['mg full'] <= 3*its['dense'] + 2
    # (half precision rounds every entry of the inverse at 2^-11: a few more
    # Krylov steps on a cold start, the same answer)
    assert its['mg full'] <= its['mg half'] <= its['mg full'] + 6
    assert its['mg tri'] <= 25 and its['mg cheb'] <= 40
    # the multigrid needs its hierarchy
    from dolfin_navier_scipy_amd import _capi
    system = sad.SaddleSystem(F, J)
    with pytest.raises(_capi.DnsError):
        system.setup_precond(cheb_degree=4, schur='mg')
    with pytest.raises(_capi.DnsError):
        system.set_schur_mg([prols[0][:-1, :]])
    system.close()


@pytest.mark.parametrize('fact', ['triangular', 'full'])
def test_driven_cavity_stokes_and_step(sad, fact):
    """BASELINE config 0 (`tests/mini_setup.py` plumbing case): enclosed flow,
    pressure pinned by dropping the last pressure dof (dnsps:178-182).  The
    steady Stokes system `[[A, JT],[J, 0]]` (no mass shift: the hardest system
    for the Krylov solver, snu:903-907) and one implicit step, both through the
    drop-in `lau`, against the oracle's direct solves"""
    from dolfin_navier_scipy_amd import lin_alg_utils as lau
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='drivencavity', N=24, Re=50)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    assert femp['ppin'] == -1 and NP == 25*25 - 1
    ref = saddle_oracle.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'],
                                         rhsp=rhsd['fp'])
    lau.clear_cache()
    got = lau.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'], rhsp=rhsd['fp'],
                               krplsprms=dict(factorization=fact))
    assert np.linalg.norm(got[:NV] - ref[:NV]) <= 1e-8*np.linalg.norm(ref[:NV])
    assert np.linalg.norm(got[NV:] - ref[NV:]) <= 1e-6*np.linalg.norm(ref[NV:])
    assert np.abs(J @ got[:NV] - rhsd['fp']).max() <= 1e-9
    dt = 1e-2
    F = (M + dt*A).tocsr()
    rhs = M @ ref[:NV] + dt*rhsd['fv']
    ref2 = saddle_oracle.solve_sadpnt_smw(amat=F, jmat=J, rhsv=rhs,
                                          rhsp=rhsd['fp'])
    got2 = lau.solve_sadpnt_smw(amat=F, jmat=J, rhsv=rhs, rhsp=rhsd['fp'],
                                krplsprms=dict(factorization=fact))
    assert np.linalg.norm(got2[:NV] - ref2[:NV]) <= \
        1e-8*np.linalg.norm(ref2[:NV])
    lau.clear_cache()


def test_stiff_penalty_entries_like_robin_control(sad, small):
    """BASELINE config 4's `A += Arob/alpha`, alpha = 1e-5 (reference
    tests/time_dep_nse_double_rotcyl_bcrob.py:37-38): a few velocity dofs carry
    entries 1e5 x larger than the rest.  The Jacobi scaling inside the
    Chebyshev polynomial has to absorb that"""
    F, J = small['F'], small['J']
    NV = F.shape[0]
    rng = np.random.default_rng(8)
    idx = rng.choice(NV, 24, replace=False)
    pen = sps.csr_matrix((np.full(idx.size, 1e5*abs(F.diagonal()).mean()),
                          (idx, idx)), shape=F.shape)
    # coupled 2x2 blocks as a boundary mass matrix would produce
    off = sps.csr_matrix((np.full(idx.size//2, 3e4*abs(F.diagonal()).mean()),
                          (idx[0::2], idx[1::2])), shape=F.shape)
    Fp = (F + small['dt']*(pen + off + off.T)).tocsr()
    ref = saddle_oracle.solve_sadpnt_smw(amat=Fp, jmat=J, rhsv=small['rhsv'],
                                         rhsp=small['rhsp']).reshape(-1)
    for fact in ('triangular', 'full'):
        system = sad.SaddleSystem(Fp, J)
        system.setup_precond(cheb_degree=6, schur='dense', factorization=fact)
        x = system.solve(small['rhsv'], small['rhsp'], rtol=1e-12, maxiter=400)
        st = system.last_stats
        assert st['status'] == 0 and st['true_relres'] <= 5e-12, (fact, st)
        assert st['iters'] <= 60, (fact, st['iters'])
        assert np.linalg.norm(x[:NV] - ref[:NV]) <= 1e-8*np.linalg.norm(ref[:NV])
        system.close()


def test_streaming_kernels_forced_on_small_systems(sad, monkeypatch):
    """the bandwidth-regime kernels (`k_spmv_stream16x`: split input for Gc,
    fused Gram-Schmidt dots on the K apply, Jacobi-sweep epilogue of the
    multigrid smoother, streamed residual / IMEX right-hand side) normally
    start at 8e5 non-zeros; `DNS_STREAM_NNZ=1` routes a small system through
    all of them -- same answers as the latency-regime kernels and the oracle"""
    from dolfin_navier_scipy_amd.fem import (
        get_sysmats, cylinder_mesh_hierarchy, pressure_prolongations,
        TaylorHood)
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=1, Re=100)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    hier = cylinder_mesh_hierarchy(N=2, refine=1)
    spaces = [TaylorHood(m) for m, _ in hier][::-1]
    prols = pressure_prolongations(spaces, [p for _, p in hier][::-1])
    dt = 1./1024
    F, R1 = (M + .5*dt*A).tocsr(), (M - .5*dt*A).tocsr()
    rng = np.random.default_rng(5)
    rhsv = M @ rng.standard_normal(NV)
    rhsp = 1e-3*(J @ rng.standard_normal(NV))
    ref = saddle_oracle.solve_sadpnt_smw(amat=F, jmat=J, rhsv=rhsv,
                                         rhsp=rhsp).reshape(-1)
    v0 = ref[:NV].copy()
    res = {}
    # 'stream': K is applied through the pair format (2x2 node blocks,
    # pair.hpp); 'stream_csr': through the CSR streaming kernel (DNS_PAIR=0)
    for mode, thr, pair in (('latency', '1000000000', '1'),
                            ('stream', '1', '1'), ('stream_csr', '1', '0')):
        monkeypatch.setenv('DNS_STREAM_NNZ', thr)
        monkeypatch.setenv('DNS_PAIR', pair)
        monkeypatch.setenv('DNS_MG_DENSE_MAX', '2000')
        system = sad.SaddleSystem(F, J)
        system.set_schur_mg(prols)
        system.setup_precond(cheb_degree=6, schur='mg', drop_tol=1e-3,
                             factorization='full')
        pbytes = system.precond_info()['pair_format_bytes']
        assert (pbytes > 0) == (mode == 'stream'), (mode, pbytes)
        for reorth in (1, 2):
            x = system.solve(rhsv, rhsp, rtol=1e-11, maxiter=400,
                             reorth=reorth, use_graph=(reorth == 2))
            st = system.last_stats
            assert st['status'] == 0 and st['true_relres'] <= 2e-11, (mode, st)
            assert np.linalg.norm(x[:NV] - ref[:NV]) <= \
                1e-9*np.linalg.norm(ref[:NV]), mode
            assert np.linalg.norm(x[NV:] - ref[NV:]) <= \
                1e-6*np.linalg.norm(ref[NV:]), mode
            if reorth == 1:
                res[mode + '_its'] = st['iters']
        # resident stepping: streamed rhs / residual kernels in the prologue
        stp = sad.ImexStepper(system, R1)
        stp.set_state(v0)
        stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
        cf = sad.ImexStepper.coeffs(a_c=1., pscale=-1./dt, extrapolate=4)
        opts = sad.solve_opts(rtol=1e-12, maxiter=300, use_graph=True,
                              reorth=2)
        stp.run(21, cf, opts)
        res[mode] = stp.get_state()
        stp.close()
        system.close()
    assert abs(res['stream_its'] - res['latency_its']) <= 1
    assert abs(res['stream_csr_its'] - res['latency_its']) <= 1
    (vl, pl), (vs, ps) = res['latency'], res['stream']
    for vq, pq in (res['stream'], res['stream_csr']):
        assert np.linalg.norm(vq - vl) <= 1e-9*np.linalg.norm(vl)
        assert np.linalg.norm(pq - pl) <= 1e-7*np.linalg.norm(pl)
    # and against the factor-once oracle loop (tiu:104-143, no convection)
    lu = saddle_oracle.SaddleLU(F, J)
    v = v0.reshape((-1, 1))
    for _ in range(21):
        vp = lu(np.vstack([R1 @ v + dt*rhsd['fv'], rhsd['fp']]))
        v, pt = vp[:NV], vp[NV:]
    assert np.linalg.norm(vs - v) <= 1e-8*np.linalg.norm(v)
    assert np.linalg.norm(ps + pt/dt) <= 1e-6*np.linalg.norm(pt/dt)


def test_debug_uploads_prints_the_host_ranges_of_every_copy():
    """`DNS_DEBUG_UPLOADS=1` (round 5, after a GPU memory fault at a host
    address whose record held no ranges): every copy between host memory and
    the device says `[ptr, ptr + bytes)` on stderr before it is enqueued, the
    blocks of `dns_saddle_solve_multi` included -- a fault address can then be
    tied to a buffer and an offset"""
    import subprocess
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import scenarios
from dolfin_navier_scipy_amd import saddle
prob = scenarios.toy_problem(nx=9, ny=4)
M, A, J = (prob['smc'][k] for k in 'MAJ')
NP, NV = J.shape
system = saddle.SaddleSystem((M + 1e-2*A).tocsr(), J)
system.setup_precond(cheb_degree=4, schur='dense')
rhs = np.random.default_rng(0).standard_normal((NV, 3))
out = system.solve_multi(rhs, rtol=1e-10)
one = system.solve(rhs[:, 0].copy(), rtol=1e-10)
print('RHS', hex(rhs.ctypes.data), rhs.nbytes)
system.close()
''' % (ROOT, HERE)
    env = dict(os.environ, DNS_DEBUG_UPLOADS='1')
    res = subprocess.run([sys.executable, '-c', code], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    err = res.stderr.decode()
    lines = [ln for ln in err.splitlines() if ln.startswith('[dns copy]')]
    assert len(lines) > 10
    kinds = set(ln.split()[2] for ln in lines)
    assert {'upload', 'download'} <= kinds
    assert any('multi rhs_v' in ln for ln in lines)
    assert any('multi out' in ln for ln in lines)
    # every line carries a well-formed half-open host range
    import re
    for ln in lines:
        m = re.search(r'host \[(0x[0-9a-f]+), (0x[0-9a-f]+)\) dev '
                      r'(0x[0-9a-f]+|\(nil\)) bytes (\d+)', ln)
        assert m, ln
        assert int(m.group(2), 16) - int(m.group(1), 16) == int(m.group(4))
    # without the switch nothing is printed
    res2 = subprocess.run([sys.executable, '-c', code],
                          env={k: v for k, v in os.environ.items()
                               if k != 'DNS_DEBUG_UPLOADS'},
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=300)
    assert res2.returncode == 0 and '[dns copy]' not in res2.stderr.decode()
