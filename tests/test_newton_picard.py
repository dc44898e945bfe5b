"""Newton/Picard trapezoidal sweeps (reference snu:1016-1047, 1304-1566).

CPU part: the oracle restatement satisfies its defining equations (the branch
cannot be executed in the reference -- dolfin + bit rot -- so it is pinned by
definition, see oracle/newton_picard_oracle.py).  GPU part: the same sweeps
with every saddle solve going through the drop-in `lin_alg_utils` (re-valued
`M + dt/2 (A + N(v))`, same pattern every step) reproduce the oracle.
"""
import numpy as np
import pytest

import scenarios
from oracle import newton_picard_oracle as npo
from oracle import imex_oracle


@pytest.fixture(scope='module')
def setup(toy_prob):
    th, smc, rhsd = toy_prob['th'], toy_prob['smc'], toy_prob['rhsd']
    inv, dbcinds, dbcvals = (toy_prob['invinds'], toy_prob['dbcinds'],
                             toy_prob['dbcvals'])
    M, A, J = smc['M'], smc['A'], smc['J']
    NP, NV = J.shape

    def appnd(vvec):
        full = np.zeros((th.vdim, 1))
        full[inv] = vvec
        full[dbcinds, 0] = dbcvals
        return full

    bcsv = np.zeros((th.vdim, 1))
    bcsv[dbcinds, 0] = dbcvals

    def conv(vfull, picard):
        """algebraic `get_v_conv_conts` (snu:109-133) + `condense_velmatsbybcs`
        (dts:610-642)"""
        if vfull.shape[0] == NV:
            vfull = appnd(vfull)
        N1, N2, fv3 = th.convection_mats(vfull, keep_pattern=True)
        Nm = N1 if picard else (N1 + N2)
        Nc = Nm[inv, :][:, inv].tocsr()
        rhsbc = -(Nm @ bcsv)[inv, :]
        return Nc, (0.*fv3[inv, :] if picard else fv3[inv, :]), rhsbc

    kw, rec, aux = scenarios.build(variant='plain', seed=0, Nts=6, tE=0.03,
                                   prob=toy_prob)
    imex_oracle.cnab(**kw)          # semi-explicit run = first lin. points
    times, vels, _ = rec.arrays()
    lin0 = {t: vels[k].reshape((-1, 1)) for k, t in enumerate(times)}
    return dict(M=M, A=A, J=J, fv=rhsd['fv'], fp=rhsd['fp'], conv=conv,
                appnd=appnd, trange=kw['trange'], iniv=kw['inivel'], lin0=lin0,
                inv=inv, NV=NV, NP=NP)


def _sweep_kwargs(s):
    return dict(M=s['M'], A=s['A'], J=s['J'], fv=s['fv'], fp=s['fp'],
                conv=s['conv'], appndbcs=s['appnd'])


def test_oracle_trapezoidal_residual_and_newton_convergence(setup):
    s = setup
    vdict, pdict, hist = npo.newton_picard(
        s['trange'], s['iniv'], s['lin0'], vel_pcrd_stps=1, vel_nwtn_stps=3,
        invinds=s['inv'], **_sweep_kwargs(s))
    kinds = [h[0] for h in hist]
    assert kinds[0] == 'picard' and 'newton' in kinds
    upd = [h[1] for h in hist]
    # Newton updates contract quadratically-ish: each at least 30x smaller
    assert upd[-1] < upd[0]*1e-3
    # converged sweep: nonlinear trapezoidal residual of every step
    M, A, J, conv, appnd = s['M'], s['A'], s['J'], s['conv'], s['appnd']
    tr = s['trange']
    for k in range(1, len(tr)):
        dt = tr[k] - tr[k-1]
        vn, vc = vdict[tr[k]], vdict[tr[k-1]]
        # full convection N(v)v at both ends (Newton form evaluated at itself)
        Nn, rcn, rbn = conv(appnd(vn), False)
        Nc, rcc, rbc = conv(appnd(vc), False)
        # N1 v + N2 v - N(v)v = N(v) v  at the linearisation point
        res = M @ (vn - vc) + .5*dt*((A @ vn) + (Nn @ vn) - rcn - rbn
                                     + (A @ vc) + (Nc @ vc) - rcc - rbc
                                     - 2*s['fv']) - dt*(J.T @ pdict[tr[k]])
        assert np.abs(res).max() <= 5e-8*np.abs(M @ vn).max()
        assert np.abs(J @ vn - s['fp']).max() <= 1e-10


def test_oracle_time_sections_and_full_sweep(setup):
    """`nsects`, `addfullsweep` (snu:1076-1090, 1576-1587): the sections share
    their end points and cover the range; the sectioned iteration converges to
    the SAME discrete trapezoidal solution as the plain one"""
    s = setup
    tr = s['trange']
    secs = npo.time_sections(tr, nsects=3, addfullsweep=True)
    assert len(secs) == 4 and np.array_equal(secs[-1], tr)
    assert secs[0][0] == tr[0] and secs[2][-1] == tr[-1]
    assert secs[0][-1] == secs[1][0] and secs[1][-1] == secs[2][0]
    assert len(npo.time_sections(tr, 1, True)) == 1           # snu:1087-1090
    plain_v, plain_p, _ = npo.newton_picard(
        tr, s['iniv'], s['lin0'], vel_pcrd_stps=1, vel_nwtn_stps=5,
        invinds=s['inv'], **_sweep_kwargs(s))
    lin0 = dict(s['lin0'])
    lin0[None] = s['lin0'][tr[0]]
    sec_v, sec_p, hist = npo.newton_picard(
        tr, s['iniv'], lin0, vel_pcrd_stps=1, vel_nwtn_stps=5,
        invinds=s['inv'], nsects=3, loc_nwtn_tol=1e-14, addfullsweep=True,
        **_sweep_kwargs(s))
    # every section has its own Picard sweep (loc_pcrd_stps); the full sweep
    # has none (the `elif` at snu:1583 does not refill the Picard count)
    assert [h[0] for h in hist].count('picard') == 3
    for t in tr[1:]:
        assert np.linalg.norm(sec_v[t] - plain_v[t]) <= \
            1e-8*np.linalg.norm(plain_v[t]), t
        assert np.linalg.norm(sec_p[t] - plain_p[t]) <= \
            1e-6*np.linalg.norm(plain_p[t]), t


def test_oracle_get_pfromv_consistency(setup):
    """reference tests/test_units_pfromv.py:45 in algebraic form: the pressure
    recomputed from a velocity satisfies the momentum equation's projection"""
    s = setup
    v = s['iniv']
    p = npo.get_pfromv(v=v, M=s['M'], A=s['A'], J=s['J'], fv=s['fv'],
                       conv=s['conv'], appndbcs=s['appnd'])
    assert p.shape == (s['NP'], 1)
    # by construction M a + J^T p~ = rhs with J a = 0
    _, rc, _ = s['conv'](s['appnd'](v), False)
    rhs = -(s['A'] @ v) - rc + s['fv']
    from oracle import saddle_oracle
    ap = saddle_oracle.solve_sadpnt_smw(amat=s['M'], jmat=s['J'], rhsv=rhs)
    assert np.allclose(-ap[s['NV']:], p)
    assert np.abs(s['J'] @ ap[:s['NV']]).max() <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize('krylovini', [None, 'upd'])
def test_gpu_sweeps_match_oracle(setup, krylovini):
    from dolfin_navier_scipy_amd import lin_alg_utils as lau, _capi
    assert _capi.device_count() > 0
    s = setup
    ref_v, ref_p, _ = npo.trapezoidal_sweep(
        s['trange'], s['iniv'], linpoints=s['lin0'], picard=False,
        **_sweep_kwargs(s))
    lau.clear_cache()
    got_v, got_p, _ = npo.trapezoidal_sweep(
        s['trange'], s['iniv'], linpoints=s['lin0'], picard=False,
        solve=lau.solve_sadpnt_smw, krylovini=krylovini, **_sweep_kwargs(s))
    # ONE resident system re-valued every step (same pattern), not one per step
    assert len(lau._cache) == 1
    for t in s['trange'][1:]:
        ev = np.linalg.norm(got_v[t] - ref_v[t])/np.linalg.norm(ref_v[t])
        ep = np.linalg.norm(got_p[t] - ref_p[t])/np.linalg.norm(ref_p[t])
        assert ev <= 1e-8 and ep <= 1e-6, (t, ev, ep)
    lau.clear_cache()


@pytest.mark.gpu
def test_gpu_get_pfromv(setup):
    from dolfin_navier_scipy_amd import lin_alg_utils as lau
    s = setup
    args = dict(v=s['iniv'], M=s['M'], A=s['A'], J=s['J'], fv=s['fv'],
                conv=s['conv'], appndbcs=s['appnd'])
    ref = npo.get_pfromv(**args)
    got = npo.get_pfromv(solve=lau.solve_sadpnt_smw, **args)
    assert np.linalg.norm(got - ref) <= 1e-7*np.linalg.norm(ref)
    lau.clear_cache()


def test_refined_solve_is_the_direct_solve(setup):
    """`saddle_oracle.RefinedSolve` (the LU of one system of a sweep serving the
    following ones through iterative refinement; used by the long-horizon GPU
    tests, where a fresh LU per step takes minutes): the same sweep as with
    the direct solve, to rounding, and with fewer factorisations than steps"""
    from oracle import saddle_oracle
    s = setup
    for picard in (True, False):
        ref_v, ref_p, ref_upd = npo.trapezoidal_sweep(
            s['trange'], s['iniv'], linpoints=s['lin0'], picard=picard,
            **_sweep_kwargs(s))
        rs = saddle_oracle.RefinedSolve()
        got_v, got_p, upd = npo.trapezoidal_sweep(
            s['trange'], s['iniv'], linpoints=s['lin0'], picard=picard,
            solve=rs, **_sweep_kwargs(s))
        for t in s['trange'][1:]:
            assert np.linalg.norm(got_v[t] - ref_v[t]) <= \
                1e-12*np.linalg.norm(ref_v[t])
            assert np.linalg.norm(got_p[t] - ref_p[t]) <= \
                1e-10*np.linalg.norm(ref_p[t])
        assert abs(upd - ref_upd) <= 1e-10*abs(ref_upd) + 1e-20
        assert rs.calls == len(s['trange']) - 1
        assert rs.factorisations < rs.calls
    # a system far from the factored one falls back to its own LU
    rs = saddle_oracle.RefinedSolve(max_inner=2)
    rng = np.random.default_rng(3)
    b = rng.standard_normal((s['NV'], 1))
    x1 = rs(amat=s['M'] + 0.01*s['A'], jmat=s['J'], rhsv=b)
    x2 = rs(amat=s['M'] + 50.*s['A'], jmat=s['J'], rhsv=b)
    ref = saddle_oracle.solve_sadpnt_smw(amat=s['M'] + 50.*s['A'], jmat=s['J'],
                                         rhsv=b)
    assert rs.factorisations == 2
    assert np.linalg.norm(x2 - ref) <= 1e-11*np.linalg.norm(ref)
