"""The GPU integrators against the golden vectors of the reference's own
`time_int_utils` and against the CPU oracle."""
import os

import numpy as np
import pytest

import scenarios
from oracle import imex_oracle

pytestmark = pytest.mark.gpu

# tolerances: velocities and pressures 1e-8 relative
# (p = -p~/dt amplifies the Krylov residual by 1/dt)
VTOL, PTOL = 1e-8, 1e-8        # v AND p


@pytest.fixture(scope='module')
def gtiu():
    from dolfin_navier_scipy_amd import time_int_utils, _capi
    assert _capi.device_count() > 0, 'HIP device required for -m gpu tests'
    return time_int_utils


@pytest.mark.parametrize('scheme', ['cnab', 'sbdf2'])
@pytest.mark.parametrize('seed,variant', list(enumerate(scenarios.VARIANTS)))
def test_integrators_match_reference_golden(gtiu, golden_dir, toy_prob, scheme,
                                            seed, variant):
    gold = np.load(os.path.join(
        golden_dir, 'imex_{0}_{1}_s{2}.npz'.format(scheme, variant, seed)))
    kw, rec, _ = scenarios.build(variant=variant, seed=seed, prob=toy_prob)
    if scheme == 'sbdf2':
        kw.pop('f_tvdp', None)
        v, p, ff = gtiu.sbdftwo(**kw)
    else:
        v, p, ff = gtiu.cnab(**kw)
    times, vels, prss = rec.arrays()
    assert ff == int(gold['ffflag'])
    assert np.allclose(times, gold['times'], rtol=0, atol=1e-15)
    for k in range(times.size):
        ev = np.linalg.norm(vels[k] - gold['vels'][k]) / \
            np.linalg.norm(gold['vels'][k])
        assert ev <= VTOL, (k, ev)
        if k > 0:
            ep = np.linalg.norm(prss[k] - gold['prss'][k]) / \
                np.linalg.norm(gold['prss'][k])
            assert ep <= PTOL, (k, ep)


def test_semi_implicit_euler_matches_golden(gtiu, golden_dir, toy_prob):
    gold = np.load(os.path.join(golden_dir, 'imex_sie_plain_s3.npz'))
    kw, rec, aux = scenarios.build(variant='plain', seed=3, prob=toy_prob)
    fvdp, cfv, appnd = kw['f_vdp'], aux['cfv'], kw['appndbcs']

    def rhsv(t, vvec):
        return cfv + fvdp(appnd(vvec.reshape((-1, 1)), []))
    vlist = gtiu.semi_implicit_euler(
        iniv=kw['inivel'], jmat=kw['J'], mmat=kw['M'], amat=kw['A'],
        rhsv=rhsv, trange=kw['trange'], data_trange=gold['data_trange'],
        fp=aux['cfp'])
    got = np.array([np.asarray(v).reshape(-1) for v in vlist])
    assert got.shape == gold['vlist'].shape
    for k in range(got.shape[0]):
        assert np.linalg.norm(got[k] - gold['vlist'][k]) <= \
            VTOL*np.linalg.norm(gold['vlist'][k])


def test_step_residual_and_divergence(gtiu, toy_prob):
    """algebraic AB2 residual (reference tests/test_units_residuals.py:121-124)
    and J v = fp on the GPU trajectory"""
    kw, rec, aux = scenarios.build(variant='plain', seed=5, Nts=6, tE=0.03,
                                   prob=toy_prob)
    M, A, J = kw['M'], kw['A'], kw['J']
    inv = toy_prob['invinds']
    fvdp = kw['f_vdp']
    v, p, ff = gtiu.cnab(**kw)
    times, vels, prss = rec.arrays()
    dt = times[1] - times[0]
    cfv, cfp = aux['cfv'].reshape(-1), aux['cfp'].reshape(-1)
    for k in range(2, times.size):
        v1, v2, v0 = vels[k-1][inv], vels[k][inv], vels[k-2][inv]
        n1 = fvdp(vels[k-1].reshape((-1, 1))).reshape(-1)
        n0 = fvdp(vels[k-2].reshape((-1, 1))).reshape(-1)
        res = M @ v2 + .5*dt*(A @ v2) - dt*(J.T @ prss[k]) \
            - (M @ v1 - .5*dt*(A @ v1) + .5*dt*(3*n1 - n0) + dt*cfv)
        assert np.allclose(res, 0., atol=1e-8), np.abs(res).max()
        assert np.allclose(J @ v2 - cfp, 0., atol=1e-9)


def test_blowup_guard(gtiu, toy_prob):
    kw, rec, _ = scenarios.build(variant='plain', seed=0, Nts=12, prob=toy_prob)
    kw['check_ff_maxv'] = 1e-3      # |v| is O(10): trips at the first slice
    v, p, ff = gtiu.cnab(**kw)
    assert ff == 1
    kw2, rec2, _ = scenarios.build(variant='plain', seed=0, Nts=12,
                                   prob=toy_prob)
    kw2['check_ff_maxv'] = 1e-3
    vo, po, ffo = imex_oracle.cnab(**kw2)
    assert ffo == 1 and len(rec.times) == len(rec2.times)


def test_full_size_cnab_against_oracle(gtiu):
    """cylinder wake N=2 (NV=9356, NP=1289), Re=80, dt=1/512: a handful of
    steps of the BASELINE configuration, GPU vs CPU oracle"""
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=80)
    th, inv = femp['V'], femp['invinds']
    dbcinds, dbcvals = femp['dbcinds'], femp['dbcvals']
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    from oracle import saddle_oracle
    vp0 = saddle_oracle.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'],
                                         rhsp=rhsd['fp'])   # snu:903-907
    inivel, inip = vp0[:NV], -vp0[NV:]

    def appnd(vvec, bcs):
        full = np.full((th.vdim, 1), np.nan)
        full[inv] = vvec
        full[dbcinds, 0] = dbcvals
        return full

    def make_kw(rec):
        return dict(trange=np.linspace(0, 6/512., 7), inivel=inivel,
                    inip=inip, bcs_ini=[], M=M, A=A, J=J,
                    f_vdp=lambda vf: -th.convection_vec(vf)[inv, :],
                    f_tdp=lambda t: rhsd['fv'], g_tdp=lambda t: rhsd['fp'],
                    scalep=-1., getbcs=lambda t, v, p, mode=None: [],
                    applybcs=lambda b: (0., 0., 0.), appndbcs=appnd,
                    savevp=rec, check_ff_maxv=1e8, verbose=False)
    rg, ro = scenarios.Recorder(), scenarios.Recorder()
    vg, pg, _ = gtiu.cnab(**make_kw(rg))
    vo, po, _ = imex_oracle.cnab(**make_kw(ro))
    mnorm = lambda x: np.sqrt((x.T @ (M @ x)).item())
    assert mnorm(vg - vo) <= VTOL*mnorm(vo)
    assert np.linalg.norm(pg - po) <= PTOL*np.linalg.norm(po)


@pytest.mark.parametrize('fact', ['triangular', 'full'])
@pytest.mark.parametrize('order', [0, 1, 2, 3, 4])
def test_pipelined_run_warm_start_orders(gtiu, order, fact):
    """`dns_imex_run` (grouped graphs, ring of five state buffers, replayed
    graphs advancing the host's view of the ring) against the oracle's
    factor-once loop, for every warm-start order; 45 steps = pipelined batches
    of unequal length, frozen convection history"""
    from dolfin_navier_scipy_amd import saddle
    from dolfin_navier_scipy_amd.fem import get_sysmats
    from oracle import saddle_oracle
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=1, Re=60)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    dt = 1./256
    rng = np.random.default_rng(11)
    vp0 = saddle_oracle.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'],
                                         rhsp=rhsd['fp'])
    v0 = vp0[:NV]
    nfc = 1e-2*rng.standard_normal((NV, 1))
    F, R1 = (M + .5*dt*A).tocsr(), (M - .5*dt*A).tocsr()
    nsteps = 45
    # oracle: tiu:104-143 with a frozen convection history
    lu = saddle_oracle.SaddleLU(F, J)
    v = v0.copy()
    for _ in range(nsteps):
        rhs = R1 @ v + 1.5*dt*nfc - .5*dt*nfc + dt*rhsd['fv']
        vp = lu(np.vstack([rhs, rhsd['fp']]))
        v, pt = vp[:NV], vp[NV:]
    system = saddle.SaddleSystem(F, J)
    system.setup_precond(cheb_degree=4, schur='dense', factorization=fact)
    stp = saddle.ImexStepper(system, R1)
    stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
    stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt, extrapolate=order)
    opts = saddle.solve_opts(rtol=1e-12, maxiter=300, restart=60,
                             use_graph=True, reorth=2)
    _, its, last = stp.run(nsteps, cf, opts)
    vg, pg = stp.get_state()
    stp.close()
    system.close()
    assert last['status'] == 0
    mnorm = lambda x: np.sqrt((x.T @ (M @ x)).item())
    assert mnorm(vg - v) <= VTOL*mnorm(v)
    assert np.linalg.norm(pg + pt/dt) <= \
        PTOL*np.linalg.norm(pt/dt)


def test_run_reports_unconverged_steps(gtiu, toy_prob):
    """a run in which solves end at `maxiter` must not come back
    as a clean trajectory -- `run()` raises like `step()`, the record says how
    many steps and which one first"""
    from dolfin_navier_scipy_amd import saddle, _capi
    M, A, J = (toy_prob['smc'][k] for k in 'MAJ')
    NP, NV = J.shape
    dt = 5e-3
    rng = np.random.default_rng(3)
    system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
    system.setup_precond(cheb_degree=2, schur='jacobi')
    stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
    stp.set_state(rng.standard_normal(NV))
    stp.set_rhs(M @ rng.standard_normal(NV), 1e-2*rng.standard_normal(NP))
    cf = saddle.ImexStepper.coeffs(a_c=1., pscale=-1./dt, extrapolate=0)
    opts = saddle.solve_opts(rtol=1e-13, maxiter=3, restart=3, use_graph=True,
                             reorth=2)
    with pytest.raises(_capi.NotConverged):
        stp.run(6, cf, opts)
    assert stp.last_run['unconverged'] >= 1 and stp.last_run['first_bad'] == 0
    _, _, last = stp.run(2, cf, opts, raise_on_fail=False)
    assert last['status'] == _capi.DNS_NOT_CONVERGED
    stp.close()
    system.close()


def test_nan_state_is_a_breakdown_not_a_converged_step(gtiu, toy_prob):
    """a blown-up state (NaN residual norm) must end the run with an error:
    `!(norm > tol)` is true for a NaN, so a careless convergence test would
    report such steps as converged in zero Krylov steps"""
    from dolfin_navier_scipy_amd import saddle, _capi
    M, A, J = (toy_prob['smc'][k] for k in 'MAJ')
    NP, NV = J.shape
    dt = 5e-3
    rng = np.random.default_rng(5)
    system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
    system.setup_precond(cheb_degree=4, schur='dense', factorization='full')
    stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
    v0 = rng.standard_normal(NV)
    v0[NV//2] = np.nan
    stp.set_state(v0)
    stp.set_rhs(M @ rng.standard_normal(NV), 1e-2*rng.standard_normal(NP))
    cf = saddle.ImexStepper.coeffs(a_c=1., pscale=-1./dt, extrapolate=4)
    for graph in (False, True):
        opts = saddle.solve_opts(rtol=1e-10, maxiter=50, use_graph=graph,
                                 reorth=2)
        with pytest.raises(_capi.DnsError):
            stp.run(12, cf, opts)
    # a plain solve with a NaN right-hand side likewise
    b = M @ rng.standard_normal(NV)
    b[3] = np.nan
    with pytest.raises(_capi.DnsError):
        system.solve(b, rtol=1e-10)
    stp.close()
    system.close()
