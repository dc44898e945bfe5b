"""Pin the CPU oracle against the golden vectors recorded from the reference.

The fixtures `tests/golden/imex_*.npz` hold inputs/outputs of the reference's
own `time_int_utils.cnab/sbdftwo/semi_implicit_euler` (see
`tests/golden/make_golden.py`).  The oracle restatement must reproduce them,
and they must satisfy the algebraic step residuals that the reference's
`tests/test_units_residuals.py:93-95,106-109,121-124` asserts.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sps

import scenarios
from oracle import imex_oracle, saddle_oracle

SCHEMES = ('cnab', 'sbdf2')


def load_system(golden_dir):
    dat = np.load(os.path.join(golden_dir, 'imex_toy_system.npz'))
    mats = {}
    for name in 'MAJ':
        mats[name] = sps.csr_matrix(
            (dat[name+'_data'], dat[name+'_indices'], dat[name+'_indptr']),
            shape=tuple(dat[name+'_shape']))
    return mats


def test_fixture_matrices_equal_assembler(golden_dir, toy_prob):
    mats = load_system(golden_dir)
    for name in 'MAJ':
        assert abs(mats[name] - toy_prob['smc'][name]).max() < 1e-15


@pytest.mark.parametrize('scheme', SCHEMES)
@pytest.mark.parametrize('seed,variant', list(enumerate(scenarios.VARIANTS)))
def test_oracle_reproduces_reference(golden_dir, toy_prob, scheme, seed,
                                     variant):
    gold = np.load(os.path.join(
        golden_dir, 'imex_{0}_{1}_s{2}.npz'.format(scheme, variant, seed)))
    kw, rec, _ = scenarios.build(variant=variant, seed=seed, prob=toy_prob)
    assert np.array_equal(kw['inivel'], gold['inivel'])
    if scheme == 'sbdf2':
        kw.pop('f_tvdp', None)
        v, p, ff = imex_oracle.sbdftwo(**kw)
    else:
        v, p, ff = imex_oracle.cnab(**kw)
    times, vels, prss = rec.arrays()
    assert ff == int(gold['ffflag'])
    assert np.allclose(times, gold['times'], rtol=0, atol=1e-15)
    scale_v = np.abs(gold['vels']).max()
    scale_p = np.abs(gold['prss']).max()
    assert np.abs(vels - gold['vels']).max() <= 1e-10*scale_v
    assert np.abs(prss - gold['prss']).max() <= 1e-8*scale_p
    assert np.abs(v - gold['vfinal']).max() <= 1e-10*scale_v


def test_oracle_semi_implicit_euler(golden_dir, toy_prob):
    gold = np.load(os.path.join(golden_dir, 'imex_sie_plain_s3.npz'))
    kw, rec, aux = scenarios.build(variant='plain', seed=3, prob=toy_prob)
    fvdp, cfv, appnd = kw['f_vdp'], aux['cfv'], kw['appndbcs']

    def rhsv(t, vvec):
        return cfv + fvdp(appnd(vvec.reshape((-1, 1)), []))
    vlist = imex_oracle.semi_implicit_euler(
        iniv=kw['inivel'], jmat=kw['J'], mmat=kw['M'], amat=kw['A'],
        rhsv=rhsv, trange=kw['trange'], data_trange=gold['data_trange'],
        fp=aux['cfp'])
    got = np.array([np.asarray(v).reshape(-1) for v in vlist])
    assert got.shape == gold['vlist'].shape
    assert np.abs(got - gold['vlist']).max() <= 1e-10*np.abs(gold['vlist']).max()


@pytest.mark.parametrize('variant,seed', [('plain', 0)])
def test_golden_ab2_step_residual(golden_dir, variant, seed):
    """algebraic AB2 residual of reference tests/test_units_residuals.py:121-124
    evaluated on the golden CNAB trajectory (plain variant: static BCs)"""
    gold = np.load(os.path.join(
        golden_dir, 'imex_cnab_{0}_s{1}.npz'.format(variant, seed)))
    mats = load_system(golden_dir)
    M, A, J = mats['M'], mats['A'], mats['J']
    NP, NV = J.shape
    times = gold['times']
    dt = times[1] - times[0]
    fed = gold['fvdp_fed']      # [N(v0), N(tv1), N(v1), N(v1) again, N(v2)...]
    # the integrator evaluates f_vdp at: v0, pred v1, v1 (end of heun),
    # then at every step start: v1, v2, ...
    nfc = {0: fed[0], 1: fed[3], 2: fed[4]}
    # recover inner velocities from the stored full vectors via the nan-free
    # dofs: inner = those that change in time or were never Dirichlet
    kw, _, aux = scenarios.build(variant='plain', seed=seed)
    inv = aux['prob']['invinds']
    v1, v2 = gold['vels'][1][inv], gold['vels'][2][inv]
    p2 = gold['prss'][2]
    cfv = gold['cfv'].reshape(-1)
    res = M @ v2 + .5*dt*(A @ v2) - dt*(J.T @ p2) \
        - (M @ v1 - .5*dt*(A @ v1) + .5*dt*(3*nfc[1] - nfc[0]) + dt*cfv)
    assert np.allclose(res, 0., atol=1e-8)
    assert np.allclose(J @ v2 - gold['cfp'].reshape(-1), 0., atol=1e-10)


def test_saddle_oracle_residual_and_smw():
    rng = np.random.default_rng(5)
    prob = scenarios.toy_problem(nx=11, ny=4)
    M, A, J = (prob['smc'][k] for k in 'MAJ')
    NP, NV = J.shape
    rhsv = rng.standard_normal((NV, 2))
    rhsp = rng.standard_normal((NP, 2))
    amat = M + 0.01*A
    vp = saddle_oracle.solve_sadpnt_smw(amat=amat, jmat=J, jmatT=J.T,
                                        rhsv=rhsv, rhsp=rhsp)
    K = saddle_oracle.saddle_matrix(amat, J)
    rhs = np.vstack([rhsv, rhsp])
    assert np.linalg.norm(K @ vp - rhs) <= 1e-11*np.linalg.norm(rhs)
    U = 1e-3*rng.standard_normal((NV, 3))
    V = rng.standard_normal((3, NV))
    vp2 = saddle_oracle.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv,
                                         rhsp=rhsp, umat=U, vmat=V)
    Kd = K.toarray()
    Kd[:NV, :NV] -= U @ V
    assert np.linalg.norm(Kd @ vp2 - rhs) <= 1e-10*np.linalg.norm(rhs)
    # projector: J Pi = 0, Pi^T J^T = 0
    f = rng.standard_normal((NV, 1))
    pf = saddle_oracle.app_prj_via_sadpnt(amat=M, jmat=J, rhsv=f)
    assert np.linalg.norm(J @ pf) <= 1e-10*np.linalg.norm(f)
    ptf = saddle_oracle.app_prj_via_sadpnt(amat=M, jmat=J, rhsv=f,
                                           transposedprj=True)
    # <Pi x, y> = <x, Pi^T y>
    g = rng.standard_normal((NV, 1))
    pg = saddle_oracle.app_prj_via_sadpnt(amat=M, jmat=J, rhsv=g)
    assert abs((pg.T @ f - g.T @ ptf).item()) <= 1e-9*np.linalg.norm(f)


def load_lau_calls(golden_dir):
    """the reference integrators' own calls to the boundary, recorded while
    `make_golden.py` ran them: list of `(tag, amat, rhsv, rhsp, vp)`"""
    dat = np.load(os.path.join(golden_dir, 'lau_calls.npz'))
    J = load_system(golden_dir)['J']
    NP, NV = J.shape
    calls = []
    for k in range(int(dat['ncalls'])):
        amat = sps.csr_matrix((dat['amat_data_%d' % k],
                               dat['amat_indices_%d' % k],
                               dat['amat_indptr_%d' % k]), shape=(NV, NV))
        calls.append((str(dat['tag_%d' % k]), amat, dat['rhsv_%d' % k],
                      dat['rhsp_%d' % k], dat['vp_%d' % k]))
    return J, calls


def test_recorded_boundary_calls_are_saddle_solutions(golden_dir):
    """every `(amat, jmat, rhsv, rhsp) -> vp` pair the reference's `tiu` sent
    through `lau` (tiu:402,466,605) satisfies the saddle-point system"""
    J, calls = load_lau_calls(golden_dir)
    assert len(calls) == 13
    NP, NV = J.shape
    for tag, amat, rhsv, rhsp, vp in calls:
        K = saddle_oracle.saddle_matrix(amat, J).tocsr()
        b = np.vstack([rhsv.reshape((NV, -1)), rhsp.reshape((NP, -1))])
        assert np.linalg.norm(K @ vp - b) <= 1e-11*np.linalg.norm(b), tag
        again = saddle_oracle.solve_sadpnt_smw(amat=amat, jmat=J, rhsv=rhsv,
                                               rhsp=rhsp)
        assert np.array_equal(again, vp), tag
