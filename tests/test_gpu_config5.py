"""BASELINE config 5: double rotating cylinder with Robin-penalised boundary
control (`/root/reference/tests/time_dep_nse_double_rotcyl_bcrob.py:37-71`):
mesh `2D-double-rotcyl_lvl1`, Re=60, `A += Arob/alpha` with `alpha = 1e-5`,
`fvtd(t) = sin(2 pi t / tE) (B_1 + B_2) / alpha`, explicit nonlinearity (CNAB),
`tE = 15`, `Nts = 6000`, Stokes start.  The first 300 steps and the config's
FULL horizon of 6000 steps through the product `solve_nse` (device resident:
rhs table for `fvtd`, device convection) against the CPU oracle on the same
matrices (factor-once CNAB loop with the host convection, tiu:104-143)."""
import numpy as np
import pytest

from oracle import snu_oracle as so

pytestmark = pytest.mark.gpu

NSTEPS = 300


@pytest.fixture(scope='module')
def rotcyl():
    from dolfin_navier_scipy_amd.fem import get_sysmats
    from dolfin_navier_scipy_amd import _capi
    assert _capi.device_count() > 0, 'HIP device required for -m gpu tests'
    femp, sm, rhsd = get_sysmats(problem='gen_bccont', Re=60, bccontrol=True)
    return femp, sm, rhsd


def test_geometry_and_robin_matrices(rotcyl):
    femp, sm, rhsd = rotcyl
    parts = femp['parts']
    # the boundary parts of `2D-double-rotcyl_lvl1_facet_region.xml.gz`
    # (physical entities 1 / 2 / 3+4 / 5 / 6): 20 / 30 / 70 / 28 / 28 edges
    assert [len(parts[k]) for k in ('inflow', 'outflow', 'walls')] == \
        [20, 30, 70]
    assert len(parts[('control', 0)]) == len(parts[('control', 1)]) == 28
    NP, NV = sm['J'].shape
    assert (NV, NP) == (25620, 3291)
    Arob, Brob = sm['Arob'], sm['Brob']
    assert Brob.shape == (NV, 2)
    # sum of the boundary mass = 2 components x circumference of the two
    # 28-gons; the rotation shape function is tangential: no net force
    per = 2*28*2*0.5*np.sin(np.pi/28)
    assert abs(Arob.sum() - 2*per) <= 1e-10*per
    assert np.abs(Brob.sum(axis=0)).max() <= 1e-10
    # g . Brob = int |g|^2 ds = r^2 * perimeter per cylinder (g nodal, P2)
    assert (Arob - Arob.T).nnz == 0 or abs(Arob - Arob.T).max() < 1e-14


def _config5_kwargs(rotcyl, nsteps):
    femp, sm, rhsd = rotcyl
    palpha, tE, Nts = 1e-5, 15., 6000
    A = (sm['A'] + 1./palpha*sm['Arob']).tocsr()
    Brob = 1./palpha*sm['Brob']

    def fv_tmdp(time=0, **kw):
        return np.sin(time/tE*2*np.pi)*(Brob[:, :1] + Brob[:, 1:])
    trange = np.linspace(0., tE, Nts + 1)[:nsteps + 1]
    return dict(A=A, M=sm['M'], J=sm['J'], fv=rhsd['fv'], fp=rhsd['fp'],
                fvtd=fv_tmdp, trange=trange, V=femp['V'],
                invinds=femp['invinds'], dbcinds=femp['dbcinds'].tolist(),
                dbcvals=femp['dbcvals'].tolist())


def test_config5_first_300_steps(rotcyl):
    from dolfin_navier_scipy_amd import stokes_navier_utils as snu
    femp, sm, rhsd = rotcyl
    kw = _config5_kwargs(rotcyl, NSTEPS)
    trange = kw['trange']
    vo, po, ffo = so.solve_nse(**kw)
    (vg, pg), ffg = snu.solve_nse(
        start_ssstokes=True, return_final_vp=True, check_ff=True,
        datatrange=[trange[0], trange[-1]],
        krplsprms=dict(cheb_degree=12), **kw)
    snu.clear_cache()
    assert ffg == ffo == 0
    M = sm['M']
    mn = lambda x: float(np.sqrt((x.T @ (M @ x)).item()))
    ev, ep = mn(vg - vo)/mn(vo), np.linalg.norm(pg - po)/np.linalg.norm(po)
    print('config 5, {0} steps: v'.format(NSTEPS), ev, 'p', ep)
    assert ev <= 1e-8, ev
    assert ep <= 1e-8, ep
    # the control acts: the run without it differs
    v0, _, _ = so.solve_nse(**dict(kw, fvtd=None))
    assert mn(vo - v0) > 1e-3*mn(v0)


class _Every(object):
    """`savevp` that keeps every `stride`-th call (6000 full vectors of 26k
    doubles each would be 1.3 GB)"""

    def __init__(self, stride, inv):
        self.stride, self.inv, self.k, self.v, self.p = stride, inv, 0, {}, {}

    def __call__(self, v, p, time=None):
        if self.k % self.stride == 0:
            self.v[time] = np.array(v)[self.inv].reshape((-1, 1))
            self.p[time] = np.array(p).reshape((-1, 1))
        self.k += 1


def test_config5_full_horizon_6000_steps(rotcyl):
    """ALL 6000 steps of the config (tests/time_dep_nse_double_rotcyl_bcrob.py
    :65-71) device resident, velocity AND pressure within 1e-8 of
    the oracle's direct-solve trajectory at every 1000th step and at the end.
    The per-step error of an inexact solve does not add up: the residual of
    every solve is carried into the next right-hand side
    (`dns_imex_coeffs.carry_residual`)."""
    from dolfin_navier_scipy_amd import stokes_navier_utils as snu
    femp, sm, rhsd = rotcyl
    kw = _config5_kwargs(rotcyl, 6000)
    trange = kw['trange']
    rec = _Every(1000, femp['invinds'])
    vo, po, ffo = so.solve_nse(savevp=rec, **kw)
    marks = [trange[k] for k in range(0, 6001, 1000)]
    (vd, pd), ffg = snu.solve_nse(
        start_ssstokes=True, return_dictofvelstrs=True,
        return_dictofpstrs=True, check_ff=True, datatrange=list(marks),
        krplsprms=dict(cheb_degree=12), **kw)
    snu.clear_cache()
    assert ffg == ffo == 0
    M = sm['M']
    mn = lambda x: float(np.sqrt((x.T @ (M @ x)).item()))
    inv = femp['invinds']
    worst_v = worst_p = 0.
    for t in marks[1:]:
        vg, pg = vd[t][inv].reshape((-1, 1)), pd[t].reshape((-1, 1))
        ev = mn(vg - rec.v[t])/mn(rec.v[t])
        ep = np.linalg.norm(pg - rec.p[t])/np.linalg.norm(rec.p[t])
        print('config 5, t = {0:.2f}: v {1:.2e} p {2:.2e}'.format(t, ev, ep))
        worst_v, worst_p = max(worst_v, ev), max(worst_p, ep)
    assert worst_v <= 1e-8, worst_v
    assert worst_p <= 1e-8, worst_p
