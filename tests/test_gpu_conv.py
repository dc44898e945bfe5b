"""Device convection against the host assembly and inside
the CNAB loop against the oracle with the host callback."""
import numpy as np
import pytest

import scenarios
from oracle import imex_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def conv_setup(toy_prob):
    from dolfin_navier_scipy_amd import convection, _capi
    assert _capi.device_count() > 0
    th = toy_prob['th']
    cv = convection.ConvectionP2.from_taylor_hood(
        th, toy_prob['invinds'], toy_prob['dbcinds'], toy_prob['dbcvals'])
    yield cv, toy_prob
    cv.close()


def _full(prob, v):
    full = np.zeros((prob['th'].vdim, 1))
    full[prob['invinds']] = np.asarray(v).reshape((-1, 1))
    full[prob['dbcinds'], 0] = prob['dbcvals']
    return full


def test_convection_vector_matches_host_assembly(conv_setup):
    cv, prob = conv_setup
    th, inv = prob['th'], prob['invinds']
    rng = np.random.default_rng(0)
    for k in range(3):
        v = rng.standard_normal((inv.size, 1))
        ref = th.convection_vec(_full(prob, v))[inv, :]
        got = cv.apply(v, scale=1.0)
        assert np.abs(got - ref).max() <= 1e-13*np.abs(ref).max()
        got = cv.apply(v, scale=-1.0)
        assert np.abs(got + ref).max() <= 1e-13*np.abs(ref).max()
    # N(u)u is quadratic: N(2u)(2u) = 4 N(u)u for homogeneous data
    zero = 0*prob['dbcvals']
    cv.set_dbcvals(zero)
    a = cv.apply(v)
    b = cv.apply(2*v)
    assert np.abs(b - 4*a).max() <= 1e-12*np.abs(b).max()
    cv.set_dbcvals(prob['dbcvals'])


def test_full_size_convection_and_bad_maps(conv_setup):
    from dolfin_navier_scipy_amd import convection, _capi
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=100)
    th, inv = femp['V'], femp['invinds']
    cv = convection.ConvectionP2.from_taylor_hood(th, inv, femp['dbcinds'],
                                                  femp['dbcvals'])
    v = np.random.default_rng(1).standard_normal((inv.size, 1))
    full = np.zeros((th.vdim, 1))
    full[inv] = v
    full[femp['dbcinds'], 0] = femp['dbcvals']
    ref = th.convection_vec(full)[inv, :]
    assert np.abs(cv.apply(v) - ref).max() <= 1e-13*np.abs(ref).max()
    # the one-lane-per-cell kernel of the streaming regime (DNS_CONV_LANE_MIN,
    # read when the operator is created): the same values, bit for bit
    import os
    got8 = cv.apply(v)
    old = os.environ.get('DNS_CONV_LANE_MIN')
    os.environ['DNS_CONV_LANE_MIN'] = '1'
    try:
        cvl = convection.ConvectionP2.from_taylor_hood(
            th, inv, femp['dbcinds'], femp['dbcvals'])
    finally:
        if old is None:
            del os.environ['DNS_CONV_LANE_MIN']
        else:
            os.environ['DNS_CONV_LANE_MIN'] = old
    got1 = cvl.apply(v)
    assert np.array_equal(got1, got8)
    cvl.close()
    cv.close()
    # a dof that is neither inner nor Dirichlet is rejected on the host
    with pytest.raises(_capi.DnsError):
        convection.ConvectionP2(th._vdofs(), th.glam, th.area, th.vdim,
                                inv[:-1], femp['dbcinds'], femp['dbcvals'])


def test_cnab_with_device_convection_matches_oracle(conv_setup):
    """no per-step host convection: `f_vdp` is only used by the Heun start"""
    from dolfin_navier_scipy_amd import time_int_utils as gtiu
    cv, prob = conv_setup
    kw, rec, _ = scenarios.build(variant='plain', seed=0, Nts=10, tE=0.05,
                                 prob=prob)
    calls = []
    host_fvdp = kw.pop('f_vdp')

    def counting_fvdp(vfull):
        calls.append(1)
        return host_fvdp(vfull)
    v, p, ff = gtiu.cnab(f_vdp=counting_fvdp, device_convection=cv,
                         invinds=prob['invinds'], **kw)
    assert len(calls) == 3          # tiu:379,456,475 -- the Heun start only
    kw2, rec2, _ = scenarios.build(variant='plain', seed=0, Nts=10, tE=0.05,
                                   prob=prob)
    vo, po, ffo = imex_oracle.cnab(**kw2)
    times, vels, prss = rec.arrays()
    t2, v2, p2 = rec2.arrays()
    for k in range(times.size):
        assert np.linalg.norm(vels[k] - v2[k]) <= 1e-8*np.linalg.norm(v2[k])
    assert np.linalg.norm(p - po) <= 1e-8*np.linalg.norm(po)


def test_resident_run_with_convection_matches_stepwise(conv_setup):
    """`dns_imex_run` (pipelined graphs, convection history advancing on the
    device) == the same steps taken one by one with host convection"""
    from dolfin_navier_scipy_amd import saddle
    cv, prob = conv_setup
    th, inv = prob['th'], prob['invinds']
    M, A, J = (prob['smc'][k] for k in 'MAJ')
    dt = 2e-3
    rng = np.random.default_rng(2)
    v0 = 0.1*rng.standard_normal((inv.size, 1))
    v0 += _full(prob, 0*v0)[inv]*0
    fv, fp = prob['rhsd']['fv'], prob['rhsd']['fp']
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt)
    opts = saddle.solve_opts(rtol=1e-12, use_graph=True, reorth=True)
    n0 = -th.convection_vec(_full(prob, v0))[inv, :]

    def make():
        system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
        system.setup_precond(cheb_degree=4, schur='dense')
        st = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
        st.set_state(v0, nfc_c=n0)
        st.set_rhs(dt*fv, fp)
        return system, st
    nsteps = 40
    sys_a, st_a = make()
    st_a.set_convection(cv, scale=-1.0)
    st_a.run(nsteps, cf, opts)
    va, pa = st_a.get_state()
    sys_b, st_b = make()
    vb = v0
    for k in range(nsteps):
        nfc = -th.convection_vec(_full(prob, vb))[inv, :]
        st_b.step(cf, nfc_new=nfc, opts=opts)
        vb, pb = st_b.get_state()
    assert np.linalg.norm(va - vb) <= 1e-9*np.linalg.norm(vb)
    assert np.linalg.norm(pa - pb) <= 1e-8*np.linalg.norm(pb)
    for obj in (st_a, st_b, sys_a, sys_b):
        obj.close()
