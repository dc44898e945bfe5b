"""Device-resident Newton/Picard trapezoidal sweeps (`dns_trap_*`,
`dns_conv_assemble`) against the host assembler and the oracle restatement of
snu:1016-1047 / snu:1402-1566."""
import numpy as np
import pytest
import scipy.sparse as sps

import scenarios
from oracle import imex_oracle
from oracle import newton_picard_oracle as npo

pytestmark = pytest.mark.gpu


def _setup(prob, Nts=6, tE=0.03):
    th, smc, rhsd = prob['th'], prob['smc'], prob['rhsd']
    inv, dbcinds, dbcvals = prob['invinds'], prob['dbcinds'], prob['dbcvals']
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
        # algebraic `get_v_conv_conts` (snu:109-133), as in test_newton_picard
        if vfull.shape[0] == NV:
            vfull = appnd(vfull)
        N1, N2, fv3 = th.convection_mats(vfull, keep_pattern=True)
        Nm = N1 if picard else (N1 + N2)
        Nc = Nm[inv, :][:, inv].tocsr()
        rhsbc = -(Nm @ bcsv)[inv, :]
        return Nc, (0.*fv3[inv, :] if picard else fv3[inv, :]), rhsbc

    kw, rec, aux = scenarios.build(variant='plain', seed=0, Nts=Nts, tE=tE,
                                   prob=prob)
    imex_oracle.cnab(**kw)          # semi-explicit run = first lin. points
    times, vels, _ = rec.arrays()
    lin0 = {t: vels[k].reshape((-1, 1))[inv] for k, t in enumerate(times)}
    return dict(M=M, A=A, J=J, fv=rhsd['fv'], fp=rhsd['fp'], conv=conv,
                appnd=appnd, trange=kw['trange'], iniv=kw['inivel'], lin0=lin0,
                inv=inv, NV=NV, NP=NP, th=th, dbcinds=dbcinds, dbcvals=dbcvals)


@pytest.fixture(scope='module')
def setup(toy_prob):
    from dolfin_navier_scipy_amd import _capi
    assert _capi.device_count() > 0
    return _setup(toy_prob)


@pytest.fixture()
def cvop(setup):
    from dolfin_navier_scipy_amd import convection
    s = setup
    cv = convection.ConvectionP2.from_taylor_hood(s['th'], s['inv'],
                                                  s['dbcinds'], s['dbcvals'])
    yield cv
    cv.close()


@pytest.mark.parametrize('newton', [False, True])
def test_device_convection_matrices_match_host_assembly(setup, cvop, newton):
    from dolfin_navier_scipy_amd import newton_picard as dnp
    s = setup
    pattern = dnp.union_pattern(s['M'], s['A'], cvop.connectivity())
    cvop.bind_pattern(pattern)
    rng = np.random.default_rng(4)
    for _ in range(2):
        u = rng.standard_normal((s['NV'], 1))
        N, rhsbc, rhscon = cvop.assemble(u, newton=newton)
        Nref, rcref, rbref = s['conv'](u, not newton)
        scale = abs(Nref).max()
        assert abs(N - Nref).max() <= 1e-13*scale
        assert np.abs(rhsbc - rbref).max() <= 1e-13*max(np.abs(rbref).max(), 1)
        # rhscon is N(u)u whatever the linearisation
        _, fv3, _ = s['conv'](u, False)
        assert np.abs(rhscon - fv3).max() <= 1e-13*np.abs(fv3).max()


def test_conv_equals_N1u_equals_N2u(setup, cvop):
    """the reference's own known-answer test for its convection matrices,
    tests/test_units_fenicsci.py:84-85: `N(u)u == N1(u) u == N2(u) u` -- on the
    full space, i.e. with homogeneous Dirichlet data after condensation"""
    from dolfin_navier_scipy_amd import newton_picard as dnp
    s = setup
    cvop.set_dbcvals(0*s['dbcvals'])
    cvop.bind_pattern(dnp.union_pattern(cvop.connectivity()))
    u = np.random.default_rng(9).standard_normal((s['NV'], 1))
    N1, _, nuu = cvop.assemble(u, newton=False)
    N12, _, _ = cvop.assemble(u, newton=True)
    N2 = N12 - N1
    ref = np.abs(nuu).max()
    assert np.abs(N1 @ u - nuu).max() <= 1e-12*ref
    assert np.abs(N2 @ u - nuu).max() <= 1e-12*ref
    cvop.set_dbcvals(s['dbcvals'])


def test_bind_pattern_rejects_a_pattern_that_is_too_small(setup, cvop):
    from dolfin_navier_scipy_amd import _capi
    s = setup
    with pytest.raises(_capi.DnsError):
        cvop.bind_pattern(sps.identity(s['NV'], format='csr'))
    with pytest.raises(_capi.DnsError):
        cvop.bind_pattern(sps.identity(s['NV'] + 1, format='csr'))


@pytest.mark.parametrize('picard', [True, False])
def test_device_sweep_matches_oracle(setup, cvop, picard):
    from dolfin_navier_scipy_amd import newton_picard as dnp, saddle
    s = setup
    tr = s['trange']
    lin_full = {t: s['appnd'](v) for t, v in s['lin0'].items()}
    ref_v, ref_p, ref_upd = npo.trapezoidal_sweep(
        tr, s['iniv'], M=s['M'], A=s['A'], J=s['J'], fv=s['fv'], fp=s['fp'],
        conv=s['conv'], appndbcs=s['appnd'], linpoints=s['lin0'],
        picard=picard)
    stp = dnp.TrapezoidalStepper(s['M'], s['A'], s['J'], cvop, nslots=tr.size,
                                 dt=tr[1] - tr[0], precond=dict(cheb_degree=4))
    stp.set_rhs(s['fv'], s['fp'])
    for k, t in enumerate(tr):
        stp.write_linpoint(0, k, s['lin0'][t])
    opts = saddle.solve_opts(rtol=1e-12, maxiter=400, use_graph=True)
    got_v, got_p, upd, st = stp.sweep(tr, s['iniv'], 0, picard, opts=opts)
    for t in tr[1:]:
        ev = np.linalg.norm(got_v[t] - ref_v[t])/np.linalg.norm(ref_v[t])
        ep = np.linalg.norm(got_p[t] - ref_p[t])/np.linalg.norm(ref_p[t])
        assert ev <= 1e-8 and ep <= 1e-8, (t, ev, ep)
    assert abs(upd - ref_upd) <= 1e-6*abs(ref_upd) + 1e-18
    # the sweep's velocities sit in the other trajectory buffer
    assert np.allclose(stp.read_traj(1, tr.size - 1), got_v[tr[-1]])
    stp.close()


@pytest.mark.parametrize('picard', [True, False])
def test_library_time_loop_equals_the_per_step_calls(toy_prob, picard):
    """`dns_trap_run` (the sweep's time loop inside the library; the update
    norm of a step rides in the element launch of the next one, dns::UpdJob)
    against one `dns_trap_step` call per step: same update norm, same
    trajectory -- read slot by slot, the LAST slot first (its update-norm job
    is still pending when the batch ends)"""
    from dolfin_navier_scipy_amd import convection, newton_picard as dnp, saddle
    s = _setup(toy_prob, Nts=40, tE=0.1)
    tr = s['trange']
    cv = convection.ConvectionP2.from_taylor_hood(s['th'], s['inv'],
                                                  s['dbcinds'], s['dbcvals'])
    stp = dnp.TrapezoidalStepper(s['M'], s['A'], s['J'], cv, nslots=tr.size,
                                 dt=tr[1] - tr[0], precond=dict(cheb_degree=4))
    stp.set_rhs(s['fv'], s['fp'])
    for k, t in enumerate(tr):
        stp.write_linpoint(0, k, s['lin0'][t])
    opts = saddle.solve_opts(rtol=1e-12, maxiter=400, use_graph=True)
    rec_v, _, upd_rec, st_rec = stp.sweep(tr, s['iniv'], 0, picard, opts=opts,
                                          record=True)
    assert st_rec['cycle'] is not None          # (pipelined batches)
    traj_rec = [stp.read_traj(1, k) for k in range(1, tr.size)]
    for k, t in enumerate(tr[1:], start=1):
        assert np.array_equal(traj_rec[k - 1], rec_v[t])
    # poison the output buffer, then the library's loop
    for k in range(tr.size):
        stp.write_linpoint(1, k, 0*s['iniv'] + 7.0)
    _, _, upd_run, st_run = stp.sweep(tr, s['iniv'], 0, picard, opts=opts,
                                      record=False)
    last = stp.read_traj(1, tr.size - 1)
    scale = np.abs(traj_rec[-1]).max()
    assert np.abs(last - traj_rec[-1]).max() <= 1e-9*scale
    for k in range(1, tr.size):
        assert np.abs(stp.read_traj(1, k) - traj_rec[k - 1]).max() <= \
            1e-9*scale, k
    assert abs(upd_run - upd_rec) <= 1e-8*abs(upd_rec) + 1e-20
    assert st_run['iters'] > 0
    # a slot read while its step's update-norm job is still pending (no poll
    # in between): the read launches the job first
    stp.write_linpoint(1, 7, 0*s['iniv'] + 7.0)
    stp.set_pipeline(3)
    stp.run(tr[1] - tr[0], 0, 5, 3, not picard, opts=opts)
    got = stp.read_traj(1, 7)
    stp.poll()
    stp.set_pipeline(0)
    assert np.array_equal(got, stp.state()[0])
    stp.close()
    cv.close()


def test_async_trajectory_writer(setup, cvop, tmp_path):
    """the trajectory of a sweep travels to the host (and to
    `.npy` files, what `dou.save_npa` writes per step in the reference,
    snu:1012-1014) on a copy stream of its own while the NEXT sweep runs; the
    solver's stream only waits before it overwrites the exported buffer"""
    from dolfin_navier_scipy_amd import newton_picard as dnp, saddle
    s = setup
    tr = s['trange']
    stp = dnp.TrapezoidalStepper(s['M'], s['A'], s['J'], cvop, nslots=tr.size,
                                 dt=tr[1] - tr[0], precond=dict(cheb_degree=4))
    stp.set_rhs(s['fv'], s['fp'])
    for k, t in enumerate(tr):
        stp.write_linpoint(0, k, s['lin0'][t])
    opts = saddle.solve_opts(rtol=1e-12, maxiter=400, use_graph=True)
    v1, _, _, _ = stp.sweep(tr, s['iniv'], 0, True, opts=opts)   # -> traj[1]
    names = {t: str(tmp_path / 'v_{0:.6f}.npy'.format(t)) for t in tr}
    writer = stp.save_trajectory_async(1, tr[1:], lambda t: names[t], slot0=1)
    host = stp.export_async(1)                     # a second export, in memory
    # the next sweep linearises about traj[1] and writes traj[0]: it runs while
    # the exports are in flight; the one after it writes traj[1] and has to
    # wait for them
    v2, _, _, _ = stp.sweep(tr, s['iniv'], 1, False, opts=opts)
    v3, _, _, _ = stp.sweep(tr, s['iniv'], 0, False, opts=opts)
    writer.join()
    stp.export_wait()
    for k, t in enumerate(tr):
        if k == 0:
            continue
        assert np.array_equal(np.load(names[t]), v1[t])
        assert np.array_equal(host[k].reshape((-1, 1)), v1[t])
    # ... and the sweeps were not disturbed
    assert np.allclose(stp.read_traj(1, tr.size - 1), v3[tr[-1]])
    assert not np.array_equal(v3[tr[-1]], v1[tr[-1]])
    stp.close()


def test_conv_kat_at_full_size():
    """`N(u)u == N1(u) u == N2(u) u` (tests/test_units_fenicsci.py:84-85) and
    the device convection vector against the host assembly on the cylinder
    mesh of BASELINE config 2 (N=2, NV=9356), not only on the toy mesh"""
    from dolfin_navier_scipy_amd import convection, newton_picard as dnp
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=100.)
    th, inv = femp['V'], femp['invinds']
    NV = inv.size
    cv = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'])
    rng = np.random.default_rng(21)
    u = rng.standard_normal((NV, 1))
    # vector against the host assembly (with the inflow data)
    full = np.zeros((th.vdim, 1))
    full[inv] = u
    full[femp['dbcinds'], 0] = femp['dbcvals']
    ref = th.convection_vec(full)[inv, :]
    got = cv.apply(u, scale=1.0)
    assert np.abs(got - ref).max() <= 1e-12*np.abs(ref).max()
    # the reference's KAT (homogeneous data: the identity holds on the full
    # space only)
    cv.set_dbcvals(0*np.asarray(femp['dbcvals'], dtype=float))
    cv.bind_pattern(dnp.union_pattern(cv.connectivity()))
    N1, _, nuu = cv.assemble(u, newton=False)
    N12, _, _ = cv.assemble(u, newton=True)
    scale = np.abs(nuu).max()
    assert np.abs(N1 @ u - nuu).max() <= 1e-12*scale
    assert np.abs((N12 - N1) @ u - nuu).max() <= 1e-12*scale
    cv.close()


def test_conv_split_into_inner_and_boundary_parts():
    """The reference's second known-answer test of the convection matrices
    (`tests/test_units_fenicsci.py:132-188`, `test_conv_asquad`): on the driven
    cavity (N = 15) with the field u = ((1-x)x(1-y)y x + 2, (1-x)x(1-y)y y + 1),
    split into its inner part u_i and its boundary part u_gamma,

        N(u)[inner, :] u  ==  H (u_i x u_i) + ((N1 + N2)(u_gamma) u_i)[inner]
                                + fv(u_gamma)[inner]

    with `N1, N2, fv = get_convmats(u0 = u_gamma)` -- the linearisation about a
    field that lives on the boundary only is exactly what `get_v_conv_conts`
    feeds the Newton/Picard systems with (snu:109-133).  Mirrored on the DEVICE
    assembly: the matrices by `dns_conv_assemble` about u_gamma, the vectors by
    the device convection kernel.  `H` (`ass_convmat_asmatquad`, the quadratic
    tensor of the reduced-order models: out of scope) enters
    the identity only through `H (u_i x u_i) = N(u_i) u_i`, which the device
    kernel evaluates directly."""
    from dolfin_navier_scipy_amd import convection, newton_picard as dnp
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='drivencavity', N=15, nu=1e-2)
    th, inv = femp['V'], femp['invinds']
    dbcinds = np.asarray(femp['dbcinds'])
    xy = th.nodecoords
    bub = (1 - xy[:, 0])*xy[:, 0]*(1 - xy[:, 1])*xy[:, 1]
    uvec = np.zeros((th.vdim, 1))
    uvec[0::2, 0] = bub*xy[:, 0] + 2
    uvec[1::2, 0] = bub*xy[:, 1] + 1
    u_i = uvec[inv]
    gam = uvec[dbcinds, 0]
    # (the split is a split: test_units_fenicsci.py:186-188)
    uvec_i, uvec_g = 0*uvec, uvec.copy()
    uvec_i[inv] = uvec[inv]
    uvec_g[inv] = 0
    assert np.linalg.norm(uvec - uvec_g - uvec_i) < 1e-14
    cv = convection.ConvectionP2.from_taylor_hood(th, inv, dbcinds, gam)
    cv.bind_pattern(dnp.union_pattern(cv.connectivity()))
    # N(u)[inner, :] u with u's own boundary values
    classicalconv = cv.apply(u_i, scale=1.0)
    assert np.abs(classicalconv - th.convection_vec(uvec)[inv]).max() <= \
        1e-12*np.abs(classicalconv).max()
    # (N1 + N2)(u_gamma) on the inner dofs and fv(u_gamma) = N(u_gamma) u_gamma
    N12, _, fv_gamma = cv.assemble(0*u_i, newton=True)
    # H (u_i x u_i) = N(u_i) u_i: homogeneous boundary values
    cv.set_dbcvals(0*gam)
    inner_inner = cv.apply(u_i, scale=1.0)
    quadconv = inner_inner + N12 @ u_i + fv_gamma.reshape((-1, 1))
    scale = np.abs(classicalconv).max()
    assert np.allclose(classicalconv, quadconv)          # (:184)
    assert np.abs(classicalconv - quadconv).max() <= 1e-12*scale
    cv.close()


def test_device_newton_picard_driver_matches_oracle(setup, cvop):
    from dolfin_navier_scipy_amd import newton_picard as dnp, saddle
    s = setup
    tr = s['trange']
    lin_full = {t: s['appnd'](v) for t, v in s['lin0'].items()}
    ref_v, ref_p, ref_hist = npo.newton_picard(
        tr, s['iniv'], lin_full, vel_pcrd_stps=1, vel_nwtn_stps=2,
        invinds=s['inv'], M=s['M'], A=s['A'], J=s['J'], fv=s['fv'],
        fp=s['fp'], conv=s['conv'], appndbcs=s['appnd'])
    stp = dnp.TrapezoidalStepper(s['M'], s['A'], s['J'], cvop, nslots=tr.size,
                                 dt=tr[1] - tr[0], precond=dict(cheb_degree=4))
    stp.set_rhs(s['fv'], s['fp'])
    opts = saddle.solve_opts(rtol=1e-12, maxiter=400, use_graph=True)
    got_v, got_p, hist = dnp.newton_picard(
        stp, tr, s['iniv'], s['lin0'], vel_pcrd_stps=1, vel_nwtn_stps=2,
        opts=opts)
    assert [h[0] for h in hist] == [h[0] for h in ref_hist]
    for (_, a), (_, b) in zip(hist, ref_hist):
        assert abs(a - b) <= 1e-5*abs(b) + 1e-16
    for t in tr[1:]:
        ev = np.linalg.norm(got_v[t] - ref_v[t])/np.linalg.norm(ref_v[t])
        ep = np.linalg.norm(got_p[t] - ref_p[t])/np.linalg.norm(ref_p[t])
        assert ev <= 1e-8 and ep <= 1e-8, (t, ev, ep)
    stp.close()


def test_device_time_sections_and_full_sweep_match_oracle(setup, cvop):
    """`nsects = 3`, `addfullsweep` (snu:1076-1090, 1576-1587) through the
    device stepper: same sweep sequence, update norms and iterates as the
    oracle's restatement"""
    from dolfin_navier_scipy_amd import newton_picard as dnp, saddle
    s = setup
    tr = s['trange']
    lin_full = {t: s['appnd'](v) for t, v in s['lin0'].items()}
    lin_full[None] = lin_full[tr[0]]
    kw = dict(vel_pcrd_stps=1, vel_nwtn_stps=2, nsects=3, loc_nwtn_tol=1e-13,
              addfullsweep=True)
    ref_v, ref_p, ref_hist = npo.newton_picard(
        tr, s['iniv'], lin_full, invinds=s['inv'], M=s['M'], A=s['A'],
        J=s['J'], fv=s['fv'], fp=s['fp'], conv=s['conv'],
        appndbcs=s['appnd'], **kw)
    stp = dnp.TrapezoidalStepper(s['M'], s['A'], s['J'], cvop, nslots=tr.size,
                                 dt=tr[1] - tr[0], precond=dict(cheb_degree=4))
    stp.set_rhs(s['fv'], s['fp'])
    opts = saddle.solve_opts(rtol=1e-12, maxiter=400, use_graph=True)
    lin0 = dict(s['lin0'])
    lin0[None] = s['lin0'][tr[0]]
    got_v, got_p, hist = dnp.newton_picard(stp, tr, s['iniv'], lin0,
                                           opts=opts, **kw)
    assert [h[0] for h in hist] == [h[0] for h in ref_hist]
    # one Picard sweep per section, none in the full sweep (the `elif` at
    # snu:1583 does not refill its Picard count); a section stops early once
    # its update norm is below `loc_nwtn_tol`
    assert [h[0] for h in hist].count('picard') == 3 and len(hist) >= 8
    for (_, a), (_, b) in zip(hist, ref_hist):
        assert abs(a - b) <= 1e-5*abs(b) + 1e-15
    for t in tr[1:]:
        ev = np.linalg.norm(got_v[t] - ref_v[t])/np.linalg.norm(ref_v[t])
        ep = np.linalg.norm(got_p[t] - ref_p[t])/np.linalg.norm(ref_p[t])
        assert ev <= 1e-8 and ep <= 1e-8, (t, ev, ep)
    stp.close()


def test_full_size_picard_and_newton_sweep_against_oracle():
    """cylinder wake N=2 (NV=9356, NP=1289), Re=100, dt=1/512: a Picard and a
    Newton sweep of 6 steps, device path vs the oracle's restatement with the
    host assembler (BASELINE config 3 at its real size)"""
    from dolfin_navier_scipy_amd import convection, saddle
    from dolfin_navier_scipy_amd import newton_picard as dnp
    from dolfin_navier_scipy_amd.fem import get_sysmats
    from oracle import saddle_oracle
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=100)
    th, inv = femp['V'], femp['invinds']
    dbcinds, dbcvals = femp['dbcinds'], femp['dbcvals']
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    vp0 = saddle_oracle.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'],
                                         rhsp=rhsd['fp'])
    iniv = vp0[:NV]

    def appnd(vvec):
        full = np.zeros((th.vdim, 1))
        full[inv] = vvec
        full[dbcinds, 0] = dbcvals
        return full

    bcsv = np.zeros((th.vdim, 1))
    bcsv[dbcinds, 0] = dbcvals

    def conv(vfull, picard):
        if vfull.shape[0] == NV:
            vfull = appnd(vfull)
        N1, N2, fv3 = th.convection_mats(vfull, keep_pattern=True)
        Nm = N1 if picard else (N1 + N2)
        return (Nm[inv, :][:, inv].tocsr(),
                0.*fv3[inv, :] if picard else fv3[inv, :],
                -(Nm @ bcsv)[inv, :])

    tr = np.arange(7)/512.
    lin0 = {t: iniv for t in tr}
    cv = convection.ConvectionP2.from_taylor_hood(th, inv, dbcinds, dbcvals)
    stp = dnp.TrapezoidalStepper(M, A, J, cv, nslots=tr.size, dt=tr[1] - tr[0])
    stp.set_rhs(rhsd['fv'], rhsd['fp'])
    opts = saddle.solve_opts(rtol=1e-12, maxiter=400, use_graph=True, reorth=2)
    mnorm = lambda x: np.sqrt((x.T @ (M @ x)).item())
    lin = lin0
    which = 0
    for k, t in enumerate(tr):
        stp.write_linpoint(which, k, lin[t])
    for picard in (True, False):
        ref_v, ref_p, ref_upd = npo.trapezoidal_sweep(
            tr, iniv, M=M, A=A, J=J, fv=rhsd['fv'], fp=rhsd['fp'], conv=conv,
            appndbcs=appnd, linpoints=lin, picard=picard)
        got_v, got_p, upd, st = stp.sweep(tr, iniv, which, picard, opts=opts)
        for t in tr[1:]:
            assert mnorm(got_v[t] - ref_v[t]) <= 1e-8*mnorm(ref_v[t]), t
            assert np.linalg.norm(got_p[t] - ref_p[t]) <= \
                1e-8*np.linalg.norm(ref_p[t]), t
        assert abs(upd - ref_upd) <= 1e-6*abs(ref_upd) + 1e-20
        lin = ref_v                      # next sweep: about the oracle's result
        which = 1 - which
        for k, t in enumerate(tr):
            stp.write_linpoint(which, k, lin[t])
    stp.close()
    cv.close()


# ---------------------------------------------------------------------------
# long horizon: developed vortex shedding, N = 2, Re = 100, dt = 1/512
# ---------------------------------------------------------------------------
HORIZON = 2048
SWEEP_RTOL = 1e-11       # (within 1e-8 of the oracle over the horizon, below)


@pytest.fixture(scope='module')
def shedding():
    return shedding_setup()


def shedding_setup(horizon=HORIZON):
    """cylinder wake N=2, Re=100: the Stokes state advanced 4096 CNAB steps on
    the device (t = 8: periodic shedding), then the next `horizon` CNAB
    velocities -- the linearisation points of a first Picard sweep
    (snu:1427-1431)"""
    from dolfin_navier_scipy_amd import convection, saddle
    from dolfin_navier_scipy_amd.fem import get_sysmats
    from oracle import saddle_oracle
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=100)
    th, inv = femp['V'], femp['invinds']
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    dt = 1./512
    v0 = saddle_oracle.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'],
                                        rhsp=rhsd['fp'])[:NV]
    cvop = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'])
    system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
    system.setup_precond(cheb_degree=6, schur='dense', fp32_store=True,
                         drop_tol=1e-3, factorization='full')
    stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
    nfc = cvop.apply(v0, scale=-1.0)
    stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
    stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
    stp.set_convection(cvop, scale=-1.0)
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt, extrapolate=4)
    o = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
    stp.run(4096, cf, o)
    vs = [stp.get_state()[0]]
    for _ in range(horizon):
        stp.run(1, cf, o)
        vs.append(stp.get_state()[0])
    stp.close()
    system.close()
    cvop.close()
    # the wake IS shedding: the lift-like cross-stream momentum changes sign
    return dict(femp=femp, sm=sm, rhsd=rhsd, vs=vs, dt=dt, NV=NV, NP=NP)


def _device_conv_callback(cv, NV):
    """the oracle's `conv(v, picard)` answered by a device operator of its own
    (verified against the host assembler to 1e-13 above; the host assembler
    needs 0.3 s per call at this size)"""
    def conv(v, picard):
        N, rhsbc, rhscon = cv.assemble(v.reshape(-1)[:NV], newton=not picard)
        return N, (0.*rhscon if picard else rhscon), rhsbc
    return conv


def _host_conv_callback(femp, NV):
    """the same callback from the HOST assembler (`TaylorHood.convection_mats`,
    the restatement of dts:325-376 the CPU tier checks): nothing of the
    device in it"""
    th, inv = femp['V'], femp['invinds']
    bcsv = np.zeros((th.vdim, 1))
    bcsv[femp['dbcinds'], 0] = femp['dbcvals']

    def conv(v, picard):
        vfull = bcsv.copy()
        vfull[inv, 0] = np.asarray(v).reshape(-1)[:NV]
        N1, N2, fv3 = th.convection_mats(vfull, keep_pattern=True)
        Nm = N1 if picard else (N1 + N2)
        return (Nm[inv, :][:, inv].tocsr(),
                (0.*fv3[inv, :] if picard else fv3[inv, :]),
                -(Nm @ bcsv)[inv, :])
    return conv


class _ConvAtMarks(object):
    """the oracle's convection callback over a long sweep: the HOST assembler
    for the two evaluations that make up the system of every marked step
    (`N_n` about the linearisation point and `N_c` about the state the step
    starts from: calls `2k - 1` and `2k - 2` of `trapezoidal_sweep`, which
    evaluates once in front of its loop and twice per step), the device
    operator for the steps in between (0.3 s per host call at this size:
    all 2 x 192 calls would take two minutes).  At the marks the oracle's
    step owes nothing to the device assembly."""

    def __init__(self, device_conv, host_conv, marks):
        self.dev, self.host = device_conv, host_conv
        self.host_calls = set()
        for k in marks:
            self.host_calls.update((2*k - 2, 2*k - 1))
        self.n, self.used_host = 0, 0

    def __call__(self, v, picard):
        host = self.n in self.host_calls
        self.n += 1
        if host:
            self.used_host += 1
            return self.host(v, picard)
        return self.dev(v, picard)


def test_sweeps_over_2048_steps_of_developed_shedding(shedding):
    """BASELINE config 3 over a horizon: a Picard sweep, then a Newton sweep,
    over 2048 steps of developed shedding.  The reference factorises the
    current operator in every step (snu:1484-1512); here the system matrix is
    re-valued every step and the preconditioner follows the refresh policy of
    `TrapezoidalStepper.sweep`.  Asserted: <= 3 Krylov steps per time step in
    EVERY batch of 64 steps; both sweeps within 1e-8 (v in the M-norm, p) of
    the oracle over their first `NORACLE` steps -- compared at marked steps
    whose systems the oracle builds with the HOST assembler (`_ConvAtMarks`);
    the step equation of both sweeps, written with host-assembled matrices,
    satisfied to 1e-9 at steps all along the horizon.  (The whole
    horizon against the oracle takes the host ten minutes:
    `scripts/sweep_horizon_parity.py`, recorded in profiles/r04_sweeps/.)"""
    import time
    from dolfin_navier_scipy_amd import convection, saddle
    from dolfin_navier_scipy_amd import newton_picard as dnp
    from oracle import saddle_oracle
    s = shedding
    femp, sm, rhsd, vs, dt = s['femp'], s['sm'], s['rhsd'], s['vs'], s['dt']
    th, inv = femp['V'], femp['invinds']
    M, A, J = sm['M'], sm['A'], sm['J']
    NV, NP = s['NV'], s['NP']
    tr = dt*np.arange(HORIZON + 1)
    cv = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'])
    stp = dnp.TrapezoidalStepper(
        M, A, J, cv, nslots=HORIZON + 1, dt=dt,
        precond=dict(cheb_degree=6, drop_tol=1e-3, factorization='full'),
        precond_linpoint=vs[0], refresh_iters=3.0)
    stp.set_rhs(rhsd['fv'], rhsd['fp'])
    for k in range(HORIZON + 1):
        stp.write_linpoint(0, k, vs[k])
    opts = saddle.solve_opts(rtol=SWEEP_RTOL, maxiter=400, use_graph=True,
                             reorth=2)
    mnorm = lambda x: np.sqrt((x.T @ (M @ x)).item())
    # the oracle's operators: a device convection operator of its own
    cvo = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'])
    cvo.bind_pattern(stp.pattern)
    conv = _device_conv_callback(cvo, NV)
    hconv = _host_conv_callback(femp, NV)
    lin = {t: vs[k] for k, t in enumerate(tr)}
    NORACLE = 192
    marks = list(range(32, NORACLE + 1, 32))
    import scipy.sparse as sps
    fv, fp = rhsd['fv'], rhsd['fp']

    def step_residual(vdict, pdict, lindict, picard):
        """the trapezoidal step equation (snu:1034-1035 + continuity) at steps
        all along the horizon, from the device's own iterates"""
        worst = 0.
        for k in range(256, HORIZON + 1, 256):
            vc, vn, pn = vdict[tr[k-1]], vdict[tr[k]], pdict[tr[k]]
            # (matrices of the HOST assembler: the device's iterates are
            # judged by an equation the device had no part in writing)
            Nc, rcc, rbc = hconv(vc, picard)
            Nn, rcn, rbn = hconv(lindict[tr[k]], picard)
            rhs = M @ vc + .5*dt*((fv + rbn + rcn) + (fv + rbc + rcc)
                                  - (A + Nc) @ vc)
            res_v = (M + .5*dt*(A + Nn)) @ vn + J.T @ (-dt*pn) - rhs
            res_p = J @ vn - fp
            worst = max(worst, np.sqrt(
                np.linalg.norm(res_v)**2 + np.linalg.norm(res_p)**2)
                / np.sqrt(np.linalg.norm(rhs)**2 + np.linalg.norm(fp)**2))
        return worst

    def against_oracle(vdict, pdict, lindict, picard):
        oconv = _ConvAtMarks(conv, hconv, marks)
        ref_v, ref_p, _ = npo.trapezoidal_sweep(
            tr[:NORACLE + 1], vs[0], M=M, A=A, J=J, fv=fv, fp=fp, conv=oconv,
            appndbcs=lambda v: v, linpoints=lindict, picard=picard,
            solve=saddle_oracle.RefinedSolve())
        assert oconv.used_host == 2*len(marks) and oconv.n == 2*NORACLE + 1
        wv = max(mnorm(vdict[tr[k]] - ref_v[tr[k]])/mnorm(ref_v[tr[k]])
                 for k in marks)
        wp = max(np.linalg.norm(pdict[tr[k]] - ref_p[tr[k]])
                 / np.linalg.norm(ref_p[tr[k]]) for k in marks)
        return wv, wp

    which, lindict = 0, lin
    for name, picard in (('picard', True), ('newton', False)):
        t0 = time.perf_counter()
        got_v, got_p, upd, st = stp.sweep(tr, vs[0], which, picard, opts=opts)
        t_gpu = time.perf_counter() - t0
        t0 = time.perf_counter()
        wv, wp = against_oracle(got_v, got_p, lindict, picard)
        t_cpu = time.perf_counter() - t0
        wres = step_residual(got_v, got_p, lindict, picard)
        print('{0}, {1} steps: {2:.2f} Krylov steps per time step (worst batch '
              '{3:.2f}), {4} refreshes, {5} batches replayed; parity over {6} '
              'steps v {7:.2e} p {8:.2e}; step residual along the horizon '
              '{9:.2e}; device {10:.1f} s (states recorded), oracle {11:.1f} s'
              .format(name, HORIZON, st['iters']/float(HORIZON),
                      max(st['batches']), st['refreshes'],
                      st['replayed_batches'], NORACLE, wv, wp, wres, t_gpu,
                      t_cpu))
        assert max(st['batches']) <= 3.0, st['batches']
        assert wv <= 1e-8 and wp <= 1e-8, (wv, wp)
        assert wres <= 1e-9, wres
        # the next sweep linearises about this one (the device's own
        # trajectory buffer: traj[1 - which])
        lindict = {t: got_v[t] for t in tr}
        which = 1 - which
    stp.close()
    cv.close()
    cvo.close()


def test_refresh_policy_rebuilds_a_stale_preconditioner(shedding):
    """the preconditioner set up for `M + dt/2 A` alone (no convection inside)
    needs 3 Krylov steps per time step in the developed wake (4 with the
    cubic warm start); with the bound at 2.8 the policy sees the first batch,
    rebuilds about the current operator, and the batches behind it stay
    below the bound -- with the same iterates as a sweep that never refreshes
    (the preconditioner does not change the answer)"""
    from dolfin_navier_scipy_amd import convection, saddle
    from dolfin_navier_scipy_amd import newton_picard as dnp
    s = shedding
    femp, sm, rhsd, vs, dt = s['femp'], s['sm'], s['rhsd'], s['vs'], s['dt']
    M, A, J = sm['M'], sm['A'], sm['J']
    nst = 320
    tr = dt*np.arange(nst + 1)
    opts = saddle.solve_opts(rtol=SWEEP_RTOL, maxiter=400, use_graph=True,
                             reorth=2)
    out = {}
    BOUND = 2.8
    for bound in (None, BOUND):
        cv = convection.ConvectionP2.from_taylor_hood(
            femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
        stp = dnp.TrapezoidalStepper(
            M, A, J, cv, nslots=nst + 1, dt=dt,
            precond=dict(cheb_degree=6, drop_tol=1e-3, factorization='full'),
            precond_linpoint=None, refresh_iters=bound)
        stp.set_rhs(rhsd['fv'], rhsd['fp'])
        for k in range(nst + 1):
            stp.write_linpoint(0, k, vs[k])
        v, p, upd, st = stp.sweep(tr, vs[0], 0, True, opts=opts)
        out[bound] = (v, p, st)
        stp.close()
        cv.close()
    st0, st1 = out[None][2], out[BOUND][2]
    print('stale preconditioner: batches', [round(b, 2) for b in st0['batches']],
          '-> with the policy', [round(b, 2) for b in st1['batches']],
          st1['refreshes'], 'refresh(es)')
    assert st0['refreshes'] == 0 and max(st0['batches']) > BOUND
    assert 1 <= st1['refreshes'] <= 2
    # (the batches behind the rebuild need fewer Krylov steps than the stale
    # set-up's; single batches may still touch 3)
    assert np.mean(st1['batches'][2:]) <= np.mean(st0['batches'][2:]) - 0.2
    assert st1['iters'] < st0['iters']
    vt, vr = out[BOUND][0][tr[-1]], out[None][0][tr[-1]]
    assert np.linalg.norm(vt - vr) <= 1e-8*np.linalg.norm(vr)
