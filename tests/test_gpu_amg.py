"""The multigrid Schur block on meshes that are NOT refinements of anything
this repository built: the reference's `karman2D-rotcyl_lvl3` / `lvl4`
(`/root/reference/tests/mesh/`, set-up `problem_setups.py:773-987`;
lvl4: NV = 133 334, NP = 17 135 -- SURVEY 8d).  Their pressure spaces are
beyond the dense Schur inverse and come with no nested spaces, so every
drop-in used to fall back to `diag(J D^-1 J^T)^-1`; now `schur='auto'` builds
an algebraic hierarchy (`amg.py`) and hands it to the same device cycle."""
import numpy as np
import pytest

from oracle import snu_oracle as so

pytestmark = pytest.mark.gpu


def _setup(level):
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(
        problem='gen_bccont', nu=1e-3, charvel=0.2, bccontrol=False,
        meshparams=dict(meshname='karman2D-rotcyl_lvl{0}'.format(level),
                        geodata='karman2D-rotcyl-bm_geo_cntrlbc'))
    return femp, sm, rhsd


def _mnorm(M, x):
    return float(np.sqrt((x.T @ (M @ x)).item()))


def test_cnab_on_karman_rotcyl_lvl3_through_solve_nse():
    """whole `solve_nse` (Stokes start, initial pressure, Heun start, CNAB
    loop) on lvl3 (NV = 54 526, NP = 7054 > `schur_dense_max`): every system of
    the call -- `A` alone, `M`, `M + dt A`, `M + dt/2 A` -- gets its algebraic
    hierarchy; velocities and pressures against the oracle's direct solves"""
    from dolfin_navier_scipy_amd import stokes_navier_utils as snu
    from dolfin_navier_scipy_amd import time_int_utils as gtiu
    femp, sm, rhsd = _setup(3)
    NP, NV = sm['J'].shape
    assert NP == 7054
    nts, dt = 24, 2e-3
    kw = dict(A=sm['A'], M=sm['M'], J=sm['J'], fv=rhsd['fv'], fp=rhsd['fp'],
              trange=np.linspace(0., nts*dt, nts + 1), V=femp['V'],
              invinds=femp['invinds'], dbcinds=femp['dbcinds'].tolist(),
              dbcvals=femp['dbcvals'].tolist())
    vo, po, ffo = so.solve_nse(**kw)
    (vg, pg), ffg = snu.solve_nse(start_ssstokes=True, return_final_vp=True,
                                  check_ff=True, **kw)
    rec = dict(gtiu.LAST_RUN)
    snu.clear_cache()
    assert ffg == ffo == 0
    inv = femp['invinds']
    ev = _mnorm(sm["M"], vg - vo)/_mnorm(sm['M'], vo)
    ep = np.linalg.norm(pg - po)/np.linalg.norm(po)
    per = rec['krylov_steps']/float(rec['time_steps'])
    print('karman2D-rotcyl_lvl3, {0} CNAB steps: v {1:.2e} p {2:.2e}; '
          'hierarchy {3}, {4:.2f} Krylov steps per time step'.format(
              nts, ev, ep, rec['schur_hierarchy'], per))
    assert rec['schur_hierarchy']['kind'] == 'algebraic'
    assert rec['schur_hierarchy']['levels'][0] == NP
    assert ev <= 1e-8 and ep <= 1e-8, (ev, ep)
    # (24 steps right behind the impulsive start, tolerance 1e-12: the warm
    # start has nothing to extrapolate yet; the Jacobi block these systems
    # used to get does not converge within 400 steps here at all)
    assert per <= 20.0, per


def test_cnab_on_karman_rotcyl_lvl4_krylov_steps_and_parity():
    """lvl4 (NP = 17 135): the CNAB loop with the algebraic hierarchy in the
    DEVELOPED run -- 1024 device-resident steps from the steady Stokes state
    (itself a device solve through the AMG block on the steady operator `A`:
    its residual is asserted), then 32 steps of `solve_nse` from that state
    against the oracle's factor-once loop from the same state: v and p within
    1e-8.  Krylov steps per time step: at most 4 in the developed run at the
    bench's tolerance (1e-10; 3.1 measured); the drop-in's 32 steps run two
    decades tighter and restart the warm-start history behind their Heun
    step (9.6 measured).  (Right behind an impulsive start the count is 10-15 whatever
    the Schur block is -- with the EXACT dense inverse on lvl3 as well,
    `scripts/amg_diag.py`: the warm start has nothing to extrapolate yet.)"""
    from dolfin_navier_scipy_amd import lin_alg_utils as lau
    from dolfin_navier_scipy_amd import time_int_utils as gtiu
    from dolfin_navier_scipy_amd import stokes_navier_utils as snu
    from dolfin_navier_scipy_amd import saddle, convection
    import scipy.sparse as sps
    femp, sm, rhsd = _setup(4)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    assert (NV, NP) == (133334, 17135)
    # steady Stokes start on the device (snu:903-907)
    vp0 = lau.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'],
                               rhsp=rhsd['fp'])
    K0 = sps.bmat([[A, J.T], [J, None]], format='csr')
    b0 = np.vstack([rhsd['fv'], rhsd['fp']])
    res0 = np.linalg.norm(K0 @ vp0 - b0)/np.linalg.norm(b0)
    assert res0 <= 1e-10, res0
    lau.clear_cache()
    # the developed run: 1024 resident steps
    dt, spin = 1e-3, 1024
    F = (M + .5*dt*A).tocsr()
    system = saddle.SaddleSystem(F, J)
    schur = saddle.choose_schur(system, F, J, schur='auto')
    assert schur == 'mg' and system.schur_hierarchy['kind'] == 'algebraic'
    # (the count asked for is what a solve NEEDS to reach the tolerance: the
    # solves stop there -- with oversolve they would run whole cycles)
    system.set_option('oversolve', 0)
    system.setup_precond(cheb_degree=8, schur=schur, drop_tol=7e-3,
                         fhat='explicit', factorization='full')
    cv = convection.ConvectionP2.from_taylor_hood(
        femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
    stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
    nfc = cv.apply(vp0[:NV], scale=-1.0)
    stp.set_state(vp0[:NV], nfc_c=nfc, nfc_o=nfc)
    stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
    stp.set_convection(cv, scale=-1.0)
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt, extrapolate=3)
    opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True,
                             reorth=2)
    stp.run(spin - 128, cf, opts)
    _, its, _ = stp.run(128, cf, opts)
    per_spin = its/128.
    v1, p1 = stp.get_state()
    stp.close()
    cv.close()
    system.close()
    iniv = np.zeros((femp['V'].vdim, 1))
    iniv[femp['invinds'], :] = v1
    iniv[femp['dbcinds'], 0] = femp['dbcvals']
    nts = 32
    kw = dict(A=A, M=M, J=J, fv=rhsd['fv'], fp=rhsd['fp'], iniv=iniv,
              inip=p1, trange=np.linspace(0., nts*dt, nts + 1),
              V=femp['V'], invinds=femp['invinds'],
              dbcinds=femp['dbcinds'].tolist(),
              dbcvals=femp['dbcvals'].tolist())
    vo, po, ffo = so.solve_nse(**kw)
    (vg, pg), ffg = snu.solve_nse(return_final_vp=True, check_ff=True, **kw)
    rec = dict(gtiu.LAST_RUN)
    snu.clear_cache()
    assert ffg == ffo == 0
    ev = _mnorm(M, vg - vo)/_mnorm(M, vo)
    ep = np.linalg.norm(pg - po)/np.linalg.norm(po)
    per = rec['krylov_steps']/float(rec['time_steps'])
    print('karman2D-rotcyl_lvl4: {0:.2f} Krylov steps per time step at the '
          'end of the {1}-step spin-up (rtol 1e-10); {2} CNAB steps of '
          'solve_nse behind it: v {3:.2e} p {4:.2e}, {5:.2f} Krylov steps per '
          'time step (rtol 1e-12); hierarchy {6}; Stokes start residual '
          '{7:.1e}'.format(per_spin, spin, nts, ev, ep, per,
                           rec['schur_hierarchy'], res0))
    assert rec['schur_hierarchy']['kind'] == 'algebraic'
    assert rec['schur_hierarchy']['contraction_estimate'] < 0.4
    assert ev <= 1e-8 and ep <= 1e-8, (ev, ep)
    # the developed run at the bench's tolerance: 3.1 measured
    assert per_spin <= 4.0, per_spin
    # the 32 steps of the drop-in: two decades tighter (1e-12) and with the
    # warm-start history restarted behind the Heun step -- 9.6 measured
    assert per <= 12.0, per
