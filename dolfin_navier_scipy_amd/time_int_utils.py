"""MI355X drop-in for the semi-explicit integrators of the reference
(`dolfin_navier_scipy/time_int_utils.py`): same names, same keyword
interfaces, same callbacks, same return values -- but the constant system
`[[M + theta*dt*A, J^T], [J, 0]]`, the velocity/pressure iterates and the
convection history live in HBM, the right-hand side of every step is built by
a fused HIP SpMV kernel and the saddle-point solve is the block-preconditioned
Krylov iteration of `csrc/` (instead of `spsla.factorized`, tiu:89-91,134).

 * `cnab`                (tiu:23-145)   Heun start + CN / AB2
 * `sbdftwo`             (tiu:260-355)  Heun start + SBDF2
 * `semi_implicit_euler` (tiu:566-635)
 * `_onestepheun`        (tiu:366-477)  two boundary solves via `lin_alg_utils`
 * `_inittimegrid`       (tiu:480-489)

Per step the host still evaluates the reference's callbacks (`f_vdp`, `getbcs`,
`applybcs`, `f_tdp`, `g_tdp`, `dynamic_rhs`, `savevp`) because they are the
caller's Python code; everything between them is on the device.
"""
import logging

import numpy as np
import scipy.sparse as sps

from . import lin_alg_utils as lau
from .saddle import SaddleSystem, ImexStepper, solve_opts, choose_schur

__all__ = ['cnab', 'sbdftwo', 'semi_implicit_euler', 'SOLVER']

# solver settings of the time loops; `rtol` is relative to ||rhs||.  `None`:
# `lin_alg_utils.default_rtol` of the system at hand -- 1e-12, and 1e-13 where
# penalised rows inflate ||rhs|| (BASELINE config 5: Robin penalty 1/alpha =
# 1e5; its pressure meets 1e-8 against the direct solve at 1e-13 (1.7e-10), not
# at 1e-12 (1e-8) -- tests/test_gpu_config5.py)
SOLVER = dict(method='gmres', rtol=None, maxiter=400, restart=60,
              cheb_degree=6, drop_tol=1e-3, factorization='full', reorth=2,
              schur='auto', extrapolate='auto', device=0, check_every=2,
              use_graph=True, carry_residual=True)


# record of the last time loop that ran (diagnostics / tests): which Schur
# block its system got, the algebraic or geometric hierarchy behind it, time
# steps and Krylov steps of the loop
LAST_RUN = {}


def _record_run(name, system, stepper):
    LAST_RUN.clear()
    try:        # (runs in a `finally`: never in the way of the real error)
        LAST_RUN.update(
            integrator=name, time_steps=stepper.total_steps,
            krylov_steps=stepper.total_iters,
            schur_hierarchy=getattr(system, 'schur_hierarchy', None),
            precond=system.precond_info())
    except Exception:
        pass


def _checkuniformgrid(trange):
    steps = np.diff(np.asarray(trange, dtype=np.float64))
    if not np.allclose(np.linalg.norm(np.diff(steps)), 0):
        raise NotImplementedError()


def _inittimegrid(trange, ntimeslices=10):
    _checkuniformgrid(trange)
    dt = trange[1] - trange[0]
    rest = np.array(trange[2:])
    chunk = np.floor(rest.size/ntimeslices).astype(np.int32)
    slices = [rest[k*chunk:(k+1)*chunk].tolist() for k in range(ntimeslices)]
    slices.append(rest[ntimeslices*chunk:].tolist())
    return dt, slices


def _col(vec, n):
    """callbacks may return scalars (`applybcs -> 0., 0., 0.`, snu:1104)"""
    arr = np.asarray(vec, dtype=np.float64)
    if arr.ndim == 0 or arr.size == 1:
        return np.full((n, 1), float(arr.reshape(-1)[0]) if arr.size else 0.)
    return arr.reshape((n, 1))


def _wrap_callbacks(NV, dynamic_rhs, f_tvdp, f_vdp):
    zerorhs = np.zeros((NV, 1))
    if dynamic_rhs is None:
        def dynamic_rhs(t, vc=None, memory={}, mode=None):
            return zerorhs, memory
    if f_tvdp is not None:
        inner = dynamic_rhs

        def dynamic_rhs(t, vc=None, memory={}, mode=None):
            val, mem = inner(t, vc=vc, memory=memory, mode=mode)
            return val + f_tvdp(t, vc), mem
    if f_vdp is None:
        def f_vdp(vvec):
            return zerorhs
    return dynamic_rhs, f_vdp


def _solver_settings(solver):
    prm = dict(SOLVER)
    prm.update(solver or {})
    return prm


def _device_system(fmat, J, prm):
    NP = J.shape[0]
    system = SaddleSystem(fmat, J, device=prm['device'])
    # dense inverse up to `schur_dense_max` pressure dofs; beyond it the
    # multigrid block, on nested pressure spaces if the caller has them
    # (`prolongations`: refined meshes), else on an algebraic hierarchy
    schur = choose_schur(system, fmat, J, schur=prm['schur'],
                         prolongations=prm.get('prolongations'),
                         dense_max=lau.DEFAULTS['schur_dense_max'])
    if prm['extrapolate'] == 'auto':
        # warm start: quartic where one Krylov step per time step does it
        # (dense Schur block); cubic with the multigrid block, whose solves
        # run their cycle's two columns (`saddle.streaming_precond_defaults`)
        prm['extrapolate'] = 3 if schur == 'mg' else 4
    # the full block factorisation needs the explicit polynomial matrix
    fact = prm['factorization']
    if fmat.shape[0] > 1000000 or not 2 <= prm['cheb_degree'] <= 12:
        fact = 'triangular'
    system.setup_precond(cheb_degree=prm['cheb_degree'], schur=schur,
                         drop_tol=prm['drop_tol'], factorization=fact)
    rtol = prm['rtol'] if prm['rtol'] is not None else lau.default_rtol(fmat)
    opts = solve_opts(method=prm['method'], rtol=rtol,
                      maxiter=prm['maxiter'], restart=prm['restart'],
                      check_every=prm['check_every'],
                      use_graph=prm['use_graph'], reorth=prm['reorth'])
    return system, opts


def _onestepheun(vc=None, pc=None, tc=None, tn=None, M=None, A=None, J=None,
                 scalep=1., dfv_c=None, dynamic_rhs=None, drm={},
                 bcs_c=None, applybcs=None, appndbcs=None, getbcs=None,
                 f_tdp=None, f_vdp=None, g_tdp=None, krylov=None,
                 krpslvprms={}):
    """IMEX-Euler predictor / trapezoidal corrector (tiu:366-477); both
    saddle solves go through `lin_alg_utils.solve_sadpnt_smw` on the GPU.
    The corrector keeps the reference's `amat=M` (tiu:466)."""
    NP, NV = J.shape
    dt = tn - tc
    JT = sps.csr_matrix(J.T)
    bfv_c, _, mbc_c = applybcs(bcs_c)
    fv_c = f_tdp(tc)
    nfc_c = f_vdp(appndbcs(vc, bcs_c))
    tdfv_n, drm = dynamic_rhs(tn, vc=vc, memory=drm, mode='heunpred')
    tbcs = getbcs(tn, appndbcs(vc, bcs_c), pc, mode='heunpred')
    tbfv_n, tbfp_n, tmbc_n = applybcs(tbcs)
    fv_n, fp_n = f_tdp(tn), g_tdp(tn)
    tfv = M @ vc + dt*(fv_n + tbfv_n + tdfv_n) + dt*nfc_c - (tmbc_n - mbc_c)
    tvp_n = lau.solve_sadpnt_smw(amat=M + dt*A, jmat=J, jmatT=JT, rhsv=tfv,
                                 rhsp=_col(fp_n + tbfp_n, NP), krylov=krylov,
                                 krpslvprms=krpslvprms)
    tv_n = tvp_n[:NV, :]
    tp_n = 1./dt*scalep*tvp_n[NV:, :]
    dfv_n, drm = dynamic_rhs(tn, vc=tv_n, memory=drm, mode='heuncorr')
    tnfc_n = f_vdp(appndbcs(tv_n, tbcs))
    bcs_n = getbcs(tn, appndbcs(tv_n, tbcs), tp_n, mode='heuncorr')
    bfv_n, bfp_n, mbc_n = applybcs(bcs_n)
    rhs_n = M @ vc - (mbc_n - mbc_c) - .5*dt*(A @ (vc + tv_n)) \
        + .5*dt*(fv_c + fv_n + bfv_n + bfv_c + dfv_n + dfv_c + nfc_c + tnfc_n)
    vp_n = lau.solve_sadpnt_smw(amat=M, jmat=J, jmatT=JT, rhsv=rhs_n,
                                rhsp=_col(fp_n + bfp_n, NP), krylov=krylov,
                                krpslvprms=krpslvprms)
    v_n = vp_n[:NV].reshape((NV, 1))
    p_n = 1./dt*scalep*vp_n[NV:].reshape((NP, 1))
    nfc_n = f_vdp(appndbcs(v_n, bcs_n))
    return (v_n, p_n, bcs_n, bfv_n, mbc_c, mbc_n, fv_n, nfc_c, nfc_n, dfv_n,
            drm)


def cnab(trange=None, inivel=None, inip=None, bcs_ini=[],
         M=None, A=None, J=None, f_vdp=None, f_tdp=None, g_tdp=None,
         f_tvdp=None, scalep=-1., getbcs=None, applybcs=None, appndbcs=None,
         savevp=None, dynamic_rhs=None, dynamic_rhs_memory={},
         check_ff_maxv=None, ntimeslices=10, verbose=True, solver=None,
         device_convection=None, invinds=None, resident=None):
    """Crank-Nicolson / Adams-Bashforth-2 on the GPU (reference tiu:23-145)

    `solver`: optional dict overriding `SOLVER` (method, rtol, cheb_degree...).
    `device_convection`: a `convection.ConvectionP2` (with `invinds`, the inner
    dofs of the full velocity vector) -- the loop then evaluates `-N(v)v` on
    the device and `f_vdp` is not called per step.
    `resident`: dict, with `device_convection` only -- what the caller
    guarantees about its callbacks so that whole time slices run on the device
    without a host round trip per step:
      `bcs_time_only`   `getbcs(t, v, p)` ignores `v` and `p` (prescribed
                        boundary motion); static boundaries need no flag
      `static_dbcvals`  values of the operator's leading (static) Dirichlet
                        dofs; the controlled values `bcs` follow them
      `savevp_times`    the only times `savevp` has to see (None: all)
    The per-step data the callbacks return (`f_tdp`, `g_tdp`, `applybcs`) are
    tabulated per slice and uploaded (`dns_imex_set_rhs_table`,
    `dns_conv_set_dbc_table`); the blow-up guard stays at the slice starts.
    Not possible (falls back to one host round trip per step) with a
    `dynamic_rhs` or `f_tvdp`, which depend on the state.
    Returns `v_n, p_n, ffflag` like the reference.
    """
    prm = _solver_settings(solver)
    state_dependent = dynamic_rhs is not None or f_tvdp is not None
    dt, listofts = _inittimegrid(trange, ntimeslices=ntimeslices)
    NP, NV = J.shape
    ffflag = 0
    if device_convection is not None and f_vdp is None:
        f_vdp = device_convection.host_callback(invinds)   # Heun start only
    dynamic_rhs, f_vdp = _wrap_callbacks(NV, dynamic_rhs, f_tvdp, f_vdp)
    dfv_c, drm = dynamic_rhs(trange[0], vc=inivel, memory=dynamic_rhs_memory,
                             mode='init')
    savevp(appndbcs(inivel, bcs_ini), inip, time=trange[0])
    (v_n, p_n, bcs_n, bfv_n, mbc_c, mbc_n, fv_n, nfc_c, nfc_n, dfv_n,
     drm) = _onestepheun(vc=inivel, pc=inip, tc=trange[0], tn=trange[1],
                         M=M, A=A, J=J, scalep=scalep, dfv_c=dfv_c,
                         dynamic_rhs=dynamic_rhs, drm=drm, bcs_c=bcs_ini,
                         applybcs=applybcs, appndbcs=appndbcs, getbcs=getbcs,
                         f_tdp=f_tdp, f_vdp=f_vdp, g_tdp=g_tdp)
    savevp(appndbcs(v_n, bcs_n), p_n, time=trange[1])

    # the constant system of the loop, factor-once in the reference (tiu:89-91)
    M, A = sps.csr_matrix(M), sps.csr_matrix(A)
    system, opts = _device_system((M + .5*dt*A).tocsr(), J, prm)
    stepper = ImexStepper(system, (M - .5*dt*A).tocsr())
    cf = ImexStepper.coeffs(a_c=1., a_p=0., cn_c=1.5*dt, cn_o=-.5*dt,
                            pscale=scalep/dt, extrapolate=prm['extrapolate'],
                            carry_residual=prm['carry_residual'])
    stepper.set_state(v_n, ptilde_c=p_n*dt/scalep, nfc_c=nfc_c)
    rsd = dict(resident or {})
    statvals = list(rsd.get('static_dbcvals', []) or [])
    moving = len(bcs_ini) > 0
    if device_convection is not None:
        stepper.set_convection(device_convection, scale=-1.0)
        if moving or statvals:
            device_convection.set_dbcvals(statvals + list(bcs_n))
    on_device = (device_convection is not None and resident is not None
                 and not state_dependent
                 and (not moving or rsd.get('bcs_time_only', False)))
    savetimes = rsd.get('savevp_times', None)
    savetimes = None if savetimes is None else set(savetimes)
    try:
        for kck, ctrange in enumerate(listofts):
            nrmvc = stepper.vnorm()
            if verbose:
                logging.info('time {0}/{1} -- |v| {2:.2e}'.format(
                    kck, ntimeslices, nrmvc))
            if nrmvc > check_ff_maxv or np.isnan(nrmvc):
                logging.warning('BREAK: |v| is `NaN` or |v| > threshhold')
                ffflag = 1
                break
            if on_device and len(ctrange) > 0:
                # the whole slice: tabulate what the callbacks return, upload,
                # replay; the host sees the state at the save times only
                ns = len(ctrange)
                gvt, gpt = np.empty((ns, NV)), np.empty((ns, NP))
                dbt = np.empty((ns, len(statvals) + len(bcs_n))) \
                    if moving else None
                for s, ctime in enumerate(ctrange):
                    bcs_c, bfv_c, mbc_c = bcs_n, bfv_n, mbc_n
                    fv_c = fv_n
                    bcs_n = getbcs(ctime, None, None, mode='abtwo')
                    bfv_n, bfp_n, mbc_n = applybcs(bcs_n)
                    fv_n, fp_n = f_tdp(ctime), g_tdp(ctime)
                    gvt[s] = _col(-(mbc_n - mbc_c)
                                  + .5*dt*(fv_c + fv_n + bfv_n + bfv_c),
                                  NV)[:, 0]
                    gpt[s] = _col(fp_n + bfp_n, NP)[:, 0]
                    if moving:       # N(v_c) sees the CURRENT boundary values
                        dbt[s] = statvals + list(bcs_c)
                stepper.set_rhs_table(gvt, gpt)
                if moving:
                    device_convection.set_dbc_table(dbt)
                done = 0
                for s, ctime in enumerate(ctrange):
                    if (savetimes is None or ctime in savetimes
                            or s == ns - 1):
                        stepper.run(s + 1 - done, cf, opts)
                        done = s + 1
                        v_n, p_n = stepper.get_state()
                        bcs_at = dbt[s + 1][len(statvals):].tolist() \
                            if (moving and s + 1 < ns) else bcs_n
                        if savetimes is None or ctime in savetimes:
                            savevp(appndbcs(v_n, bcs_at), p_n, time=ctime)
                if moving:
                    device_convection.set_dbcvals(statvals + list(bcs_n))
                stepper.set_rhs(_col(0., NV), _col(0., NP))
                continue
            for ctime in ctrange:
                v_c, p_c = v_n, p_n
                bcs_c, bfv_c, mbc_c = bcs_n, bfv_n, mbc_n
                fv_c, dfv_c = fv_n, dfv_n
                if device_convection is not None and (moving or statvals):
                    device_convection.set_dbcvals(statvals + list(bcs_c))
                nfc_new = None if device_convection is not None \
                    else f_vdp(appndbcs(v_c, bcs_c))
                bcs_n = getbcs(ctime, appndbcs(v_c, bcs_c), p_c, mode='abtwo')
                bfv_n, bfp_n, mbc_n = applybcs(bcs_n)
                fv_n, fp_n = f_tdp(ctime), g_tdp(ctime)
                dfv_n, drm = dynamic_rhs(ctime, vc=v_c, memory=drm,
                                         mode='abtwo')
                # everything that is not `M v - dt/2 A v` or convection
                gvec = -(mbc_n - mbc_c) \
                    + .5*dt*(fv_c + fv_n + bfv_n + bfv_c + dfv_n + dfv_c)
                stepper.set_rhs(_col(gvec, NV), _col(fp_n + bfp_n, NP))
                stepper.step(cf, nfc_new=nfc_new, opts=opts)
                v_n, p_n = stepper.get_state()
                savevp(appndbcs(v_n, bcs_n), p_n, time=ctime)
    finally:
        _record_run('cnab', system, stepper)
        stepper.close()
        system.close()
    return v_n, p_n, ffflag


def sbdftwo(trange=None, inivel=None, inip=None, bcs_ini=[],
            M=None, A=None, J=None, f_vdp=None, f_tdp=None, g_tdp=None,
            check_ff=False, check_ff_maxv=None, scalep=-1.,
            getbcs=None, applybcs=None, appndbcs=None, savevp=None,
            dynamic_rhs=None, dynamic_rhs_memory={},
            ntimeslices=10, verbose=True, solver=None,
            device_convection=None, invinds=None, resident=None):
    """SBDF2 on the GPU (reference tiu:260-355, quirks kept: pressure scaled
    by `1/dt`, blow-up guard on the previous velocity)

    `device_convection`, `invinds`, `resident`: as in `cnab` -- `-N(v)v` from
    the device operator, and whole time slices without a host round trip per
    step when the callbacks depend on the time only (rhs / boundary-value
    tables)."""
    prm = _solver_settings(solver)
    state_dependent = dynamic_rhs is not None
    dt, listofts = _inittimegrid(trange, ntimeslices=ntimeslices)
    NP, NV = J.shape
    if device_convection is not None and f_vdp is None:
        f_vdp = device_convection.host_callback(invinds)   # Heun start only
    dynamic_rhs, f_vdp = _wrap_callbacks(NV, dynamic_rhs, None, f_vdp)
    dfv_c, drm = dynamic_rhs(trange[0], vc=inivel, memory=dynamic_rhs_memory,
                             mode='init')
    savevp(appndbcs(inivel, bcs_ini), inip, time=trange[0])
    v_c = inivel
    (v_n, p_n, bcs_n, bfv_n, mbc_c, mbc_n, fv_n, nfc_c, nfc_n, dfv_n,
     drm) = _onestepheun(vc=v_c, pc=inip, tc=trange[0], tn=trange[1],
                         M=M, A=A, J=J, scalep=scalep, dfv_c=dfv_c,
                         dynamic_rhs=dynamic_rhs, drm=drm, bcs_c=bcs_ini,
                         applybcs=applybcs, appndbcs=appndbcs, getbcs=getbcs,
                         f_tdp=f_tdp, f_vdp=f_vdp, g_tdp=g_tdp)
    savevp(appndbcs(v_n, bcs_n), p_n, time=trange[1])

    M, A = sps.csr_matrix(M), sps.csr_matrix(A)
    system, opts = _device_system((M + 2./3*dt*A).tocsr(), J, prm)
    stepper = ImexStepper(system, M)
    cf = ImexStepper.coeffs(a_c=4./3, a_p=-1./3, cn_c=4./3*dt, cn_o=-2./3*dt,
                            pscale=scalep/dt, extrapolate=prm['extrapolate'],
                            carry_residual=prm['carry_residual'])
    stepper.set_state(v_n, v_p=v_c, ptilde_c=p_n*dt/scalep, nfc_c=nfc_c)
    rsd = dict(resident or {})
    statvals = list(rsd.get('static_dbcvals', []) or [])
    moving = len(bcs_ini) > 0
    if device_convection is not None:
        stepper.set_convection(device_convection, scale=-1.0)
        if moving or statvals:
            device_convection.set_dbcvals(statvals + list(bcs_n))
    on_device = (device_convection is not None and resident is not None
                 and not state_dependent
                 and (not moving or rsd.get('bcs_time_only', False)))
    savetimes = rsd.get('savevp_times', None)
    savetimes = None if savetimes is None else set(savetimes)
    ffflag = 0
    try:
        for kck, ctrange in enumerate(listofts):
            nrmvc = np.linalg.norm(v_c)
            if nrmvc > check_ff_maxv or np.isnan(nrmvc):
                ffflag = 1
                break
            if on_device and len(ctrange) > 0:
                ns = len(ctrange)
                gvt, gpt = np.empty((ns, NV)), np.empty((ns, NP))
                dbt = np.empty((ns, len(statvals) + len(bcs_n))) \
                    if moving else None
                for s, ctime in enumerate(ctrange):
                    mbc_p = mbc_c
                    bcs_c, mbc_c = bcs_n, mbc_n
                    bcs_n = getbcs(ctime, None, None, mode='abtwo')
                    bfv_n, bfp_n, mbc_n = applybcs(bcs_n)
                    fv_n, fp_n = f_tdp(ctime), g_tdp(ctime)
                    gvt[s] = _col(-(mbc_n - 4/3*mbc_c + 1/3*mbc_p)
                                  + 2/3*dt*bfv_n + 2/3*dt*fv_n, NV)[:, 0]
                    gpt[s] = _col(fp_n + bfp_n, NP)[:, 0]
                    if moving:
                        dbt[s] = statvals + list(bcs_c)
                stepper.set_rhs_table(gvt, gpt)
                if moving:
                    device_convection.set_dbc_table(dbt)
                done = 0
                v_start = v_n
                for s, ctime in enumerate(ctrange):
                    wanted = savetimes is None or ctime in savetimes
                    # the blow-up guard of the next slice looks at the velocity
                    # BEFORE this slice's last step (tiu:317,322)
                    if not (wanted or s >= ns - 2):
                        continue
                    stepper.run(s + 1 - done, cf, opts)
                    done = s + 1
                    v_s, p_s = stepper.get_state()
                    if s == ns - 2:
                        v_c = v_s
                    if s == ns - 1:
                        v_n, p_n = v_s, p_s
                    if wanted:
                        bcs_at = dbt[s + 1][len(statvals):].tolist() \
                            if (moving and s + 1 < ns) else bcs_n
                        savevp(appndbcs(v_s, bcs_at), p_s, time=ctime)
                if ns == 1:
                    v_c = v_start
                if moving:
                    device_convection.set_dbcvals(statvals + list(bcs_n))
                stepper.set_rhs(_col(0., NV), _col(0., NP))
                continue
            for ctime in ctrange:
                v_p, mbc_p = v_c, mbc_c
                v_c, p_c = v_n, p_n
                bcs_c, mbc_c = bcs_n, mbc_n
                dfv_c = dfv_n
                if device_convection is not None and (moving or statvals):
                    device_convection.set_dbcvals(statvals + list(bcs_c))
                nfc_new = None if device_convection is not None \
                    else f_vdp(appndbcs(v_c, bcs_c))
                bcs_n = getbcs(ctime, appndbcs(v_c, bcs_c), p_c, mode='abtwo')
                bfv_n, bfp_n, mbc_n = applybcs(bcs_n)
                fv_n, fp_n = f_tdp(ctime), g_tdp(ctime)
                dfv_n, drm = dynamic_rhs(ctime, vc=v_c, memory=drm,
                                         mode='abtwo')
                gvec = -(mbc_n - 4/3*mbc_c + 1/3*mbc_p) + 2/3*dt*bfv_n \
                    + 2/3*dt*(fv_n + dfv_n)
                stepper.set_rhs(_col(gvec, NV), _col(fp_n + bfp_n, NP))
                stepper.step(cf, nfc_new=nfc_new, opts=opts)
                v_n, p_n = stepper.get_state()
                savevp(appndbcs(v_n, bcs_n), p_n, time=ctime)
    finally:
        _record_run('sbdftwo', system, stepper)
        stepper.close()
        system.close()
    return v_n, p_n, ffflag


def semi_implicit_euler(iniv=None, jmat=None, mmat=None, amat=None, rhsv=None,
                        trange=None, data_trange=None, fp=None, solver=None,
                        device_convection=None, constant_rhs=None):
    """`M v' + A v + J^T p = rhs(t, v)`, `J v = fp` with the linear part
    implicit (reference tiu:566-635); list of velocities at `data_trange`

    `device_convection` (a `convection.ConvectionP2` over the inner dofs) with
    `constant_rhs` (NV x 1 or None): the right-hand side is
    `rhs(t, v) = constant_rhs - N(v)v`, evaluated on the device -- `rhsv` is
    not called and the loop runs resident between the data points."""
    prm = _solver_settings(solver)
    record = list(np.copy(trange if data_trange is None else data_trange))
    record.pop(0)
    NP, NV = jmat.shape
    fpz = np.zeros((NP, 1)) if fp is None else fp
    dt = trange[1] - trange[0]
    mmat, amat = sps.csr_matrix(mmat), sps.csr_matrix(amat)
    system, opts = _device_system((mmat + dt*amat).tocsr(), jmat, prm)
    stepper = ImexStepper(system, mmat)
    cf = ImexStepper.coeffs(a_c=1., a_p=0., cn_c=dt, cn_o=0., pscale=1.,
                            extrapolate=prm['extrapolate'],
                            carry_residual=prm['carry_residual'])
    stepper.set_state(iniv)
    stepper.set_rhs(np.zeros((NV, 1)), fpz)
    out = [iniv]
    cv = iniv
    try:
        if device_convection is not None:
            stepper.set_convection(device_convection, scale=-1.0)
            if constant_rhs is not None:
                stepper.set_rhs(dt*_col(constant_rhs, NV), fpz)
            todo = 0
            for ct in trange[1:]:
                todo += 1
                if len(record) > 0 and ct == record[0]:
                    stepper.run(todo, cf, opts)
                    todo = 0
                    out.append(stepper.get_state()[0])
                    record.pop(0)
            if todo:
                stepper.run(todo, cf, opts)
            return out
        for ct in trange[1:]:
            # b_v = M v + dt*rhs(t, v): `rhs` plays the role of the convection
            stepper.step(cf, nfc_new=rhsv(ct, cv), opts=opts)
            cv, _ = stepper.get_state()
            if len(record) > 0 and ct == record[0]:
                out.append(cv)
                record.pop(0)
    finally:
        _record_run('semi_implicit_euler', system, stepper)
        stepper.close()
        system.close()
    return out
