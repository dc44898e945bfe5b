"""ORACLE (test infrastructure, never shipped, never the thing measured).

CPU (NumPy/SciPy-SuperLU) restatement of the reference's semi-explicit time
integrators, `/root/reference/dolfin_navier_scipy/time_int_utils.py`:

 * `time_grid`            <- `_inittimegrid`/`_checkuniformgrid`  tiu:358-363,480-489
 * `heun_start`           <- `_onestepheun`                        tiu:366-477
 * `cnab`                 <- `cnab`                                tiu:23-145
 * `sbdftwo`              <- `sbdftwo`                             tiu:260-355
 * `semi_implicit_euler`  <- `semi_implicit_euler`                 tiu:566-635

Keyword interfaces equal the reference's so that a test can feed one set of
kwargs to the reference module (when generating the golden fixtures), to this
oracle, and to the HIP path.  Quirks are reproduced on purpose and flagged
`QUIRK`.

Pinning: `tests/golden/make_golden.py` executes the reference's own
`time_int_utils.py` (loaded by file path, with `oracle/saddle_oracle.py`
standing in for the un-vendored `lau`) on seeded inputs and stores inputs and
outputs in `tests/golden/imex_*.npz`; `tests/test_oracle_golden.py` checks this
restatement against those vectors.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.
"""
import numpy as np
import scipy.sparse as sps

from . import saddle_oracle as lau

__all__ = ['time_grid', 'heun_start', 'cnab', 'sbdftwo',
           'semi_implicit_euler']


def time_grid(trange, ntimeslices=10):
    """uniform-grid check and slicing of `trange[2:]` (tiu:480-489)"""
    tr = np.asarray(trange, dtype=np.float64)
    steps = np.diff(tr)
    if not np.allclose(np.linalg.norm(np.diff(steps)), 0):
        raise NotImplementedError('only equidistant time grids')
    dt = trange[1] - trange[0]
    rest = tr[2:]
    chunk = int(np.floor(rest.size/ntimeslices))
    slices = [rest[k*chunk:(k+1)*chunk].tolist() for k in range(ntimeslices)]
    slices.append(rest[ntimeslices*chunk:].tolist())
    return dt, slices


def _defaults(NV, dynamic_rhs, f_tvdp, f_vdp):
    zerorhs = np.zeros((NV, 1))
    if dynamic_rhs is None:
        def dynamic_rhs(t, vc=None, memory={}, mode=None):
            return zerorhs, memory
    if f_tvdp is not None:
        _inner = dynamic_rhs

        def dynamic_rhs(t, vc=None, memory={}, mode=None):
            val, mem = _inner(t, vc=vc, memory=memory, mode=mode)
            return val + f_tvdp(t, vc), mem
    if f_vdp is None:
        def f_vdp(vvec):
            return zerorhs
    return dynamic_rhs, f_vdp


def heun_start(vc=None, pc=None, tc=None, tn=None, M=None, A=None, J=None,
               scalep=1., dfv_c=None, dynamic_rhs=None, drm={},
               bcs_c=None, applybcs=None, appndbcs=None, getbcs=None,
               f_tdp=None, f_vdp=None, g_tdp=None, solve=None):
    """first step: IMEX-Euler predictor + trapezoidal corrector (tiu:366-477)

    QUIRK (tiu:463-466): the corrector is solved with `amat=M`, not with
    `M + dt/2 A`, although `-dt/2 A (vc + tv)` sits on the right-hand side.
    """
    solve = lau.solve_sadpnt_smw if solve is None else solve
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
    # predictor
    prhs = M @ vc + dt*(fv_n + tbfv_n + tdfv_n) + dt*nfc_c - (tmbc_n - mbc_c)
    tvp = solve(amat=M + dt*A, jmat=J, jmatT=JT, rhsv=prhs,
                rhsp=fp_n + tbfp_n)
    tv_n = tvp[:NV, :]
    tp_n = 1./dt*scalep*tvp[NV:, :]
    # corrector
    dfv_n, drm = dynamic_rhs(tn, vc=tv_n, memory=drm, mode='heuncorr')
    tnfc_n = f_vdp(appndbcs(tv_n, tbcs))
    bcs_n = getbcs(tn, appndbcs(tv_n, tbcs), tp_n, mode='heuncorr')
    bfv_n, bfp_n, mbc_n = applybcs(bcs_n)
    crhs = M @ vc - (mbc_n - mbc_c) - .5*dt*(A @ (vc + tv_n)) \
        + .5*dt*(fv_c + fv_n + bfv_n + bfv_c + dfv_n + dfv_c + nfc_c + tnfc_n)
    vp = solve(amat=M, jmat=J, jmatT=JT, rhsv=crhs, rhsp=fp_n + bfp_n)
    v_n = vp[:NV].reshape((NV, 1))
    p_n = 1./dt*scalep*vp[NV:].reshape((NP, 1))
    nfc_n = f_vdp(appndbcs(v_n, bcs_n))
    return (v_n, p_n, bcs_n, bfv_n, mbc_c, mbc_n, fv_n, nfc_c, nfc_n, dfv_n,
            drm)


def cnab(trange=None, inivel=None, inip=None, bcs_ini=[],
         M=None, A=None, J=None, f_vdp=None, f_tdp=None, g_tdp=None,
         f_tvdp=None, scalep=-1., getbcs=None, applybcs=None, appndbcs=None,
         savevp=None, dynamic_rhs=None, dynamic_rhs_memory={},
         check_ff_maxv=None, ntimeslices=10, verbose=False):
    """Crank-Nicolson (diffusion) / Adams-Bashforth 2 (convection) (tiu:23-145)
    """
    dt, slices = time_grid(trange, ntimeslices=ntimeslices)
    NP, NV = J.shape
    ffflag = 0
    dynamic_rhs, f_vdp = _defaults(NV, dynamic_rhs, f_tvdp, f_vdp)
    dfv_c, drm = dynamic_rhs(trange[0], vc=inivel, memory=dynamic_rhs_memory,
                             mode='init')
    savevp(appndbcs(inivel, bcs_ini), inip, time=trange[0])
    (v_n, p_n, bcs_n, bfv_n, mbc_c, mbc_n, fv_n, nfc_c, nfc_n, dfv_n,
     drm) = heun_start(vc=inivel, pc=inip, tc=trange[0], tn=trange[1],
                       M=M, A=A, J=J, scalep=scalep, dfv_c=dfv_c,
                       dynamic_rhs=dynamic_rhs, drm=drm, bcs_c=bcs_ini,
                       applybcs=applybcs, appndbcs=appndbcs, getbcs=getbcs,
                       f_tdp=f_tdp, f_vdp=f_vdp, g_tdp=g_tdp)
    savevp(appndbcs(v_n, bcs_n), p_n, time=trange[1])
    klu = lau.SaddleLU(M + .5*dt*A, J)           # tiu:89-91, factor once
    for tslice in slices:
        nrmvc = np.linalg.norm(v_n)
        if nrmvc > check_ff_maxv or np.isnan(nrmvc):     # tiu:99-103
            ffflag = 1
            break
        for ctime in tslice:
            v_c, p_c = v_n, p_n
            bcs_c, bfv_c, mbc_c = bcs_n, bfv_n, mbc_n
            fv_c, dfv_c = fv_n, dfv_n
            nfc_o = nfc_c
            nfc_c = f_vdp(appndbcs(v_c, bcs_c))
            bcs_n = getbcs(ctime, appndbcs(v_c, bcs_c), p_c, mode='abtwo')
            bfv_n, bfp_n, mbc_n = applybcs(bcs_n)
            fv_n, fp_n = f_tdp(ctime), g_tdp(ctime)
            dfv_n, drm = dynamic_rhs(ctime, vc=v_c, memory=drm, mode='abtwo')
            rhs_n = M @ v_c - .5*dt*(A @ v_c) - (mbc_n - mbc_c) \
                + .5*dt*(3*nfc_c - nfc_o) \
                + .5*dt*(fv_c + fv_n + bfv_n + bfv_c + dfv_n + dfv_c)
            vp_n = klu(np.vstack([rhs_n, fp_n + bfp_n]).flatten())
            v_n = vp_n[:NV].reshape((NV, 1))
            p_n = 1./dt*scalep*vp_n[NV:].reshape((NP, 1))
            savevp(appndbcs(v_n, bcs_n), p_n, time=ctime)
    return v_n, p_n, ffflag


def sbdftwo(trange=None, inivel=None, inip=None, bcs_ini=[],
            M=None, A=None, J=None, f_vdp=None, f_tdp=None, g_tdp=None,
            check_ff=False, check_ff_maxv=None, scalep=-1.,
            getbcs=None, applybcs=None, appndbcs=None, savevp=None,
            dynamic_rhs=None, dynamic_rhs_memory={},
            ntimeslices=10, verbose=False):
    """semi-implicit BDF2 with extrapolated convection (tiu:260-355)

    QUIRKS: the pressure is rescaled by `1/dt` although the implicit weight
    is `2/3 dt` (tiu:351); the blow-up guard looks at `v_c`, the value *before*
    the last step (tiu:311); `bfv`, `fv`, `dfv` enter at the new time only
    (tiu:342-346).
    """
    dt, slices = time_grid(trange, ntimeslices=ntimeslices)
    NP, NV = J.shape
    dynamic_rhs, f_vdp = _defaults(NV, dynamic_rhs, None, f_vdp)
    dfv_c, drm = dynamic_rhs(trange[0], vc=inivel, memory=dynamic_rhs_memory,
                             mode='init')
    savevp(appndbcs(inivel, bcs_ini), inip, time=trange[0])
    v_c = inivel
    (v_n, p_n, bcs_n, bfv_n, mbc_c, mbc_n, fv_n, nfc_c, nfc_n, dfv_n,
     drm) = heun_start(vc=v_c, pc=inip, tc=trange[0], tn=trange[1],
                       M=M, A=A, J=J, scalep=scalep, dfv_c=dfv_c,
                       dynamic_rhs=dynamic_rhs, drm=drm, bcs_c=bcs_ini,
                       applybcs=applybcs, appndbcs=appndbcs, getbcs=getbcs,
                       f_tdp=f_tdp, f_vdp=f_vdp, g_tdp=g_tdp)
    savevp(appndbcs(v_n, bcs_n), p_n, time=trange[1])
    klu = lau.SaddleLU(M + 2./3*dt*A, J)         # tiu:304-306
    ffflag = 0
    for tslice in slices:
        nrmvc = np.linalg.norm(v_c)
        if nrmvc > check_ff_maxv or np.isnan(nrmvc):
            ffflag = 1
            break
        for ctime in tslice:
            v_p, mbc_p = v_c, mbc_c
            v_c, p_c = v_n, p_n
            bcs_c, mbc_c = bcs_n, mbc_n
            dfv_c = dfv_n
            nfc_p = nfc_c
            nfc_c = f_vdp(appndbcs(v_c, bcs_c))
            bcs_n = getbcs(ctime, appndbcs(v_c, bcs_c), p_c, mode='abtwo')
            bfv_n, bfp_n, mbc_n = applybcs(bcs_n)
            fv_n, fp_n = f_tdp(ctime), g_tdp(ctime)
            dfv_n, drm = dynamic_rhs(ctime, vc=v_c, memory=drm, mode='abtwo')
            rhs_n = 1/3*(M @ (4*v_c - v_p)) \
                - (mbc_n - 4/3*mbc_c + 1/3*mbc_p) \
                + 2/3*dt*bfv_n + 2/3*dt*(2*nfc_c - nfc_p) \
                + 2/3*dt*(fv_n + dfv_n)
            vp_n = klu(np.vstack([rhs_n, fp_n + bfp_n]).flatten())
            v_n = vp_n[:NV].reshape((NV, 1))
            p_n = 1./dt*scalep*vp_n[NV:].reshape((NP, 1))
            savevp(appndbcs(v_n, bcs_n), p_n, time=ctime)
    return v_n, p_n, ffflag


def semi_implicit_euler(iniv=None, jmat=None, mmat=None, amat=None, rhsv=None,
                        trange=None, data_trange=None, fp=None):
    """`M v' + A v + J^T p = rhs(t, v)`, `J v = fp`; linear part implicit
    (tiu:566-635); returns the list of velocities at `data_trange`"""
    record = list(np.copy(trange if data_trange is None else data_trange))
    record.pop(0)
    NP, NV = jmat.shape
    fpz = np.zeros((NP, 1)) if fp is None else fp
    dt = trange[1] - trange[0]
    _, klu = lau.solve_sadpnt_smw(amat=mmat + dt*amat, jmat=jmat,
                                  rhsv=0*iniv, return_alu=True)
    out = [iniv]
    cv = iniv
    for ct in trange[1:]:
        rhs = (mmat @ cv).reshape((-1, 1)) + dt*rhsv(ct, cv)
        cv = klu(np.vstack([rhs, fpz]))[:NV]
        if len(record) > 0 and ct == record[0]:
            out.append(cv)
            record.pop(0)
    return out
