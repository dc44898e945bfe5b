"""Algebraic half of the reference's `stokes_navier_utils` on MI355X.

`solve_nse` (reference snu:548-1600) is, once FEniCS has produced the
matrices, pure index/linear algebra: restrict the system to the inner dofs
that carry no controlled Dirichlet value (snu:729-770), build the closures the
integrators call back (`applybcs`, `rhsv`, `rhsp`, `nonlvfunc`, `getbcs`,
`_appbcs`; snu:1003-1006, 1103-1157), and dispatch to `cnab` / `sbdftwo`
(explicit nonlinearity) or to the Newton/Picard trapezoidal sweeps
(snu:1304-1587).  This module is that half, with every saddle-point solve,
every convection evaluation and the time loops on the device:

 * `solve_nse`              snu:548-1600   (same keyword names and meaning)
 * `solve_steadystate_nse`  snu:240-546    (what `tests/mini_setup.py` runs)
 * `get_v_conv_conts`       snu:40-133
 * `get_pfromv`             snu:1602-1633
 * `m_innerproduct`         snu:136-143
 * `_localizecdbinds`, `_comp_cntrl_bcvals`, `_cntrl_stffnss_rhs`  snu:146-218

What stands in for dolfin: `V` is any object with the P2 element data of the
velocity space -- `vdim`, `_vdofs()` (cell -> 12 dofs), `glam`, `area` -- e.g.
`fem.TaylorHood`; file based result dictionaries (`dou.save_npa`) are replaced
by in-memory dictionaries of arrays (the device trajectory store);
paraview output and the observer/feedback helpers are out of scope
(SURVEY.md section 2).
"""
import logging

import numpy as np
import scipy.sparse as sps

from . import lin_alg_utils as lau
from . import time_int_utils as tiu
from . import bcs as dbc
from .convection import ConvectionP2

__all__ = ['solve_nse', 'solve_steadystate_nse', 'get_v_conv_conts',
           'get_pfromv', 'm_innerproduct', 'clear_cache']

_conv_cache = {}


_warn_ignored = lau.warn_ignored


def clear_cache():
    for cv in _conv_cache.values():
        cv.close()
    _conv_cache.clear()
    lau.clear_cache()


def m_innerproduct(M, v1, v2=None):
    """inner product with a spd sparse matrix (snu:136-143)"""
    if v2 is None:
        v2 = v1
    return np.dot(v1.T, M @ v2)


def _unroll(bcinds, bcvals):
    """`dts.unroll_dlfn_dbcs` for index/value lists (dts:26-46)"""
    if bcinds is None or len(bcinds) == 0:
        return [], []
    if not isinstance(bcinds[0], (list, tuple, np.ndarray)):
        return list(bcinds), list(bcvals)
    ui, uv = [], []
    for k, cbci in enumerate(bcinds):
        ui.extend(list(cbci))
        uv.extend(list(bcvals[k]))
    return ui, uv


def _localizecdbinds(cdbinds, V, invinds):
    """where the full-space dof numbers `cdbinds` sit inside the vector of
    inner dofs -- the matrices at hand have lost the constant Dirichlet dofs
    already, the controlled ones are still numbered with respect to `V`
    (snu:146-161).  `invinds` is ascending, so a position is a binary search;
    without a space the reference indexes `invinds` by itself, kept as it is."""
    inner = np.asarray(invinds)
    if V is None:
        inner = inner[inner]
    return np.searchsorted(inner, np.asarray(cdbinds), side='left')


def _ask_control(func, time, vel, p, mode, memory):
    """`(value, new memory)` of one control function; `mode=` is only passed
    to functions that take it (the reference's protocol, snu:170-177)"""
    try:
        return func(time, vel=vel, p=p, mode=mode, memory=memory)
    except TypeError:
        return func(time, vel=vel, p=p, memory=memory)


def _comp_cntrl_bcvals(diricontbcvals=[], diricontfuncs=[], mode=None,
                       diricontfuncmems=[], time=None, vel=None, p=None, **kw):
    """values of the controlled Dirichlet dofs at `time`: per controlled
    boundary the scalar its control function returns times the boundary's shape
    values; the functions' memories are updated in place (snu:164-183).  Lists
    left at `None` mean "no controls": like the reference, a `TypeError` ends
    the walk and what has been collected so far is returned."""
    out = []
    try:
        for k, shape in enumerate(diricontbcvals):
            gain, diricontfuncmems[k] = _ask_control(
                diricontfuncs[k], time, vel, p, mode, diricontfuncmems[k])
            out.extend(gain*sv for sv in shape)
    except TypeError:
        pass
    return out


def _cntrl_stffnss_rhs(loccntbcinds=None, cntrlldbcvals=None, A=None, J=None,
                       invinds_loc=None, **kw):
    """contribution of the controlled boundary values to the right-hand sides,
    `-A[:, cnt] vals` (inner rows) and `-J[:, cnt] vals` (snu:186-202,
    dts:475-573 with `get_rhs_only`)"""
    NV = A.shape[0]
    if loccntbcinds is None or len(loccntbcinds) == 0:
        return np.zeros((len(invinds_loc), 1)), np.zeros((J.shape[0], 1))
    aux = np.zeros((NV, 1))
    aux[loccntbcinds, 0] = cntrlldbcvals
    return -(A @ aux)[invinds_loc, :], -(J @ aux)


def _conv_operator(V, invinds, dbcinds, device=0):
    """device convection operator of the space for these index sets (cached:
    the element tables and inverted indices are built once)"""
    inv = np.ascontiguousarray(invinds, dtype=np.int32)
    dbi = np.ascontiguousarray(dbcinds, dtype=np.int32)
    key = (id(V), inv.size, dbi.size, hash(inv.tobytes()), hash(dbi.tobytes()),
           device)
    cv = _conv_cache.get(key)
    if cv is None:
        cv = ConvectionP2.from_taylor_hood(V, inv, dbi, np.zeros(dbi.size),
                                           device=device)
        cv.bind_pattern(cv.connectivity())
        cv._Vref = V                 # keep `id(V)` alive with the entry
        _conv_cache[key] = cv
        while len(_conv_cache) > 4:
            _conv_cache.pop(next(iter(_conv_cache))).close()
    return cv


def get_v_conv_conts(vvec=None, V=None, invinds=None, dbcvals=[], dbcinds=[],
                     semi_explicit=False, Picard=False, retparts=False,
                     device=0):
    """condensed linearised convection, on the device (snu:40-133)

    Returns `(convc_mat, rhs_con, rhsv_conbc)`: the matrix `N1(v)` (Picard) or
    `N1(v) + N2(v)` (Newton) on the inner dofs, `N(v)v` on the inner dofs, and
    `-N[:, bc] bcvals`.  `semi_explicit`: `(0., -N(v)v, 0.)`.  If `vvec`
    already contains the boundary values (length `V.vdim`) they define the
    convection field and `dbcvals` only enter the rhs contribution."""
    bci, bcv = _unroll(dbcinds, dbcvals)
    cv = _conv_operator(V, invinds, bci, device=device)
    vvec = np.asarray(vvec, dtype=np.float64)
    if vvec.size == V.vdim:
        ve = vvec.reshape(-1)
        vin, linvals = ve[invinds], ve[bci]
    else:
        vin, linvals = vvec.reshape(-1), np.asarray(bcv, dtype=np.float64)
    if semi_explicit:
        cv.set_dbcvals(linvals)
        return 0., cv.apply(vin, scale=-1.0), 0.
    rhsvals = np.asarray(bcv, dtype=np.float64)
    if Picard:
        N, rbc, _ = cv.assemble(vin, newton=False, dbcvals_lin=linvals,
                                dbcvals_rhs=rhsvals)
        return N, None, rbc
    if retparts:
        N1, rbc1, rcon = cv.assemble(vin, newton=False, dbcvals_lin=linvals,
                                     dbcvals_rhs=rhsvals)
        N12, rbc12, _ = cv.assemble(vin, newton=True, dbcvals_lin=linvals,
                                    dbcvals_rhs=rhsvals)
        N2 = sps.csr_matrix((N12.data - N1.data, N1.indices, N1.indptr),
                            shape=N1.shape)
        return (N1, N2), rcon, (rbc1, rbc12 - rbc1)
    N, rbc, rcon = cv.assemble(vin, newton=True, dbcvals_lin=linvals,
                               dbcvals_rhs=rhsvals)
    return N, rcon, rbc


def get_pfromv(v=None, V=None, M=None, A=None, J=None, fv=None, fp=None,
               decouplevp=False, solve_M=None, symmetric=False, cgtol=1e-8,
               stokes_flow=False, dbcinds=None, dbcvals=None, invinds=None,
               **kwargs):
    """the pressure that belongs to a velocity (snu:1602-1633): one saddle
    solve with `amat=M` -- or, `decouplevp and symmetric`, CG on the Schur
    complement with the caller's `solve_M` (amat omitted, snu:1622)"""
    if stokes_flow:
        rhs_con = 0.
    else:
        _, rhs_con, _ = get_v_conv_conts(vvec=v, V=V, invinds=invinds,
                                         dbcinds=dbcinds, dbcvals=dbcvals)
    rhsv = -(A @ v) - rhs_con + fv
    if decouplevp and symmetric:
        vp = lau.solve_sadpnt_smw(jmat=J, jmatT=J.T, decouplevp=decouplevp,
                                  solve_A=solve_M, symmetric=symmetric,
                                  cgtol=cgtol, rhsv=rhsv)
    else:
        vp = lau.solve_sadpnt_smw(amat=M, jmat=J, jmatT=J.T,
                                  decouplevp=decouplevp, solve_A=solve_M,
                                  symmetric=symmetric, cgtol=cgtol, rhsv=rhsv)
    return -vp[J.shape[1]:, :]


def _restrict(A, M, J, JT, fv, fp, V, invinds, dbcinds, dbcvals,
              diricontbcinds):
    """index restriction of snu:729-770 / snu:370-390: the dofs carrying a
    controlled Dirichlet value leave the system"""
    JT = sps.csr_matrix(J.T) if JT is None else JT
    loccntbcinds, glbcntbcinds = [], []
    if diricontbcinds is None or len(diricontbcinds) == 0:
        dbcntinvinds = np.asarray(invinds)
    else:
        for cdbidbv in diricontbcinds:
            loccntbcinds.extend(_localizecdbinds(cdbidbv, V, invinds).tolist())
            glbcntbcinds.extend(list(cdbidbv))
        dbcntinvinds = np.setdiff1d(invinds, glbcntbcinds).astype(np.int32)
    locinvinds = _localizecdbinds(dbcntinvinds, V, invinds).tolist()
    cnv = dbcntinvinds.size
    NP = J.shape[0]
    fv = np.zeros((len(invinds), 1)) if fv is None else fv
    fp = np.zeros((NP, 1)) if fp is None else fp
    A, M, J, JT = (sps.csr_matrix(A), sps.csr_matrix(M), sps.csr_matrix(J),
                   sps.csr_matrix(JT))
    out = dict(cmmat=M[locinvinds, :][:, locinvinds].tocsr(),
               camat=A[locinvinds, :][:, locinvinds].tocsr(),
               cjt=JT[locinvinds, :].tocsr(), cj=J[:, locinvinds].tocsr(),
               cfv=np.asarray(fv)[locinvinds], cfp=fp, cnv=cnv,
               loccntbcinds=loccntbcinds, glbcntbcinds=glbcntbcinds,
               dbcntinvinds=dbcntinvinds, locinvinds=locinvinds,
               A=A, M=M, J=J)
    for mat in (out['cmmat'], out['camat'], out['cjt'], out['cj']):
        mat.sort_indices()
    return out


def solve_steadystate_nse(A=None, J=None, JT=None, M=None, fv=None, fp=None,
                          V=None, invinds=None, dbcvals=None, dbcinds=None,
                          diricontbcinds=None, diricontbcvals=None,
                          diricontfuncs=None, diricontfuncmems=None,
                          return_vp=False, ppin=None,
                          return_nwtnupd_norms=False, only_stokes=False,
                          vel_pcrd_stps=10, vel_pcrd_tol=1e-4,
                          vel_nwtn_stps=20, vel_nwtn_tol=5e-15,
                          vel_start_nwtn=None, verbose=False, krylov=None,
                          krpslvprms={}, krplsprms={}, **kw):
    """steady Navier-Stokes by Picard + Newton iterations (snu:240-546); every
    linear system `[[A + N(v_k), J^T],[J, 0]]` is solved on the GPU through
    `lin_alg_utils.solve_sadpnt_smw` (snu:401,458,497)"""
    _warn_ignored('solve_steadystate_nse', kw)
    dbcinds, dbcvals = _unroll(dbcinds, dbcvals)
    rs = _restrict(A, M, J, JT, fv, fp, V, invinds, dbcinds, dbcvals,
                   diricontbcinds)
    cmmat, camat, cj, cjt = rs['cmmat'], rs['camat'], rs['cj'], rs['cjt']
    cfv, cfp, cnv = rs['cfv'], rs['cfp'], rs['cnv']
    glb, dbcnt = rs['glbcntbcinds'], rs['dbcntinvinds']
    cdict = dict(A=rs['A'], J=rs['J'], loccntbcinds=rs['loccntbcinds'],
                 invinds_loc=rs['locinvinds'], diricontbcvals=diricontbcvals,
                 diricontfuncs=diricontfuncs,
                 diricontfuncmems=diricontfuncmems)
    norm_nwtnupd_list = []
    slv = dict(krylov=krylov, krpslvprms=krpslvprms, krplsprms=krplsprms)

    def _appbcs(vvec, ccntrlldbcvals):
        return dbc.append_bcs_vec(vvec, vdim=V.vdim, invinds=dbcnt,
                                  bcinds=[dbcinds, glb],
                                  bcvals=[dbcvals, ccntrlldbcvals])
    bcsets = dict(invinds=dbcnt, dbcinds=[dbcinds, glb])
    if vel_start_nwtn is None or only_stokes:
        cdbcvals_c = _comp_cntrl_bcvals(time=None, vel=None, p=None,
                                        mode='init', **cdict)
        ccfv, ccfp = _cntrl_stffnss_rhs(cntrlldbcvals=cdbcvals_c, **cdict)
        vp_k = lau.solve_sadpnt_smw(amat=camat, jmat=cj, jmatT=cjt,
                                    rhsv=cfv+ccfv, rhsp=cfp+ccfp, **slv)
        vp_k[cnv:] = -vp_k[cnv:]          # pressure was flipped for symmetry
        vel_k, p_k = vp_k[:cnv, ], vp_k[cnv:, ]
    else:
        cdbcvals_c = vel_start_nwtn[glb, :].flatten().tolist()
        vel_k = vel_start_nwtn[dbcnt, :]
        p_k = np.zeros((J.shape[0], 1))
        vp_k = np.vstack([vel_k, p_k])
    norm_nwtnupd = None
    for k in range(vel_pcrd_stps):
        if only_stokes:
            break
        cdbcvals_n = _comp_cntrl_bcvals(vel=_appbcs(vel_k, cdbcvals_c), p=p_k,
                                        **cdict)
        ccfv_n, ccfp_n = _cntrl_stffnss_rhs(cntrlldbcvals=cdbcvals_n, **cdict)
        # the old boundary values define the convection field, the new ones
        # the rhs contribution (snu:446-455)
        pcrdcnvmat, _, rhsv_conbc = get_v_conv_conts(
            vvec=_appbcs(vel_k, cdbcvals_c), V=V, Picard=True,
            dbcvals=[dbcvals, cdbcvals_n], **bcsets)
        vp_k = lau.solve_sadpnt_smw(amat=camat+pcrdcnvmat, jmat=cj, jmatT=cjt,
                                    rhsv=cfv+ccfv_n+rhsv_conbc,
                                    rhsp=cfp+ccfp_n, **slv)
        normpicupd = np.sqrt(m_innerproduct(cmmat, vel_k-vp_k[:cnv, ]))[0]
        if verbose:
            logging.info('Picard iteration: {0} -- norm of update: {1}'.
                         format(k+1, normpicupd))
        vel_k = vp_k[:cnv, ]
        vp_k[cnv:] = -vp_k[cnv:]
        if normpicupd < vel_pcrd_tol:
            break
    for vel_newtk in range(vel_nwtn_stps):
        if only_stokes:
            break
        cdbcvals_n = _comp_cntrl_bcvals(vel=_appbcs(vel_k, cdbcvals_c), p=p_k,
                                        **cdict)
        ccfv_n, ccfp_n = _cntrl_stffnss_rhs(cntrlldbcvals=cdbcvals_n, **cdict)
        convc_mat, rhs_con, rhsv_conbc = get_v_conv_conts(
            vvec=_appbcs(vel_k, cdbcvals_c), V=V,
            dbcvals=[dbcvals, cdbcvals_n], **bcsets)
        vp_k = lau.solve_sadpnt_smw(amat=camat+convc_mat, jmat=cj, jmatT=cjt,
                                    rhsv=cfv+ccfv_n+rhs_con+rhsv_conbc,
                                    rhsp=cfp+ccfp_n, **slv)
        norm_nwtnupd = np.sqrt(m_innerproduct(cmmat, vel_k - vp_k[:cnv, :]))[0]
        norm_nwtnupd_list.append(norm_nwtnupd)
        vel_k = vp_k[:cnv, ]
        vp_k[cnv:] = -vp_k[cnv:]
        p_k = vp_k[cnv:, ]
        cdbcvals_c = cdbcvals_n
        if verbose:
            logging.info('Steady State NSE: Newton iteration: {0} -- norm of '
                         'update: {1}'.format(vel_newtk, norm_nwtnupd))
        if norm_nwtnupd < vel_nwtn_tol:
            break
    else:
        if vel_nwtn_stps > 0 and not only_stokes:
            raise UserWarning('Steady State NSE: Newton has not converged')
    vwc = _appbcs(vel_k, cdbcvals_c).reshape((V.vdim, 1))
    retthing = (vwc, vp_k[cnv:, :]) if return_vp else vwc
    if return_nwtnupd_norms:
        return retthing, norm_nwtnupd_list
    return retthing


def solve_nse(A=None, M=None, J=None, JT=None, fv=None, fp=None, fvtd=None,
              fvss=0., fvtvd=None, iniv=None, inip=None, lin_vel_point=None,
              stokes_flow=False, trange=None, t0=None, tE=None, Nts=None,
              time_int_scheme='cnab', V=None, invinds=None, dbcinds=None,
              dbcvals=None, diricontbcinds=None, diricontbcvals=None,
              diricontfuncs=None, diricontfuncmems=None, ppin=None,
              vel_nwtn_stps=20, vel_nwtn_tol=5e-15, nsects=1,
              loc_nwtn_tol=5e-15, loc_pcrd_stps=True, addfullsweep=False,
              vel_pcrd_stps=4, krylov=None, krpslvprms={}, krplsprms={},
              treat_nonl_explicit=True, use_custom_nonlinearity=False,
              custom_nonlinear_vel_function=None, datatrange=None,
              dataoutpnts=None, return_final_vp=False, return_vp_dict=False,
              return_y_list=False, return_dictofvelstrs=False,
              return_dictofpstrs=False, cv_mat=None, check_ff=False,
              check_ff_maxv=1e8, verbose=False, start_ssstokes=False,
              closed_loop=False, dynamic_feedback=False, static_feedback=False,
              b_mat=None, feedbackthroughdict=None,
              vp_output=False, vp_out_fun=None, vp_output_dict=None,
              solver=None, device=0, bcs_time_only=False,
              applybcs_literal=True, **kw):
    """time-dependent Navier-Stokes on the device (reference snu:548-1600)

    Keyword names and meaning follow the reference.  `V`: object with the P2
    element data (module docstring).  Differences: result dictionaries hold
    arrays instead of file names.  `closed_loop` in the Newton/Picard sweeps
    (snu:1367-1384, 1461-1483, 1036-1040): `b_mat` (NV x r, inner dofs) and
    `feedbackthroughdict` -- `{t: dict(mtxtb=(NV, r) array, w=(NV, 1) array)}`,
    key `None` with `static_feedback`, key `0` for the initial instance
    otherwise (the name the reference's loop reads; it is bound nowhere in
    the reference at this commit) -- add `b_mat (b_mat^T w)` to the right-hand
    sides and the low-rank term `dt/2 b_mat mtxtb^T` to the system
    (`TrapezoidalStepper.step(feedback=)`, Sherman-Morrison-Woodbury on the
    device).  In the explicit schemes the reference's static feedback does
    nothing (snu:1261-1262) and neither does it here; `dynamic_feedback`
    (observer helpers of `tiu`, SURVEY section 2: out of scope) raises.
    Extra keywords: `solver`
    (overrides `time_int_utils.SOLVER`), `device`, `bcs_time_only` (the
    control functions `diricontfuncs` ignore `vel`/`p`: the explicit loop may
    then run device resident over whole time slices), `applybcs_literal`
    (default True: the `applybcs` closure returns zeros exactly as the
    reference's does with snu:1112 commented out; False: the controlled values
    are written into the auxiliary vector, see `bcs.make_applybcs`).
    """
    if dynamic_feedback:
        raise NotImplementedError('observer-based (dynamic) feedback is '
                                  'outside the MI355X path (SURVEY.md '
                                  'section 2)')
    if closed_loop and not treat_nonl_explicit and \
            (b_mat is None or feedbackthroughdict is None):
        raise ValueError('`closed_loop` in the Newton/Picard sweeps needs '
                         '`b_mat` and `feedbackthroughdict`')
    _warn_ignored('solve_nse', kw)
    if trange is None:
        trange = np.linspace(t0, tE, Nts+1)
    trange = np.asarray(trange, dtype=np.float64)
    if treat_nonl_explicit and lin_vel_point is not None:
        raise UserWarning('cant use `lin_vel_point` and explicit treatment of '
                          'the nonlinearity')
    dbcinds, dbcvals = _unroll(dbcinds, dbcvals)
    rs = _restrict(A, M, J, JT, fv, fp, V, invinds, dbcinds, dbcvals,
                   diricontbcinds)
    cmmat, camat, cj, cjt = rs['cmmat'], rs['camat'], rs['cj'], rs['cjt']
    cfv, cfp, cnv = rs['cfv'], rs['cfp'], rs['cnv']
    glb, dbcnt = rs['glbcntbcinds'], rs['dbcntinvinds']
    loccnt, locinv = rs['loccntbcinds'], rs['locinvinds']
    Afull, Mfull, Jfull = rs['A'], rs['M'], rs['J']
    NP = cj.shape[0]
    vdim = cnv if V is None else V.vdim
    cdict = dict(A=Afull, J=Jfull, loccntbcinds=loccnt, invinds_loc=locinv,
                 diricontbcvals=diricontbcvals, diricontfuncs=diricontfuncs,
                 diricontfuncmems=diricontfuncmems)
    slv = dict(krylov=krylov, krpslvprms=krpslvprms, krplsprms=krplsprms)
    if datatrange is None and dataoutpnts is None:
        datatrange = np.copy(trange).tolist()
    elif datatrange is None:
        h = (trange.size-1)/(dataoutpnts-1)
        datatrange = trange[[int(np.floor(h*i))
                             for i in range(dataoutpnts)]].tolist()
    else:
        datatrange = list(datatrange)

    # ---- the initial value (snu:833-925) --------------------------------
    if iniv is None:
        if not start_ssstokes:
            raise ValueError('No initial value given')
        inicdbcvals = _comp_cntrl_bcvals(time=trange[0], vel=None, p=None,
                                         mode='stokes', **cdict)
        ccfv, ccfp = _cntrl_stffnss_rhs(cntrlldbcvals=inicdbcvals, **cdict)
        vp_stokes = lau.solve_sadpnt_smw(amat=camat, jmat=cj, jmatT=cjt,
                                         rhsv=cfv+ccfv+fvss, rhsp=cfp+ccfp,
                                         **slv)
        iniv = vp_stokes[:cnv].reshape((-1, 1))
    else:
        iniv = np.asarray(iniv, dtype=np.float64).reshape((-1, 1))
        inicdbcvals = (iniv[glb].flatten()).tolist()
        iniv = iniv[dbcnt] if iniv.shape[0] == vdim else iniv
        ccfv, ccfp = _cntrl_stffnss_rhs(cntrlldbcvals=inicdbcvals, **cdict)
    bcsets = dict(invinds=dbcnt, dbcinds=[dbcinds, glb])
    if inip is None:
        # QUIRK kept (snu:909-915): `A=cmmat` -- the mass matrix in the place
        # of the stiffness matrix; only seeds saved/observed pressures at t0
        inip = get_pfromv(v=iniv, V=V, M=cmmat, A=cmmat, J=cj,
                          fv=cfv+ccfv+fvss, fp=cfp+ccfp,
                          stokes_flow=stokes_flow,
                          dbcvals=[dbcvals, inicdbcvals], **bcsets)

    def _appbcs(vvec, ccntrlldbcvals):
        return dbc.append_bcs_vec(vvec, vdim=vdim, invinds=dbcnt,
                                  bcinds=[dbcinds, glb],
                                  bcvals=[dbcvals, ccntrlldbcvals])

    if stokes_flow:
        vel_nwtn_stps, vel_pcrd_stps = 1, 0

    vp_dict, ylist, veldict, pdict = {}, [], {}, {}

    def _record(vfull, pvec, time=None):
        """`_svpplz` (snu:1161-1226) without the file/paraview side"""
        if vp_output:
            vp_output_dict.update({time: vp_out_fun(vfull, pvec, time=None)})
        if return_vp_dict:
            vp_dict.update({time: dict(p=pvec, v=vfull)})
            return
        try:
            if not time == datatrange[0]:
                return
            datatrange.pop(0)
        except IndexError:
            return
        if return_dictofvelstrs:
            veldict[time] = vfull
            pdict[time] = pvec
        elif return_y_list:
            if cv_mat is None:
                ylist.append(vfull)
            else:
                try:
                    ylist.append(cv_mat.dot(vfull[dbcnt]))
                except ValueError:
                    ylist.append(cv_mat.dot(vfull))

    # =====================================================================
    # explicit treatment of the nonlinearity: Heun start + CNAB / SBDF2
    # (snu:1100-1302)
    # =====================================================================
    if lin_vel_point is None:
        applybcs = dbc.make_applybcs(Afull, Jfull, Mfull, loccnt, locinv,
                                     device=device,
                                     reference_literal=applybcs_literal)
        if fvtd is None:
            def rhsv(t):
                return cfv
        else:
            def rhsv(t):
                return cfv + np.asarray(fvtd(t)).reshape((-1, 1))

        def rhsp(t):
            return cfp

        cvop = None
        if stokes_flow:
            f_vdp = None
        elif use_custom_nonlinearity:
            def f_vdp(vvec):              # minus: it goes to the rhs
                return -custom_nonlinear_vel_function(vvec)
        else:
            cvop = _conv_operator(V, dbcnt, list(dbcinds) + list(glb),
                                  device=device)
            cvop.set_dbcvals(list(dbcvals) + list(inicdbcvals))

            def f_vdp(vvec):
                # `get_v_conv_conts(semi_explicit=True)` (snu:1136-1140): the
                # boundary values ride in `vvec` (it comes from `_appbcs`)
                ve = np.asarray(vvec).reshape(-1)
                cvop.set_dbcvals(ve[list(dbcinds) + list(glb)])
                return cvop.apply(ve[dbcnt], scale=-1.0)

        def getbcs(time, vvec, pvec, mode=None):
            return _comp_cntrl_bcvals(time=time, vel=vvec, p=pvec,
                                      diricontbcvals=diricontbcvals,
                                      diricontfuncs=diricontfuncs,
                                      diricontfuncmems=diricontfuncmems,
                                      mode=mode)
        timintsc = {'cnab': tiu.cnab, 'sbdf2': tiu.sbdftwo}[time_int_scheme]
        icd = dict(f_tdp=rhsv, inivel=iniv, verbose=verbose, M=cmmat,
                   A=camat, J=cj, f_vdp=f_vdp, getbcs=getbcs,
                   applybcs=applybcs, appndbcs=_appbcs, savevp=_record,
                   solver=solver)
        if time_int_scheme == 'cnab':
            icd.update(f_tvdp=fvtvd)
        static_bcs = len(loccnt) == 0
        if cvop is not None and (static_bcs or bcs_time_only):
            # the loop may evaluate N(v)v itself and, the callbacks being
            # functions of the time only, run whole time slices resident
            # (CNAB and SBDF2 alike)
            icd.update(device_convection=cvop, invinds=dbcnt,
                       resident=dict(
                           bcs_time_only=bcs_time_only,
                           static_dbcvals=list(dbcvals),
                           savevp_times=(None if return_vp_dict
                                         else list(datatrange))))
        v_end, p_end, ffflag = timintsc(trange=trange, inip=inip, scalep=-1.,
                                        g_tdp=rhsp, bcs_ini=inicdbcvals,
                                        check_ff_maxv=check_ff_maxv, **icd)

        def _flag(thing):
            return (thing, ffflag) if check_ff else thing
        if treat_nonl_explicit:
            if return_vp_dict:
                return _flag(vp_dict)
            if return_final_vp:
                return _flag((v_end, p_end))
            if return_dictofvelstrs:
                return _flag((veldict, pdict) if return_dictofpstrs
                             else veldict)
            if return_y_list:
                return _flag(ylist)
            return None
        # the explicit run seeds the Newton iteration (snu:1300)
        if not veldict:
            raise UserWarning('seeding the Newton sweeps with an explicit run '
                              'needs `return_dictofvelstrs=True`')
        cur_linvel_point = veldict
    else:
        cur_linvel_point = lin_vel_point

    # =====================================================================
    # Newton / Picard trapezoidal sweeps (snu:1304-1587)
    # =====================================================================
    from . import newton_picard as dnp
    tables = None
    if len(loccnt) > 0:
        # controlled Dirichlet values in the sweeps (snu:1433-1466): per time
        # instance the stiffness / divergence / mass columns of the controlled
        # dofs go to the right-hand sides and the convection sees the values
        # -- tabulated, the sweeps stay on the device
        if not bcs_time_only:
            raise NotImplementedError(
                'Newton/Picard sweeps with controlled Dirichlet values that '
                'depend on the state: pass `bcs_time_only=True` if the control '
                'functions depend on the time only')
        cdbs = [_comp_cntrl_bcvals(time=t, vel=None, p=None, **cdict)
                for t in trange]
        fvt, fpt, mbt, dbt = [], [], [], []
        for t, cdb in zip(trange, cdbs):
            ccfv_t, ccfp_t = _cntrl_stffnss_rhs(cntrlldbcvals=cdb, **cdict)
            ft = cfv + ccfv_t
            if fvtd is not None:
                ft = ft + np.asarray(fvtd(t)).reshape((-1, 1))
            fvt.append(ft)
            fpt.append(cfp + ccfp_t)
            mbt.append(dbc.condense_velmatsbybcs_rhs(
                Mfull, invinds=locinv, dbcinds=loccnt, dbcvals=cdb))
            dbt.append(np.array(list(dbcvals) + list(cdb)).reshape((-1, 1)))
        tables = dict(fv=np.hstack(fvt), fp=np.hstack(fpt),
                      mbc=np.hstack(mbt), dbc=np.hstack(dbt))
        cvop = _conv_operator(V, dbcnt, list(dbcinds) + list(glb),
                              device=device)
        cvop.set_dbcvals(list(dbcvals) + list(cdbs[0]))
    else:
        cvop = _conv_operator(V, dbcnt, list(dbcinds), device=device)
        cvop.set_dbcvals(dbcvals)
    dt = trange[1] - trange[0]
    # the preconditioner is set up once (the system is re-valued every step):
    # about the initial state, M + dt/2 (A + N1(iniv)), unless the caller's
    # `solver['precond_linpoint']` says otherwise (False: M + dt/2 A)
    plp = (solver or {}).get('precond_linpoint', None)
    if plp is None and not stokes_flow:
        plp = iniv
    elif plp is False:
        plp = None
    ts = dnp.TrapezoidalStepper(cmmat, camat, cj, cvop,
                                nslots=trange.size, dt=dt, device=device,
                                precond=(solver or {}).get('precond'),
                                precond_linpoint=plp)
    try:
        fvtab = None
        if fvtd is not None and tables is None:
            fvtab = np.hstack([cfv + np.asarray(fvtd(t)).reshape((-1, 1))
                               for t in trange])
        ts.set_rhs(cfv, cfp)

        def _lp(t):
            try:
                val = cur_linvel_point[t]
            except KeyError:
                val = cur_linvel_point[None]
            val = np.asarray(val, dtype=np.float64).reshape((-1, 1))
            return val[dbcnt] if val.shape[0] == vdim else val
        linpoints = {t: _lp(t) for t in trange}
        try:
            if None in cur_linvel_point:
                linpoints[None] = _lp(None)
        except TypeError:
            pass
        feedback = None
        if closed_loop:
            # snu:1367-1384, 1461-1483
            import scipy.sparse as sps
            bdense = np.asarray(b_mat.todense()) if sps.issparse(b_mat) \
                else np.asarray(b_mat, dtype=np.float64)

            def _fbentry(t):
                if static_feedback:
                    return feedbackthroughdict[None]
                return feedbackthroughdict[0 if t == trange[0] else t]

            def feedback(t):
                return bdense, np.asarray(_fbentry(t)['mtxtb']).T
            fbcols = np.hstack([
                b_mat @ (b_mat.T @ np.asarray(
                    _fbentry(t)['w'], dtype=np.float64).reshape((-1, 1)))
                for t in trange])
            if tables is not None:
                tables['fv'] = tables['fv'] + fbcols
            else:
                base = fvtab if fvtab is not None else \
                    np.tile(np.asarray(cfv).reshape((-1, 1)), (1, trange.size))
                fvtab = base + fbcols
        vdict, pdict_, hist = dnp.newton_picard(
            ts, trange, iniv, linpoints, vel_pcrd_stps=vel_pcrd_stps,
            vel_nwtn_stps=vel_nwtn_stps, vel_nwtn_tol=vel_nwtn_tol,
            rhs_table=fvtab, nsects=nsects, loc_nwtn_tol=loc_nwtn_tol,
            loc_pcrd_stps=loc_pcrd_stps, addfullsweep=addfullsweep,
            tables=tables, opts=(solver or {}).get('opts'),
            feedback=feedback)
    finally:
        ts.close()
    tlast = trange[-1]
    v_old, p_old = vdict[tlast], pdict_[tlast]
    cat = {t: [] for t in trange} if tables is None else \
        {t: cdb for t, cdb in zip(trange, cdbs)}
    if return_final_vp:
        return (_appbcs(v_old, cat[tlast]), p_old)
    if return_dictofvelstrs:
        vfull = {t: _appbcs(v, cat[t]) for t, v in vdict.items()}
        return (vfull, pdict_) if return_dictofpstrs else vfull
    if return_vp_dict:
        return {t: dict(v=_appbcs(vdict[t], cat[t]), p=pdict_.get(t))
                for t in vdict}
    return hist
