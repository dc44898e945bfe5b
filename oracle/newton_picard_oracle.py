"""ORACLE (test infrastructure, never shipped, never the thing measured).

CPU restatement of the Newton/Picard time sweeps of the reference's
`stokes_navier_utils.solve_nse` -- the branch taken with
`treat_nonl_explicit=False`:

 * `get_mats_rhs_ts`   <- `_get_mats_rhs_ts`            snu:1016-1047
 * `trapezoidal_sweep` <- the time loop                 snu:1402-1566
 * `newton_picard`     <- the sweep loop / Picard-then-Newton switch
                                                         snu:1304-1334,1562-1587
 * `get_pfromv`        <- `get_pfromv`                  snu:1602-1633
 * `m_innerproduct`    <- `m_innerproduct`              snu:136-143

in purely algebraic form: the FEniCS pieces (`get_v_conv_conts`, snu:40-133)
enter as a callback `conv(vfull, Picard) -> (N_condensed, rhs_con, rhsv_conbc)`,
the per-time-step `.npy` files that hold the linearisation points
(snu:1424-1431) become a dict `t -> v`.  The low-rank feedback terms
(`umat/vmat`, snu:1036-1042) and the moving-boundary mass term (`mbcs`,
snu:1044-1045) of `_get_mats_rhs_ts` are restated too; time-dependent data
enter as callables of the time (`fv(t)`, `fp(t)`, `mbcs(t)`, `feedback(t)`).

PARITY UNPINNED against the reference itself: this branch needs dolfin
(snu:7) and is bit-rotted on Python >= 3.8 (`time.clock`, snu:1412;
`np.float`, snu:1560), so it cannot be executed here and the reference ships
no vectors for it.  The restatement is pinned by the mathematical definition
instead: `tests/test_newton_picard.py` checks the trapezoidal-rule residual of
every step and second-order agreement with the semi-explicit integrator.

Only `tests/` may import this module.
"""
import numpy as np

from . import saddle_oracle as lau

__all__ = ['get_mats_rhs_ts', 'trapezoidal_sweep', 'newton_picard',
           'get_pfromv', 'm_innerproduct']


def m_innerproduct(M, v1, v2=None):
    v2 = v1 if v2 is None else v2
    return np.dot(v1.T, M @ v2)


def get_mats_rhs_ts(mmat=None, dt=None, var_c=None, coeffmat_c=None,
                    coeffmat_n=None, fv_c=None, fv_n=None,
                    umat_c=None, vmat_c=None, umat_n=None, vmat_n=None,
                    mbcs_c=None, mbcs_n=None):
    """trapezoidal rule: `(M + dt/2 C_n) v_n = M v_c + dt/2 (f_n + f_c - C_c v_c)`
    (snu:1034-1035); with feedback the system matrix is `solvmat - umat vmat`
    (`umat = dt/2 umat_n`, snu:1036-1042; QUIRK kept: the rhs term carries a
    PLUS, "do we really need a PLUS here??", snu:1039-1040); moving boundary
    values add `mbcs_n - mbcs_c` (snu:1044-1045)"""
    solvmat = mmat + 0.5*dt*coeffmat_n
    rhs = mmat @ var_c + 0.5*dt*(fv_n + fv_c - coeffmat_c @ var_c)
    if umat_n is not None:
        umat = 0.5*dt*umat_n
        vmat = vmat_n
        if umat_c is not None and vmat_c is not None:
            rhs = rhs + 0.5*dt*umat_c.dot(vmat_c.dot(var_c))
    else:
        umat, vmat = None, None
    if mbcs_c is not None and mbcs_n is not None:
        rhs = rhs + mbcs_n - mbcs_c
    if umat is None and mbcs_c is None:
        return solvmat, rhs
    return solvmat, rhs, umat, vmat


def trapezoidal_sweep(trange, iniv, M=None, A=None, J=None, fv=None, fp=None,
                      conv=None, appndbcs=None, linpoints=None, picard=False,
                      solve=None, krylovini=None, feedback=None, mbcs=None):
    """one sweep over `trange` with the convection linearised about
    `linpoints[t]` (snu:1402-1566); returns `vdict, pdict, norm_nwtnupd`

    `krylovini='upd'` feeds the extrapolated previous solutions as `x0`
    (snu:1493-1503).  `fv`, `fp` may be callables of the time; `feedback(t) ->
    (umat, vmat)` and `mbcs(t)` switch the extra terms of `_get_mats_rhs_ts`
    on (`conv`/`appndbcs` then take the time as a second/third argument)."""
    solve = lau.solve_sadpnt_smw if solve is None else solve
    NP, NV = J.shape
    JT = J.T.tocsr()
    timedep = callable(fv) or callable(fp) or feedback is not None \
        or mbcs is not None
    if timedep:
        return _sweep_timedep(trange, iniv, M, A, J, JT, fv, fp, conv,
                              appndbcs, linpoints, picard, solve, feedback,
                              mbcs)
    v_old = iniv
    vdict, pdict = {trange[0]: iniv}, {}
    N_c, rhs_con_c, rhsbc_c = conv(appndbcs(v_old), picard)       # snu:1351
    fvn_c = fv + rhsbc_c + (0. if picard else rhs_con_c)          # snu:1364
    norm_nwtnupd = 0.
    vp_old = np.vstack([v_old, np.zeros((NP, 1))])
    vp_new = vp_old
    cts_old = trange[1] - trange[0]
    for tk, t in enumerate(trange[1:]):
        cts = t - trange[tk]
        prev_v = linpoints[t]                                      # snu:1425
        N_n, rhs_con_n, rhsbc_n = conv(prev_v, picard)            # snu:1443
        rhscon_n = 0. if picard else rhs_con_n
        fvn_n = fv + rhsbc_n + rhscon_n                            # snu:1459
        solvmat, rhsv = get_mats_rhs_ts(mmat=M, dt=cts, var_c=v_old,
                                        coeffmat_c=A + N_c,
                                        coeffmat_n=A + N_n,
                                        fv_c=fvn_c, fv_n=fvn_n)
        kw = {}
        if krylovini == 'upd':
            vp_oldold, vp_old = vp_old, vp_new
            kw['krpslvprms'] = dict(
                x0=vp_old + cts*(vp_old - vp_oldold)/cts_old)
            cts_old = cts
        vp_new = solve(amat=solvmat, jmat=J, jmatT=JT, rhsv=rhsv, rhsp=fp,
                       **kw)                                       # snu:1505
        v_old = vp_new[:NV, ]
        N_c, rhs_con_c, rhsbc_c = conv(appndbcs(v_old), picard)   # snu:1529
        rhscon_c = 0. if picard else rhs_con_c
        fvn_c = fvn_n - rhscon_n - rhsbc_n + rhsbc_c + rhscon_c   # snu:1537
        vdict[t] = v_old
        pdict[t] = -1/cts*vp_new[NV:, ]                            # snu:1542
        pv = prev_v if prev_v.shape[0] == NV else None
        if pv is not None:
            norm_nwtnupd += cts*m_innerproduct(M, v_old - pv).item()
    return vdict, pdict, norm_nwtnupd


def _sweep_timedep(trange, iniv, M, A, J, JT, fv, fp, conv, appndbcs,
                   linpoints, picard, solve, feedback, mbcs):
    """the same loop with everything that may depend on the time spelled out
    (snu:1402-1566): `fv(t)` = cfv + ccfv(t) [+ fvtd(t)], `fp(t)`, boundary
    values through `conv(v, picard, t)` / `appndbcs(v, t)`, `mbcs(t)`,
    `feedback(t)`"""
    NP, NV = J.shape
    _fv = fv if callable(fv) else (lambda t: fv)
    _fp = fp if callable(fp) else (lambda t: fp)

    def _conv(v, t):
        try:
            return conv(v, picard, t)
        except TypeError:
            return conv(v, picard)

    def _app(v, t):
        try:
            return appndbcs(v, t)
        except TypeError:
            return appndbcs(v)
    t0 = trange[0]
    v_old = iniv
    vdict, pdict = {t0: iniv}, {}
    N_c, rhs_con_c, rhsbc_c = _conv(_app(v_old, t0), t0)          # snu:1351
    fvn_c = _fv(t0) + rhsbc_c + (0. if picard else rhs_con_c)     # snu:1364
    umat_c, vmat_c = feedback(t0) if feedback is not None else (None, None)
    mbcs_c = mbcs(t0) if mbcs is not None else None
    norm_nwtnupd = 0.
    for tk, t in enumerate(trange[1:]):
        cts = t - trange[tk]
        prev_v = linpoints[t]
        N_n, rhs_con_n, rhsbc_n = _conv(prev_v, t)                # snu:1443
        rhscon_n = 0. if picard else rhs_con_n
        fvn_n = _fv(t) + rhsbc_n + rhscon_n                        # snu:1459
        umat_n, vmat_n = feedback(t) if feedback is not None \
            else (None, None)
        mbcs_n = mbcs(t) if mbcs is not None else None
        out = get_mats_rhs_ts(mmat=M, dt=cts, var_c=v_old,
                              coeffmat_c=A + N_c, coeffmat_n=A + N_n,
                              fv_c=fvn_c, fv_n=fvn_n, umat_c=umat_c,
                              vmat_c=vmat_c, umat_n=umat_n, vmat_n=vmat_n,
                              mbcs_c=mbcs_c, mbcs_n=mbcs_n)
        solvmat, rhsv = out[0], out[1]
        umat, vmat = (out[2], out[3]) if len(out) > 2 else (None, None)
        vp_new = solve(amat=solvmat, jmat=J, jmatT=JT, rhsv=rhsv,
                       rhsp=_fp(t), umat=umat, vmat=vmat)          # snu:1505
        v_old = vp_new[:NV, ]
        umat_c, vmat_c, mbcs_c = umat_n, vmat_n, mbcs_n            # snu:1525
        N_c, rhs_con_c, rhsbc_c = _conv(_app(v_old, t), t)        # snu:1529
        rhscon_c = 0. if picard else rhs_con_c
        # QUIRK-free form of snu:1536-1537 (which carries `cfv + ccfv_n` over):
        # f_c at the new time instance
        fvn_c = fvn_n - rhscon_n - rhsbc_n + rhsbc_c + rhscon_c
        vdict[t] = v_old
        pdict[t] = -1/cts*vp_new[NV:, ]
        pv = prev_v if prev_v.shape[0] == NV else None
        if pv is not None:
            norm_nwtnupd += cts*m_innerproduct(M, v_old - pv).item()
    return vdict, pdict, norm_nwtnupd


def time_sections(trange, nsects=1, addfullsweep=False):
    """the local time ranges of the sweeps (snu:1076-1086): `nsects` sections
    of `floor(len(trange)/nsects)` steps that share their end points, the
    last one taking the rest; `addfullsweep` appends the whole range"""
    trange = np.asarray(trange)
    lensect = int(np.floor(trange.size/nsects))
    loctrngs = [trange[k*lensect:(k+1)*lensect+1] for k in range(nsects-1)]
    loctrngs.append(trange[(nsects-1)*lensect:])
    if addfullsweep:
        loctrngs.append(trange)
    if nsects == 1:
        loctrngs = [trange]
    return loctrngs


def newton_picard(trange, iniv, linpoints0, vel_pcrd_stps=1, vel_nwtn_stps=2,
                  vel_nwtn_tol=1e-14, invinds=None, nsects=1,
                  loc_nwtn_tol=5e-15, loc_pcrd_stps=True, addfullsweep=False,
                  **kw):
    """Picard sweeps first, then Newton sweeps, each linearised about the
    previous sweep's trajectory (snu:1304-1334, 1574), section by section
    (snu:1076-1090): every section starts from the end of the one before,
    iterates until `loc_nwtn_tol` with its own Picard count (`loc_pcrd_stps`)
    and the optional full sweep at the end restarts from the true initial
    value with `vel_nwtn_tol` (snu:1579-1587).  Linearisation points of times
    not yet computed come from `linpoints0[None]` (snu:1427-1431)."""
    def appnd(v, t):
        try:
            return kw['appndbcs'](v, t)          # time-dependent boundary values
        except TypeError:
            return kw['appndbcs'](v)
    loctrngs = time_sections(trange, nsects, addfullsweep)
    if nsects == 1:
        loc_nwtn_tol, addfullsweep = vel_nwtn_tol, False            # snu:1087
    vel_loc_pcrd_steps = vel_pcrd_stps                               # snu:1091
    realiniv = np.copy(iniv)
    cur = dict(linpoints0)
    newtk, norm_nwtnupd = 0, 1.
    hist = []
    vall, pall = {}, {}
    for si, loctrng in enumerate(loctrngs):
        while newtk < vel_nwtn_stps and norm_nwtnupd > loc_nwtn_tol:
            if vel_pcrd_stps > 0:
                vel_pcrd_stps -= 1
                picard = True
            else:
                picard = False
                newtk += 1
            lin = {t: (cur[t] if t in cur else cur[None]) for t in loctrng}
            inner = {t: (v if v.shape[0] == len(invinds) else v[invinds, :])
                     for t, v in lin.items()}
            vdict, pdict, _ = trapezoidal_sweep(loctrng, iniv, linpoints=lin,
                                                picard=picard, **kw)
            norm_nwtnupd = sum(
                (loctrng[k+1]-loctrng[k])*m_innerproduct(
                    kw['M'], vdict[loctrng[k+1]] - inner[loctrng[k+1]]).item()
                for k in range(len(loctrng)-1))                     # snu:1557-1560
            hist.append(('picard' if picard else 'newton', norm_nwtnupd))
            cur.update({t: appnd(v, t) for t, v in vdict.items()})  # snu:1574
            vall.update(vdict)
            pall.update(pdict)
        iniv = vall[loctrng[-1]]                                     # snu:1576
        if addfullsweep and si == len(loctrngs) - 2:                 # snu:1579
            iniv = realiniv
            loc_nwtn_tol = vel_nwtn_tol
        elif loc_pcrd_stps:
            vel_pcrd_stps = vel_loc_pcrd_steps
        norm_nwtnupd, newtk = 1., 0                                  # snu:1586
    return vall, pall, hist


def get_pfromv(v=None, M=None, A=None, J=None, fv=None, conv=None,
               appndbcs=None, solve=None):
    """`[[M, J^T],[J, 0]] [.; p~] = [-A v - N(v)v + fv; 0]`, returns `-p~`
    (snu:1629-1633)"""
    solve = lau.solve_sadpnt_smw if solve is None else solve
    _, rhs_con, _ = conv(appndbcs(v), False)
    vp = solve(amat=M, jmat=J, jmatT=J.T.tocsr(), rhsv=-(A @ v) - rhs_con + fv)
    return -vp[J.shape[1]:, :]
