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
(snu:1424-1431) become a dict `t -> v`.  No control boundaries, no feedback
(`umat/vmat` of snu:1036-1042 are covered by `saddle_oracle`'s SMW test).

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
                    coeffmat_n=None, fv_c=None, fv_n=None):
    """trapezoidal rule: `(M + dt/2 C_n) v_n = M v_c + dt/2 (f_n + f_c - C_c v_c)`
    (snu:1034-1035)"""
    solvmat = mmat + 0.5*dt*coeffmat_n
    rhs = mmat @ var_c + 0.5*dt*(fv_n + fv_c - coeffmat_c @ var_c)
    return solvmat, rhs


def trapezoidal_sweep(trange, iniv, M=None, A=None, J=None, fv=None, fp=None,
                      conv=None, appndbcs=None, linpoints=None, picard=False,
                      solve=None, krylovini=None):
    """one sweep over `trange` with the convection linearised about
    `linpoints[t]` (snu:1402-1566); returns `vdict, pdict, norm_nwtnupd`

    `krylovini='upd'` feeds the extrapolated previous solutions as `x0`
    (snu:1493-1503)."""
    solve = lau.solve_sadpnt_smw if solve is None else solve
    NP, NV = J.shape
    JT = J.T.tocsr()
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


def newton_picard(trange, iniv, linpoints0, vel_pcrd_stps=1, vel_nwtn_stps=2,
                  vel_nwtn_tol=1e-14, invinds=None, **kw):
    """Picard sweeps first, then Newton sweeps, each linearised about the
    previous sweep's trajectory (snu:1304-1334, 1574)"""
    linpoints = linpoints0
    newtk, norm_nwtnupd = 0, 1.
    hist = []
    vdict = pdict = None
    while newtk < vel_nwtn_stps and norm_nwtnupd > vel_nwtn_tol:
        if vel_pcrd_stps > 0:
            vel_pcrd_stps -= 1
            picard = True
        else:
            picard = False
            newtk += 1
        inner = {t: (v if v.shape[0] == len(invinds) else v[invinds, :])
                 for t, v in linpoints.items()}
        vdict, pdict, _ = trapezoidal_sweep(trange, iniv, linpoints=linpoints,
                                            picard=picard, **kw)
        norm_nwtnupd = sum(
            (trange[k+1]-trange[k])*m_innerproduct(
                kw['M'], vdict[trange[k+1]] - inner[trange[k+1]]).item()
            for k in range(len(trange)-1))                          # snu:1557-1560
        hist.append(('picard' if picard else 'newton', norm_nwtnupd))
        appnd = kw['appndbcs']
        linpoints = {t: appnd(v) for t, v in vdict.items()}        # snu:1574
    return vdict, pdict, hist


def get_pfromv(v=None, M=None, A=None, J=None, fv=None, conv=None,
               appndbcs=None, solve=None):
    """`[[M, J^T],[J, 0]] [.; p~] = [-A v - N(v)v + fv; 0]`, returns `-p~`
    (snu:1629-1633)"""
    solve = lau.solve_sadpnt_smw if solve is None else solve
    _, rhs_con, _ = conv(appndbcs(v), False)
    vp = solve(amat=M, jmat=J, jmatT=J.T.tocsr(), rhsv=-(A @ v) - rhs_con + fv)
    return -vp[J.shape[1]:, :]
