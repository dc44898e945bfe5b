"""Does recycling cut the Krylov steps per time step at scale?  (round-5 item 4)

The CNAB loop of `refined_bench.py` driven from Python through
`SaddleSystem.solve(x0=...)` so that the START VECTOR of every solve can be
chosen freely (the device stepper's own warm start is the quartic
extrapolation):

  q4        quartic extrapolation of the last five solutions (what the device
            stepper does)
  q5        quintic (six solutions)
  q4+d1     q4, then the correction of the previous solve d = x_prev - x0_prev
            as ONE deflation direction: x0 += alpha d, alpha = <K d, r0> /
            <K d, K d>  (what keeping Z y and K Z y of the previous time step
            amounts to)
  q4+d2     the corrections of the last two solves, 2 x 2 least squares
  q3+d1     cubic + one direction

per variant: mean / max Krylov steps per time step, mean initial relative
residual, over `nsteps` steps behind a common device spin-up.

    python scripts/recycle_probe.py [refine] [nts] [nsteps] [spinup]
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dolfin_navier_scipy_amd import saddle, convection  # noqa: E402
from dolfin_navier_scipy_amd.fem import (  # noqa: E402
    get_sysmats, cylinder_mesh_hierarchy, pressure_prolongations, TaylorHood)

def fit_weights(d, m):
    """value at t = 0 of the degree-d least-squares polynomial through the
    points t = -1 .. -m: fixed weights like the interpolating extrapolation,
    with a noise amplification sqrt(sum c^2) of 2.5 (d = 3, m = 8) instead
    of 8.3 (d = 3, m = 4)"""
    t = -np.arange(1, m + 1.)
    V = np.vander(t, d + 1, increasing=True)
    return np.linalg.solve(V.T @ V, V.T)[0]


EXTRAP = {0: [1.], 1: [2., -1.], 2: [3., -3., 1.], 3: [4., -6., 4., -1.],
          4: [5., -10., 10., -5., 1.], 5: [6., -15., 20., -15., 6., -1.]}


def main(refine=2, nts=2048, nsteps=100, spinup=256, rtol=1e-10):
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=refine,
                                 Re=100.)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    th, inv = femp['V'], femp['invinds']
    dt = 1./nts
    hier = cylinder_mesh_hierarchy(N=2, refine=refine)
    spaces = [TaylorHood(m) for m, _ in hier][::-1]
    prols = pressure_prolongations(spaces, [p for _, p in hier][::-1])
    F = (M + .5*dt*A).tocsr()
    R1 = (M - .5*dt*A).tocsr()
    K = sps.bmat([[F, J.T], [J, None]], format='csr')
    system = saddle.SaddleSystem(F, J)
    system.set_schur_mg(prols, smooth_steps=2)
    dflt = saddle.streaming_precond_defaults(NV + NP)
    system.setup_precond(cheb_degree=dflt['cheb_degree'], schur='mg',
                         drop_tol=dflt['drop_tol'],
                         fhat='explicit' if NV + NP >= 100000 else 'auto',
                         factorization='full')
    cv = convection.ConvectionP2.from_taylor_hood(th, inv, femp['dbcinds'],
                                                  femp['dbcvals'])
    # common spin-up on the device
    v0 = np.zeros((NV, 1))
    stp = saddle.ImexStepper(system, R1)
    nfc = cv.apply(v0, scale=-1.0)
    stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
    stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
    stp.set_convection(cv, scale=-1.0)
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt, extrapolate=4)
    opts = saddle.solve_opts(rtol=rtol, maxiter=400, use_graph=True, reorth=2)
    stp.run(spinup, cf, opts)
    # eight more steps by hand to fill the history (exact-ish solves)
    vs, ps = stp.get_state()
    stp.close()
    fv, fp = dt*rhsd['fv'].reshape(-1), rhsd['fp'].reshape(-1)
    tight = saddle.solve_opts(rtol=1e-13, maxiter=400, use_graph=True,
                              reorth=2)

    def rhs_of(v, nc, no):
        return np.concatenate([R1 @ v + dt*(1.5*nc - .5*no) + fv, fp])

    hist0 = []
    v = vs.reshape(-1).copy()
    nc = cv.apply(v.reshape(-1, 1), scale=-1.0).reshape(-1)
    no = nc.copy()
    x_prev = np.concatenate([v, -dt*ps.reshape(-1)])
    for k in range(12):
        b = rhs_of(v, nc, no)
        x = system.solve(b[:NV], b[NV:], x0=x_prev, opts=tight)
        hist0.insert(0, x.copy())
        x_prev = x
        v = x[:NV]
        no = nc
        nc = cv.apply(v.reshape(-1, 1), scale=-1.0).reshape(-1)
    state0 = (v.copy(), nc.copy(), no.copy(), [h.copy() for h in hist0])
    sopts = saddle.solve_opts(rtol=rtol, maxiter=400, use_graph=True, reorth=2)
    fixed_c = int(os.environ.get('RECYCLE_FIXED_C', '0'))
    if fixed_c:
        # "oversolve": every solve runs exactly `fixed_c` columns whatever its
        # residual; what is reported is where the residual levels settle
        sopts = saddle.solve_opts(rtol=1e-30, maxiter=fixed_c, restart=fixed_c,
                                  use_graph=True, reorth=2)
    results = {}
    variants = os.environ.get(
        'RECYCLE_VARIANTS', 'q4,q5,q4+d1,q4+d2,q3+d1,q5+d1').split(',')
    for variant in variants:
        # 'lsL': least squares over the last L solutions (Fischer's projection
        # onto the span of the history, QR of K X in double precision)
        ls = int(variant[2:].split('+')[0]) if variant.startswith('ls') else 0
        fit = None
        if variant.startswith('f'):           # 'f<d>m<m>'
            fit = fit_weights(int(variant[1]), int(variant.split('m')[1]))
        order = 3 if (ls or fit is not None) else int(variant[1])
        ndefl = int(variant.split('+d')[1]) if '+d' in variant else 0
        v, nc, no, hist = state0[0].copy(), state0[1].copy(), \
            state0[2].copy(), [h.copy() for h in state0[3]]
        corr = []                   # previous corrections, newest first
        its, r0s, r0d, fins = [], [], [], []
        t0 = time.time()
        for k in range(nsteps):
            b = rhs_of(v, nc, no)
            x0 = sum(c*h for c, h in zip(EXTRAP[order], hist))
            if fit is not None:
                x0 = sum(c*h for c, h in zip(fit, hist))
            if ls:
                X = np.stack(hist[:ls], axis=1)
                # difference basis: newest, first differences, ... (the
                # solutions themselves are parallel to eight digits)
                Xd = X.copy()
                for lev in range(1, ls):
                    Xd[:, lev:] = Xd[:, lev-1:-1] - Xd[:, lev:]
                Wd = K @ Xd
                Q, Rr = np.linalg.qr(Wd)
                coef = np.linalg.solve(Rr, Q.T @ b)
                x0 = Xd @ coef
            bn = np.linalg.norm(b)
            r0 = b - K @ x0
            r0s.append(np.linalg.norm(r0)/bn)
            x0e = x0
            if ndefl and len(corr) >= ndefl:
                D = np.stack(corr[:ndefl], axis=1)
                W = K @ D
                alpha, *_ = np.linalg.lstsq(W, r0, rcond=None)
                x0 = x0 + D @ alpha
                r0 = r0 - W @ alpha
            r0d.append(np.linalg.norm(r0)/bn)
            x = system.solve(b[:NV], b[NV:], x0=x0, opts=sopts,
                             raise_on_fail=not fixed_c)
            its.append(system.last_stats['iters'])
            fins.append(system.last_stats['true_relres'])
            corr.insert(0, x - x0e)
            corr = corr[:2]
            hist.insert(0, x.copy())
            hist = hist[:12]
            v = x[:NV]
            no = nc
            nc = cv.apply(v.reshape(-1, 1), scale=-1.0).reshape(-1)
        its = np.array(its[8:])          # (the deflation history has filled)
        results[variant] = dict(
            krylov_mean=float(its.mean()), krylov_max=int(its.max()),
            hist=np.bincount(its).tolist(),
            r0_extrap=float(np.mean(r0s[8:])),
            r0_deflated=float(np.mean(r0d[8:])),
            r0_last20_max=float(np.max(r0s[-20:])),
            final_last20_max=float(np.max(fins[-20:])),
            wall_s=round(time.time() - t0, 1))
        print(variant, results[variant], file=sys.stderr)
    out = dict(refine=refine, n=int(NV + NP), dt=dt, rtol=rtol,
               nsteps=nsteps, spinup=spinup, variants=results)
    print(json.dumps(out))
    cv.close()
    system.close()


if __name__ == '__main__':
    a = [int(x) for x in sys.argv[1:5]]
    main(*a)
