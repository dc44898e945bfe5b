"""The Newton/Picard sweeps of tests/test_gpu_newton_picard.py over their WHOLE
horizon (2048 steps of developed shedding, N=2, Re=100, dt=1/512) against the
oracle's restatement (oracle/newton_picard_oracle.py with the refined direct
solve) -- ten minutes of host time, hence a script and not a test.

    python scripts/sweep_horizon_parity.py [nsteps] [rtol] [extrapolate]

Prints one JSON line: Krylov steps per time step, refreshes, and the distance
to the oracle (v in the M-norm, p in l2, relative) at every 128th step.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_newton_picard as tnp  # noqa: E402
from oracle import newton_picard_oracle as npo  # noqa: E402
from oracle import saddle_oracle  # noqa: E402
from dolfin_navier_scipy_amd import convection, saddle  # noqa: E402
from dolfin_navier_scipy_amd import newton_picard as dnp  # noqa: E402


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    rtol = float(sys.argv[2]) if len(sys.argv) > 2 else tnp.SWEEP_RTOL
    extrap = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    s = tnp.shedding_setup(nsteps)
    femp, sm, rhsd, vs, dt = s['femp'], s['sm'], s['rhsd'], s['vs'], s['dt']
    th, inv = femp['V'], femp['invinds']
    M, A, J = sm['M'], sm['A'], sm['J']
    NV = s['NV']
    tr = dt*np.arange(nsteps + 1)
    cv = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'])
    stp = dnp.TrapezoidalStepper(
        M, A, J, cv, nslots=nsteps + 1, dt=dt,
        precond=dict(cheb_degree=6, drop_tol=1e-3, factorization='full'),
        precond_linpoint=vs[0], refresh_iters=3.0)
    stp.set_rhs(rhsd['fv'], rhsd['fp'])
    for k in range(nsteps + 1):
        stp.write_linpoint(0, k, vs[k])
    opts = saddle.solve_opts(rtol=rtol, maxiter=400, use_graph=True, reorth=2)
    cvo = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'])
    cvo.bind_pattern(stp.pattern)
    conv = tnp._device_conv_callback(cvo, NV)
    mnorm = lambda x: np.sqrt((x.T @ (M @ x)).item())
    out = dict(nsteps=nsteps, rtol=rtol, extrapolate=extrap, sweeps=[])
    which, lindict = 0, {t: vs[k] for k, t in enumerate(tr)}
    for name, picard in (('picard', True), ('newton', False)):
        got_v, got_p, upd, st = stp.sweep(tr, vs[0], which, picard, opts=opts,
                                          extrapolate=extrap)
        rs = saddle_oracle.RefinedSolve()
        t0 = time.perf_counter()

        def solve(**kw):
            if rs.calls % 128 == 0:
                sys.stderr.write('{0}: oracle step {1} of {2}, {3:.0f} s, {4} '
                                 'LUs\n'.format(name, rs.calls, nsteps,
                                                time.perf_counter() - t0,
                                                rs.factorisations))
                sys.stderr.flush()
            return rs(**kw)
        ref_v, ref_p, ref_upd = npo.trapezoidal_sweep(
            tr, vs[0], M=M, A=A, J=J, fv=rhsd['fv'], fp=rhsd['fp'], conv=conv,
            appndbcs=lambda v: v, linpoints=lindict, picard=picard,
            solve=solve)
        marks = list(range(128, nsteps + 1, 128))
        ev = [mnorm(got_v[tr[k]] - ref_v[tr[k]])/mnorm(ref_v[tr[k]])
              for k in marks]
        ep = [float(np.linalg.norm(got_p[tr[k]] - ref_p[tr[k]])
                    / np.linalg.norm(ref_p[tr[k]])) for k in marks]
        rec = dict(sweep=name, krylov_per_step=st['iters']/float(nsteps),
                   worst_batch=max(st['batches']), refreshes=st['refreshes'],
                   replayed_batches=st['replayed_batches'],
                   v_rel_Mnorm_max=max(ev), p_rel_l2_max=max(ep),
                   v_rel_Mnorm_at_marks=ev, p_rel_l2_at_marks=ep,
                   update_norm=upd, update_norm_oracle=float(ref_upd),
                   oracle_seconds=time.perf_counter() - t0,
                   oracle_factorisations=rs.factorisations)
        out['sweeps'].append(rec)
        sys.stderr.write('{sweep}: {krylov_per_step:.2f} Krylov steps per time '
                         'step, parity v {v_rel_Mnorm_max:.2e} p '
                         '{p_rel_l2_max:.2e}\n'.format(**rec))
        lindict = {t: got_v[t] for t in tr}
        which = 1 - which
    print(json.dumps(out))


if __name__ == '__main__':
    main()
