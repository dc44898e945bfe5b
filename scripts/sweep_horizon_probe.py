"""Newton/Picard sweeps over a long horizon of DEVELOPED vortex shedding
(cylinder wake N=2, Re=100, dt=1/512): Krylov steps per time step batch by
batch, with the refresh policy of `TrapezoidalStepper.sweep` on and off.

    python scripts/sweep_horizon_probe.py [nsteps] [spinup] [rtol] [bound]

The linearisation points are the CNAB trajectory of the same steps (what the
reference's first Picard sweep linearises about, snu:1427-1431).
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench  # noqa: E402
from dolfin_navier_scipy_amd import saddle, convection  # noqa: E402
from dolfin_navier_scipy_amd import newton_picard as dnp  # noqa: E402


def developed_state_and_linpoints(femp, sm, rhsd, dt, nsteps, spinup,
                                  device=0):
    """Stokes state advanced `spinup` CNAB steps on the device (shedding has
    developed by t ~ 4), then the next `nsteps` CNAB velocities one by one"""
    M, A, J = sm['M'], sm['A'], sm['J']
    th, inv = femp['V'], femp['invinds']
    v0, _, _ = bench.initial_state(
        sm, rhsd, lambda F, Jm: saddle.SaddleSystem(F, Jm, device=device))
    cvop = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'], device=device)
    system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J, device=device)
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
    if spinup > 0:
        stp.run(spinup, cf, o)
    vs = [stp.get_state()[0]]
    for _ in range(nsteps):
        stp.run(1, cf, o)
        vs.append(stp.get_state()[0])
    stp.close()
    system.close()
    cvop.close()
    return vs


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    spinup = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    rtol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-10
    bound = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
    extrap = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    dt = 1./512
    femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
    M, A, J = sm['M'], sm['A'], sm['J']
    th, inv = femp['V'], femp['invinds']
    t0 = time.perf_counter()
    vs = developed_state_and_linpoints(femp, sm, rhsd, dt, nsteps, spinup)
    print('CNAB trajectory: {0} + {1} steps in {2:.1f} s'.format(
        spinup, nsteps, time.perf_counter() - t0), file=sys.stderr)
    trange = dt*np.arange(nsteps + 1)
    out = dict(nsteps=nsteps, spinup=spinup, rtol=rtol, bound=bound,
               extrapolate=extrap, runs=[])
    for refresh in (None, bound):
        cvop = convection.ConvectionP2.from_taylor_hood(
            th, inv, femp['dbcinds'], femp['dbcvals'])
        ts = dnp.TrapezoidalStepper(
            M, A, J, cvop, nslots=nsteps + 1, dt=dt,
            precond=dict(cheb_degree=6, drop_tol=1e-3, factorization='full'),
            precond_linpoint=vs[0], refresh_iters=refresh)
        ts.set_rhs(rhsd['fv'], rhsd['fp'])
        for k in range(nsteps + 1):
            ts.write_linpoint(0, k, vs[k])
        opts = saddle.solve_opts(rtol=rtol, maxiter=400, use_graph=True,
                                 reorth=2)
        which = 0
        for name, picard in (('picard', True), ('newton', False)):
            t0 = time.perf_counter()
            _, _, upd, st = ts.sweep(trange, vs[0], which, picard, opts=opts,
                                     record=False, extrapolate=extrap)
            wall = time.perf_counter() - t0
            rec = dict(sweep=name, refresh_bound=refresh,
                       steps_per_s=nsteps/wall,
                       krylov_per_step=st['iters']/float(nsteps),
                       refreshes=st['refreshes'],
                       replayed_batches=st['replayed_batches'],
                       worst_batch=max(st['batches']),
                       batches=[round(b, 2) for b in st['batches']],
                       update_norm=upd)
            out['runs'].append(rec)
            print('{sweep:7s} refresh {refresh_bound}: {steps_per_s:8.0f} '
                  'steps/s, {krylov_per_step:.2f} Krylov steps per time step '
                  '(worst batch {worst_batch:.2f}), {refreshes} refreshes, '
                  '{replayed_batches} batches replayed'.format(**rec),
                  file=sys.stderr)
            which = 1 - which
        ts.close()
        cvop.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
