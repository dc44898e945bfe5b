"""Newton sweep of the bench's secondary workload with the preconditioner rebuilt
about the NEWTON operator of the start state (`refresh_precond` behind a few
Newton steps) against the set-up about `M + dt/2 (A + N1(v0))` the bench keeps:
1.85 -> 1.75 Krylov steps per time step, 9.17k -> 9.38k steps/s (MI355X).

    python scripts/newton_precond_probe.py
"""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import bench
from dolfin_navier_scipy_amd import saddle, convection
from dolfin_navier_scipy_amd import newton_picard as dnp
femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
v0, _, _ = bench.initial_state(sm, rhsd, lambda F, J: saddle.SaddleSystem(F, J))
M, A, J = sm['M'], sm['A'], sm['J']
th, inv = femp['V'], femp['invinds']
dt = 1./512; nsteps = 256
cvop = convection.ConvectionP2.from_taylor_hood(th, inv, femp['dbcinds'], femp['dbcvals'])
trange = dt*np.arange(nsteps + 1)
ts = dnp.TrapezoidalStepper(M, A, J, cvop, nslots=nsteps + 1, dt=dt,
                            precond=dict(cheb_degree=6, drop_tol=1e-3, factorization='full'),
                            precond_linpoint=v0)
ts.set_rhs(rhsd['fv'], rhsd['fp'])
for k in range(nsteps + 1):
    ts.write_linpoint(0, k, v0)
opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
ts.sweep(trange[:9], v0, 0, True, opts=opts, record=False)
t0 = time.perf_counter(); _, _, upd, st = ts.sweep(trange, v0, 0, True, opts=opts, record=False); w = time.perf_counter() - t0
print('picard', nsteps/w, st['iters']/nsteps)
for refresh in (False, True):
    if refresh:
        # the Newton operator of the first steps on the device, then the set-up about it
        ts.sweep(trange[:9], v0, 1, False, opts=opts, record=False)
        t1 = time.perf_counter(); ts.refresh_precond(); print('refresh s', time.perf_counter() - t1)
        ts.sweep(trange[:9], v0, 1, False, opts=opts, record=False)
    t0 = time.perf_counter(); _, _, upd, st = ts.sweep(trange, v0, 1, False, opts=opts, record=False); w = time.perf_counter() - t0
    print('newton refresh', refresh, nsteps/w, st['iters']/nsteps, upd)
