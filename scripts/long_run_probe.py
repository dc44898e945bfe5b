"""iterations and steps/s of the benchmark configuration far into the run
(developed vortex shedding), chunk by chunk

    python scripts/long_run_probe.py [tend] [Re] [level] [nts]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench  # noqa: E402
from dolfin_navier_scipy_amd import saddle, convection, _capi  # noqa: E402

tend = float(sys.argv[1]) if len(sys.argv) > 1 else 24.
Re = float(sys.argv[2]) if len(sys.argv) > 2 else 100.
level = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nts = int(sys.argv[4]) if len(sys.argv) > 4 else 512
dt = 1./nts
femp, sm, rhsd = bench.build_problem(N=level, Re=Re)
M, A, J = sm['M'], sm['A'], sm['J']
th, inv = femp['V'], femp['invinds']
v0, pt0, st0 = bench.initial_state(sm, rhsd,
                                   lambda F, Jm: saddle.SaddleSystem(F, Jm))
system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
system.setup_precond(cheb_degree=6, schur='dense', drop_tol=1e-3,
                     factorization='full')
cv = convection.ConvectionP2.from_taylor_hood(th, inv, femp['dbcinds'],
                                              femp['dbcvals'])
stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
nfc = cv.apply(v0, scale=-1.0)
stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
stp.set_convection(cv, scale=-1.0)
cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                               pscale=-1./dt, extrapolate=4)
opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
chunk = 2*nts
t = 0.
while t < tend - 1e-9:
    _capi.device_synchronize(0)
    t0 = time.perf_counter()
    ds, its, last = stp.run(chunk, cf, opts)
    _capi.device_synchronize(0)
    wall = time.perf_counter() - t0
    t += chunk*dt
    v = stp.get_state()[0]
    print('t = {0:5.1f}: {1:7.0f} steps/s, {2:.2f} its/step, relres {3:.1e}, '
          'max |v| {4:.3f}'.format(t, chunk/wall, its/float(chunk),
                                   last['true_relres'], np.abs(v).max()))
