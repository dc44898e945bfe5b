"""Newton/Picard sweep at the benchmark size: iterations and time per step of
the device path, SuperLU factor+solve per step of the CPU path beside it.

    python scripts/picard_probe.py [nts_run] [cheb] [refresh]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spsla

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench  # noqa: E402
from dolfin_navier_scipy_amd import saddle, convection  # noqa: E402
from dolfin_navier_scipy_amd import newton_picard as dnp  # noqa: E402

nrun = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cheb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dt = 1./512
femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
M, A, J = sm['M'], sm['A'], sm['J']
NP, NV = J.shape
th, inv = femp['V'], femp['invinds']
v0, pt0, st0 = bench.initial_state(sm, rhsd, lambda F, Jm: saddle.SaddleSystem(F, Jm))
cvop = convection.ConvectionP2.from_taylor_hood(
    th, inv, femp['dbcinds'], femp['dbcvals'])
trange = dt*np.arange(nrun + 1)

# linearisation points of the first sweep: the semi-explicit (CNAB) trajectory
system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
system.setup_precond(cheb_degree=6, schur='dense', fp32_store=True,
                     drop_tol=3e-3)
stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
nfc = cvop.apply(v0, scale=-1.0)
stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
stp.set_convection(cvop, scale=-1.0)
cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                               pscale=-1./dt, extrapolate=3)
o_imex = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
lin0 = {trange[0]: v0}
for k in range(1, nrun + 1):
    stp.run(1, cf, o_imex)
    lin0[trange[k]] = stp.get_state()[0]
stp.close()
system.close()

t0 = time.perf_counter()
ts = dnp.TrapezoidalStepper(M, A, J, cvop, nslots=nrun + 1, dt=dt,
                            precond=dict(cheb_degree=cheb))
ts.set_rhs(rhsd['fv'], rhsd['fp'])
print('setup {0:.2f} s'.format(time.perf_counter() - t0))
for k, t in enumerate(trange):
    ts.write_linpoint(0, k, lin0[t])
opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=0)
which = 0
for name, picard in (('picard', True), ('newton', False), ('newton', False)):
    t0 = time.perf_counter()
    _, _, upd, st = ts.sweep(trange, v0, which, picard, opts=opts,
                             record=False)
    wall = time.perf_counter() - t0
    print('{0}: {1} steps, {2:.3f} ms/step wall, {3:.3f} ms/step device, '
          '{4:.1f} its/step, update norm {5:.3e}'.format(
              name, nrun, 1e3*wall/nrun, 1e3*st['device_seconds']/nrun,
              st['iters']/nrun, upd))
    which = 1 - which
vg, pg = ts.state()

# CPU: what `lau`'s direct path costs per step (SuperLU factor + solve)
Nc, _, _ = cvop.assemble(v0, newton=False)
Fm = sps.csr_matrix((ts.mvals + .5*dt*(ts.avals + Nc.data), ts.pattern.indices,
                     ts.pattern.indptr), shape=(NV, NV))
K = sps.bmat([[Fm, J.T], [J, None]], format='csc')
rhs = np.ones(NV + NP)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    lu = spsla.splu(K)
    x = lu.solve(rhs)
print('CPU splu factor+solve: {0:.1f} ms/step'.format(
    1e3*(time.perf_counter() - t0)/reps))
