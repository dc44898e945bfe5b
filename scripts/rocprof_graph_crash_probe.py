"""which captured graph makes rocprofv3 crash?  python3 scratch/prof_bisect.py <mode> [dist]
 mode 'solve'  : only system.solve() with use_graph (graphs = GMRES cycles)
 mode 'step'   : stepper.step() per step (graphs = prologue + first cycle)
 mode 'run'    : stepper.run() (graphs = groups of steps)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from dolfin_navier_scipy_amd import saddle, convection, comm as dcomm
mode = sys.argv[1]
dist = len(sys.argv) > 2 and sys.argv[2] == 'dist'
conv_on = not (len(sys.argv) > 3 and sys.argv[3] == 'noconv')
femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
M, A, J = sm['M'], sm['A'], sm['J']
NP, NV = J.shape
dt = 1./512
system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
if dist:
    with bench.stdout_to_stderr():
        cm = dcomm.Comm.rccl(0, 1, 0, dcomm.rccl_unique_id())
    system.set_comm(cm)
system.setup_precond(cheb_degree=6, schur='dense', fhat='explicit', fp32_store=True, drop_tol=1e-3, factorization='full')
rng = np.random.default_rng(0)
if mode == 'solve':
    for k in range(6):
        system.solve(M @ rng.standard_normal(NV), rhsd['fp'], rtol=1e-10, use_graph=True, reorth=2)
    print('solve ok', system.last_stats['iters'])
    sys.exit(0)
cv = convection.ConvectionP2.from_taylor_hood(femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
v0 = np.zeros((NV, 1))
nfc = cv.apply(v0, scale=-1.0)
stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
if conv_on:
    stp.set_convection(cv, scale=-1.0)
cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt, pscale=-1./dt, extrapolate=4)
opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
if mode == 'step':
    for k in range(30):
        stp.step(cf, opts=opts)
    print('step ok')
else:
    stp.run(int(os.environ.get("NSTEPS", "64")), cf, opts)
    print('run ok')
