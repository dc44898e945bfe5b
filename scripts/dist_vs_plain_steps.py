"""the SAME CNAB steps through a plain handle and through the row-partitioned
code path on one RCCL rank, step by step (synchronous steps): Krylov steps,
start residual and final residual per step -- the two are the same algorithm
up to rounding, or the stepper around the solve differs"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from dolfin_navier_scipy_amd import saddle, convection, comm as dcomm
from dolfin_navier_scipy_amd.fem import (get_sysmats, cylinder_mesh_hierarchy,
                                         pressure_prolongations, TaylorHood)
ref = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pipe = len(sys.argv) > 3 and sys.argv[3] == 'pipelined'
femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=ref, Re=100.)
M, A, J = sm['M'], sm['A'], sm['J']
NP, NV = J.shape
dt = 1./(512*2**ref)
F = (M + .5*dt*A).tocsr()
R1 = (M - .5*dt*A).tocsr()
hier = cylinder_mesh_hierarchy(N=2, refine=ref)
spaces = [TaylorHood(m) for m, _ in hier][::-1]
prols = pressure_prolongations(spaces, [p for _, p in hier][::-1])
res = {}
for kind in ('plain', 'rccl1'):
    cm = None
    system = saddle.SaddleSystem(F, J)
    if kind != 'plain':
        cm = dcomm.Comm.rccl(0, 1, 0, dcomm.rccl_unique_id())
        system.set_comm(cm)
    system.set_schur_mg(prols)
    system.setup_precond(cheb_degree=8, schur='mg', drop_tol=7e-3,
                         fhat='explicit', factorization='full')
    cv = convection.ConvectionP2.from_taylor_hood(
        femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
    stp = saddle.ImexStepper(system, R1)
    v0 = np.zeros((NV, 1))
    nfc = cv.apply(v0, scale=-1.0)
    stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
    stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
    stp.set_convection(cv, scale=-1.0)
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt, extrapolate=3)
    opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=pipe,
                             reorth=2)
    rows = []
    if pipe:
        for chunk in range(nsteps//32):
            ds, its, last = stp.run(32, cf, opts)
            rows.append((its/32., last['est_relres'], last['true_relres']))
    else:
        for k in range(nsteps):
            st = stp.step(cf, opts=opts)
            h = system.residual_history()
            rows.append((st['iters'], h[0]/st['bnorm'], st['true_relres']))
    res[kind] = (rows, stp.get_state())
    stp.close()
    cv.close()
    system.close()
    if cm is not None:
        cm.close()
for k, (a, b) in enumerate(zip(res['plain'][0], res['rccl1'][0])):
    print(k, 'plain', ' '.join('%.3g' % x for x in a), '| rccl1',
          ' '.join('%.3g' % x for x in b))
va, vb = res['plain'][1][0], res['rccl1'][1][0]
print('final states differ by', np.linalg.norm(va - vb)/np.linalg.norm(va))
