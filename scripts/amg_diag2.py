import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from dolfin_navier_scipy_amd import saddle, convection
from dolfin_navier_scipy_amd.fem import get_sysmats
which, dt = sys.argv[1], float(sys.argv[2])
if which.startswith('c'):
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=int(which[1:]), Re=100.)
else:
    femp, sm, rhsd = get_sysmats(problem='gen_bccont', nu=1e-3, charvel=0.2, bccontrol=False,
        meshparams=dict(meshname='karman2D-rotcyl_lvl'+which, geodata='karman2D-rotcyl-bm_geo_cntrlbc'))
M, A, J = sm['M'], sm['A'], sm['J']; NP, NV = J.shape
F = (M + .5*dt*A).tocsr(); R1 = (M - .5*dt*A).tocsr()
cv = convection.ConvectionP2.from_taylor_hood(femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
system = saddle.SaddleSystem(F, J)
schur = saddle.choose_schur(system, F, J, schur='amg', dense_max=1500)
system.setup_precond(cheb_degree=8, schur=schur, drop_tol=7e-3, fhat='explicit', factorization='full')
stp = saddle.ImexStepper(system, R1)
v0 = np.zeros((NV, 1)); nfc = cv.apply(v0, scale=-1.0)
stp.set_state(v0, nfc_c=nfc, nfc_o=nfc); stp.set_rhs(dt*rhsd['fv'], rhsd['fp']); stp.set_convection(cv, scale=-1.0)
cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt, pscale=-1./dt, extrapolate=4)
opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=False, reorth=2)
for chunk in (8, 24, 32, 64, 128, 256):
    ds, its, last = stp.run(chunk, cf, saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2))
    st = stp.step(cf, opts=opts)
    h = system.residual_history()
    v, p = stp.get_state()
    print(which, 'after +%d steps: %.2f krylov/step; one step: iters %d, r0/|b| %.2e, hist %s |v| %.3e |p| %.3e' % (
        chunk, its/float(chunk), st['iters'], h[0]/st['bnorm'], ['%.1e' % (x/st['bnorm']) for x in h[:8]], np.linalg.norm(v), np.linalg.norm(p)))
