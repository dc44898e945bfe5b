import sys, os, time, json
sys.path.insert(0, '/root/repo')
import numpy as np
from dolfin_navier_scipy_amd import saddle, convection
from dolfin_navier_scipy_amd.fem import get_sysmats
level, dt = sys.argv[1], float(sys.argv[2])
femp, sm, rhsd = get_sysmats(problem='gen_bccont', nu=1e-3, charvel=0.2, bccontrol=False,
    meshparams=dict(meshname='karman2D-rotcyl_lvl'+level, geodata='karman2D-rotcyl-bm_geo_cntrlbc'))
M, A, J = sm['M'], sm['A'], sm['J']; NP, NV = J.shape
F = (M + .5*dt*A).tocsr(); R1 = (M - .5*dt*A).tocsr()
d = F.diagonal(); dm=M.diagonal()
print('NV',NV,'NP',NP,'diag(F)/diag(M) min/max', (d/dm).min(), (d/dm).max(), file=sys.stderr)
cv = convection.ConvectionP2.from_taylor_hood(femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
for kind, deg, drop in (('dense',8,7e-3),('amg',8,7e-3),('amg',8,1e-3),('amg',12,1e-3),('dense',12,1e-4)):
    if kind=='dense' and NP>8000: continue
    system = saddle.SaddleSystem(F, J)
    schur = saddle.choose_schur(system, F, J, schur=kind, dense_max=100000 if kind=='dense' else 1500)
    system.setup_precond(cheb_degree=deg, schur=schur, drop_tol=drop, fhat='explicit', factorization='full')
    stp = saddle.ImexStepper(system, R1)
    v0 = np.zeros((NV, 1)); nfc = cv.apply(v0, scale=-1.0)
    stp.set_state(v0, nfc_c=nfc, nfc_o=nfc); stp.set_rhs(dt*rhsd['fv'], rhsd['fp']); stp.set_convection(cv, scale=-1.0)
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt, pscale=-1./dt, extrapolate=4)
    opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
    stp.run(128, cf, opts)
    ds, its, last = stp.run(64, cf, opts)
    pi = system.precond_info()
    print(kind, deg, drop, 'bounds', system.cheb_bounds(), 'krylov/step', its/64., 'steps/s', 64/ds, 'nnz_Gc/nnz_K', pi['nnz_Gc']/pi['nnz_K'], file=sys.stderr)
    # cold solve history
    rng=np.random.default_rng(0); b=rng.standard_normal(NV)
    x=system.solve(b, np.zeros(NP), rtol=1e-10, maxiter=300, raise_on_fail=False)
    h=system.residual_history()
    print('   cold solve iters', system.last_stats['iters'], 'hist', ['%.1e'%v for v in (h/h[0])[:12]], file=sys.stderr)
    stp.close(); system.close()
