"""residual history of single CNAB steps (why does a step need k iterations?)

    python scripts/extrap_probe.py [nsteps] [extrap] [cheb]

prints, for the last steps, the relative residual after every Arnoldi step of
the solve: the first entry is what the warm start leaves, the ratio of
consecutive entries what the preconditioner achieves per iteration.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench  # noqa: E402
from dolfin_navier_scipy_amd import saddle, convection  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
extrap = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cheb = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dt = 1./512
femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
M, A, J = sm['M'], sm['A'], sm['J']
th, inv = femp['V'], femp['invinds']


def factory(F, Jm):
    return saddle.SaddleSystem(F, Jm)


v0, pt0, st0 = bench.initial_state(sm, rhsd, factory)
vfull = np.zeros((th.vdim, 1))
vfull[inv] = v0
vfull[femp['dbcinds'], 0] = femp['dbcvals']
nfc = -th.convection_vec(vfull)[inv, :]
system = factory((M + .5*dt*A).tocsr(), J)
system.setup_precond(cheb_degree=cheb, schur='dense', fhat='auto',
                     fp32_store=True, drop_tol=3e-3)
stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
cvop = convection.ConvectionP2.from_taylor_hood(
    th, inv, femp['dbcinds'], femp['dbcvals'])
stp.set_convection(cvop, scale=-1.0)
cf = saddle.ImexStepper.coeffs(a_c=1., a_p=0., cn_c=1.5*dt, cn_o=-.5*dt,
                               pscale=-1./dt, extrapolate=extrap)
opts = saddle.solve_opts(method='gmres', rtol=1e-10, maxiter=400, restart=60,
                         check_every=2, use_graph=False, reorth=False)
for k in range(nsteps):
    st = stp.step(cf, opts=opts)
    if k >= nsteps - 6 or k in (3, 10, 30):
        h = system.residual_history()/st['bnorm']
        print(k, st['iters'], ' '.join('{0:.1e}'.format(x) for x in h))
