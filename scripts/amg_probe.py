"""Krylov steps per time step of the CNAB loop on the reference's
karman2D-rotcyl meshes with the algebraic multigrid Schur block (and the
Jacobi block it replaces):  python scripts/amg_probe.py <level> <dt> <nsteps>"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dolfin_navier_scipy_amd import saddle, convection, amg  # noqa: E402
from dolfin_navier_scipy_amd.fem import get_sysmats  # noqa: E402

level, dt, nsteps = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
geo_prols = None
if level.startswith('c'):
    # the cylinder wake refined `c<r>` times: a mesh that HAS nested pressure
    # spaces, for the geometric hierarchy next to the algebraic one
    from dolfin_navier_scipy_amd.fem import (cylinder_mesh_hierarchy,
                                             pressure_prolongations,
                                             TaylorHood)
    ref = int(level[1:])
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=ref,
                                 Re=100.)
    hier = cylinder_mesh_hierarchy(N=2, refine=ref)
    spaces = [TaylorHood(m) for m, _ in hier][::-1]
    geo_prols = pressure_prolongations(spaces, [p for _, p in hier][::-1])
else:
    femp, sm, rhsd = get_sysmats(
        problem='gen_bccont', nu=1e-3, charvel=0.2, bccontrol=False,
        meshparams=dict(meshname='karman2D-rotcyl_lvl{0}'.format(level),
                        geodata='karman2D-rotcyl-bm_geo_cntrlbc'))
M, A, J = sm['M'], sm['A'], sm['J']
NP, NV = J.shape
F = (M + .5*dt*A).tocsr()
R1 = (M - .5*dt*A).tocsr()
out = dict(level=level, NV=NV, NP=NP, dt=dt)
for kind in os.environ.get('AMG_KINDS', 'amg,jacobi').split(','):
    for deg, drop in ((8, 7e-3),):
        system = saddle.SaddleSystem(F, J)
        t0 = time.time()
        schur = saddle.choose_schur(
            system, F, J, schur='auto' if kind == 'geo' else kind,
            prolongations=geo_prols if kind == 'geo' else None,
            dense_max=1500)
        system.setup_precond(cheb_degree=deg, schur=schur, drop_tol=drop,
                             fhat='explicit', factorization='full')
        tset = time.time() - t0
        cv = convection.ConvectionP2.from_taylor_hood(
            femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
        stp = saddle.ImexStepper(system, R1)
        v0 = np.zeros((NV, 1))
        nfc = cv.apply(v0, scale=-1.0)
        stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
        stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
        stp.set_convection(cv, scale=-1.0)
        for rtol in (1e-10,):
            for ex in (3, 4):
                cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                               pscale=-1./dt, extrapolate=ex)
                opts = saddle.solve_opts(rtol=rtol, maxiter=400, use_graph=True,
                                         reorth=2)
                stp.run(64, cf, opts)                      # spin-up
                t0 = time.time()
                ds, its, last = stp.run(nsteps, cf, opts)
                key = '{0} deg{1} rtol{2:g} ex{3}'.format(kind, deg, rtol, ex)
                out[key] = dict(krylov_per_step=its/float(nsteps),
                                steps_per_s=nsteps/ds, setup_s=round(tset, 2),
                                hierarchy=system.schur_hierarchy,
                                relres=last['true_relres'],
                                run=dict(stp.last_run))
                print(key, out[key], file=sys.stderr)
        stp.close()
        cv.close()
        system.close()
print(json.dumps(out))
