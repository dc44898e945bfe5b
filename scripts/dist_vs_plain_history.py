"""residual history of ONE cold solve, plain handle against the row-partitioned
code path on one RCCL rank (same matrices, same preconditioner settings): the
two are the same Krylov process up to rounding, or something differs"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from dolfin_navier_scipy_amd import saddle, comm as dcomm
from dolfin_navier_scipy_amd.fem import (get_sysmats, cylinder_mesh_hierarchy,
                                         pressure_prolongations, TaylorHood)
ref = int(sys.argv[1]) if len(sys.argv) > 1 else 2
femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=ref, Re=100.)
M, A, J = sm['M'], sm['A'], sm['J']
NP, NV = J.shape
dt = 1./(512*2**ref)
F = (M + .5*dt*A).tocsr()
hier = cylinder_mesh_hierarchy(N=2, refine=ref)
spaces = [TaylorHood(m) for m, _ in hier][::-1]
prols = pressure_prolongations(spaces, [p for _, p in hier][::-1])
rng = np.random.default_rng(0)
b = rng.standard_normal(NV)
out = {}
for kind in ('plain', 'rccl1', 'rccl1_rows'):
    cm = None
    if kind == 'plain':
        system = saddle.SaddleSystem(F, J)
    else:
        cm = dcomm.Comm.rccl(0, 1, 0, dcomm.rccl_unique_id())
        if kind == 'rccl1_rows':
            system = saddle.SaddleSystem.from_rows_of(F, J, cm)
        else:
            system = saddle.SaddleSystem(F, J)
            system.set_comm(cm)
    system.set_schur_mg(prols)
    system.setup_precond(cheb_degree=8, schur='mg', drop_tol=7e-3,
                         fhat='explicit', factorization='full')
    x = system.solve(b, np.zeros(NP), rtol=1e-10, maxiter=100, reorth=2,
                     use_graph=False)
    h = system.residual_history()
    print(kind, system.last_stats['iters'], ['%.3e' % (v/h[0]) for v in h[:10]],
          system.cheb_bounds())
    out[kind] = x
    system.close()
    if cm is not None:
        cm.close()
for k in ('rccl1', 'rccl1_rows'):
    print(k, 'vs plain', np.linalg.norm(out[k] - out['plain'])/np.linalg.norm(out['plain']))
