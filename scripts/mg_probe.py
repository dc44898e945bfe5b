"""Multigrid Schur block on refined meshes: Krylov steps and time per solve
against the dense Schur inverse (where it fits) and SuperLU.

    python scripts/mg_probe.py [max refine] [with_cpu]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spsla

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from dolfin_navier_scipy_amd import saddle  # noqa: E402
from dolfin_navier_scipy_amd.fem import (  # noqa: E402
    get_sysmats, cylinder_mesh_hierarchy, pressure_prolongations, TaylorHood)

rmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2
with_cpu = len(sys.argv) > 2 and sys.argv[2] == '1'
dt = 1./512
rmin = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for refine in range(rmin, rmax + 1):
    t0 = time.perf_counter()
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=refine,
                                 Re=100.)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    hier = cylinder_mesh_hierarchy(N=2, refine=refine)
    spaces = [TaylorHood(m) for m, _ in hier][::-1]
    parents = [p for _, p in hier][::-1]
    prols = pressure_prolongations(spaces, parents)
    t_asm = time.perf_counter() - t0
    F = (M + .5*dt*A).tocsr()
    rng = np.random.default_rng(refine)
    rhsv = M @ rng.standard_normal(NV)
    rhsp = 1e-3*(J @ rng.standard_normal(NV))
    print('refine {0}: NV {1} NP {2} (assembly {3:.1f} s)'.format(
        refine, NV, NP, t_asm), flush=True)
    ref = None
    if with_cpu and NV < 200000:
        t0 = time.perf_counter()
        K = sps.bmat([[F, J.T], [J, None]], format='csc')
        lu = spsla.splu(K)
        ref = lu.solve(np.concatenate([rhsv, rhsp]))
        print('   CPU SuperLU factor + solve {0:.2f} s'.format(
            time.perf_counter() - t0), flush=True)
    for schur, fact, fhat, deg in (('dense', 'full', 'explicit', 6),
                                   ('mg', 'triangular', 'explicit', 6),
                                   ('mg', 'full', 'explicit', 6),
                                   ('mg', 'triangular', 'cheb', 6)):
        if schur == 'dense' and NP > 6000:
            continue
        if fhat == 'explicit' and NV > 1000000:
            continue
        t0 = time.perf_counter()
        system = saddle.SaddleSystem(F, J)
        if schur == 'mg':
            system.set_schur_mg(prols, smooth_steps=2)
        system.setup_precond(cheb_degree=deg, schur=schur, fhat=fhat,
                             drop_tol=1e-3, factorization=fact)
        t_set = time.perf_counter() - t0
        x = system.solve(rhsv, rhsp, rtol=1e-10, maxiter=600, reorth=1,
                         use_graph=True, raise_on_fail=False)
        st = system.last_stats
        t0 = time.perf_counter()
        x = system.solve(rhsv, rhsp, rtol=1e-10, maxiter=600, reorth=1,
                         use_graph=True, raise_on_fail=False)
        t_solve = time.perf_counter() - t0
        err = ''
        if ref is not None:
            err = ' err vs LU {0:.1e}'.format(
                np.linalg.norm(x[:NV] - ref[:NV])/np.linalg.norm(ref[:NV]))
        print('   {0:5s} {1:10s} {2:8s}: set-up {3:6.2f} s, {4:3d} its, status '
              '{5}, relres {6:.1e}, solve {7:7.2f} ms{8}'.format(
                  schur, fact, fhat, t_set, st['iters'], st['status'],
                  st['true_relres'], 1e3*t_solve, err), flush=True)
        system.close()
