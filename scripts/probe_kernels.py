"""per-kernel cost of the GMRES cycle on the benchmark system (graph replay)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dolfin_navier_scipy_amd import saddle
from dolfin_navier_scipy_amd.fem import get_sysmats

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=100)
M, A, J = sm['M'], sm['A'], sm['J']
dt = 1/512.
system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
for fhat in ('explicit', 'cheb'):
    system.setup_precond(cheb_degree=deg, schur='dense', fhat=fhat)
    rng = np.random.default_rng(0)
    system.solve(M @ rng.standard_normal(M.shape[0]), rtol=1e-10)
    print(fhat, 'iters', system.last_stats['iters'])
    for name in system.PROBES:
        print('  %-12s %.2f us' % (name, system.probe(name)))
    if fhat == 'explicit':
        names = system.PROBES
        for a in range(4):
            for b in range(4):
                if a != b:
                    print('  pair %-12s -> %-12s %.2f us/launch' % (
                        names[a], names[b], system.probe(1000 + 10*a + b)))
