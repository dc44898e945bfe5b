"""one call of bench.py's `picard_sweep_figures` (N = 2, Re = 100, 256 steps,
Picard then Newton) -- the workload of scripts/profile_sweeps_timeline.sh"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from dolfin_navier_scipy_amd import lin_alg_utils as lau  # noqa: E402

femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
dt = 1./512
vp = lau.solve_sadpnt_smw(amat=sm['A'], jmat=sm['J'], rhsv=rhsd['fv'],
                          rhsp=rhsd['fp'])
v0 = vp[:sm['J'].shape[1]]
lau.clear_cache()
graph = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    res = bench.picard_sweep_figures(femp, sm, rhsd, v0, dt, 0,
                                     use_graph=graph)
    print(json.dumps({k: res[k] for k in ('picard', 'newton')}))
