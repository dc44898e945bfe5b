"""sweep rates of the bench with the quartic / cubic / quadratic warm start:
    python scripts/sweep_extrap_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from dolfin_navier_scipy_amd import lin_alg_utils as lau  # noqa: E402
from dolfin_navier_scipy_amd import newton_picard as dnp  # noqa: E402

femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
dt = 1./512
vp = lau.solve_sadpnt_smw(amat=sm['A'], jmat=sm['J'], rhsv=rhsd['fv'],
                          rhsp=rhsd['fp'])
v0 = vp[:sm['J'].shape[1]]
lau.clear_cache()
orig = dnp.TrapezoidalStepper.sweep
out = {}
for ex in (4, 3, 2):
    def sweep(self, *a, **kw):
        kw['extrapolate'] = ex
        return orig(self, *a, **kw)
    dnp.TrapezoidalStepper.sweep = sweep
    best = None
    for rep in range(3):
        res = bench.picard_sweep_figures(femp, sm, rhsd, v0, dt, 0)
        if best is None or res['picard']['steps_per_s'] > \
                best['picard']['steps_per_s']:
            best = res
    out['extrapolate_{0}'.format(ex)] = {k: best[k] for k in ('picard',
                                                              'newton')}
    print(ex, out['extrapolate_{0}'.format(ex)], file=sys.stderr)
dnp.TrapezoidalStepper.sweep = orig
print(json.dumps(out))
