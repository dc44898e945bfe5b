"""where a 256-step sweep of the bench spends its wall time: the synchronous
first steps, the batches (with their polls), the rest.
    DNS_DEBUG=1 python scripts/sweep_startup_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from dolfin_navier_scipy_amd import saddle, convection, _capi  # noqa: E402
from dolfin_navier_scipy_amd import lin_alg_utils as lau  # noqa: E402
from dolfin_navier_scipy_amd import newton_picard as dnp  # noqa: E402

femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
dt = 1./512
vp = lau.solve_sadpnt_smw(amat=sm['A'], jmat=sm['J'], rhsv=rhsd['fv'],
                          rhsp=rhsd['fp'])
NV = sm['J'].shape[1]
v0 = vp[:NV]
lau.clear_cache()
nsteps = 256
cvop = convection.ConvectionP2.from_taylor_hood(
    femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
trange = dt*np.arange(nsteps + 1)
ts = dnp.TrapezoidalStepper(sm['M'], sm['A'], sm['J'], cvop, nslots=nsteps + 1,
                            dt=dt, precond=dict(cheb_degree=6, drop_tol=1e-3,
                                                factorization='full'),
                            precond_linpoint=v0)
ts.set_rhs(rhsd['fv'], rhsd['fp'])
for k in range(nsteps + 1):
    ts.write_linpoint(0, k, v0)
opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
ts.sweep(trange[:9], v0, 0, True, opts=opts, record=False)
# instrument: wall time of the pieces
marks = []
orig = dict(step=ts.step, run=ts.run, poll=ts.poll, checkpoint=ts.checkpoint,
            restore=ts.restore)


def timed(name):
    f = orig[name]

    def g(*a, **kw):
        t0 = time.perf_counter()
        r = f(*a, **kw)
        marks.append((name, time.perf_counter() - t0,
                      a[3] if name == 'run' else 1))
        return r
    return g


for name in orig:
    setattr(ts, name, timed(name))
which = 0
for picard in (True, False):
    del marks[:]
    t0 = time.perf_counter()
    _, _, upd, st = ts.sweep(trange, v0, which, picard, opts=opts,
                             record=False)
    wall = time.perf_counter() - t0
    tot = {}
    for name, sec, cnt in marks:
        a = tot.setdefault(name, [0, 0., 0])
        a[0] += 1
        a[1] += sec
        a[2] += cnt
    print('picard' if picard else 'newton', 'wall ms %.2f' % (1e3*wall),
          'steps/s %.0f' % (nsteps/wall), 'cycle', st['cycle'],
          'replayed', st['replayed_batches'], 'batches', len(st['batches']))
    for name, (n, sec, cnt) in sorted(tot.items()):
        print('   %-10s calls %3d  steps %4d  ms %.3f' % (name, n, cnt, 1e3*sec))
    print('   other (python, policy) ms %.3f' % (
        1e3*(wall - sum(a[1] for a in tot.values()))))
    which = 1 - which
ts.close()
cvop.close()
