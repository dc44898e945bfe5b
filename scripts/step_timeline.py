"""Timeline of ONE time step inside a graph-replayed run, from a rocprofv3
kernel trace: the dispatches of the last `nsteps` steps in start order with
their duration and the gap to the dispatch before (what a dependent graph
edge costs), then the same grouped by kernel instance.

    python scripts/step_timeline.py <dir with *_kernel_trace.csv> <marker kernel> [nsteps]

`marker kernel`: a kernel that runs exactly once per time step (the first
kernel of a step, e.g. k_imex_bvec / k_step_one2 / k_conv_cells_lane).
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, marker = sys.argv[1], sys.argv[2]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
trace = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'),
                         recursive=True))[-1]
rows = []
with open(trace) as fh:
    for r in csv.DictReader(fh):
        name = (r.get('Kernel_Name') or r.get('Name'))
        name = name.replace('void dns::', '').replace('dns::', '').split('(')[0]
        g = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
        wg = int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 256)) or 256)
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name,
                     g//max(1, wg)))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < nsteps + 2:
    sys.exit('marker {0} seen {1} times only'.format(marker, len(marks)))
first, last = marks[-nsteps - 1], marks[-1]
window = rows[first:last]
span = (rows[last][0] - rows[first][0])/1e3
busy = sum(e - s for s, e, _, _ in window)/1e3
print('# {0}: last {1} steps, {2:.1f} us per step, {3:.1f} us of it inside '
      'kernels ({4:.0f} %), {5:.1f} dispatches per step'.format(
          os.path.basename(trace), nsteps, span/nsteps, busy/nsteps,
          100*busy/span, len(window)/nsteps))
grp = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for s, e, name, nb in window:
    g = grp[(name, nb)]
    g[0] += 1
    g[1] += (e - s)/1e3
    if prev_end is not None:
        g[2] += max(0.0, (s - prev_end)/1e3)
    prev_end = max(prev_end or e, e)
print('%-44s %7s %9s %8s %9s %9s' % ('kernel instance', 'blocks', 'per step',
                                     'avg us', 'us/step', 'gap/step'))
for (name, nb), (c, t, gap) in sorted(grp.items(), key=lambda kv: -kv[1][1]):
    print('%-44s %7d %9.2f %8.2f %9.2f %9.2f' % (name[:44], nb, c/nsteps, t/c,
                                                 t/nsteps, gap/nsteps))
print('# one step in dispatch order (the last one): start offset us, '
      'duration us, gap before us')
one = rows[marks[-2]:marks[-1]]
t0 = one[0][0]
prev_end = None
for s, e, name, nb in one:
    gap = 0.0 if prev_end is None else (s - prev_end)/1e3
    print('%9.1f %7.2f %6.2f  %s [%d]' % ((s - t0)/1e3, (e - s)/1e3, gap,
                                          name[:60], nb))
    prev_end = e
