"""Per-instance table of a rocprofv3 kernel trace (eager-mode run of
refined_bench.py): the dispatches grouped by (kernel, grid size) -- one group
per operator a kernel instance is launched on -- with calls, average duration,
the operator's non-zeros inferred from the row-block grid (2048 products per
row block) or taken from the run's own `precond_info`, algorithmic GB/s
(12 nnz + 20 rows) and waves per SIMD from the VGPR count.

    python scripts/instance_table.py <dir with *_kernel_trace.csv> <bench json>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, jf = sys.argv[1], sys.argv[2]
trace = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'),
                         recursive=True))[-1]
info = json.loads(open(jf).read().strip().splitlines()[-1])
grp = defaultdict(lambda: [0, 0.0, 0, 0, 0])
with open(trace) as fh:
    for r in csv.DictReader(fh):
        name = r.get('Kernel_Name') or r.get('Name')
        g = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
        wg = int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 256)) or 256)
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        key = (name, g // max(1, wg))
        e = grp[key]
        e[0] += 1
        e[1] += dur
        e[2] = int(r.get('VGPR_Count', 0) or 0)
        e[3] = int(r.get('Accum_VGPR_Count', 0) or 0)
        e[4] = int(r.get('LDS_Block_Size', 0) or 0)
rows = []
tot = sum(e[1] for e in grp.values())
for (name, nblk), (calls, ns, vg, ag, lds) in grp.items():
    if calls < 5:
        continue
    avg = ns/calls/1e3
    # unified register file of gfx950: 512 per SIMD lane set, allocation
    # granule 8; at most 8 waves
    regs = max(8, ((vg + ag + 7)//8)*8)
    waves = min(8, 512//regs) if regs else 8
    if lds:
        waves = min(waves, max(1, (160*1024//lds)*4//4))
    short = name.replace('void dns::', '').replace('dns::', '')
    short = short.split('(')[0]
    nnz_est = None
    if 'stream16' in short or 'pair16' in short:
        nnz_est = nblk*2048
    rows.append((ns, short, nblk, calls, avg, vg, ag, waves, nnz_est))
rows.sort(reverse=True)
print('# {0}: refine {1}, n = {2}, {3:.0f} steps/s, {4:.2f} Krylov steps per '
      'time step (eager launches)'.format(
          os.path.basename(trace), info.get('refine'), info.get('n'),
          info.get('gpu_steps_per_s', 0), info.get('krylov_iters_per_step', 0)))
print('%-46s %8s %7s %9s %6s %5s %6s %14s %9s' % (
    'kernel instance', 'blocks', 'calls', 'avg us', 'share', 'VGPR', 'w/SIMD',
    'nnz (blocks*2048)', 'alg GB/s'))
for ns, short, nblk, calls, avg, vg, ag, waves, nnz in rows[:28]:
    gbs = ''
    if nnz and nblk not in (1024,):      # (capped grids: fused-dots instances)
        gbs = '%.0f' % (12.0*nnz/avg/1e3)
    print('%-46s %8d %7d %9.1f %5.1f%% %5d %6d %14s %9s' % (
        short[:46], nblk, calls, avg, 100*ns/tot, vg + ag, waves,
        nnz if nnz else '', gbs))
