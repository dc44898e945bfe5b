"""A/B of the LDS-streaming SpMV with 32-bit and 16-bit column indices in ONE
process (interleaved rounds), on the refined benchmark matrix"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dolfin_navier_scipy_amd import saddle

refine = int(sys.argv[1]) if len(sys.argv) > 1 else 4
_, sm, _ = bench.build_problem(N=2, Re=100., refine=refine)
K = bench.saddle_csr((sm['M'] + .5/512*sm['A']).tocsr(), sm['J'])
nb = bench.spmv_bytes(K)
names = {17: 'stream, 2 loads per lane in flight (earlier kernel)',
         1: 'stream (int32 cols)', 2: 'stream16 (u16 offsets, pair loads)',
         5: 'stream16, single loads (before)',
         7: 'diag: single loads, gather from an 8 KB range of x',
         3: 'diag: stream16 without the x gather',
         4: 'diag: ... and without the LDS row reduction'}
res = {k: [] for k in names}
for rnd in range(4):
    for k in names:
        secs, _ = saddle.spmv_bench(K, variant=k, reps=15, warmup=2)
        res[k].append(nb/secs/1e9)
for k in sorted(names):
    v = np.array(res[k])
    print('%-24s median %7.0f GB/s (algorithmic)  min %7.0f max %7.0f' % (
        names[k], np.median(v), v.min(), v.max()))
x = np.random.default_rng(0).standard_normal(K.shape[1])
y1 = saddle.spmv(K, x, variant='stream')
y2 = saddle.spmv(K, x, variant='stream16')
print('max |y16 - y32| =', np.abs(y1 - y2).max())
y3 = saddle.spmv(K, x, variant=5)
print('max |ysingle - y32| =', np.abs(y1 - y3).max())

# the 2x2-blocked pair format (needs even sizes)
NV = sm['M'].shape[0]
if NV % 2 == 0:
    rates = []
    for rnd in range(4):
        yp, secs, fb = saddle.spmv_pair(K, NV, x, reps=15, warmup=2)
        rates.append(nb/secs/1e9)
    print('%-24s median %7.0f GB/s (algorithmic)  min %7.0f max %7.0f; '
          'format %.1f MB (CSR16 %.1f MB)' % (
              'pair format (2x2 blocks)', np.median(rates), min(rates),
              max(rates), fb/1e6, (10*K.nnz + 4*K.shape[0])/1e6))
    print('max |ypair - y32| =', np.abs(y1 - yp).max(), 'of', np.abs(y1).max())
    rates = []
    for rnd in range(3):
        _, secs, _ = saddle.spmv_pair(K, NV, x, reps=15, warmup=-2)
        rates.append(nb/secs/1e9)
    print('%-24s median %7.0f GB/s (algorithmic)' % (
        'diag: pair format without the x gather', np.median(rates)))
