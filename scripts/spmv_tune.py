"""A/B of the LDS-streaming SpMV variants in ONE process (interleaved rounds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dolfin_navier_scipy_amd import saddle

refine = int(sys.argv[1]) if len(sys.argv) > 1 else 4
_, sm, _ = bench.build_problem(N=2, Re=100., refine=refine)
K = bench.saddle_csr((sm['M'] + .5/512*sm['A']).tocsr(), sm['J'])
nb = bench.spmv_bytes(K)
names = {16: 'unr2', 17: 'unr2 rps', 18: 'unr2 rps G8', 19: 'unr2 G8',
         20: 'unr1 rps', 21: 'unr4 rps', 22: 'unr2 rps G2',
         23: 'tile4096 unr2 rps', 1: 'default stream'}
res = {k: [] for k in names}
for rnd in range(4):
    for k in names:
        secs, _ = saddle.spmv_bench(K, variant=k, reps=15, warmup=2)
        res[k].append(nb/secs/1e9)
for k in sorted(names):
    v = np.array(res[k])
    print('%-20s median %7.0f GB/s  min %7.0f max %7.0f' % (
        names[k], np.median(v), v.min(), v.max()))
