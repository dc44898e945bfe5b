"""SpMV roofline run alone (short; used under rocprofv3 --pmc)

    python scripts/spmv_roofline.py [refine] [reps]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dolfin_navier_scipy_amd import saddle

refine = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
_, sm, _ = bench.build_problem(N=2, Re=100., refine=refine)
K = bench.saddle_csr((sm['M'] + .5/512*sm['A']).tocsr(), sm['J'])
out = {}
for variant in ('stream16', 'stream', 'vector'):
    secs, chk = saddle.spmv_bench(K, variant=variant, reps=reps, warmup=3)
    out[variant] = dict(avg_us=secs*1e6, GBs=bench.spmv_bytes(K)/secs/1e9)
NV = sm['M'].shape[0]
if NV % 2 == 0:
    import numpy as np
    x = np.sin(0.37*np.arange(K.shape[1]))
    _, secs, fb = saddle.spmv_pair(K, NV, x, reps=reps, warmup=3)
    out['pair'] = dict(avg_us=secs*1e6, GBs=bench.spmv_bytes(K)/secs/1e9,
                       format_bytes=int(fb))
out.update(bytes=bench.spmv_bytes(K), nnz=int(K.nnz), rows=int(K.shape[0]))
print(json.dumps(out))
