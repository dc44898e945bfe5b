"""A/B of two instances of the pair SpMV in ONE process (alternating launches
on the same box): DNS_PAIR_AB=0 the shipped instance, 2 the one without the
occupancy bound.   python scripts/pair_ab.py [refine] [reps] [rounds]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dolfin_navier_scipy_amd import saddle

refine = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
_, sm, _ = bench.build_problem(N=2, Re=100., refine=refine)
K = bench.saddle_csr((sm['M'] + .5/512*sm['A']).tocsr(), sm['J'])
NV = sm['M'].shape[0]
x = np.sin(0.37*np.arange(K.shape[1]))
variants = os.environ.get('PAIR_AB_VARIANTS', '0,2').split(',')
out = {v: [] for v in variants}
ref = None
for r in range(rounds):
    for ab in variants:
        # "<instance>" or "<instance>:<grid cap>"
        os.environ['DNS_PAIR_AB'] = ab.split(':')[0]
        os.environ['DNS_PAIR_GRID'] = ab.split(':')[1] if ':' in ab else '65535'
        y, secs, fb = saddle.spmv_pair(K, NV, x, reps=reps, warmup=3)
        if ref is None:
            ref = y
        assert np.array_equal(y, ref)
        out[ab].append(secs*1e6)
print(json.dumps({k: dict(us=v, median=float(np.median(v))) for k, v in out.items()}))
