#!/bin/bash
# refined-mesh legs after hoisting the guard test behind the first loads
TAG=${1:-r05_guard_hoist}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for spec in "2 2048 400" "3 4096 300" "4 8192 150"; do
  set -- $spec
  timeout -k 10 400 python3 $R/refined_bench.py $1 $2 $3 0 > $OUT/refine$1.json 2> $OUT/refine$1.err
  python3 - $OUT/refine$1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['gpu_steps_per_s'],1), d['krylov_iters_per_step'], round(d['roofline_step']['frac'],3), d['true_relres_last'])
PY
done
