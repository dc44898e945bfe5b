#!/bin/bash
# A/B of the oversolve policy on the mid-size legs (one box, back to back):
#   bash scripts/oversolve_ab.sh <tag>
TAG=${1:-r05_oversolve}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for spec in "2 2048 400" "3 4096 400" "4 8192 100"; do
  set -- $spec
  for ov in 0 1; do
    DNS_OVERSOLVE=$ov timeout -k 10 400 python refined_bench.py $1 $2 $3 0 > $OUT/refine$1_over$ov.json 2> $OUT/refine$1_over$ov.err || echo "refine $1 over $ov failed"
    python - <<PY
import json
d=json.loads(open('$OUT/refine$1_over$ov.json').read().strip().splitlines()[-1])
print('refine $1 oversolve $ov: %.0f steps/s, %.2f Krylov/step, relres %.1e, frac %.3f, run %s' % (d['gpu_steps_per_s'], d['krylov_iters_per_step'], d['true_relres_last'], d['roofline_step']['frac'], d['run_record']))
PY
  done
done
