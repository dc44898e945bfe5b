#!/bin/bash
# A/B of the two round-5 levers of the mid-size legs, one box, back to back:
# oversolve (DNS_OVERSOLVE=0/1) x warm start (quartic / cubic)
#   bash scripts/oversolve_ab.sh <tag>
TAG=${1:-r05_oversolve}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
: > $OUT/table.txt
for spec in "2 2048 400" "3 4096 400" "4 8192 100"; do
  set -- $spec
  for ov in 0 1; do
    for ex in 4 3; do
      DNS_OVERSOLVE=$ov MG_EXTRAP=$ex timeout -k 10 400 python refined_bench.py $1 $2 $3 0 > $OUT/refine$1_over${ov}_ex$ex.json 2> $OUT/refine$1_over${ov}_ex$ex.err || echo "refine $1 over $ov ex $ex failed"
      python - <<PY | tee -a $OUT/table.txt
import json
d=json.loads(open('$OUT/refine$1_over${ov}_ex$ex.json').read().strip().splitlines()[-1])
print('refine $1 (n = %d) oversolve $ov warm start order $ex: %5.0f steps/s, %.2f Krylov columns per time step, true relres %.1e, op-list frac %.3f, replayed %d of %d, timed attempts %d' % (d['n'], d['gpu_steps_per_s'], d['krylov_iters_per_step'], d['true_relres_last'], d['roofline_step']['frac'], d['run_record']['replayed'], d['steps'], d['timed_attempts']))
PY
    done
  done
done
