#!/bin/bash
# workgroups (= partials) of the K apply with fused dots: bash scripts/stream_grid_ab.sh <tag>
TAG=${1:-r05_stream_grid}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for spec in "3 4096 300" "4 8192 150" "2 2048 300"; do
  set -- $spec
  for g in 1024 2048 4096; do
    DNS_STREAM_GRID=$g timeout -k 10 400 python3 $R/refined_bench.py $1 $2 $3 0 > $OUT/refine$1_grid$g.json 2> $OUT/refine$1_grid$g.err
    python3 - $OUT/refine$1_grid$g.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['gpu_steps_per_s'],1), d['krylov_iters_per_step'], round(d['roofline_step']['frac'],3))
PY
  done
done
