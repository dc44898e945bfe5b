#!/bin/bash
# aggressive coarsening (two prolongations as one): bash scripts/mg_collapse_ab.sh <tag>
TAG=${1:-r05_mg_collapse}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python3 $R/refined_bench.py 2 2048 300 0 > $OUT/$name.json 2> $OUT/$name.err
  python3 - $OUT/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['gpu_steps_per_s'],1), d['krylov_iters_per_step'], round(d['roofline_step']['frac'],3), d['true_relres_last'])
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
run base X=1
run collapse2 MG_COLLAPSE=2
run collapse2_nu3 MG_COLLAPSE=2 MG_NU=3
run collapse2_nu4 MG_COLLAPSE=2 MG_NU=4
run base_nu3 MG_NU=3
