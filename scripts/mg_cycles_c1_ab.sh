#!/bin/bash
# one Krylov column with two V-cycles per Schur application against two columns with one
TAG=${1:-r05_mg_cycles_c1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
run() {
  name=$1; r=$2; nts=$3; n=$4; shift 4
  env "$@" timeout -k 10 300 python3 $R/refined_bench.py $r $nts $n 0 > $OUT/$name.json 2> $OUT/$name.err
  python3 - $OUT/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['gpu_steps_per_s'],1), d['krylov_iters_per_step'], d['true_relres_last'], d['run_record'])
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
run r2_base 2 2048 300 X=1
run r2_cyc2_cmin1 2 2048 300 DNS_MG_CYCLES=2 DNS_OVERSOLVE_CMIN=1
run r2_cyc2 2 2048 300 DNS_MG_CYCLES=2
run r2_cmin1 2 2048 300 DNS_OVERSOLVE_CMIN=1
run r3_cyc2_cmin1 3 4096 300 DNS_MG_CYCLES=2 DNS_OVERSOLVE_CMIN=1
