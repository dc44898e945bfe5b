#!/bin/bash
TAG=${1:-r05_mg_cycles_c1b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
run() {
  name=$1; r=$2; nts=$3; n=$4; cpu=$5; shift 5
  env "$@" timeout -k 10 500 python3 $R/refined_bench.py $r $nts $n $cpu > $OUT/$name.json 2> $OUT/$name.err
  python3 - $OUT/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['gpu_steps_per_s'],1), d['krylov_iters_per_step'], d['true_relres_last'], [v for k,v in d.items() if k.startswith('parity')], round(d['roofline_step']['frac'],3))
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
run r4_cyc2_cmin1 4 8192 150 0 DNS_MG_CYCLES=2 DNS_OVERSOLVE_CMIN=1
run r4_cyc2_cmin1_over 4 8192 150 0 DNS_MG_CYCLES=2 DNS_OVERSOLVE_CMIN=1 DNS_OVERSOLVE=1
run r2_cyc2_cmin1_dbg 2 2048 400 1 DNS_MG_CYCLES=2 DNS_OVERSOLVE_CMIN=1 DNS_DEBUG=1
run r2_cyc3_cmin1 2 2048 400 0 DNS_MG_CYCLES=3 DNS_OVERSOLVE_CMIN=1
run r3_cyc2_cmin1_cpu 3 4096 300 1 DNS_MG_CYCLES=2 DNS_OVERSOLVE_CMIN=1
grep "imex_run\|batch" $OUT/r2_cyc2_cmin1_dbg.err | tail -12 | cut -c1-250
