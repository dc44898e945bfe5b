#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/abcyc
export DNS_MG_CHEB=1 DNS_MG_CHEB_ALPHA=3
for cfg in "2 2048 200 6" "3 4096 100 6" "4 8192 40 8"; do
  set -- $cfg
  for cy in "1 0.3" "2 0.3" "2 0.5" "2 0.15"; do
    set -- $cfg $cy
    export MG_DEG=$4 DNS_MG_CYCLES=$5 DNS_MG_RHO=$6
    if [ "$1" = "4" ]; then export MG_FHAT=explicit; else unset MG_FHAT; fi
    timeout -k 10 500 python refined_bench.py $1 $2 $3 0 > gpurun_out/abcyc/r$1_c$5_$6.json 2> gpurun_out/abcyc/r$1_c$5_$6.err
    python - <<PY
import json
try:
    r=json.loads(open("gpurun_out/abcyc/r$1_c$5_$6.json").read().strip().splitlines()[-1])
    print("refine $1 deg $4 cycles $5 rho $6: %.0f steps/s, %.2f its/step, relres %.1e setup %.1f s" % (r['gpu_steps_per_s'], r['krylov_iters_per_step'], r['true_relres_last'], r['setup_s']))
except Exception as e:
    print("refine $1 cycles $5 rho $6: FAILED", e)
PY
  done
done
