#!/bin/bash
# A/B of the multigrid smoother: DNS_MG_CHEB=0 (damped Jacobi) / 1 (Chebyshev pair)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/abcheb
for cfg in "2 2048 200" "3 4096 100" "4 8192 40"; do
  set -- $cfg
  for ch in "0 4" "1 4" "1 8" "1 3"; do
    set -- $cfg $ch
    export DNS_MG_CHEB=$4 DNS_MG_CHEB_ALPHA=$5
    if [ "$1" = "4" ]; then export MG_FHAT=explicit; else unset MG_FHAT; fi
    timeout -k 10 400 python refined_bench.py $1 $2 $3 0 > gpurun_out/abcheb/r$1_c$4_a$5.json 2> gpurun_out/abcheb/r$1_c$4_a$5.err
    python - <<PY
import json
try:
    r=json.loads(open("gpurun_out/abcheb/r$1_c$4_a$5.json").read().strip().splitlines()[-1])
    print("refine $1 cheb $4 alpha $5: %.0f steps/s, %.2f its/step, frac %.3f, setup %.1f s" % (r['gpu_steps_per_s'], r['krylov_iters_per_step'], r['roofline_step']['frac'], r['setup_s']))
except Exception as e:
    print("refine $1 cheb $4 alpha $5: FAILED", e)
PY
  done
done
