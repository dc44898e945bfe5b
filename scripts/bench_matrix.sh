#!/bin/bash
# usage: run_bench_matrix.sh "<args1>" "<args2>" ...
for a in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu --no-refined --roofline-refine 0 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$a', '| steps/s %.0f ms/step %.4f its %.2f setup %.3f relres %.2e' % (d['value'], d['ms_per_step'], c['krylov_iters_per_step'], c['precond_setup_s'], c['true_relres_last']))"
done
