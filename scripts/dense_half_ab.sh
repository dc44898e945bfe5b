#!/bin/bash
# refined-mesh CNAB runs with the 5000-row multigrid level sparse / as a dense
# inverse in half precision:  bash scripts/dense_half_ab.sh <tag> [refines]
TAG=${1:-r05_dense_half}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for r in ${2:-2 3}; do for hm in 0 6000; do
  n=400; nts=2048
  [ $r -ge 3 ] && { n=300; nts=4096; }
  [ $r -ge 4 ] && { n=200; nts=8192; }
  DNS_MG_DENSE_HALF_MAX=$hm timeout -k 10 500 python3 $R/refined_bench.py $r $nts $n 0 > $OUT/refine${r}_half${hm}.json 2> $OUT/refine${r}_half${hm}.err
  python3 - $OUT/refine${r}_half${hm}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['gpu_steps_per_s'],1), d['krylov_iters_per_step'], round(d['roofline_step']['frac'],3), 'setup', round(d['setup_s'],2), d['true_relres_last'])
PY
done; done
