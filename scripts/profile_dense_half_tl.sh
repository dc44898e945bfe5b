#!/bin/bash
# graph-replay step timelines of refine 2 with the 5000-row level sparse and
# as a dense half-precision inverse:  bash scripts/profile_dense_half_tl.sh <tag>
TAG=${1:-r05_dense_half_tl}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export MG_SPINUP=256 ROC_AQL_QUEUE_SIZE=131072
for hm in 0 6000; do
  export DNS_MG_DENSE_HALF_MAX=$hm
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref_$hm -o r2 -- python3 refined_bench.py 2 2048 100 0 > $OUT/refine2_half${hm}_bench.json 2> $OUT/refine2_half${hm}.err
  python3 scripts/step_timeline.py $OUT/ref_$hm k_imex_bvec 50 > $OUT/refine2_half${hm}_step_timeline.txt
  rm -rf $OUT/ref_$hm
  head -60 $OUT/refine2_half${hm}_step_timeline.txt
done
