#!/bin/bash
# kernel statistics of a refine-2 run with the 5000-row level as a dense
# half-precision inverse:  bash scripts/profile_dense_half.sh <tag>
TAG=${1:-r05_dense_half_prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export MG_SPINUP=256
export DNS_MG_DENSE_HALF_MAX=6000
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref2 -o r2 -- python3 refined_bench.py 2 2048 60 0 eager > $OUT/refine2_eager_bench.json 2> $OUT/refine2_eager.err
python3 scripts/prof_stats.py "$OUT/ref2/*kernel_stats.csv" 24 > $OUT/kernel_stats.txt
rm -rf $OUT/ref2
cut -c1-160 $OUT/kernel_stats.txt
