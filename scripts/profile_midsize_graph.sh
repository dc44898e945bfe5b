#!/bin/bash
# graph-replay kernel trace of the mid-size legs + timeline of a step
#   bash scripts/profile_midsize_graph.sh <tag> [steps]
TAG=${1:-r05_midsize_graph}
NST=${2:-100}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export MG_SPINUP=128 ROC_AQL_QUEUE_SIZE=131072
for spec in "2 2048 k_imex_bvec" "3 4096 k_imex_bvec"; do
  set -- $spec
  L=$1; NTS=$2; MARK=$3
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref$L -o r$L -- python3 refined_bench.py $L $NTS $NST 0 > $OUT/refine${L}_graph_bench.json 2> $OUT/refine${L}_graph.err
  python scripts/step_timeline.py $OUT/ref$L $MARK 50 > $OUT/refine${L}_step_timeline.txt
  cp $(find $OUT/ref$L -name "*kernel_stats.csv" | head -1) $OUT/refine${L}_graph_kernel_stats.csv
  rm -rf $OUT/ref$L
  head -50 $OUT/refine${L}_step_timeline.txt
done
