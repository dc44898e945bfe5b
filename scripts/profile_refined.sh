#!/bin/bash
# eager-mode rocprofv3 kernel stats + per-instance table of the refined runs
#   bash scripts/profile_refined.sh <tag>   ->  gpurun_out/<tag>/refine{3,4}_*
TAG=${1:-r03_bandwidth}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export MG_SPINUP=256
for cfg in "3 4096 60 6 auto" "4 8192 30 8 explicit"; do
  set -- $cfg
  export MG_DEG=$4 MG_FHAT=$5
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref$1 -o r$1 -- python3 refined_bench.py $1 $2 $3 0 eager > $OUT/refine$1_eager_bench.json 2> $OUT/refine$1_eager.err
  python scripts/instance_table.py $OUT/ref$1 $OUT/refine$1_eager_bench.json > $OUT/refine$1_instance_table.txt
  cp $(find $OUT/ref$1 -name "*kernel_stats.csv" | head -1) $OUT/refine$1_eager_kernel_stats.csv
  rm -rf $OUT/ref$1
  timeout -k 10 400 python refined_bench.py $1 $2 200 0 > $OUT/refine$1_graph_bench.json 2> $OUT/refine$1_graph.err
done
ls $OUT
