#!/bin/bash
# eager-mode rocprofv3 kernel stats + per-instance tables of the mid-size
# legs (refine 2, n = 173k; refine 3, n = 693k):
#   bash scripts/profile_midsize.sh <tag> [steps]
TAG=${1:-r05_midsize}
NST=${2:-60}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export MG_SPINUP=256
for spec in "2 2048" "3 4096"; do
  set -- $spec
  L=$1; NTS=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref$L -o r$L -- python3 refined_bench.py $L $NTS $NST 0 eager > $OUT/refine${L}_eager_bench.json 2> $OUT/refine${L}_eager.err
  python scripts/instance_table.py $OUT/ref$L $OUT/refine${L}_eager_bench.json > $OUT/refine${L}_instance_table.txt
  cp $(find $OUT/ref$L -name "*kernel_stats.csv" | head -1) $OUT/refine${L}_eager_kernel_stats.csv
  rm -rf $OUT/ref$L
  timeout -k 10 300 python refined_bench.py $L $NTS 200 0 > $OUT/refine${L}_bench.json 2> $OUT/refine${L}.err || echo "refine$L failed"
  tail -c 600 $OUT/refine${L}_bench.json
done
