#!/bin/bash
# eager-mode rocprofv3 kernel stats + per-instance table of the refine-4 run
# (n = 2.78M) alone:  bash scripts/profile_refine3.sh <tag> [steps]
TAG=${1:-r04_refine3}
NST=${2:-30}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export MG_SPINUP=256 MG_DEG=8 MG_FHAT=explicit
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref3 -o r4 -- python3 refined_bench.py 3 4096 $NST 0 eager > $OUT/refine3_eager_bench.json 2> $OUT/refine3_eager.err
python scripts/instance_table.py $OUT/ref3 $OUT/refine3_eager_bench.json > $OUT/refine3_instance_table.txt
cp $(find $OUT/ref3 -name "*kernel_stats.csv" | head -1) $OUT/refine3_eager_kernel_stats.csv
rm -rf $OUT/ref3
grep -E "conv|bvec|imex" $OUT/refine3_instance_table.txt
