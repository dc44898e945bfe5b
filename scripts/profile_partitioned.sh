#!/bin/bash
# eager-mode rocprofv3 kernel statistics of the partitioned code path on ONE
# RCCL rank (latency regime n = 10.6k, bandwidth regime n = 693k) -- the table
# to hold against the un-partitioned step's (profiles/r03_b_final/step_eager*,
# profiles/r03_bandwidth/refine3_eager*)
# usage (GPU box): bash scripts/profile_partitioned.sh <tag>
set -o pipefail
TAG=${1:-r03_partitioned}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== n10k eager"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_n10k -o n10k -- \
  python bench.py --partitioned-only --gpus 1 --level 2 --steps 200 --warmup 20 --spinup 0 --eager \
  > $OUT/n10k_eager.json 2> $OUT/n10k_eager.err || echo "n10k profile failed"
echo "== n693k eager"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_n693k -o n693k -- \
  python bench.py --partitioned-only --gpus 1 --level 2 --refine 3 --nts 4096 --steps 60 --warmup 10 --spinup 0 --eager \
  > $OUT/n693k_eager.json 2> $OUT/n693k_eager.err || echo "n693k profile failed"
for f in $(find $OUT -name '*kernel_stats.csv'); do echo $f; python scripts/prof_stats.py $f 30; done
