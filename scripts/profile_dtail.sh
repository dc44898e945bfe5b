#!/bin/bash
# graph-mode rocprofv3 kernel statistics of the partitioned step on ONE RCCL
# rank (n = 10.6k) with and without the convection cells in the lazy tail
# usage (GPU box): bash scripts/profile_dtail.sh <tag>
TAG=${1:-r04_dtail}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in 1 0; do
  export DNS_DIST_TAIL=$t
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tail$t -o p -- \
    python3 $R/bench.py --partitioned-only --gpus 1 --level 2 --steps 200 --warmup 20 --spinup 64 --start rest --no-parity \
    > $OUT/tail$t.json 2> $OUT/tail$t.err || echo "profile tail$t failed"
  find $OUT/tail$t -name "*kernel_trace.csv" -delete
  echo "== DNS_DIST_TAIL=$t"
  python3 $R/scripts/prof_stats.py "$OUT/tail$t/*kernel_stats.csv" 24
done
