#!/bin/bash
# graph-mode rocprofv3 kernel statistics of the Newton/Picard sweeps (N = 2,
# Re = 100, developed shedding):  bash scripts/profile_sweeps.sh <tag> [steps]
TAG=${1:-r04_sweeps_prof}
NST=${2:-512}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROC_AQL_QUEUE_SIZE=131072
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o s -- \
  python3 $R/scripts/sweep_horizon_probe.py $NST 1024 1e-10 3.0 3 > $OUT/probe.json 2> $OUT/probe.err || echo "profile failed"
find $OUT/prof -name "*kernel_trace.csv" -delete
python3 $R/scripts/prof_stats.py "$OUT/prof/*kernel_stats.csv" 40
grep -E "picard|newton" $OUT/probe.err
