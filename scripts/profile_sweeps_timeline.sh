#!/bin/bash
# per-step timeline of the Newton/Picard sweeps inside their graph replays:
#   bash scripts/profile_sweeps_timeline.sh <tag>
TAG=${1:-r05_sweeps_tl}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROC_AQL_QUEUE_SIZE=131072
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o s -- \
  python3 $R/scripts/sweep_once.py 2 > $OUT/sweeps.json 2> $OUT/sweeps.err || echo "profile failed"
python3 $R/scripts/step_timeline.py $OUT/prof k_conv_step_cells 60 > $OUT/sweep_step_timeline.txt
python3 $R/scripts/prof_stats.py "$OUT/prof/*kernel_stats.csv" 30 > $OUT/sweep_kernel_stats.txt
find $OUT/prof -name "*kernel_trace.csv" -delete
cat $OUT/sweeps.json; head -70 $OUT/sweep_step_timeline.txt
