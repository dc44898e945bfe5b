#!/bin/bash
# rocprofv3 kernel stats (csv) of one command on the GPU box:
#   bash scripts/prof.sh <tag> python3 <script> [args]   -> gpurun_out/<tag>/
# (the program itself follows the tag: no env / bash -c wrappers under rocprofv3)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- "$@" > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "rocprofv3 exit $?"
find $OUT -name "*kernel_trace.csv" -size +3M -delete
python scripts/prof_stats.py "$OUT/**/*kernel_stats.csv" 40 2>/dev/null || python scripts/prof_stats.py "$OUT/*/*kernel_stats.csv" 40
