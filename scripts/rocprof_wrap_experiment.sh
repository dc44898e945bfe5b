#!/bin/bash
# ONE experiment with four predictions for the rocprofv3 SIGSEGV under hipGraph
# replay (profiles/r03_rocprof_graph_crash): the recorded stack is libc memcpy
# <- rocprofiler-sdk <- libamdhip64 <- hipGraphLaunch with a page-aligned fault
# address, and the runs that crash are the ones that have pushed >~ 1e4 AQL
# packets through ONE HSA queue.  Hypothesis: a graph launch writes its AQL
# packets in batches (DEBUG_HIP_GRAPH_BATCH_SIZE), and the profiler's queue
# interception copies a batch that WRAPS around the end of the ring buffer
# (ROC_AQL_QUEUE_SIZE packets) as one contiguous block -- off the end of the
# ring's mapping.  Predictions, N = 2 un-partitioned `run` (7 kernels / step):
#   A  default queue,            800 steps ( 5.6k kernels)  -> ok
#   B  ROC_AQL_QUEUE_SIZE=4096,  800 steps                  -> SIGSEGV
#   C  ROC_AQL_QUEUE_SIZE=131072, 3000 steps (21k kernels)  -> ok
#   D  DEBUG_HIP_GRAPH_BATCH_SIZE=1, 3000 steps             -> ok
# usage (GPU box): bash scripts/rocprof_wrap_experiment.sh <tag>
TAG=${1:-r04_rocprof_wrap}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() {   # name, steps  (the knobs are exported by the caller)
    export NSTEPS=$2
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o p -- \
        python3 $R/scripts/rocprof_graph_crash_probe.py run > $OUT/$1.out 2> $OUT/$1.err
    rc=$?
    calls=$(python3 - <<PY
import csv, glob
n = 0
for f in glob.glob('$OUT/$1/**/*kernel_stats.csv', recursive=True):
    n += sum(int(r['Calls']) for r in csv.DictReader(open(f)))
print(n)
PY
)
    sig=$(grep -c "SIGSEGV" $OUT/$1.err)
    echo "$1: steps $2 ROC_AQL_QUEUE_SIZE=${ROC_AQL_QUEUE_SIZE:-default} DEBUG_HIP_GRAPH_BATCH_SIZE=${DEBUG_HIP_GRAPH_BATCH_SIZE:-default} -> exit $rc, SIGSEGV lines $sig, kernels in the stats $calls, program said: $(tail -1 $OUT/$1.out)" | tee -a $OUT/summary.txt
    find $OUT/$1 -name "*kernel_trace.csv" -delete
}
unset ROC_AQL_QUEUE_SIZE DEBUG_HIP_GRAPH_BATCH_SIZE
one A_default_800 800
export ROC_AQL_QUEUE_SIZE=4096
one B_queue4096_800 800
export ROC_AQL_QUEUE_SIZE=131072
one C_queue131072_3000 3000
unset ROC_AQL_QUEUE_SIZE
export DEBUG_HIP_GRAPH_BATCH_SIZE=1
one D_batch1_3000 3000
unset DEBUG_HIP_GRAPH_BATCH_SIZE
head -12 $OUT/B_queue4096_800.err > $OUT/B_stack_head.txt
cat $OUT/summary.txt
