#!/bin/bash
# The measurement bundle of a round, ONE gpurun call on one box:
#   bash scripts/final_profiles.sh <tag>      -> gpurun_out/<tag>/
# (copy what is to be judged into profiles/<tag>/ afterwards)
set -o pipefail
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# PMC passes FIRST: bench.py prints roofline.traffic only from a stamp taken on the sources as they are
cd /tmp && export TMPDIR=/tmp
echo "== PMC passes (separate), SpMV alone"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $R/scripts/spmv_roofline.py 4 10 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python3 $R/scripts/spmv_roofline.py 4 10 > $OUT/pmc_write.json 2> $OUT/pmc_write.err || echo "pmc write failed"
cd $R
python scripts/pmc_summary.py "$OUT/pmc_*/*counter_collection.csv" > $OUT/pmc_summary.json
python scripts/stamp_traffic.py $OUT/pmc_summary.json > $OUT/traffic_stamp.json
cd $R
echo "== bench (default)"; timeout -k 10 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
echo "== bench (driver window)"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_window.json 2> $OUT/bench_driver_window.err || echo "driver window failed"
echo "== refined legs"
timeout -k 10 300 python refined_bench.py 2 2048 200 0 > $OUT/refine2_bench.json 2> $OUT/refine2.err || echo "refine2 failed"
timeout -k 10 300 python refined_bench.py 3 4096 400 0 > $OUT/refine3_bench.json 2> $OUT/refine3.err || echo "refine3 failed"
timeout -k 10 400 python refined_bench.py 4 8192 200 0 > $OUT/refine4_bench.json 2> $OUT/refine4.err || echo "refine4 failed"
echo "== SBDF2, partitioned path on one RCCL rank"
timeout -k 10 300 python bench.py --scheme sbdf2 --no-cpu --no-refined --no-picard --roofline-refine 0 > $OUT/bench_sbdf2.json 2> $OUT/bench_sbdf2.err || echo "sbdf2 failed"
timeout -k 10 300 python bench.py --partitioned-only --gpus 1 --level 2 --steps 400 --warmup 40 --spinup 256 > $OUT/partitioned_one_rank_n10k.json 2> $OUT/partitioned_one_rank_n10k.err || echo "partitioned n10k failed"
timeout -k 10 300 python bench.py --partitioned-only --gpus 1 --level 2 --refine 2 --nts 2048 --steps 200 --warmup 20 --spinup 256 > $OUT/partitioned_one_rank_n173k.json 2> $OUT/partitioned_one_rank_n173k.err || echo "partitioned n173k failed"
timeout -k 10 300 python bench.py --partitioned-only --gpus 1 --level 2 --refine 3 --nts 4096 --steps 200 --warmup 20 --spinup 256 > $OUT/partitioned_one_rank_n693k.json 2> $OUT/partitioned_one_rank_n693k.err || echo "partitioned n693k failed"
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 kernel stats of the bench command (graph replay)"
# (a ring of 131072 AQL packets never wraps inside these runs: rocprofv3 7.2
# dies on a graph launch whose packet batch wraps the ring, profiles/r04_rocprof_wrap)
export ROC_AQL_QUEUE_SIZE=131072
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 $R/bench.py --no-cpu --no-refined --no-picard --steps 200 --warmup 20 --spinup 64 > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err || echo "profiled bench failed"
echo "== rocprofv3 kernel stats of the step alone (graph replay / eager)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step_graph -o step -- python3 $R/bench.py --profile-step --steps 2000 --warmup 40 --spinup 256 > $OUT/step_graph.json 2> $OUT/step_graph.err || echo "step graph profile failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step_eager -o step -- python3 $R/bench.py --profile-step --eager --steps 400 --warmup 40 > $OUT/step_eager.json 2> $OUT/step_eager.err || echo "step eager profile failed"
unset ROC_AQL_QUEUE_SIZE
cd $R
python - <<PY
import json, glob, csv
out='$OUT'
def last(f):
    try:
        return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        return None
for name in ('bench','bench_driver_window','bench_profiled'):
    d=last(out+'/'+name+'.json')
    if d:
        r=d['roofline']; c=d['config']
        print(name, 'value %.0f its %.3f roof %.0f GB/s (%.3f) avg_us %s step frac %.3f refined %s' % (
            d['value'], c['krylov_iters_per_step'], r['achieved'] or 0, r['frac'] or 0,
            (r.get('detail') or {}).get('avg_us'), r['step']['frac'],
            (r.get('step_refined') or {}).get('frac')))
for f in glob.glob(out+'/prof_bench/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_spmv_stream16<' in row['Name']:
            print('rocprof', row['Name'][:60], row['Calls'], row['AverageNs'])
print(open(out+'/pmc_summary.json').read()[:1500])
PY
# trim: the traces are large, the stats are what is kept
find $OUT -name "*kernel_trace.csv" -size +3M -delete
find $OUT -name "*counter_collection.csv" -size +3M -delete
ls $OUT
