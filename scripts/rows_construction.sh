#!/bin/bash
# Rank-local construction (dns_saddle_create_rows) against whole-matrix
# construction on the same communicator: NR gloo-staged ranks on this box's ONE
# GPU, bench.py's partitioned leg on the mesh refined REFINE times, set-up laps
# (DNS_DEBUG) in the .err files.   bash scripts/rows_construction.sh <tag> <refine> <nts> <nranks>
TAG=${1:-rows_construction}
REFINE=${2:-3}
NTS=${3:-4096}
NR=${4:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export DNS_BENCH_REHEARSE_ONE_GPU=1 GLOO_SOCKET_IFNAME=lo DNS_DEBUG=1
# (the ranks share this box's 16 cores: each gets its share of host threads)
export DNS_HOST_THREADS=$((16 / NR))
PORT=29711
for kind in rows whole; do
  timeout -k 10 ${LIMIT:-500} python -m torch.distributed.run --nnodes=1 --nproc-per-node $NR --master-addr 127.0.0.1 --master-port $PORT \
    bench.py --partitioned-only --gpus $NR --level 2 --refine $REFINE --nts $NTS --steps 10 --warmup 2 --spinup 8 --construction $kind \
    > $OUT/refine${REFINE}_${NR}ranks_$kind.json 2> $OUT/refine${REFINE}_${NR}ranks_$kind.err || echo "$kind failed"
  PORT=$((PORT + 20))
done
python - <<PY
import json
for kind in ('rows', 'whole'):
    try:
        d = json.loads(open('$OUT/refine${REFINE}_${NR}ranks_%s.json' % kind).read().strip().splitlines()[-1])
    except Exception as exc:
        print(kind, 'no line:', exc)
        continue
    c = d['construction']
    print(kind, 'n', d['unknowns'], 'create+setup %.2f s' % c['create_and_setup_s'],
          'host kept %.1f MB, set-up %.1f MB' % (c['host_matrix_bytes_kept']/1e6, c['host_matrix_bytes_setup']/1e6),
          'device %.1f MB' % (d['matrix_bytes_per_rank_max']/1e6), 'rss', c['process_peak_rss_mb'],
          'parity', d['parity']['v_rel_Mnorm'], d['parity']['p_rel_l2'], d.get('error'))
PY
grep -h "setup" $OUT/refine${REFINE}_${NR}ranks_rows.err | head -40
