#!/bin/bash
# Rehearsal of `bench.py --gpus 2` as the driver launches it, both ranks on this
# box's ONE GPU through the gloo-staged communicator (RCCL refuses two ranks per
# device): the N > 1 line with its parity record and collective timings, and
# the same launch with ONE wrong halo entry (DNS_TEST_CORRUPT_HALO), which has
# to turn the leg red.   bash scripts/rehearse_n2.sh <tag>
TAG=${1:-r04_partitioned}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export DNS_BENCH_REHEARSE_ONE_GPU=1 GLOO_SOCKET_IFNAME=lo
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 2 --steps 20 --warmup 5 --partitioned-timeout 400 \
  > $OUT/rehearsal_two_ranks_one_gpu_gloo.json 2> $OUT/rehearsal.err || echo "rehearsal failed"
export DNS_TEST_CORRUPT_HALO=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 \
  bench.py --gpus 2 --steps 20 --warmup 5 --no-bandwidth --no-strong --no-ensemble --partitioned-timeout 300 \
  > $OUT/rehearsal_corrupted_halo.json 2> $OUT/rehearsal_corrupted.err || echo "corrupted rehearsal failed"
unset DNS_TEST_CORRUPT_HALO
python - <<PY
import json
for name in ('rehearsal_two_ranks_one_gpu_gloo', 'rehearsal_corrupted_halo'):
    try:
        d = json.loads(open('$OUT/%s.json' % name).read().strip().splitlines()[-1])
    except Exception as exc:
        print(name, 'no line:', exc)
        continue
    c = d['config']
    w = c['weak_scaling']
    print(name, 'value', d['value'], '|', c['parallelism'][:60])
    print('   parity', d.get('parity'))
    print('   weak error', w.get('error'), '| start', w.get('start_state'))
    for leg in ('weak_scaling_bandwidth', 'strong_scaling'):
        l = c.get(leg)
        if isinstance(l, dict):
            print('  ', leg, l.get('steps_per_s'), l.get('parity'), l.get('error'))
    t = w.get('collectives_device_time')
    if t:
        print('   device us per call', {k: t[k]['us_per_call'] for k in ('allreduce', 'allgatherv', 'halo_exchange')})
PY
