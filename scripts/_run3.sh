set -o pipefail
mkdir -p gpurun_out/r03_part6
timeout -k 10 600 python -m pytest tests/test_gpu_zz_dist.py -x -q > gpurun_out/r03_part6/dist_tests.log 2>&1; echo "dist tests rc $?"; tail -4 gpurun_out/r03_part6/dist_tests.log
timeout -k 10 200 python bench.py --partitioned-only --gpus 1 --level 2 --steps 400 --warmup 40 --spinup 256 > gpurun_out/r03_part6/n10k.json 2> gpurun_out/r03_part6/n10k.err || echo "n10k failed"
DNS_BENCH_REHEARSE_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --partitioned-only --gpus 2 --level 2 --steps 100 --warmup 20 --spinup 64 > gpurun_out/r03_part6/two_rank_gloo.json 2> gpurun_out/r03_part6/two_rank_gloo.err || echo "two-rank failed"
python - <<'PY'
import json
for f in ('n10k','two_rank_gloo'):
    try:
        r=json.loads([l for l in open('gpurun_out/r03_part6/%s.json'%f).read().strip().splitlines() if l.startswith('{')][-1])
        print(f, round(r['steps_per_s'],1), 'its', r['krylov_iters_per_step'], 'relres', r['true_relres_last'], r['collectives_timed_window'], r['steps'])
    except Exception as e:
        print(f, 'no result', e)
PY
