"""Host logic of `bench.py` and of the published op list (`perfmodel.py`):
what can be checked without a GPU."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench                                             # noqa: E402
from dolfin_navier_scipy_amd import perfmodel            # noqa: E402


def test_weak_ladder_keeps_rows_per_rank_about_constant():
    """N > 1 headline: the mesh grows with the ranks"""
    assert bench.weak_ladder(1) == (2, 0)
    assert bench.weak_ladder(2) == (3, 0)
    assert bench.weak_ladder(4) == (2, 1)
    assert bench.weak_ladder(8) == (3, 1)
    rel = dict(bench.WEAK_LADDER)
    for world in (1, 2, 3, 4, 6, 8):
        per_rank = rel[bench.weak_ladder(world)]/world
        assert 0.6 <= per_rank <= 1.5, (world, per_rank)


def test_child_env_drops_the_launcher_agent_store(monkeypatch):
    """a child run hosts its own rendezvous store: with
    TORCHELASTIC_USE_AGENT_STORE in its environment rank 0 would wait for the
    launcher's agent on a port nobody serves (the round-1 rehearsal hung on
    exactly this)"""
    monkeypatch.setenv('TORCHELASTIC_USE_AGENT_STORE', 'True')
    monkeypatch.setenv('TORCHELASTIC_RUN_ID', 'x')
    monkeypatch.setenv('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    monkeypatch.setenv('RANK', '3')
    env = bench.child_env(MASTER_PORT=29517, MASTER_ADDR='127.0.0.1')
    assert not [k for k in env if k.startswith('TORCHELASTIC_')]
    assert env['MASTER_PORT'] == '29517' and env['MASTER_ADDR'] == '127.0.0.1'
    assert env['RANK'] == '3' and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_run_child_reports_results_failures_and_time_limits():
    py = sys.executable
    ok = bench.run_child([py, '-c', 'print("noise"); print(\'{"a": 1}\')'],
                         dict(os.environ), 30., True)
    assert ok == {'a': 1}
    bad = bench.run_child([py, '-c', 'import sys; sys.exit(3)'],
                          dict(os.environ), 30., True)
    assert 'error' in bad and 'code 3' in bad['error']
    quiet = bench.run_child([py, '-c', 'pass'], dict(os.environ), 30., False)
    assert quiet is None
    slow = bench.run_child([py, '-c', 'import time; time.sleep(60)'],
                           dict(os.environ), 1., True)
    assert 'error' in slow and 'killed' in slow['error']


def _info(schur='dense'):
    info = dict(NV=9356, NP=1289, nnz_K=300000, nnz_Gc=1100000, nnz_JG=350000,
                nnz_F=215000, nnz_J=43000, fp32_store=True, cheb_degree=6,
                schur=schur, mg_nu=2, mg_levels=[])
    if schur == 'mg':
        info['mg_levels'] = [dict(n=1289, nnz_S=90000, nnz_P=5000),
                             dict(n=340, nnz_S=0, nnz_P=0)]
    return info


@pytest.mark.parametrize('schur', ['dense', 'mg'])
def test_step_roofline_is_the_sum_of_its_published_ops(schur):
    info = _info(schur)
    roof = perfmodel.step_roofline(info, nnz_R1=215000, ncells=4600,
                                   iters=1.5, ms_per_step=0.05)
    total = sum(op['count']*op['bytes'] for op in roof['ops'])
    assert abs(total - roof['bytes_per_step']) <= 1e-6*total
    assert abs(roof['achieved'] - total/0.05e-3/1e9) <= 1e-6*roof['achieved']
    assert abs(roof['frac'] - roof['achieved']/8000.) <= 1e-12
    # CSR SpMV = 12 nnz + 4 (r+1) + 8 c + 8 r
    assert perfmodel.spmv_bytes(10, 4, 5) == 120 + 20 + 40 + 32
    # what the kernels move (fp32 Gc) is less than the fp64 / int32 op list
    assert roof['bytes_moved_estimate'] < roof['bytes_per_step']
    # one more Krylov step costs exactly the ops of that step
    more = perfmodel.step_roofline(info, 215000, 4600, 2.5, 0.05)
    extra = sum(c*b for _, c, b in perfmodel.krylov_iteration_ops(info, 2))
    n = info['NV'] + info['NP']
    assert abs(more['bytes_per_step'] - roof['bytes_per_step']
               - 0.5*sum(c*b for _, c, b in
                         perfmodel.krylov_iteration_ops(info, 1))
               - 0.5*extra
               - 8*n*1.0            # (x = x0 + Z y reads one more Z_j)
               ) <= 1e-6*total


def test_spmv_bytes_of_the_bench_match_the_model():
    import scipy.sparse as sps
    A = sps.random(50, 40, density=0.1, format='csr', random_state=1)
    assert bench.spmv_bytes(A) == perfmodel.spmv_bytes(A.nnz, 50, 40)


def test_multi_rank_launch_plumbing_on_cpu(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run,
    one process per rank), with `--dry-run` children: parent ranks over gloo,
    child processes per rank with their own rendezvous port and without the
    launcher's TORCHELASTIC_* variables, the failure all-reduce, ONE JSON line
    on rank 0's stdout with the weak-scaling headline, the strong-scaling and
    the ensemble legs"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run',
           '--partitioned-timeout', '120']
    env = dict(os.environ, GLOO_SOCKET_IFNAME='lo')
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines()
             if ln.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['scaling'] == 'weak'
    assert rec['unit'] == 'timesteps/s' and rec['higher_is_better'] is True
    cfg = rec['config']
    assert cfg['weak_scaling']['level'] == 3            # the ladder's mesh
    assert rec['value'] == cfg['weak_scaling']['steps_per_s']
    assert cfg['strong_scaling']['level'] == 2
    assert cfg['collectives'] == {'allreduce': 1}
    assert cfg['ensemble']['error'] is None
    assert 'FALLBACK' not in cfg['parallelism']
    # the N > 1 line carries the parity record of its headline leg, and the
    # latency-regime legs start where the N = 1 headline starts
    assert rec['parity']['ok'] is True and rec['parity']['tol'] == 1e-8
    assert cfg['weak_scaling']['start_state'] == 'stokes'
    assert cfg['strong_scaling']['start_state'] == 'stokes'
    assert cfg['weak_scaling_bandwidth']['start_state'] == 'rest'


def _json_lines(raw):
    return [ln for ln in raw.decode().splitlines() if ln.startswith('{')]


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver's N=1
    record shows the script being started): the script starts its two rank
    processes itself and prints an `n_gpus: 2` line carrying the RCCL
    first-contact record -- never the one-GPU line"""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT')}
    env['GLOO_SOCKET_IFNAME'] = 'lo'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                          '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--dry-run', '--partitioned-timeout', '120'],
                         env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert 'without a launcher' in out.stderr.decode()
    lines = _json_lines(out.stdout)
    assert len(lines) == 1, out.stdout.decode()
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['scaling'] == 'weak'
    st = rec['config']['rccl_selftest']
    assert st['complete'] is True and st['world'] == 2
    assert sorted(st['primitives']) == sorted(
        name + suffix for name, _ in bench.SELFTEST_LEGS[::2]
        for suffix in ('_eager', '_graph'))
    assert 'FALLBACK' not in rec['config']['parallelism']


def test_a_launch_for_another_n_is_refused():
    """`--gpus 8` inside a launch of 2 ranks (or of 1) prints no line"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'),
                          '--gpus', '8', '--dry-run'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=120)
    assert out.returncode != 0
    assert not _json_lines(out.stdout)
    assert 'refusing' in out.stderr.decode()


def test_self_launch_refuses_more_ranks_than_gpus(monkeypatch):
    import argparse
    monkeypatch.setattr(bench, 'visible_gpu_count', lambda: 1)
    monkeypatch.delenv('DNS_BENCH_REHEARSE_ONE_GPU', raising=False)
    args = argparse.Namespace(gpus=4, dry_run=False, time_budget=10.)
    with pytest.raises(SystemExit) as exc:
        bench.self_launch(args)
    assert exc.value.code == 2


def test_run_child_keeps_the_last_record_of_a_killed_child():
    """the self-test prints its record after every primitive: what a hung
    child had reported says which primitive did not come back"""
    code = ('import sys, time, json; print(json.dumps(dict(stage="entering '
            'allreduce_graph"))); sys.stdout.flush(); time.sleep(60)')
    res = bench.run_child([sys.executable, '-c', code], dict(os.environ), 2.,
                          True)
    assert 'killed' in res['error']
    assert res['partial'] == {'stage': 'entering allreduce_graph'}


def test_the_printed_line_stays_under_the_drivers_record(tmp_path):
    """the driver keeps an 8 KB tail of the line: the published op lists and
    the long explanations go to a side file, secondary records keep their
    figures of merit, what the bench contract names stays"""
    ops = [dict(op='SpMV {0}'.format(i), count=1.5, bytes=123456789.123)
           for i in range(40)]
    big = dict(steps_per_s=1234.56789123, krylov_iters_per_step=2.0,
               parity=dict(v_rel_Mnorm=1.23456789e-10, ok=True),
               what='x'*900, roofline_step=dict(frac=0.5, ops=list(ops)),
               padding={str(i): 0.123456789123*i for i in range(400)})
    out = dict(metric='timesteps/sec', value=27796.437291, unit='timesteps/s',
               config=dict(workload='w'*200, parallelism='single',
                           row_partitioned=dict(big),
                           weak_scaling_bandwidth_base=dict(big),
                           newton_picard_sweeps=dict(
                               picard=dict(steps_per_s=11000.123456)),
                           refined_mesh=dict(gpu_steps_per_s=4200.987654321)),
               roofline=dict(bound='hbm', achieved=6480.123456789, peak=8000.,
                             frac=0.810015432, kernel='k'*150,
                             step=dict(frac=0.14, ops=list(ops))),
               cpu_baseline=dict(value=394.123, cores=1, kind='port',
                                 sample='s'*170))
    assert len(json.dumps(out)) > 2*bench.LINE_LIMIT
    side = str(tmp_path / 'ops.json')
    line, moved = bench.compact_line(out, side=[side])
    assert len(json.dumps(line)) <= bench.LINE_LIMIT
    assert line['config']['workload'] == 'w'*200
    assert line['cpu_baseline']['sample'] == 's'*170
    assert line['roofline']['kernel'] == 'k'*150
    assert line['roofline']['frac'] == 0.810015
    assert line['value'] == 27796.4
    assert line['roofline']['step']['ops'].startswith('profiles/bench_ops')
    rp = line['config']['row_partitioned']
    assert rp['steps_per_s'] == 1234.57 and rp['parity']['ok'] is True
    assert line['config']['newton_picard_sweeps']['picard']['steps_per_s'] \
        == 11000.1
    with open(side) as fh:
        kept = json.load(fh)
    assert kept['roofline.step.ops'][3]['op'] == 'SpMV 3'
    assert any(k.endswith('.what') for k in kept)      # (long strings)
    # (secondary records beyond the limit keep their figures of merit only)
    assert 'config.row_partitioned' in kept and 'padding' not in rp
    # a line that is short already is left alone (but for the rounding)
    small, moved2 = bench.compact_line(dict(value=1.0, config=dict(a='b')),
                                       side=[side])
    assert small == dict(value=1.0, config=dict(a='b')) and not moved2
