"""World-size-2 (gloo, CPU) tests of the multi-rank path's host side:
the block partition the C library uses, the host halves of the staged
collectives, and the row-partitioned GMRES model against the serial one."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME='lo')   # loopback, deterministically
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    from dolfin_navier_scipy_amd import comm as dcomm
    import dist_model
    import krylov_model as km
    import scenarios

    # 1. partition of the C library == partition of the model, covers [0, n)
    for n in (0, 1, 7, 1286, 1493, 10645):
        st = dist_model.starts_for(n, world)
        a, b = dcomm.partition_range(n, world, rank)
        assert (a, b) == (st[rank], st[rank+1]), (n, a, b, st)
        assert st[0] == 0 and st[-1] == n

    # 2. host collectives
    buf = np.arange(5, dtype=np.float64)*(rank + 1)
    dcomm.host_allreduce(buf)
    assert np.allclose(buf, np.arange(5)*sum(range(1, world + 1)))
    n = 11
    st = dist_model.starts_for(n, world)
    vec = np.full(n, -1.0)
    vec[st[rank]:st[rank+1]] = 100*rank + np.arange(st[rank], st[rank+1])
    dcomm.host_allgatherv(vec, st, rank)
    want = np.concatenate([100*r + np.arange(st[r], st[r+1])
                           for r in range(world)])
    assert np.array_equal(vec, want)

    # 3. row-partitioned GMRES == serial GMRES
    prob = scenarios.toy_problem(nx=11, ny=4)
    M, A, J = (prob['smc'][k] for k in 'MAJ')
    F = (M + .5*5e-3*A).tocsr()
    NP, NV = J.shape

    # 3a. halo plan of the row partition (`dns_halo_lists`, host only) + the
    # staged halo exchange: after it, every rank holds exactly the entries of
    # a vector that its row block of F references -- and its SpMV rows agree
    # with the serial product
    stv = dist_model.starts_for(NV, world)
    need = dcomm.halo_lists(F, stv[rank], stv[rank+1], world, rank, stv, NV)
    cols = np.unique(F[stv[rank]:stv[rank+1], :].indices)
    for q in range(world):
        want_q = cols[(cols >= stv[q]) & (cols < stv[q+1])] if q != rank \
            else np.zeros(0, dtype=np.int32)
        assert np.array_equal(need[q], want_q), (rank, q)
    # what the others need from me: they computed it the same way
    send_lists = [dcomm.halo_lists(F, stv[q], stv[q+1], world, q, stv, NV)[rank]
                  if q != rank else np.zeros(0, dtype=np.int32)
                  for q in range(world)]
    xfull = np.sin(0.3*np.arange(NV))
    mine = np.full(NV, np.nan)
    mine[stv[rank]:stv[rank+1]] = xfull[stv[rank]:stv[rank+1]]
    sc = [len(l) for l in send_lists]
    sd = np.concatenate([[0], np.cumsum(sc)[:-1]]).astype(int).tolist()
    rc = [len(l) for l in need]
    rd = np.concatenate([[0], np.cumsum(rc)[:-1]]).astype(int).tolist()
    sbuf = np.concatenate([mine[l] for l in send_lists]) if sum(sc) else \
        np.zeros(0)
    rbuf = dcomm.host_alltoallv(sbuf, sc, sd, rc, rd, rank)
    for q in range(world):
        mine[need[q]] = rbuf[rd[q]:rd[q] + rc[q]]
    assert not np.isnan(mine[cols]).any()
    halo = sum(rc)
    assert 0 < halo < NV - (stv[rank+1] - stv[rank])     # a halo, not all
    yloc = F[stv[rank]:stv[rank+1], :] @ np.nan_to_num(mine)
    assert np.allclose(yloc, (F @ xfull)[stv[rank]:stv[rank+1]], rtol=0,
                       atol=1e-14)
    rng = np.random.default_rng(3)
    b = np.concatenate([M @ rng.standard_normal(NV),
                        1e-3*rng.standard_normal(NP)])
    prec = km.BlockTriPrecond(F, J, degree=3)

    class GlooComm(object):
        def __init__(self):
            self.rank, self.nranks = rank, world

        def allreduce(self, x):
            return dcomm.host_allreduce(np.ascontiguousarray(x))

        def allgatherv(self, x, starts):
            return dcomm.host_allgatherv(x, starts, rank)

    xs, hs, its_s = dist_model.PartitionedSaddle(
        F, J, prec, dist_model.SerialComm()).gmres(b, rtol=1e-10)
    xd, hd, its_d = dist_model.PartitionedSaddle(
        F, J, prec, GlooComm()).gmres(b, rtol=1e-10)
    assert its_d == its_s
    assert np.linalg.norm(xd - xs) <= 1e-10*np.linalg.norm(xs)
    xm, hm, its_m = km.gmres(km.saddle(F, J), b, prec, rtol=1e-10,
                             reorth=False)
    assert abs(its_m - its_d) <= 1
    assert np.linalg.norm(xd - xm) <= 1e-8*np.linalg.norm(xm)
    # every rank ends with the same iterate
    chk = np.array([xd.sum(), float(its_d)])
    tot = dcomm.host_allreduce(chk.copy())
    assert np.allclose(tot, world*chk)
    open(os.path.join(outdir, 'ok{0}'.format(rank)), 'w').write('ok')
    dist.destroy_process_group()


def test_world2_partition_collectives_and_model(tmp_path):
    from spawn_util import spawn_ranks
    spawn_ranks(_worker, 2, str(tmp_path))
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')
