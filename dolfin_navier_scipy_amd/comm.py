"""Communicators of the row-partitioned solve (`dns_comm_*` of the C-ABI).

 * `Comm.rccl(...)`  -- RCCL over xGMI, one process per GPU (production)
 * `Comm.gloo(...)`  -- host-staged collectives through `torch.distributed`
                        (gloo): lets two ranks share ONE GPU, which RCCL
                        refuses; used by the tests to exercise the real HIP
                        kernels with rank > 0 row blocks
 * `host_allreduce`, `host_allgatherv` -- the host halves of the gloo backend,
   testable without a GPU
"""
import ctypes as ct

import numpy as np

from . import _capi as C

__all__ = ['Comm', 'partition_range', 'host_allreduce', 'host_allgatherv',
           'host_alltoallv', 'halo_lists', 'rccl_unique_id']


def partition_range(n, nranks, rank):
    """the block partition the library uses for `n` rows"""
    a, b = ct.c_int32(0), ct.c_int32(0)
    C.check(C.load_library().dns_partition_range(n, nranks, rank,
                                                 ct.byref(a), ct.byref(b)))
    return a.value, b.value


def host_allreduce(buf, group=None):
    """in-place sum over ranks of a float64 NumPy array (gloo)"""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(buf)
    dist.all_reduce(t, group=group)
    return buf


def host_allgatherv(buf, starts, rank, group=None):
    """in-place: rank r owns `buf[starts[r]:starts[r+1]]`; afterwards every
    rank holds every block (variable block sizes -> padded all_gather)"""
    import torch
    import torch.distributed as dist
    nranks = len(starts) - 1
    width = max(starts[r+1] - starts[r] for r in range(nranks))
    mine = np.zeros(max(width, 1))
    cnt = starts[rank+1] - starts[rank]
    mine[:cnt] = buf[starts[rank]:starts[rank+1]]
    outs = [torch.zeros(max(width, 1), dtype=torch.float64)
            for _ in range(nranks)]
    dist.all_gather(outs, torch.from_numpy(mine), group=group)
    for r in range(nranks):
        buf[starts[r]:starts[r+1]] = outs[r].numpy()[:starts[r+1]-starts[r]]
    return buf


def host_alltoallv(send, scounts, sdispls, rcounts, rdispls, rank,
                   group=None):
    """the halo exchange on the host (gloo has no all-to-all on CPU tensors:
    every rank publishes its packed send buffer and its counts, everybody
    picks the pieces addressed to it); returns the packed receive buffer"""
    import torch
    import torch.distributed as dist
    nranks = len(scounts)
    meta = torch.tensor(list(scounts) + list(sdispls), dtype=torch.int64)
    metas = [torch.zeros_like(meta) for _ in range(nranks)]
    dist.all_gather(metas, meta, group=group)
    width = torch.tensor([len(send)], dtype=torch.int64)
    widths = [torch.zeros_like(width) for _ in range(nranks)]
    dist.all_gather(widths, width, group=group)
    wmax = max(1, max(int(w.item()) for w in widths))
    mine = np.zeros(wmax)
    mine[:len(send)] = send
    outs = [torch.zeros(wmax, dtype=torch.float64) for _ in range(nranks)]
    dist.all_gather(outs, torch.from_numpy(mine), group=group)
    recv = np.zeros(max(1, int(sum(rcounts))))
    for q in range(nranks):
        if q == rank or rcounts[q] == 0:
            continue
        cnt_q = int(metas[q][rank].item())          # what q sends to me
        dsp_q = int(metas[q][nranks + rank].item())
        if cnt_q != rcounts[q]:
            raise RuntimeError('halo plan mismatch between ranks {0} and {1}: '
                               '{2} sent, {3} expected'.format(
                                   q, rank, cnt_q, rcounts[q]))
        recv[rdispls[q]:rdispls[q] + cnt_q] = \
            outs[q].numpy()[dsp_q:dsp_q + cnt_q]
    return recv


def halo_lists(pattern, row0, row1, nranks, rank, col_starts, ncols_part):
    """`dns_halo_lists`: for each other rank the sorted distinct columns
    (below `ncols_part`) the rows `[row0, row1)` of `pattern` reference in
    that rank's column range (host only, no GPU)"""
    view = C.CsrView(pattern)
    cs = np.ascontiguousarray(col_starts, dtype=np.int32)
    counts = np.zeros(nranks, dtype=np.int32)
    total = ct.c_int64(0)
    lib = C.load_library()
    C.check(lib.dns_halo_lists(view.byref(), row0, row1, nranks, rank,
                               cs.ctypes.data_as(C.c_int32_p), ncols_part,
                               counts.ctypes.data_as(C.c_int32_p), None, 0,
                               ct.byref(total)))
    lists = np.zeros(max(1, total.value), dtype=np.int32)
    C.check(lib.dns_halo_lists(view.byref(), row0, row1, nranks, rank,
                               cs.ctypes.data_as(C.c_int32_p), ncols_part,
                               counts.ctypes.data_as(C.c_int32_p),
                               lists.ctypes.data_as(C.c_int32_p),
                               lists.size, ct.byref(total)))
    out, pos = [], 0
    for q in range(nranks):
        out.append(lists[pos:pos + counts[q]].copy())
        pos += counts[q]
    return out


def rccl_unique_id():
    buf = ct.create_string_buffer(C.UNIQUE_ID_BYTES)
    C.check(C.load_library().dns_comm_unique_id(buf))
    return bytes(buf.raw)


class Comm(object):
    def __init__(self, handle, rank, nranks, keep=None):
        self._h, self.rank, self.nranks = handle, rank, nranks
        self._keep = keep           # callback objects must outlive the handle
        self.lib = C.load_library()

    @classmethod
    def rccl(cls, device, nranks, rank, unique_id):
        h = ct.c_void_p()
        C.check(C.load_library().dns_comm_create_rccl(
            device, nranks, rank, unique_id, ct.byref(h)))
        return cls(h, rank, nranks)

    @classmethod
    def rccl_from_torch(cls, device):
        """bootstrap from an initialised `torch.distributed` process group:
        rank 0 creates the RCCL id, everybody receives it"""
        import torch.distributed as dist
        rank, nranks = dist.get_rank(), dist.get_world_size()
        box = [rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls.rccl(device, nranks, rank, box[0])

    @classmethod
    def gloo(cls, device, group=None):
        """host-staged backend on an initialised gloo process group"""
        import torch.distributed as dist
        lib = C.load_library()
        rank = dist.get_rank(group)
        nranks = dist.get_world_size(group)

        def _allreduce(ctx, dev, count):
            try:
                buf = np.empty(count)
                C.check(lib.dns_device_read(device, dev, buf.ctypes.data,
                                            buf.nbytes))
                host_allreduce(buf, group)
                C.check(lib.dns_device_write(device, dev, buf.ctypes.data,
                                             buf.nbytes))
                return 0
            except Exception as exc:        # never unwind through C
                print('allreduce callback failed:', exc)
                return 1

        def _allgatherv(ctx, dev, starts, nr):
            try:
                st = [starts[k] for k in range(nr + 1)]
                buf = np.empty(st[-1])
                C.check(lib.dns_device_read(device, dev, buf.ctypes.data,
                                            buf.nbytes))
                host_allgatherv(buf, st, rank, group)
                C.check(lib.dns_device_write(device, dev, buf.ctypes.data,
                                             buf.nbytes))
                return 0
            except Exception as exc:
                print('allgatherv callback failed:', exc)
                return 1

        def _alltoallv(ctx, send, scnt, sdsp, recv, rcnt, rdsp, nr):
            try:
                sc = [scnt[k] for k in range(nr)]
                sd = [sdsp[k] for k in range(nr)]
                rc = [rcnt[k] for k in range(nr)]
                rd = [rdsp[k] for k in range(nr)]
                nsend = max(sd[k] + sc[k] for k in range(nr))
                sbuf = np.empty(max(nsend, 1))
                if nsend:
                    C.check(lib.dns_device_read(device, send,
                                                sbuf.ctypes.data, 8*nsend))
                rbuf = host_alltoallv(sbuf[:nsend], sc, sd, rc, rd, rank,
                                      group)
                nrecv = max(rd[k] + rc[k] for k in range(nr))
                if nrecv:
                    C.check(lib.dns_device_write(device, recv,
                                                 rbuf.ctypes.data, 8*nrecv))
                return 0
            except Exception as exc:
                print('alltoallv callback failed:', exc)
                return 1

        ar, ag = C.ALLREDUCE_CB(_allreduce), C.ALLGATHERV_CB(_allgatherv)
        a2a = C.ALLTOALLV_CB(_alltoallv)
        h = ct.c_void_p()
        C.check(lib.dns_comm_create_callbacks(device, nranks, rank, ar, ag,
                                              None, ct.byref(h)))
        C.check(lib.dns_comm_set_alltoallv_cb(h, a2a))
        return cls(h, rank, nranks, keep=(ar, ag, a2a))

    def stats(self):
        """calls and bytes SENT by this rank since the communicator exists"""
        buf = (ct.c_int64*5)()
        C.check(self.lib.dns_comm_stats2(self._h, buf))
        return dict(allreduce=buf[0], allgatherv=buf[1], halo_exchange=buf[2],
                    halo_bytes=buf[3], allgather_bytes=buf[4])

    def set_timing(self, on=True):
        """event pairs around every collective issued as a plain launch
        (`dns_comm_set_timing`); switching on resets the sums"""
        C.check(self.lib.dns_comm_set_timing(self._h, int(bool(on))))

    def timing(self):
        """device milliseconds and timed calls per kind of collective since
        `set_timing(True)` (the wait for the slowest peer included)"""
        ms, calls = (ct.c_double*3)(), (ct.c_int64*3)()
        C.check(self.lib.dns_comm_timing(self._h, ms, calls))
        names = ('allreduce', 'allgatherv', 'halo_exchange')
        return {k: dict(device_ms=ms[i], calls=calls[i],
                        us_per_call=(1e3*ms[i]/calls[i] if calls[i] else None))
                for i, k in enumerate(names)}

    SELFTEST_PRIMITIVES = ('allreduce', 'sendrecv_ring', 'allgather_unequal',
                           'allgather_equal')

    def selftest(self, which, graph=False, count=4, reps=20):
        """`dns_comm_selftest`: ONE primitive (name or index into
        `SELFTEST_PRIMITIVES`) once checked and `reps` times timed, as plain
        launches or captured in a hipGraph; collective"""
        if not isinstance(which, int):
            which = self.SELFTEST_PRIMITIVES.index(which)
        ok, us = ct.c_int32(0), ct.c_double(0.0)
        C.check(self.lib.dns_comm_selftest(self._h, which, int(bool(graph)),
                                           count, reps, ct.byref(ok),
                                           ct.byref(us)))
        return dict(ok=bool(ok.value), us_per_call=us.value)

    def set_gather_form(self, form):
        """'auto' (in place for equal blocks, staged otherwise), 'bcast' (a
        group of broadcasts) or 'staged' (always through the staging buffer)"""
        C.check(self.lib.dns_comm_set_gather_form(
            self._h, ('auto', 'bcast', 'staged').index(form)))

    def gather_forms(self):
        """how the all-gathers were issued so far"""
        buf = (ct.c_int64*3)()
        C.check(self.lib.dns_comm_gather_forms(self._h, buf))
        return dict(allgather_inplace=buf[0], allgather_staged=buf[1],
                    broadcast_group=buf[2])

    def close(self):
        if self._h:
            self.lib.dns_comm_destroy(self._h)
            self._h = ct.c_void_p()
