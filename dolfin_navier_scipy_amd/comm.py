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
           'rccl_unique_id']


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

        ar, ag = C.ALLREDUCE_CB(_allreduce), C.ALLGATHERV_CB(_allgatherv)
        h = ct.c_void_p()
        C.check(lib.dns_comm_create_callbacks(device, nranks, rank, ar, ag,
                                              None, ct.byref(h)))
        return cls(h, rank, nranks, keep=(ar, ag))

    def stats(self):
        a, b = ct.c_int64(0), ct.c_int64(0)
        C.check(self.lib.dns_comm_stats(self._h, ct.byref(a), ct.byref(b)))
        return dict(allreduce=a.value, allgatherv=b.value)

    def close(self):
        if self._h:
            self.lib.dns_comm_destroy(self._h)
            self._h = ct.c_void_p()
