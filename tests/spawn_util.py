"""`mp.spawn` of the multi-rank tests with ONE retry for the one benign
failure: the rendezvous port picked by `free_port()` was taken by somebody else
between the probe and the bind.  Anything else -- a worker that raised,
aborted or faulted on the GPU -- propagates on the first attempt."""
import socket

_RENDEZVOUS = ('address already in use', 'eaddrinuse', 'errno 98',
               'failed to bind', 'could not bind')


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def is_rendezvous_error(exc):
    msg = str(exc).lower()
    return any(pat in msg for pat in _RENDEZVOUS)


def spawn_ranks(worker, world, *args):
    """`worker(rank, world, port, *args)` in `world` processes"""
    import torch.multiprocessing as mp
    try:
        mp.spawn(worker, args=(world, free_port()) + tuple(args),
                 nprocs=world, join=True)
    except Exception as exc:
        if not is_rendezvous_error(exc):
            raise
        mp.spawn(worker, args=(world, free_port()) + tuple(args),
                 nprocs=world, join=True)
