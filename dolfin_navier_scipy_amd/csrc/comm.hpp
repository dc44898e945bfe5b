// Communicator of the row-partitioned solve: RCCL over xGMI in production,
// host callbacks (torch.distributed/gloo in the tests) as a stand-in where two
// ranks have to share one GPU.  Three collectives:
//   allreduce  : sum of a few fp64 scalars (the dot products of an Arnoldi
//                step, one call per step)
//   alltoallv  : the HALO exchange -- every rank sends each neighbour exactly
//                the vector entries that neighbour's rows reference (index
//                lists from the column footprint of its row block): grouped
//                ncclSend/ncclRecv over xGMI
//   allgatherv : contiguous blocks of a vector that every rank needs in full
//                (the pressure-sized Schur input; the velocity once per solve)
#pragma once
#include <rccl/rccl.h>

#include <algorithm>
#include <vector>

#include "common.hpp"

struct dns_comm {
    int rank = 0, nranks = 1, device = 0;
    ncclComm_t nccl = nullptr;
    dns_allreduce_cb ar_cb = nullptr;
    dns_allgatherv_cb ag_cb = nullptr;
    dns_alltoallv_cb a2a_cb = nullptr;
    void *ctx = nullptr;
    int64_t n_allreduce = 0, n_allgather = 0, n_alltoall = 0;
    int64_t bytes_alltoall = 0, bytes_allgather = 0;   // sent by this rank

    // sendbuf/recvbuf: device; counts/displs in doubles, one entry per rank
    int alltoallv(const double *sendbuf, const std::vector<int> &scounts,
                  const std::vector<int> &sdispls, double *recvbuf,
                  const std::vector<int> &rcounts,
                  const std::vector<int> &rdispls, hipStream_t s) {
        n_alltoall++;
        for (int q = 0; q < nranks; ++q)
            if (q != rank) bytes_alltoall += 8 * (int64_t)scounts[q];
        if (nccl) {
            ncclResult_t r = ncclGroupStart();
            for (int q = 0; q < nranks && r == ncclSuccess; ++q) {
                if (q == rank) continue;
                if (scounts[q] > 0)
                    r = ncclSend(sendbuf + sdispls[q], (size_t)scounts[q],
                                 ncclDouble, q, nccl, s);
                if (r == ncclSuccess && rcounts[q] > 0)
                    r = ncclRecv(recvbuf + rdispls[q], (size_t)rcounts[q],
                                 ncclDouble, q, nccl, s);
            }
            ncclResult_t e = ncclGroupEnd();
            if (r == ncclSuccess) r = e;
            if (r != ncclSuccess)
                return dns::fail(DNS_ERR_COMM, "halo exchange (ncclSend/Recv): %s",
                                 ncclGetErrorString(r));
            return DNS_OK;
        }
        if (!a2a_cb) return dns::fail(DNS_ERR_COMM, "no alltoallv backend");
        DNS_HIP(hipStreamSynchronize(s));
        if (a2a_cb(ctx, sendbuf, scounts.data(), sdispls.data(), recvbuf,
                   rcounts.data(), rdispls.data(), nranks) != 0)
            return dns::fail(DNS_ERR_COMM, "alltoallv callback failed");
        return DNS_OK;
    }

    ~dns_comm() {
        if (nccl) (void)ncclCommDestroy(nccl);
    }

    int allreduce(double *dev, int count, hipStream_t s) {
        n_allreduce++;
        if (nccl) {
            ncclResult_t r = ncclAllReduce(dev, dev, (size_t)count, ncclDouble,
                                           ncclSum, nccl, s);
            if (r != ncclSuccess)
                return dns::fail(DNS_ERR_COMM, "ncclAllReduce: %s",
                                 ncclGetErrorString(r));
            return DNS_OK;
        }
        if (!ar_cb) return dns::fail(DNS_ERR_COMM, "no allreduce backend");
        DNS_HIP(hipStreamSynchronize(s));
        if (ar_cb(ctx, dev, count) != 0)
            return dns::fail(DNS_ERR_COMM, "allreduce callback failed");
        return DNS_OK;
    }

    int allgatherv(double *dev, const std::vector<int> &starts, hipStream_t s) {
        n_allgather++;
        bytes_allgather += 8 * (int64_t)(starts[rank + 1] - starts[rank]);
        if (nccl) {
            ncclResult_t r = ncclGroupStart();
            for (int q = 0; q < nranks && r == ncclSuccess; ++q) {
                const size_t cnt = (size_t)(starts[q + 1] - starts[q]);
                if (cnt == 0) continue;
                r = ncclBroadcast(dev + starts[q], dev + starts[q], cnt,
                                  ncclDouble, q, nccl, s);
            }
            ncclResult_t e = ncclGroupEnd();
            if (r == ncclSuccess) r = e;
            if (r != ncclSuccess)
                return dns::fail(DNS_ERR_COMM, "allgatherv (ncclBroadcast): %s",
                                 ncclGetErrorString(r));
            return DNS_OK;
        }
        if (!ag_cb) return dns::fail(DNS_ERR_COMM, "no allgatherv backend");
        DNS_HIP(hipStreamSynchronize(s));
        if (ag_cb(ctx, dev, starts.data(), nranks) != 0)
            return dns::fail(DNS_ERR_COMM, "allgatherv callback failed");
        return DNS_OK;
    }
};

namespace dns {

// contiguous block partition of `n` rows over `nranks`: starts[r] .. starts[r+1]
inline std::vector<int> partition_starts(int n, int nranks) {
    std::vector<int> st((size_t)nranks + 1);
    // (even chunks: the two velocity dofs of a node stay on one rank -- the
    // pair format of a rank's row block needs that)
    const int chunk = ((n + nranks - 1) / nranks + 1) & ~1;
    for (int r = 0; r <= nranks; ++r)
        st[r] = (int)std::min<int64_t>((int64_t)n, (int64_t)r * chunk);
    return st;
}

}  // namespace dns
