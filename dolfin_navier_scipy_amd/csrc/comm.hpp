// Communicator of the row-partitioned solve: RCCL over xGMI in production,
// host callbacks (torch.distributed/gloo in the tests) as a stand-in where two
// ranks have to share one GPU.  Two collectives are all the Krylov solve needs:
//   allreduce  : sum of a few fp64 scalars (dot products, norms)
//   allgatherv : every rank contributes its contiguous row block of a vector
//                that all ranks keep in full ("halo = everything" form of the
//                halo gather; exact neighbour lists are a later refinement)
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
    void *ctx = nullptr;
    int64_t n_allreduce = 0, n_allgather = 0;

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
    const int chunk = (n + nranks - 1) / nranks;
    for (int r = 0; r <= nranks; ++r)
        st[r] = (int)std::min<int64_t>((int64_t)n, (int64_t)r * chunk);
    return st;
}

}  // namespace dns
