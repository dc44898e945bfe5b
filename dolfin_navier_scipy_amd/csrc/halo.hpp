// Halo exchange by index lists (row-partitioned solve, dist_solve.inc; the
// partitioned right-hand side of the resident time steppers, imex_capi.inc).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <iterator>
#include <vector>

#include "comm.hpp"
#include "common.hpp"
#include "halo_host.hpp"
#include "hostcsr.hpp"
#include "kernels.hpp"

namespace dns {

__global__ void __launch_bounds__(kBlock)
k_halo_pack(int count, const int *__restrict__ idx,
            const double *__restrict__ vec, double *__restrict__ buf) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < count;
         i += gridDim.x * kBlock)
        buf[i] = vec[idx[i]];
}

__global__ void __launch_bounds__(kBlock)
k_halo_unpack(int count, const int *__restrict__ idx,
              const double *__restrict__ buf, double *__restrict__ vec) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < count;
         i += gridDim.x * kBlock)
        vec[idx[i]] = buf[i];
}

}  // namespace dns

// who sends what to whom for one vector exchange
struct dns_halo_plan {
    static int halo_grid(int n) {
        return std::max(1, std::min((n + dns::kBlock - 1) / dns::kBlock, 2048));
    }
    std::vector<int> scounts, sdispls, rcounts, rdispls;
    dns::DevBuf<int> sidx, ridx;
    dns::DevBuf<double> sbuf, rbuf;
    int nsend = 0, nrecv = 0;

    // need_of[r][q]: what rank r needs from rank q (all ranks compute all of
    // it from the replicated host pattern: no set-up communication)
    int build(const std::vector<std::vector<std::vector<int>>> &need_of,
              int rank, int nranks, hipStream_t s) {
        scounts.assign((size_t)nranks, 0);
        rcounts.assign((size_t)nranks, 0);
        sdispls.assign((size_t)nranks, 0);
        rdispls.assign((size_t)nranks, 0);
        std::vector<int> si, ri;
        for (int q = 0; q < nranks; ++q) {
            sdispls[q] = (int)si.size();
            rdispls[q] = (int)ri.size();
            if (q == rank) continue;
            const std::vector<int> &snd = need_of[q][rank];   // q needs mine
            const std::vector<int> &rcv = need_of[rank][q];   // I need q's
            si.insert(si.end(), snd.begin(), snd.end());
            ri.insert(ri.end(), rcv.begin(), rcv.end());
            scounts[q] = (int)snd.size();
            rcounts[q] = (int)rcv.size();
        }
        nsend = (int)si.size();
        nrecv = (int)ri.size();
        // TEST HOOK (tests/test_gpu_zz_dist.py, bench.py's parity leg): a send
        // list whose first entry is wrong -- the peer then reads a neighbour's
        // value where it expects its halo entry.  A partitioned run has to
        // notice (parity against the un-partitioned run), not merely converge.
        if (const char *bad = getenv("DNS_TEST_CORRUPT_HALO"))
            if (bad[0] != '0' && nsend >= 2) {
                si[0] = si[1];
                fprintf(stderr, "[dns] TEST HOOK: halo send list of rank %d "
                        "corrupted on purpose\n", rank);
            }
        DNS_TRY(sidx.alloc((size_t)std::max(1, nsend)));
        DNS_TRY(ridx.alloc((size_t)std::max(1, nrecv)));
        DNS_TRY(sbuf.alloc((size_t)std::max(1, nsend)));
        DNS_TRY(rbuf.alloc((size_t)std::max(1, nrecv)));
        if (nsend) DNS_TRY(sidx.upload(si.data(), si.size(), s));
        if (nrecv) DNS_TRY(ridx.upload(ri.data(), ri.size(), s));
        DNS_HIP(hipStreamSynchronize(s));
        return DNS_OK;
    }

    int exchange(dns_comm *comm, double *vec, hipStream_t s) {
        if (nsend > 0)
            hipLaunchKernelGGL(dns::k_halo_pack, halo_grid(nsend),
                               dns::kBlock, 0, s, nsend, sidx.p, vec, sbuf.p);
        DNS_HIP(hipGetLastError());
        DNS_TRY(comm->alltoallv(sbuf.p, scounts, sdispls, rbuf.p, rcounts,
                                rdispls, s));
        if (nrecv > 0)
            hipLaunchKernelGGL(dns::k_halo_unpack, halo_grid(nrecv),
                               dns::kBlock, 0, s, nrecv, ridx.p, rbuf.p, vec);
        DNS_HIP(hipGetLastError());
        return DNS_OK;
    }
};


