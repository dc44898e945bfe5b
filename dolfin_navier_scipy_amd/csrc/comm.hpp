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
#include "halo_host.hpp"

// block starts of an all-gather, by value in the kernel arguments
constexpr int kDnsMaxGatherRanks = 64;
struct dns_gather_starts {
    int st[kDnsMaxGatherRanks + 1];
};

namespace dns {
// behind a padded ncclAllGather: block q of `width` doubles in `stage` goes to
// vec[st[q] .. st[q+1]); the own block is already where it belongs
__global__ void __launch_bounds__(kBlock)
k_gather_unpack(dns_gather_starts gs, int nranks, int rank, int width,
                const double *__restrict__ stage, double *__restrict__ vec) {
    const long long total = (long long)width * nranks;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (long long)gridDim.x * kBlock) {
        const int q = (int)(e / width), i = (int)(e - (long long)q * width);
        if (q != rank && i < gs.st[q + 1] - gs.st[q])
            vec[gs.st[q] + i] = stage[e];
    }
}
}  // namespace dns

struct dns_comm {
    int rank = 0, nranks = 1, device = 0;
    ncclComm_t nccl = nullptr;
    dns_allreduce_cb ar_cb = nullptr;
    dns_allgatherv_cb ag_cb = nullptr;
    dns_alltoallv_cb a2a_cb = nullptr;
    void *ctx = nullptr;
    int64_t n_allreduce = 0, n_allgather = 0, n_alltoall = 0;
    int64_t bytes_alltoall = 0, bytes_allgather = 0;   // sent by this rank

    // Device time per collective (dns_comm_set_timing): an event pair on the
    // launch stream around every call that is issued as a plain launch (a
    // call inside a stream capture is left alone: captured events carry no
    // time).  What lies between the two events is the collective as the
    // stream sees it -- the wait for the slowest peer included.
    enum { kAllreduce = 0, kAllgather = 1, kHalo = 2, kKinds = 3 };
    struct TimedCall {
        hipEvent_t a, b;
        int kind;
    };
    bool timing = false;
    std::vector<TimedCall> timed;      // pool, grows up to kMaxTimed pairs
    size_t timed_used = 0;
    double timed_ms[kKinds] = {0.0, 0.0, 0.0};
    int64_t timed_calls[kKinds] = {0, 0, 0};
    static constexpr size_t kMaxTimed = 4096;

    // sums the recorded pairs into timed_ms / timed_calls (waits for them)
    int timing_collect() {
        for (size_t i = 0; i < timed_used; ++i) {
            float ms = 0.f;
            DNS_HIP(hipEventSynchronize(timed[i].b));
            DNS_HIP(hipEventElapsedTime(&ms, timed[i].a, timed[i].b));
            timed_ms[timed[i].kind] += (double)ms;
            timed_calls[timed[i].kind]++;
        }
        timed_used = 0;
        return DNS_OK;
    }
    // index of the pair whose first event has been recorded, -1: not timed
    int timing_begin(int kind, hipStream_t s) {
        if (!timing) return -1;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess ||
            cs != hipStreamCaptureStatusNone)
            return -1;
        if (timed_used == timed.size()) {
            if (timed.size() >= kMaxTimed) {
                if (timing_collect() != DNS_OK) return -1;
            } else {
                TimedCall t;
                t.kind = kind;
                if (hipEventCreate(&t.a) != hipSuccess) return -1;
                if (hipEventCreate(&t.b) != hipSuccess) {
                    (void)hipEventDestroy(t.a);
                    return -1;
                }
                timed.push_back(t);
            }
        }
        TimedCall &t = timed[timed_used];
        t.kind = kind;
        if (hipEventRecord(t.a, s) != hipSuccess) return -1;
        return (int)timed_used++;
    }
    void timing_end(int idx, hipStream_t s) {
        if (idx >= 0) (void)hipEventRecord(timed[(size_t)idx].b, s);
    }
    // (a pair whose call failed still gets its second event: collect() must
    // not wait for an event that was never recorded)
    struct Timed {
        dns_comm *c;
        int idx;
        hipStream_t s;
        Timed(dns_comm *c_, int kind, hipStream_t s_)
            : c(c_), idx(c_->timing_begin(kind, s_)), s(s_) {}
        ~Timed() { c->timing_end(idx, s); }
    };

    // sendbuf/recvbuf: device; counts/displs in doubles, one entry per rank
    int alltoallv(const double *sendbuf, const std::vector<int> &scounts,
                  const std::vector<int> &sdispls, double *recvbuf,
                  const std::vector<int> &rcounts,
                  const std::vector<int> &rdispls, hipStream_t s) {
        n_alltoall++;
        Timed timed_call(this, kHalo, s);
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
        for (TimedCall &t : timed) {
            (void)hipEventDestroy(t.a);
            (void)hipEventDestroy(t.b);
        }
        if (nccl) (void)ncclCommDestroy(nccl);
    }

    int allreduce(double *dev, int count, hipStream_t s) {
        n_allreduce++;
        Timed timed_call(this, kAllreduce, s);
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

    // ---- all-gather of contiguous, possibly unequal blocks ---------------
    // ONE ncclAllGather on blocks padded to the widest one.  Equal blocks are
    // gathered in place (send pointer = receive pointer + rank * width, the
    // in-place form of the collective); unequal ones -- the even-chunk
    // partition of an NV that the ranks do not divide -- go through a staging
    // buffer [own block | nranks blocks]: one device copy in front, one
    // kernel behind.  (Until round 5 this was a group of nranks
    // ncclBroadcasts: nranks rings instead of one.)  The staging buffer only
    // grows outside a stream capture; a captured call that finds it too
    // small takes the broadcast form, which needs none.
    static constexpr int kMaxGatherRanks = kDnsMaxGatherRanks;
    dns::DevBuf<double> ag_stage;
    // DNS_COMM_ALLGATHER / dns_comm_set_gather_form: 0 = as described, 1 =
    // "bcast" the old form, 2 = "staged" the staging buffer for equal blocks
    // too (lets ONE rank exercise the pack / unpack path)
    int ag_form = 0;
    int64_t n_ag_inplace = 0, n_ag_staged = 0, n_ag_bcast = 0;

    static bool capturing(hipStream_t s) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        return hipStreamIsCapturing(s, &cs) != hipSuccess ||
               cs != hipStreamCaptureStatusNone;
    }
    // room for an all-gather of blocks up to `width` doubles (set-up time)
    int reserve_gather(size_t width, hipStream_t s) {
        const size_t need = width * (size_t)(nranks + 1);
        if (!nccl || ag_stage.n >= need) return DNS_OK;
        if (capturing(s))
            return dns::fail(DNS_ERR_COMM, "reserve_gather inside a capture");
        DNS_HIP(hipStreamSynchronize(s));
        return ag_stage.alloc(need);
    }

    int allgatherv_bcast(double *dev, const std::vector<int> &starts,
                         hipStream_t s) {
        n_ag_bcast++;
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

    int allgatherv(double *dev, const std::vector<int> &starts, hipStream_t s) {
        n_allgather++;
        bytes_allgather += 8 * (int64_t)(starts[rank + 1] - starts[rank]);
        Timed timed_call(this, kAllgather, s);
        if (nccl) {
            int width = 0;
            bool equal = true;
            for (int q = 0; q < nranks; ++q) {
                const int cnt = starts[q + 1] - starts[q];
                width = std::max(width, cnt);
                if (cnt != starts[1] - starts[0]) equal = false;
            }
            if (width == 0) return DNS_OK;
            if (ag_form == 1 || nranks > kMaxGatherRanks)
                return allgatherv_bcast(dev, starts, s);
            if (equal && ag_form != 2) {
                n_ag_inplace++;
                ncclResult_t r = ncclAllGather(dev + starts[rank],
                                               dev + starts[0], (size_t)width,
                                               ncclDouble, nccl, s);
                if (r != ncclSuccess)
                    return dns::fail(DNS_ERR_COMM, "ncclAllGather: %s",
                                     ncclGetErrorString(r));
                return DNS_OK;
            }
            const size_t need = (size_t)width * (size_t)(nranks + 1);
            if (ag_stage.n < need) {
                if (capturing(s)) return allgatherv_bcast(dev, starts, s);
                DNS_TRY(reserve_gather((size_t)width, s));
            }
            n_ag_staged++;
            double *snd = ag_stage.p, *rcv = ag_stage.p + width;
            const int mine = starts[rank + 1] - starts[rank];
            if (mine > 0)
                DNS_HIP(hipMemcpyAsync(snd, dev + starts[rank],
                                       (size_t)mine * sizeof(double),
                                       hipMemcpyDeviceToDevice, s));
            ncclResult_t r = ncclAllGather(snd, rcv, (size_t)width, ncclDouble,
                                           nccl, s);
            if (r != ncclSuccess)
                return dns::fail(DNS_ERR_COMM, "ncclAllGather: %s",
                                 ncclGetErrorString(r));
            dns_gather_starts gs;
            for (int q = 0; q <= nranks; ++q) gs.st[q] = starts[q];
            const long long total = (long long)width * nranks;
            const int grid = (int)std::max<long long>(
                1, std::min<long long>((total + dns::kBlock - 1) / dns::kBlock,
                                       2048));
            hipLaunchKernelGGL(dns::k_gather_unpack, grid, dns::kBlock, 0, s, gs,
                               nranks, rank, width, rcv, dev);
            DNS_HIP(hipGetLastError());
            return DNS_OK;
        }
        if (!ag_cb) return dns::fail(DNS_ERR_COMM, "no allgatherv backend");
        DNS_HIP(hipStreamSynchronize(s));
        if (ag_cb(ctx, dev, starts.data(), nranks) != 0)
            return dns::fail(DNS_ERR_COMM, "allgatherv callback failed");
        return DNS_OK;
    }
};

