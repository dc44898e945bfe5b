// Host-side driver of the saddle-point solve: owns the HBM-resident system,
// the block preconditioner and the Krylov workspace; enqueues the kernels of
// kernels.hpp / bicgstab_kernels.hpp on one HIP stream.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <deque>
#include <functional>

#include "bicgstab_kernels.hpp"
#include "comm.hpp"
#include "dense_half.hpp"
#include "gmres_kernels.hpp"
#include "halo.hpp"
#include "hostcsr.hpp"
#include "kernels.hpp"
#include "pair.hpp"

namespace dns {

#define DNS_LPR_SWITCH(lpr, CALL)                                            \
    switch (lpr) {                                                           \
        case 2: { constexpr int L = 2; CALL; } break;                        \
        case 4: { constexpr int L = 4; CALL; } break;                        \
        case 8: { constexpr int L = 8; CALL; } break;                        \
        case 16: { constexpr int L = 16; CALL; } break;                      \
        case 32: { constexpr int L = 32; CALL; } break;                      \
        default: { constexpr int L = 64; CALL; } break;                      \
    }

inline int grid_for_rows(int nrows, int lpr) {
    const int rows_per_block = kBlock / lpr;
    long g = ((long)nrows + rows_per_block - 1) / rows_per_block;
    return (int)std::max(1L, std::min(g, 4096L));
}

inline int grid_for_elems(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, 2048));
}

inline int CsrDev::upload(const dns_csr *a, hipStream_t s) {
    vals32.release();     // (a copy of OTHER values: whoever wants one remakes it)
    nrows = a->nrows;
    ncols = a->ncols;
    nnz = a->nnz;
    lpr = pick_lpr(nrows > 0 ? (double)nnz / nrows : 1.0);
    DNS_TRY(rowptr.alloc((size_t)nrows + 1));
    // two spare entries: k_spmv_stream reads (value, index) PAIRS on the
    // even-aligned stream and may touch one entry past the last non-zero
    DNS_TRY(colidx.alloc((size_t)nnz + 2));
    DNS_TRY(vals.alloc((size_t)nnz + 2));
    DNS_HIP(hipMemsetAsync(colidx.p + nnz, 0, 2 * sizeof(int), s));
    DNS_HIP(hipMemsetAsync(vals.p + nnz, 0, 2 * sizeof(double), s));
    {
        // the caller's arrays outlive this call: one wait for the three
        SyncOnExit arrived(s);
        DNS_TRY(rowptr.upload_async(a->rowptr, (size_t)nrows + 1, s));
        DNS_TRY(colidx.upload_async(a->colidx, (size_t)nnz, s));
        DNS_TRY(vals.upload_async(a->vals, (size_t)nnz, s));
    }
    // row blocks of the LDS-streaming kernel: consecutive rows with at most
    // TILE non-zeros and at most kBlock rows; a longer row stands alone
    // (tables for the three tile sizes the tuning variants use)
    const int tiles[3] = {1024, 2048, 4096};
    for (int t = 0; t < 3; ++t) {
        std::vector<int> rb;
        rb.push_back(0);
        int start = 0;
        while (start < nrows) {
            int end = start;
            int64_t acc = 0;
            while (end < nrows && (end - start) < kBlock) {
                const int64_t len = a->rowptr[end + 1] - a->rowptr[end];
                if (acc + len > tiles[t]) break;
                acc += len;
                ++end;
            }
            if (end == start) end = start + 1;   // single long row
            rb.push_back(end);
            start = end;
        }
        nrowblocks_t[t] = (int)rb.size() - 1;
        DNS_TRY(rowblocks_t[t].alloc(rb.size()));
        DNS_TRY(rowblocks_t[t].upload(rb.data(), rb.size(), s));
        if (t == 1 && nnz > 0) {
            // compressed column indices of the 2048 tile (one pass, no sort):
            // window 1 starts at the block's smallest column, window 2 at the
            // smallest column beyond window 1
            const int nb = nrowblocks_t[t];
            std::vector<unsigned short> cc((size_t)nnz + 2, 0);
            std::vector<int> bases((size_t)2 * nb, 0);
            c16_rawblocks = 0;
            for (int b = 0; b < nb; ++b) {
                const int64_t k0 = a->rowptr[rb[b]], k1 = a->rowptr[rb[b + 1]];
                int lo = INT32_MAX, hi = INT32_MAX, mx = -1;
                for (int64_t k = k0; k < k1; ++k) lo = std::min(lo, a->colidx[k]);
                for (int64_t k = k0; k < k1; ++k) {
                    const int c = a->colidx[k];
                    mx = std::max(mx, c);
                    if (c - lo > 32767) hi = std::min(hi, c);
                }
                if (k1 == k0) lo = 0;
                if (hi == INT32_MAX) hi = lo;
                if (mx - hi > 32767 && mx - lo > 32767) {
                    bases[2 * b] = -1;           // raw block
                    bases[2 * b + 1] = -1;
                    c16_rawblocks++;
                    continue;
                }
                bases[2 * b] = lo;
                bases[2 * b + 1] = hi;
                for (int64_t k = k0; k < k1; ++k) {
                    const int c = a->colidx[k];
                    cc[k] = (c - lo <= 32767)
                                ? (unsigned short)(c - lo)
                                : (unsigned short)(0x8000 | (c - hi));
                }
            }
            std::vector<int> mt((size_t)8 * std::max(1, nb), 0);
            for (int b = 0; b < nb; ++b) {
                int *m = &mt[(size_t)8 * b];
                m[0] = rb[b];
                m[1] = rb[b + 1] - rb[b];
                m[2] = (int)a->rowptr[rb[b]];
                m[3] = (int)(a->rowptr[rb[b + 1]] - a->rowptr[rb[b]]);
                m[4] = bases[2 * b];
                m[5] = bases[2 * b + 1];
            }
            DNS_TRY(c16.alloc(cc.size()));
            DNS_TRY(c16base.alloc(bases.size()));
            DNS_TRY(meta16.alloc(mt.size()));
            DNS_TRY(c16.upload(cc.data(), cc.size(), s));
            DNS_TRY(c16base.upload(bases.data(), bases.size(), s));
            DNS_TRY(meta16.upload(mt.data(), mt.size(), s));
        }
    }
    return DNS_OK;
}

template <int TILE, int UNR, int RPS, int GOVR>
inline void launch_stream_g(const CsrDev &A, int t, const double *x, double *y,
                            double alpha, double beta, const double *b,
                            hipStream_t s) {
    const double avg = A.nrows > 0 ? (double)A.nnz / A.nrows : 1.0;
    const int nb = A.nrowblocks_t[t];
    const int grid = std::min(nb, 65535);
    const int *rbp = A.rowblocks_t[t].p;
    if (GOVR > 0)
        hipLaunchKernelGGL((k_spmv_stream<(GOVR > 0 ? GOVR : 4), TILE, UNR, RPS>),
                           grid, kBlock, 0, s, nb, rbp, A.rowptr.p, A.colidx.p,
                           A.vals.p, x, y, alpha, beta, b);
    else if (avg <= 6)
        hipLaunchKernelGGL((k_spmv_stream<1, TILE, UNR, RPS>), grid, kBlock, 0,
                           s, nb, rbp, A.rowptr.p, A.colidx.p, A.vals.p, x, y,
                           alpha, beta, b);
    else if (avg <= 12)
        hipLaunchKernelGGL((k_spmv_stream<2, TILE, UNR, RPS>), grid, kBlock, 0,
                           s, nb, rbp, A.rowptr.p, A.colidx.p, A.vals.p, x, y,
                           alpha, beta, b);
    else if (avg <= 48)
        hipLaunchKernelGGL((k_spmv_stream<4, TILE, UNR, RPS>), grid, kBlock, 0,
                           s, nb, rbp, A.rowptr.p, A.colidx.p, A.vals.p, x, y,
                           alpha, beta, b);
    else
        hipLaunchKernelGGL((k_spmv_stream<16, TILE, UNR, RPS>), grid, kBlock, 0,
                           s, nb, rbp, A.rowptr.p, A.colidx.p, A.vals.p, x, y,
                           alpha, beta, b);
}

// y = alpha*A*x + beta*b on `s` (device pointers)
inline int launch_spmv(const CsrDev &A, const double *x, double *y,
                       double alpha, double beta, const double *b, int variant,
                       hipStream_t s, const int *guard = nullptr) {
    // guard (vector and stream16 kernels): device flag, non-zero = skip the work
    if (A.nrows == 0) return DNS_OK;
    if (variant == 7 && A.c16.p) {
        // diagnostic: the gather reads an 8 KB range of x (cache-resident):
        // separates the address-processing cost of the gather from its
        // memory-side cost
        const int nb = A.nrowblocks_t[1];
        hipLaunchKernelGGL((k_spmv_stream16<4, 3>), std::min(nb, 65535), kBlock,
                           0, s, nb, A.meta16.p, A.rowptr.p, A.colidx.p,
                           A.c16.p, A.c16base.p, A.vals.p, x, y, alpha, beta, b);
    } else if ((variant == 3 || variant == 4) && A.c16.p) {
        // diagnostic variants of the 16-bit kernel (see k_spmv_stream16)
        const int nb = A.nrowblocks_t[1];
        const int grid = std::min(nb, 65535);
        if (variant == 3)
            hipLaunchKernelGGL((k_spmv_stream16<4, 1>), grid, kBlock, 0, s, nb,
                               A.meta16.p, A.rowptr.p, A.colidx.p,
                               A.c16.p, A.c16base.p, A.vals.p, x, y, alpha,
                               beta, b);
        else
            hipLaunchKernelGGL((k_spmv_stream16<4, 2>), grid, kBlock, 0, s, nb,
                               A.meta16.p, A.rowptr.p, A.colidx.p,
                               A.c16.p, A.c16base.p, A.vals.p, x, y, alpha,
                               beta, b);
    } else if (variant == 5 && A.c16.p) {
        // single loads (the kernel before the pair loads; A/B record,
        // scripts/spmv_c16.py)
        const int nb = A.nrowblocks_t[1];
        hipLaunchKernelGGL((k_spmv_stream16<4, 0, double, 0>),
                           std::min(nb, 65535), kBlock, 0, s, nb,
                           A.meta16.p, A.rowptr.p, A.colidx.p, A.c16.p,
                           A.c16base.p, A.vals.p, x, y, alpha, beta, b, guard);
    } else if (variant == DNS_SPMV_STREAM16 && A.c16.p) {
        // the same with 16-bit column offsets: 10 instead of 12 bytes per
        // non-zero cross the HBM; pair loads (16 bytes per lane)
        const double avg = A.nrows > 0 ? (double)A.nnz / A.nrows : 1.0;
        const int nb = A.nrowblocks_t[1];
        const int grid = std::min(nb, 65535);
        const int *rbp = A.meta16.p;
        if (avg <= 6)
            hipLaunchKernelGGL((k_spmv_stream16<1, 0, double, 1>), grid, kBlock,
                               0, s, nb, rbp, A.rowptr.p, A.colidx.p, A.c16.p,
                               A.c16base.p, A.vals.p, x, y, alpha, beta, b,
                               guard);
        else if (avg <= 12)
            hipLaunchKernelGGL((k_spmv_stream16<2, 0, double, 1>), grid, kBlock,
                               0, s, nb, rbp, A.rowptr.p, A.colidx.p, A.c16.p,
                               A.c16base.p, A.vals.p, x, y, alpha, beta, b,
                               guard);
        else if (avg <= 48)
            hipLaunchKernelGGL((k_spmv_stream16<4, 0, double, 1>), grid, kBlock,
                               0, s, nb, rbp, A.rowptr.p, A.colidx.p, A.c16.p,
                               A.c16base.p, A.vals.p, x, y, alpha, beta, b,
                               guard);
        else
            hipLaunchKernelGGL((k_spmv_stream16<16, 0, double, 1>), grid,
                               kBlock, 0, s, nb, rbp, A.rowptr.p, A.colidx.p,
                               A.c16.p, A.c16base.p, A.vals.p, x, y, alpha,
                               beta, b, guard);
    } else if (variant == DNS_SPMV_STREAM || variant == DNS_SPMV_STREAM16) {
        // tile 2048, branch-free full-tile loads (UNR = 0), row pointers
        // staged in LDS
        launch_stream_g<2048, 0, 1, 0>(A, 1, x, y, alpha, beta, b, s);
    } else if (variant >= 16 && variant < 32) {
        // tuning variants (scripts/spmv_tune.py), all on the 2048 tile
        switch (variant - 16) {
            case 0: launch_stream_g<2048, 2, 0, 0>(A, 1, x, y, alpha, beta, b, s); break;
            case 1: launch_stream_g<2048, 2, 1, 0>(A, 1, x, y, alpha, beta, b, s); break;
            case 2: launch_stream_g<2048, 2, 1, 8>(A, 1, x, y, alpha, beta, b, s); break;
            case 3: launch_stream_g<2048, 2, 0, 8>(A, 1, x, y, alpha, beta, b, s); break;
            case 4: launch_stream_g<2048, 1, 1, 0>(A, 1, x, y, alpha, beta, b, s); break;
            case 5: launch_stream_g<2048, 4, 1, 0>(A, 1, x, y, alpha, beta, b, s); break;
            case 6: launch_stream_g<2048, 2, 1, 2>(A, 1, x, y, alpha, beta, b, s); break;
            default: launch_stream_g<4096, 2, 1, 0>(A, 2, x, y, alpha, beta, b, s); break;
        }
    } else {
        const int grid = grid_for_rows(A.nrows, A.lpr);
        DNS_LPR_SWITCH(A.lpr,
                       hipLaunchKernelGGL(k_spmv_vec<L>, grid, kBlock, 0, s,
                                          A.nrows, A.rowptr.p, A.colidx.p,
                                          A.vals.p, x, y, alpha, beta, b,
                                          guard));
    }
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

// the streaming kernel with epilogues (k_spmv_stream16x); `grid_cap` bounds the
// number of workgroups (= partials of the fused dots)
template <typename VT>
inline int launch_stream16x(const CsrDev &A, const VT *vals, const double *x,
                            double *y, const StreamEpi &ep, hipStream_t s,
                            const int *guard, int grid_cap = 65535) {
    if (A.nrows == 0) return DNS_OK;
    if (!A.c16.p)
        return fail(DNS_ERR_NOT_READY, "matrix has no 16-bit column table");
    const double avg = A.nrows > 0 ? (double)A.nnz / A.nrows : 1.0;
    const int nb = A.nrowblocks_t[1];
    const int grid = std::max(1, std::min(nb, grid_cap));
    const int *rbp = A.meta16.p;
#define DNS_STREAM16X(GG)                                                      \
    do {                                                                      \
        if (ep.part)                                                          \
            hipLaunchKernelGGL((k_spmv_stream16x<GG, VT, true>), grid, kBlock, \
                               0, s, nb, rbp, A.rowptr.p, A.colidx.p,         \
                               A.c16.p, A.c16base.p, vals, x, y, ep, guard);  \
        else                                                                  \
            hipLaunchKernelGGL((k_spmv_stream16x<GG, VT, false>), grid,       \
                               kBlock, 0, s, nb, rbp, A.rowptr.p,             \
                               A.colidx.p, A.c16.p, A.c16base.p, vals, x, y,  \
                               ep, guard);                                    \
    } while (0)
    if (avg <= 6)
        DNS_STREAM16X(1);
    else if (avg <= 12)
        DNS_STREAM16X(2);
    else if (avg <= 192)   // (measured on Gc, 115 per row: 4 beats 16)
        DNS_STREAM16X(4);
    else
        DNS_STREAM16X(16);
#undef DNS_STREAM16X
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

inline int stream_grid(const CsrDev &A, int cap) {
    return std::max(1, std::min(A.nrowblocks_t[1], cap));
}

struct CtlHeader {   // leading part of DnsCtl, copied back to the host
    int jdone, predone;
    int done, status, zero, total_it, hist_len, conv;
    double beta, tol, resnorm, bnorm;
};

struct CtlHeaderAcc {   // header + the batch accumulators behind it
    CtlHeader h;
    int acc_solves, acc_fail, acc_iters, acc_maxit;
    double rho, alpha, omega;          // (BiCGStab scalars, layout of DnsCtl)
    double acc_maxrel;
    double stop_frac, acc_maxprev;
    int acc_maxneed, need_it;
    int acc_sumneed, acc_pad;
};

// a captured chunk of work, replayed with hipGraphLaunch
struct GraphEntry {
    std::vector<uint64_t> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // collectives the captured body issued (a replay counts them again)
    int64_t d_allreduce = 0, d_allgather = 0, d_alltoall = 0;
    int64_t d_bytes_a2a = 0, d_bytes_ag = 0;
};

}  // namespace dns

// the opaque handle of the C-ABI
struct dns_saddle {
    int device = 0;
    int nv = 0, np = 0, n = 0;
    size_t ld = 0;                   // leading dimension of basis vectors
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    dns::CsrDev F, J, JT, K;
    dns::DevBuf<double> dinv;        // 1/diag(F)
    // preconditioner
    bool precond_ready = false;
    dns_precond_opts popts;
    double lam_lo = 0, lam_hi = 0, theta = 0, delta = 0;
    std::vector<double> c1, c2;      // Chebyshev recurrence coefficients
    dns::DevBuf<double> sinv;        // dense Schur inverse (np x np) or diag
    bool fhat_explicit = false;
    bool fp32_store = false;         // fp32 copies of Gc values / dense Sinv
    dns::DevBuf<float> sinv32, gc32;
    int sld = 0;                      // row stride of sinv32 (np padded to a
                                      // multiple of four: 16-byte row loads)
    int to_f32(const double *in, dns::DevBuf<float> &out, size_t count);
    int schur_to_f32();
    dns::CsrDev Gc;                  // [G, -G JT], explicit polynomial F^-1
    dns::HostCsr Fh, Jh, JTh;        // host copies for preconditioner set-up
    // convection-dominated F (steady Oseen / Newton systems, snu:458,497): the
    // Chebyshev polynomial is built for the symmetric part of F (shifted to be
    // positive), K itself is untouched -- see setup_precond
    dns::HostCsr Fpc_h;
    bool pc_sym = false;
    double skew_eta = 0.0, pc_shift = 0.0;
    int build_explicit(bool dense_schur);
    dns::DevBuf<double> cheb_r, cheb_d0, cheb_d1;
    // Krylov workspace
    int nred = 64;                   // partial count of the BiCGStab kernels
    int gridS = 64, gridC = 64, gridD = 64;   // GMRES: residual / K apply /
                                              // vector kernels
    bool fuse_dots = true;
    // matrices with at least this many non-zeros go through the LDS-streaming
    // kernels (bandwidth regime); below, the sub-wave kernels (latency regime)
    int64_t stream_nnz = 1300000;
    // workgroups (= partials per scalar) of a streaming launch with reductions
    // (fused dots, residual norms): DNS_STREAM_GRID / option `stream_grid`
    int sgrid = dns::kStreamGrid;
    bool streams(const dns::CsrDev &A) const {
        return A.nnz >= stream_nnz && A.c16.p != nullptr;
    }
    // ... and the operators of the multigrid cycle from this many on.  The
    // threshold above is where `Gc` of the N = 2 system (1.1M non-zeros, one
    // launch per Krylov step, latency bound) still takes the sub-wave kernel;
    // the level operators of a refined mesh are four launches per cycle of
    // 0.3-3M non-zeros each, and as fp32 streams with 16-bit columns they
    // move half the bytes of the sub-wave kernel's fp64 / int32 rows
    // (measured at n = 173k, profiles/r05_mg_stream/)
    int64_t mg_stream_nnz = 600000;
    bool streams_mg(const dns::CsrDev &A) const {
        return A.nnz >= std::min(stream_nnz, mg_stream_nnz) &&
               A.c16.p != nullptr;
    }
    dns::DevBuf<double> V, Z, w, z, u, r, xdev, bdev;   // Z_j = P^-1 V_j
    dns::DevBuf<double> xcat;        // [r_v; z_p] packed for the streaming Gc
    dns::DevBuf<double> partA, partN, partR, partB, partC, partE;
    dns::DevBuf<double> bi_rhat, bi_p, bi_v, bi_s, bi_t, bi_y, histdev;
    dns::DevBuf<dns::DnsCtl> ctl;
    dns::DevBuf<dns::BicgCtl> bctl;
    // blocks of right-hand sides / solutions of dns_saddle_solve_multi
    // (column c at c * ld) and the residual history of every column
    dns::DevBuf<double> mrhs, msol;
    dns::PinnedBuf<double> mstage;    // host side of solve_multi's blocks
    std::vector<std::vector<double>> col_history;
    dns::DevBuf<double> scal;        // small scalar scratch
    dns::CtlHeader *hdr_host = nullptr;   // pinned
    double *scal_host = nullptr;          // pinned
    std::vector<double> history;
    int64_t spmv_count = 0;
    // row-partitioned solve (comm.hpp): this rank's row blocks of the n rows of
    // K, the nv velocity rows (Fh^-1) and the np pressure rows (Sh^-1)
    dns_comm *comm = nullptr;
    bool dist_active = false;         // false during set-up (done redundantly)
    struct dns_dist_data *dd = nullptr;   // row blocks + halo plans (dist_solve.inc)
    bool dist_sliced = false;         // K, Gc, JG hold this rank's rows only
    uint64_t dist_generation = 0;     // bumped whenever the slicing changes
                                      // (attached steppers re-slice their own
                                      // operators then)
    int setup_dist(const dns::HostCsr &Gch, const dns::HostCsr *JGh,
                   bool presliced = false);
    // partitioned set-up (every rank forms the rows of its blocks only)
    int build_explicit_part(bool dense_schur, const dns::HostCsr &DF,
                            const std::vector<double> &dv,
                            const dns::HostCsr &JTx);
    int gather_need_lists(const std::vector<std::vector<int>> &mine,
                          std::vector<std::vector<std::vector<int>>> &all);
    int gather_csr_rows(const dns::HostCsr &loc, const std::vector<int> &starts,
                        int ncols, dns::HostCsr &out);
    int restore_full_device();
    int update_values_dist();
    // rank-local construction (rank_local.inc): the handle was created from
    // this rank's rows only; Fh, Jh, JTh keep the global shape with the own
    // rows populated, no whole matrix ever reaches the host or the HBM
    bool rank_local = false;
    int64_t nnz_f_all = 0, nnz_j_all = 0;     // entries of the whole F and J
    // bytes of the host matrices the last explicit set-up had alive at its
    // end (scaled F, J^T, polynomial rows and their products), beside the
    // copies the handle keeps (dns_saddle_host_bytes)
    int64_t host_setup_bytes = 0;
    int init_rows(int dev, dns_comm *c, int nv_all, int np_all,
                  const dns_csr *f, const dns_csr *jt, const dns_csr *j);
    int fetch_rows(const dns::HostCsr &A, const std::vector<int> &starts,
                   const std::vector<int> &want, dns::HostCsr &out);
    int all_ranks_max(double mine, double *out);
    int allgather_host(std::vector<double> &v, const std::vector<int> &st);
    int estimate_bounds_rows();
    int extend_rows_for_setup(dns::HostCsr &Fx, dns::HostCsr &JTx,
                              std::vector<double> &dv);
    int skew_bound_rows(const dns::HostCsr &Fx, double *eta);
    int enqueue_cycle_dist(const double *b, double *x, int c,
                           const dns_solve_opts *o, int first,
                           bool have_resid = false);
    dns::RowMap dist_rowmap() const;
    dns::HostCsr tmp_Gch, tmp_JGh;    // handed from build_explicit to setup_dist
    // row-partitioned cycle: exchange the start vector's halo (not needed, see
    // enqueue_cycle_dist); halo plan for the preconditioned vectors when the
    // caller's is wider than the rows of K need
    bool dist_x0_exchange = false;
    // one-step cycles without the all-reduce of the residual norm
    // (k_arn_tail_lazy1); the constants 1, 0 the head reads as "norms"
    bool dist_lazy1 = true;
    int cycle_first = 0;              // > 0: length of a solve's first cycle
    dns::DevBuf<double> lazy_one;
    const struct dns_halo_plan *z_plan_override = nullptr;
    double *dist_rnew = nullptr;      // residual carry-over of the stepper
    bool tmp_presliced = false;       // ... already this rank's rows only
    bool part_setup = true;           // DNS_PART_SETUP: partitioned set-up
    std::vector<int> st_v, st_p;      // block partitions of the velocity /
                                      // pressure rows over the ranks
    dns::DevBuf<double> dsum;         // all-reduced scalars
    bool dist() const { return comm != nullptr && dist_active; }
    // RCCL calls are stream-ordered and capturable: a cycle (and a whole time
    // step with its halo exchanges) of a row-partitioned handle is replayed as
    // a hipGraph like on one GPU.  Host-callback communicators (the gloo-staged
    // test stand-in) synchronise the stream and are not.  `dist_graph_ok`
    // goes false when a capture with collectives failed once (DNS_DIST_GRAPH=0
    // starts that way): plain launches from then on.
    bool dist_graph_ok = true;
    // logical OR of a flag over the ranks of the communicator (one all-reduce
    // of one scalar, synchronises the stream; never inside a capture)
    int all_ranks_any(bool mine, bool *any);
    // a LOCAL verdict inside a collective routine of the set-up: every rank
    // enters, the failing rank returns its own status, the others
    // DNS_ERR_COMM -- nobody is left waiting in the exchange behind it
    int agree(int rc_local, const char *where);
    bool graph_capable() const {
        return !dist() || (comm->nccl != nullptr && dist_graph_ok);
    }
    int *step_counter = nullptr;      // device step counter of the attached
                                      // stepper (tables of per-step data);
                                      // bumped by the first head kernel of
                                      // a solve
    // pair format of K (pair.hpp): the K applies of the bandwidth regime run
    // on it when the velocity block has the two-component structure
    dns::PairDev Kp;
    bool pair_knob = true;            // DNS_PAIR (read once, at create)
    int build_pair();
    // six-node resident step (set by dns_imex around a solve): out-of-place
    // tail with the new residual and the convection cells, K z_j kept per
    // column in Wcols
    struct Step6 {
        bool on = false;
        dns::Tail6 t6 = {};
        dns::TailCells cells = {};
        const double *kx = nullptr;   // K x0 of the step's first kernel: the
                                      // residual r = b - kx is formed by the
                                      // first tau kernel (k_tau_first)
    } step6;
    // row-partitioned time stepper (set by dns_imex around a solve): the tail
    // of a ONE-step cycle (k_arn_tail_lazy1) also evaluates the convection
    // cells of the new velocity, in extra workgroups, from `x0copy` (the warm
    // start as the step's front kernel has copied it) and Z_0; `ran` says
    // whether the cycle that was enqueued did
    struct DistTail {
        bool on = false, ran = false;
        const double *x0copy = nullptr;
        dns::TailCells cells = {};
    } dist_tail;
    dns::DevBuf<double> Wcols;
    dns::TailExtrap tail_extrap = {}; // warm start of the next step, written
                                      // by the tail kernels (set by dns_imex)
    int prologue_nparts = 0;          // > 0: partials of ||r||^2, ||b||^2 the
                                      // caller's prologue kernel has written
    bool mg_fused = false;            // the V-cycle runs on the fused operators
    // V(1,1) on TWO fused operators per level (one GPU): down b_c = Rd b with
    // Rd = P^T (I - w S D^-1), up x = U [b; e] with U = [(I + T) w D^-1, T P]
    // -- 2 (L - 1) + 1 launches instead of the 4 (L - 1) + 1 of the fused
    // V(2,2).  A weaker cycle per launch saved: pays where the cycle is
    // launch bound and the solves run a fixed number of columns anyway
    // (oversolve); `smooth_steps = 1` of dns_saddle_set_schur_mg asks for it
    bool mg_fused11 = false;
    int mg_cycle_fused11(const double *in, double *zp, double *xacc,
                         const int *guard);
    bool mg_fused_knob = true;        // DNS_MG_FUSED (read once, at create)
    int mg_part_min = 100000;         // levels with at least this many rows are
                                      // row-partitioned over the ranks
                                      // (DNS_MG_PART_MIN), smaller ones run
                                      // replicated
    int mg_nparts = 0;                // how many levels are partitioned
    int setup_dist_mg();
    int schur_mg_apply_dist(const double *in, double *zp, const int *guard);
    int last_iters = -1;              // iteration count of the previous solve
    int pipeline_c = 0;               // > 0: one cycle of this length, no sync
    // oversolve (DnsCtl::stop_frac): the solves of a pipelined batch run their
    // cycle's columns instead of stopping at the tolerance.  On by default with
    // the multigrid Schur block (DNS_OVERSOLVE=0/1 overrides): there a cycle
    // is 2-3 columns of 13+ launch-bound nodes each, the residual of a solve
    // that stops AT the tolerance keeps the slack column of the old policy in
    // every cycle, and it is the final residuals of the last solves,
    // amplified by the warm start's coefficients, that the next start
    // residual consists of (scripts/recycle_probe.py)
    bool oversolve = false;
    int oversolve_env = -1;           // DNS_OVERSOLVE: -1 unset
    double oversolve_frac = 1e-3;     // raise `done` at this fraction of tol
    // (thresholds from the refined wake: a cycle one column shorter ends
    // where the longer one stood in front of its last column, and what the
    // next warm starts make of the larger residuals adds a factor 1.5-2.
    // The two-column cycle of the bandwidth regime sits at 0.05-0.35 on one
    // GPU and at 0.3-0.8 through the partitioned code path: with the bound
    // at 0.7 that path ran three columns, 1322 instead of 1857 steps/s at
    // n = 693k)
    double oversolve_raise = 0.9;     // batch max of final res / tol: c + 1
    double oversolve_lower = 0.25;    // ... in front of the last column: c - 1
    // shortest cycle the policy tries.  With the multigrid block the first
    // column reduces by ~0.4 and the warm start amplifies by >= 2.5: a
    // one-column cycle lets the start residuals grow step by step (measured:
    // 1 -> 1e6 x tol within a batch at n = 693k), every trial of it is a
    // replayed batch -- so it is not tried (DNS_OVERSOLVE_CMIN overrides)
    int oversolve_cmin = 1;
    int oversolve_cmin_env = -1;
    int set_stop_frac(double f);
    bool capturing = false;           // a run_cached capture is open
    // multigrid Schur block (DNS_SCHUR_MG): level 0 = the pressure space
    struct MgLevel {
        int n = 0;
        // damping of the first / second sweep of a smoothing pair: equal
        // (4 / (3 lambda_max): damped Jacobi) or the reciprocals of the two
        // Chebyshev roots on [lambda_max / alpha, lambda_max] (mg_cheb) -- the
        // same two products, error polynomial (1 - w1 t)(1 - w2 t) uniformly
        // small on the upper part of the spectrum (0.22 for alpha = 4, where
        // two Jacobi sweeps leave 0.44 at lambda_max / 4)
        double omega = 0.6, omega2 = 0.6;
        dns::CsrDev S, P, PT;             // operator, prolongation from l+1, P^T
        // fused V(2,2) cycle (mg_fused): with T = I - w D^-1 S, c(b) = w D^-1 b
        //   Apre = (I + T) w D^-1     x_pre = Apre b       (two sweeps from 0)
        //   Rr   = [P^T, -P^T S]      b_c = Rr [b; x_pre]  (residual+restriction)
        //   Qq   = [T, T P]           x' = Qq [x_pre; e] + c(b)
        //                                            (prolongation + first sweep)
        dns::CsrDev Apre, Rr, Qq;
        // host copies (kept while a communicator is attached: the row blocks
        // of a partitioned cycle are cut from them, dist_solve.inc)
        dns::HostCsr Sh, Apreh, Rrh, Qqh;
        // row-partitioned level: rows [r0, r1) of this level's operators
        // (Rr: rows [c0, c1) of the next level), halo plans of the level's
        // vectors (planF) and of the next level's correction (planC)
        bool part = false;
        int64_t nnz_S = 0, nnz_P = 0;     // sizes of the whole S and P
        std::vector<int> st;              // block partition of the rows
        dns_halo_plan planF, planC;
        bool coarse_replicated = true;    // level l+1 is not partitioned
        dns::DevBuf<double> dinv, x, b, r, x2;
    };
    std::deque<MgLevel> mg;               // (device buffers do not move)
    std::vector<dns::HostCsr> mg_prol_h;  // prolongations as handed over
    dns::DevBuf<double> mg_cinv;          // dense inverse on the coarsest level
    dns::DevBuf<float> mg_cinv32;         // ... its fp32 copy (fp32_store)
    int mg_dense_max = 2000;              // first level <= this: dense inverse
    // ... and up to this size when the inverse is stored in HALF precision
    // (dense_half.hpp: one bandwidth-bound launch instead of the five
    // latency-bound ones of a sparse level; 0 = never); DNS_MG_DENSE_HALF_MAX
    int mg_dense_half_max = 0;
    dns::DevBuf<__half> mg_cinv16;        // the inverse / mg_cscale[0]
    dns::DevBuf<double> mg_cscale;
    int mg_cld16 = 0;                     // its leading dimension (8 | ld)
    // x = (coarsest level)^-1 b in whichever store the set-up chose
    int mg_coarse_apply(const double *b, double *x, const dns::DnsCtl *gctl);
    int mg_nu = 2;
    bool mg_cheb = true;                  // DNS_MG_CHEB (read once, at create)
    double mg_cheb_alpha = 3.0;           // DNS_MG_CHEB_ALPHA (measured: 3
                                          // beats 4 and 8 at refine 2-4)
    bool mg_ready = false, mg_set = false;
    int build_mg_schur(const dns::HostCsr &S0);
    int mg_prepare(int n0);               // level count, fresh levels
    int build_mg_levels(int l0, dns::HostCsr Sl);
    // the hierarchy formed BY ROWS (mg_rows.inc): the levels that run row-
    // partitioned never exist as whole matrices; the first replicated level is
    // all-gathered, the rest is built from it as on one GPU
    int build_mg_schur_rows(const dns::HostCsr &S0loc);
    int mg_rows_parts = 0;                // levels built (and kept) by rows
    bool mg_rows_knob = true;             // DNS_MG_ROWS / option `mg_rows`
    int mg_op(const dns::CsrDev &A, const double *xa, int nsplit,
              const double *xb, const MgLevel *add, const double *b,
              double *out, const int *guard);
    int schur_mg_apply_fused(const double *in, double *zp, double *xacc,
                             const int *guard);
    int mg_cycle_fused(const double *in, double *zp, double *xacc,
                       const int *guard, double scale);
    // cycles per application of the Schur block (DNS_MG_CYCLES): two cycles
    // as two Richardson steps at the Chebyshev roots of [1 - rho, 1]
    // (DNS_MG_RHO: bound of the cycle's contraction number) -- a LINEAR map
    // of its input like one cycle (the fused Gram-Schmidt relies on that)
    // Default (mg_cycles_knob == 0): TWO cycles where the pipelined batches
    // run whole Krylov cycles (`oversolve`: multigrid Schur block, fewer than
    // 1.5e6 unknowns, one GPU) -- the second cycle costs nine small launches,
    // a second Krylov column costs those nine AND the K apply, the head, the
    // J Fh^-1 and the Fh^-1 products; with the stronger block one column per
    // time step holds (the warm start stays below the tolerance, the column
    // keeps it there): refine 2 4835 -> 5650-5750 steps/s, refine 3 1905 ->
    // 2480 (profiles/r05_mg_cycles/); ONE cycle otherwise, and -- chosen by
    // default -- in cycles of more than one column (mg_two_for)
    int mg_cycles = 1;
    int mg_cycles_knob = 0;           // 0 = auto, 1 / 2 = as said
    // (what runs: a row-partitioned solve keeps ONE cycle -- the two-cycle
    // path has no partitioned residual product, and with a hierarchy every
    // rank holds whole the second cycle did not buy the one-column regime on
    // the partitioned path: n = 173k on one rank 3916 -> 3152 steps/s at two
    // columns, n = 693k 1856 -> 2006 at 1.3, profiles/r05_mg_cycles/)
    int mg_cycles_eff() const { return (mg_cycles >= 2 && !dist()) ? 2 : 1; }
    // Chosen by default (knob 0), the second cycle runs in ONE-column Krylov
    // cycles only: a longer cycle is what a transient needs, and there a
    // second column (one cycle each) is the cheaper way to the same
    // reduction -- with two cycles in every column a 200-step window that
    // starts in a two-column phase ran at 3360 steps/s instead of 4810 (refine
    // 2).  enqueue_cycle sets `mg_two_now` for the kernels it enqueues.
    bool mg_two_now = false;
    bool mg_two_for(int c) const {
        return mg_cycles_eff() >= 2 && (mg_cycles_knob == 2 || c <= 1);
    }
    int oversolve_cmin_eff() const {
        if (oversolve_cmin_env > 0) return oversolve_cmin_env;
        if (popts.schur != DNS_SCHUR_MG) return 1;
        return mg_cycles_eff() >= 2 ? 1 : 2;
    }
    double mg_rho = 0.3;
    dns::DevBuf<double> mg_r2, mg_z2;
    int schur_mg_apply(const double *in, double *zp, double *xacc,
                       const int *guard);
    // full block factorisation: J Fh^-1 as one CSR matrix, tau = r_p - JG r_v
    dns::CsrDev JG;
    dns::DevBuf<double> tau;
    bool have_jg = false;
    bool fh_stale = false;            // F.vals changed on the device
    // ... by a row-partitioned assembly: only this rank's rows are current
    // (trap_capi.inc, RowRange); a set-up gathers the others from their
    // owners before it takes the values to the host
    bool fvals_own_rows_only = false;
    bool dinv_stale = false;          // ... and 1/diag(F) was not refreshed
                                      // (explicit Fh^-1: nobody reads it
                                      // between two set-ups)
    // (k_current: the caller's kernel has written the new values into K too)
    int device_values_changed(bool k_current = false,
                              bool own_rows_only = false);
    int ensure_F_device();            // a sliced handle gets its F block back
    int dist_v0() const;              // this rank's velocity rows [v0, v1)
    int dist_v1() const;
    int gs_fallbacks = 0;             // solves in a row whose fused Gram-Schmidt
                                      // fell back (>= 8: not tried any more)
    bool want_history = true;         // copy the residual history back
    size_t hist_cap = 0;
    std::vector<dns::GraphEntry> graphs;
    typedef int (*enqueue_fn)(void *ctx);
    // enqueue one GMRES cycle of `c` iterations (capturable: no sync inside)
    int enqueue_cycle(const double *b, double *x, int c,
                      const dns_solve_opts *o, int first, bool have_resid);
    // run `body` eagerly or as a cached graph identified by `key`
    // launch == false: capture + instantiate only (set-up time), nothing runs
    template <typename Body>
    int run_cached(const std::vector<uint64_t> &key, bool use_graph,
                   Body body, bool launch = true);
    bool have_graph(const std::vector<uint64_t> &key) const {
        for (const auto &g : graphs)
            if (g.key == key) return true;
        return false;
    }
    int64_t graph_captures = 0;       // captures + instantiations so far
    int64_t graph_generation = 0;     // bumped whenever the cache is dropped
    void drop_graphs();

    ~dns_saddle();
    int init(int dev, const dns_csr *f, const dns_csr *j, const dns_csr *jt);
    int init_device(int dev);
    int init_workspace();
    int update_values(const double *fvals);
    int setup_precond(const dns_precond_opts *o);
    int estimate_bounds();
    int build_dense_schur();
    int build_jacobi_schur();
    int invert_dense(double *a, int nn);
    int ensure_workspace(int m);
    int ensure_solver_buffers(const dns_solve_opts *o);   // basis + history
    // z = P^-1 r (device pointers); `xacc` != null: also x += z
    int apply_precond(const double *rvec, double *zout, const int *guard,
                      double *xacc);
    int apply_fhat_part(const double *rvec, const double *zp, double *zv,
                        const int *guard, double *xacc);
    int dot_host(int64_t len, const double *x, const double *y, double *out);
    int read_header();
    int solve_device(const double *b, double *x, const dns_solve_opts *o,
                     dns_solve_stats *st);
    // `prologue` enqueues caller work (e.g. the IMEX right-hand side) in front
    // of the first cycle so that it is captured into the same graph
    int gmres(const double *b, double *x, const dns_solve_opts *o,
              dns_solve_stats *st,
              const std::function<int()> &prologue = nullptr,
              uint64_t prologue_key = 0, bool prologue_has_resid = false);
    int bicgstab(const double *b, double *x, const dns_solve_opts *o,
                 dns_solve_stats *st);
    int true_residual(const double *b, const double *x, double *out);
    const int *done_ptr() const {
        return reinterpret_cast<const int *>(
            reinterpret_cast<const char *>(ctl.p) + offsetof(dns::DnsCtl, done));
    }
    const int *zero_ptr() const {
        return reinterpret_cast<const int *>(
            reinterpret_cast<const char *>(ctl.p) + offsetof(dns::DnsCtl, zero));
    }
};
