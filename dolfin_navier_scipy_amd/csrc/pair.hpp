// Pair format of the saddle-point matrix K = [[F, JT], [J, 0]] for the
// bandwidth regime.
//
// The velocity dofs of the reference's matrices come in pairs (x and y
// component of one P2 node, interleaved: dof = 2 node + component), and with
// the symmetric velocity gradient of the reference's weak form (dts:236-245)
// the velocity block is made of dense 2x2 node blocks.  A CSR kernel streams a
// column index per scalar entry and -- what costs more, DESIGN.md section 5 --
// processes one scattered 8-byte address per scalar entry, although x[2b] and
// x[2b+1] sit next to each other.
//
// Here the two velocity rows of a node are ONE "pair row" and velocity columns
// are taken in pairs; three entry streams, each with 16-bit column offsets
// from a per-row-block base (as in the CSR kernel):
//   A  (F)  : 2x2 block (a00, a01, a10, a11) at velocity pair column b
//                 y[2a]   += a00 x[2b] + a01 x[2b+1]
//                 y[2a+1] += a10 x[2b] + a11 x[2b+1]      34 B per <= 4 non-zeros
//   B  (JT) : (v0, v1) at pressure column p
//                 y[2a] += v0 x_p[p],  y[2a+1] += v1 x_p[p]  18 B per <= 2
//   C  (J)  : (vx, vy) at velocity pair column b, pressure row p
//                 y_p += vx x[2b] + vy x[2b+1]              18 B per <= 2
// and ONE 16-byte gather of x per A / C entry.  Entries that are not full
// carry explicit zeros, so the format holds any K with an even number of
// velocity dofs and an empty pressure-pressure block; for the systems of this
// package it needs ~8.6 bytes per non-zero instead of 10 and a third of the
// gather addresses.  The builder returns false otherwise (odd NV, column
// windows beyond 15 bits, ...) and the CSR kernels stay in charge.
#pragma once
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "hostcsr.hpp"
#include "kernels.hpp"
#include "pair_host.hpp"

namespace dns {

struct PairDev {
    int nv = 0, np = 0, nvp = 0, nblocks = 0, aoff = 0, poff = 0;
    int64_t nentA = 0;
    DevBuf<int> rpA, rpB, rpC, rowblocks, base;
    DevBuf<int> meta;        // per row block: r0, nr, k0A, nA, k0B, nB, b0, b1
    DevBuf<unsigned short> cA, cB, cC;
    DevBuf<double> vA, vB, vC;
    bool ready = false;
    int upload(const HostPair &P, hipStream_t s) {
        nv = P.nv;
        np = P.np;
        nvp = P.nvp;
        aoff = P.aoff;
        poff = P.poff;
        nblocks = (int)P.rowblocks.size() - 1;
        auto up_i = [&](DevBuf<int> &d, const std::vector<int> &h) -> int {
            DNS_TRY(d.alloc(std::max<size_t>(2, h.size())));
            if (!h.empty()) DNS_TRY(d.upload(h.data(), h.size(), s));
            return DNS_OK;
        };
        auto up_s = [&](DevBuf<unsigned short> &d,
                        const std::vector<unsigned short> &h) -> int {
            DNS_TRY(d.alloc(std::max<size_t>(2, h.size())));
            if (!h.empty()) DNS_TRY(d.upload(h.data(), h.size(), s));
            return DNS_OK;
        };
        auto up_d = [&](DevBuf<double> &d, const std::vector<double> &h) -> int {
            DNS_TRY(d.alloc(std::max<size_t>(4, h.size())));
            if (!h.empty()) DNS_TRY(d.upload(h.data(), h.size(), s));
            return DNS_OK;
        };
        DNS_TRY(up_i(rpA, P.rpA));
        DNS_TRY(up_i(rpB, P.rpB));
        DNS_TRY(up_i(rpC, P.rpC));
        DNS_TRY(up_i(rowblocks, P.rowblocks));
        DNS_TRY(up_i(base, P.base));
        {
            // everything a workgroup needs to know about its row block in ONE
            // 32-byte record (two scalar loads instead of a chain of three
            // dependent ones in front of the stream loads)
            std::vector<int> mt((size_t)8 * std::max(1, nblocks), 0);
            for (int b = 0; b < nblocks; ++b) {
                const int r0 = P.rowblocks[b], r1 = P.rowblocks[b + 1];
                int *m = &mt[(size_t)8 * b];
                m[0] = r0;
                m[1] = r1 - r0;
                if (r0 < P.nvp) {
                    m[2] = P.rpA[r0];
                    m[3] = P.rpA[r1] - P.rpA[r0];
                    m[4] = P.rpB[r0];
                    m[5] = P.rpB[r1] - P.rpB[r0];
                } else {
                    m[2] = P.rpC[r0 - P.nvp];
                    m[3] = P.rpC[r1 - P.nvp] - P.rpC[r0 - P.nvp];
                }
                m[6] = P.base[2 * b];
                m[7] = P.base[2 * b + 1];
            }
            DNS_TRY(meta.alloc(mt.size()));
            DNS_TRY(meta.upload(mt.data(), mt.size(), s));
        }
        DNS_TRY(up_s(cA, P.cA));
        DNS_TRY(up_s(cB, P.cB));
        DNS_TRY(up_s(cC, P.cC));
        {
            // structure of arrays: all top rows (a00, a01), then all bottom
            // rows (a10, a11) -- both halves of a block are read with
            // contiguous 16-byte lane loads
            const size_t ne = P.colA.size();
            std::vector<double> soa(P.vA.size());
            for (size_t k = 0; k < ne; ++k) {
                soa[2 * k] = P.vA[4 * k];
                soa[2 * k + 1] = P.vA[4 * k + 1];
                soa[2 * ne + 2 * k] = P.vA[4 * k + 2];
                soa[2 * ne + 2 * k + 1] = P.vA[4 * k + 3];
            }
            nentA = (int64_t)ne;
            DNS_TRY(up_d(vA, soa));
        }
        DNS_TRY(up_d(vB, P.vB));
        DNS_TRY(up_d(vC, P.vC));
        ready = true;
        return DNS_OK;
    }
    void release_all() {
        rpA.release(); rpB.release(); rpC.release();
        rowblocks.release(); base.release(); meta.release();
        cA.release(); cB.release(); cC.release();
        vA.release(); vB.release(); vC.release();
        ready = false;
        nblocks = 0;
    }
    int64_t bytes() const {
        return (int64_t)((rpA.n + rpB.n + rpC.n + rowblocks.n + base.n +
                          meta.n) * 4 +
                         (cA.n + cB.n + cC.n) * 2 + (vA.n + vB.n + vC.n) * 8);
    }
};

struct PairArgs {
    const int *rpA, *rpB, *rpC, *rowblocks, *base;
    const int4 *meta;
    const unsigned short *cA, *cB, *cC;
    const double *vA, *vB, *vC;
    int64_t nentA;
    int nblocks, nvp, nv;
    int aoff, poff;              // row block of a partitioned system: global
                                 // pair row / pressure row of its first rows
};

// y = alpha K x + beta b with the epilogues of k_spmv_stream16x that a K apply
// needs (plain, fused Gram-Schmidt dots, <b, b>); x, y, b 16-byte aligned
// DIAG != 0 (diagnostic, wrong results by design): 1 = no gather of x
// DOTS: instantiation with the fused-dots epilogue (the plain product keeps
// its accumulators out of the register file: 6 instead of 4 waves per SIMD)
template <int G, int DIAG = 0, bool DOTS = true, int ND = kStreamDots>
__global__ void __launch_bounds__(kBlock, (DOTS || DIAG) ? 1 : 7)   // DIAG 2: A/B without the bound
k_spmv_pair16x(PairArgs A, const double *__restrict__ x,
               double *__restrict__ y, StreamEpi ep,
               const int *__restrict__ guard) {
    if (guard && *guard) return;
    constexpr int NIA = kPairTileA / kBlock;    // 4
    constexpr int NIB = kPairTileB / kBlock;    // 1
    __shared__ double prod0[kPairTileA];
    __shared__ double prod1[kPairTileA];
    __shared__ double prb0[kPairTileB];
    __shared__ double prb1[kPairTileB];
    __shared__ double red[4];
    __shared__ int rpsA[kPairRowsP + 1];
    __shared__ int rpsB[kPairRowsV + 1];
    __shared__ double srow[kBlock];
    const dns_double2 *__restrict__ xx =
        reinterpret_cast<const dns_double2 *>(x);
    const double *__restrict__ xp = x + A.nv;
    const int gq = gridDim.x / 8, gr = gridDim.x % 8;
    const int cls = blockIdx.x % 8;
    const int vb = cls * gq + (cls < gr ? cls : gr) + blockIdx.x / 8;
    const bool dots = DOTS && ep.part != nullptr;
    // ND basis-vector accumulators + <y, y> (ND: what this instance is good
    // for; the launcher picks the smallest that holds ep.nvec)
    double acc[DOTS ? ND + 1 : 1];
    double accb = 0.0;
#pragma unroll
    for (int i = 0; i < (DOTS ? ND + 1 : 1); ++i) acc[i] = 0.0;
    for (int blk = vb; blk < A.nblocks; blk += gridDim.x) {
        const int4 m0 = A.meta[2 * blk], m1 = A.meta[2 * blk + 1];
        const int r0 = m0.x, nr = m0.y, r1 = r0 + nr;
        const bool vrows = r0 < A.nvp;
        const int b0 = m1.z, b1 = m1.w;
        __syncthreads();                 // previous tile fully consumed
        if (vrows) {
            const int ka0 = m0.z, kb0 = m1.x;
            const int na = m0.w, nbt = m1.y;
            if ((int)threadIdx.x <= nr) {
                rpsA[threadIdx.x] = A.rpA[r0 + threadIdx.x] - ka0;
                rpsB[threadIdx.x] = A.rpB[r0 + threadIdx.x] - kb0;
            }
            // branch-free: every load of both tiles, then the gathers (ONE
            // 16-byte address per 2x2 block), then LDS
            const dns_double2 *__restrict__ va2 =
                reinterpret_cast<const dns_double2 *>(A.vA);
            const dns_double2 *__restrict__ vb_bot = va2 + A.nentA;
            const dns_double2 *__restrict__ vb2 =
                reinterpret_cast<const dns_double2 *>(A.vB);
            dns_double2 v0[NIA], v1[NIA], w[NIB];
            int ea[NIA], eb[NIB];
            // (a tile without entries loads nothing: block-uniform branches)
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                const size_t kk = (size_t)ka0 +
                                  min((int)threadIdx.x + i * kBlock,
                                      max(na - 1, 0));
                if (na > 0) {
                    v0[i] = va2[kk];
                    v1[i] = vb_bot[kk];
                    ea[i] = A.cA[kk];
                } else {
                    v0[i].x = v0[i].y = v1[i].x = v1[i].y = 0.0;
                    ea[i] = 0;
                }
            }
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                const size_t kk = (size_t)kb0 +
                                  min((int)threadIdx.x + i * kBlock,
                                      max(nbt - 1, 0));
                if (nbt > 0) {
                    w[i] = vb2[kk];
                    eb[i] = A.cB[kk];
                } else {
                    w[i].x = w[i].y = 0.0;
                    eb[i] = 0;
                }
            }
            dns_double2 xg[NIA];
            double pg[NIB];
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                if (na > 0) xg[i] = xx[DIAG == 1 ? (r0 + (i & 3)) : (b0 + ea[i])];
                else xg[i].x = xg[i].y = 0.0;
            }
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                pg[i] = nbt > 0 ? xp[DIAG == 1 ? (i & 1) : (b1 + eb[i])] : 0.0;
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                const int slot = threadIdx.x + i * kBlock;
                prod0[slot] = fma(v0[i].x, xg[i].x, v0[i].y * xg[i].y);
                prod1[slot] = fma(v1[i].x, xg[i].x, v1[i].y * xg[i].y);
            }
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                const int slot = threadIdx.x + i * kBlock;
                prb0[slot] = w[i].x * pg[i];
                prb1[slot] = w[i].y * pg[i];
            }
            __syncthreads();
            const int g = threadIdx.x % G, rsub = threadIdx.x / G;
            for (int r = rsub; r < nr; r += kBlock / G) {
                double s0 = 0.0, s1 = 0.0;
                for (int k = rpsA[r] + g; k < rpsA[r + 1]; k += G) {
                    s0 += prod0[k];
                    s1 += prod1[k];
                }
                for (int k = rpsB[r] + g; k < rpsB[r + 1]; k += G) {
                    s0 += prb0[k];
                    s1 += prb1[k];
                }
                s0 = subwave_sum<G>(s0);
                s1 = subwave_sum<G>(s1);
                if (g == 0) {
                    const size_t row = (size_t)2 * (A.aoff + r0 + r);
                    double o0, o1;
                    if (ep.b) {
                        const dns_double2 bb =
                            *reinterpret_cast<const dns_double2 *>(ep.b + row);
                        o0 = fma(ep.alpha, s0, ep.beta * bb.x);
                        o1 = fma(ep.alpha, s1, ep.beta * bb.y);
                    } else {
                        o0 = ep.alpha * s0;
                        o1 = ep.alpha * s1;
                    }
                    dns_double2 oo;
                    oo.x = o0;
                    oo.y = o1;
                    *reinterpret_cast<dns_double2 *>(y + row) = oo;
                    if (dots) {
                        srow[2 * r] = o0;
                        srow[2 * r + 1] = o1;
                    }
                }
            }
        } else {
            const int p0 = r0 - A.nvp;
            const int kc0 = m0.z;
            const int nc = m0.w;
            for (int t = threadIdx.x; t <= nr; t += kBlock)
                rpsA[t] = A.rpC[p0 + t] - kc0;
            const dns_double2 *__restrict__ vc2 =
                reinterpret_cast<const dns_double2 *>(A.vC);
            dns_double2 w[NIA];
            int ec[NIA];
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                const size_t kk = (size_t)kc0 +
                                  min((int)threadIdx.x + i * kBlock,
                                      max(nc - 1, 0));
                if (nc > 0) {
                    w[i] = vc2[kk];
                    ec[i] = A.cC[kk];
                } else {
                    w[i].x = w[i].y = 0.0;
                    ec[i] = 0;
                }
            }
            dns_double2 xg[NIA];
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                if (nc > 0) xg[i] = xx[DIAG == 1 ? (i & 3) : (b0 + ec[i])];
                else xg[i].x = xg[i].y = 0.0;
            }
#pragma unroll
            for (int i = 0; i < NIA; ++i)
                prod0[threadIdx.x + i * kBlock] =
                    fma(w[i].x, xg[i].x, w[i].y * xg[i].y);
            __syncthreads();
            const int g = threadIdx.x % G, rsub = threadIdx.x / G;
            for (int r = rsub; r < nr; r += kBlock / G) {
                double s0 = 0.0;
                for (int k = rpsA[r] + g; k < rpsA[r + 1]; k += G)
                    s0 += prod0[k];
                s0 = subwave_sum<G>(s0);
                if (g == 0) {
                    const size_t row = (size_t)A.nv + A.poff + p0 + r;
                    const double o0 =
                        ep.b ? fma(ep.alpha, s0, ep.beta * ep.b[row])
                             : ep.alpha * s0;
                    y[row] = o0;
                    if (dots) srow[r] = o0;
                }
            }
        }
        if (DOTS && dots) {
            __syncthreads();
            const int nout = vrows ? 2 * nr : nr;
            if ((int)threadIdx.x < nout) {
                const double out = srow[threadIdx.x];
                const size_t row =
                    vrows ? (size_t)2 * (A.aoff + r0) + threadIdx.x
                          : (size_t)A.nv + A.poff + (r0 - A.nvp) + threadIdx.x;
#pragma unroll
                for (int i = 0; i < ND; ++i)
                    if (i < ep.nvec)
                        acc[DOTS ? i : 0] = fma(ep.V[(size_t)i * ep.ld + row],
                                                out, acc[DOTS ? i : 0]);
                acc[DOTS ? ND : 0] = fma(out, out, acc[DOTS ? ND : 0]);
                if (ep.part_bb) {
                    const double bv = ep.b[row];
                    accb = fma(bv, bv, accb);
                }
            }
        }
    }
    if (DOTS && dots) {
        for (int i = 0; i < ep.nvec; ++i) {
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < ND; ++q)
                if (q == i) a = acc[DOTS ? q : 0];
            a = block_sum(a, red);
            if (threadIdx.x == 0)
                ep.part[(size_t)i * ep.nparts + blockIdx.x] = a;
        }
        if (ep.with_ww) {
            const double a = block_sum(acc[DOTS ? ND : 0], red);
            if (threadIdx.x == 0)
                ep.part[(size_t)ep.nvec * ep.nparts + blockIdx.x] = a;
        }
        if (ep.part_bb) {
            const double a = block_sum(accb, red);
            if (threadIdx.x == 0) ep.part_bb[blockIdx.x] = a;
        }
    }
}

inline int pair_grid(const PairDev &A, int cap) {
    return std::max(1, std::min(A.nblocks, cap));
}

inline int launch_pair16x(const PairDev &A, const double *x, double *y,
                          const StreamEpi &ep, hipStream_t s, const int *guard,
                          int grid_cap = 65535, int diag = 0) {
    if (!A.ready) return fail(DNS_ERR_NOT_READY, "no pair format");
    if (A.nblocks == 0) return DNS_OK;
    PairArgs a;
    a.rpA = A.rpA.p; a.rpB = A.rpB.p; a.rpC = A.rpC.p;
    a.rowblocks = A.rowblocks.p; a.base = A.base.p;
    a.meta = reinterpret_cast<const int4 *>(A.meta.p);
    a.cA = A.cA.p; a.cB = A.cB.p; a.cC = A.cC.p;
    a.vA = A.vA.p; a.vB = A.vB.p; a.vC = A.vC.p;
    a.nblocks = A.nblocks; a.nvp = A.nvp; a.nv = A.nv;
    a.nentA = A.nentA;
    a.aoff = A.aoff; a.poff = A.poff;
    if (diag == 2)
        hipLaunchKernelGGL((k_spmv_pair16x<4, 2, false>), pair_grid(A, grid_cap),
                           kBlock, 0, s, a, x, y, ep, guard);
    else if (diag)
        hipLaunchKernelGGL((k_spmv_pair16x<4, 1, false>), pair_grid(A, grid_cap),
                           kBlock, 0, s, a, x, y, ep, guard);
    else if (ep.part && ep.nvec <= 3)
        hipLaunchKernelGGL((k_spmv_pair16x<4, 0, true, 3>),
                           pair_grid(A, grid_cap), kBlock, 0, s, a, x, y, ep,
                           guard);
    else if (ep.part)
        hipLaunchKernelGGL((k_spmv_pair16x<4, 0, true, kStreamDots>),
                           pair_grid(A, grid_cap), kBlock, 0, s, a, x, y, ep,
                           guard);
    else
        hipLaunchKernelGGL((k_spmv_pair16x<4, 0, false>),
                           pair_grid(A, grid_cap), kBlock, 0, s, a, x, y, ep,
                           guard);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

}  // namespace dns
