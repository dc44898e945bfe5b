// Device kernels of the gfx950 saddle-point path (fp64 values, int32 indices).
//
// Everything here is HBM/L2-bandwidth or launch-latency bound sparse work:
// no MFMA.  Wavefronts are 64 lanes; reductions use cross-lane shuffles, block
// level results go through LDS, grid level dot products are written as one
// partial per workgroup and summed by the CONSUMING kernel (deterministic, no
// atomics, no extra launch) -- see `reduce_partials`.
#pragma once
#include "common.hpp"

namespace dns {

// ---------------------------------------------------------------------------
// Krylov control block, lives in HBM; written only by workgroup 0 of the
// kernels that close an iteration, read (never written) by everyone else.
// ---------------------------------------------------------------------------
// internal status (never leaves the library): the fused Gram-Schmidt could not
// trust its norm; the host repeats from the current iterate with the explicit
// kernel
constexpr int kGsFallback = 100;

struct DnsCtl {
    int jdone;       // completed Arnoldi columns of the current cycle
    int predone;     // k_tau_guard: the column about to be closed converges
    int done;        // != 0: all later kernels of this solve return at once
    int status;      // DNS_OK / DNS_BREAKDOWN
    int zero;        // constant 0 (vector selector / never-raised guard)
    int total_it;
    int hist_len;    // entries appended to the solve's residual history
    int conv;        // 1 once ||r|| <= tol was observed
    double beta, tol, resnorm, bnorm;
    // accumulated over the solves of a pipelined batch of time steps (never
    // reset by a solve; the host zeroes them at the start of a batch)
    int acc_solves, acc_fail, acc_iters, acc_maxit;
    // BiCGStab scalars
    double rho, alpha, omega;
    double acc_maxrel;   // batch maximum of (final residual / tolerance)
    // "Oversolve" (pipelined batches, multigrid Schur block): the nodes of a
    // replayed cycle are paid for whether they work or not, so a solve does
    // not stop at the tolerance but runs the cycle's columns -- `done` is
    // raised at stop_frac x tol (0 = 1: stop at the tolerance); `conv` still
    // means "met the tolerance".  What the host's prediction of the cycle
    // length then goes by: the largest number of columns a solve NEEDED, and
    // the largest residual / tolerance in front of the last column.
    double stop_frac, acc_maxprev;
    int acc_maxneed, need_it;
    int acc_sumneed, acc_pad;    // sum of the columns the solves needed
    double hcol[kMaxRestart + 1];
    double cs[kMaxRestart], sn[kMaxRestart], g[kMaxRestart + 1];
    double y[kMaxRestart];
    double hist[kMaxRestart + 1];                       // per-cycle residuals
    double R[(kMaxRestart + 1) * kMaxRestart];          // column major
};

// A vector that may be one row of a device-resident table: row = *ctr (a step
// counter kept on the device, so that replayed graphs walk through the table
// without the host), clamped to the table; ctr == nullptr: `base` itself.
struct TabRef {
    const double *base;
    const int *ctr;
    int stride, rows;
};
__device__ __forceinline__ const double *tab_row(const TabRef &t) {
    if (!t.ctr) return t.base;
    int r = *t.ctr;
    r = r < 0 ? 0 : (r >= t.rows ? t.rows - 1 : r);
    return t.base + (size_t)r * t.stride;
}

// Warm start of the NEXT time step, written by the kernel that finishes the
// solution of this one (resident time steppers): out = e0 x_new + e1 h1 + e2 h2
// + e3 h3 + e4 h4 -- the polynomial extrapolation of the last solutions.  The
// next step's front kernel then gathers ONE vector along the rows of K instead
// of five (10.6 -> 7.4 us at n = 1e4).  out == nullptr: off.
struct TailExtrap {
    const double *h1, *h2, *h3, *h4;
    double e0, e1, e2, e3, e4;
    double *out;
};

__device__ __forceinline__ void tail_extrapolate(const TailExtrap &t, int e,
                                                 double xn) {
    double v = t.e0 * xn;
    if (t.e1 != 0.0) v = fma(t.e1, t.h1[e], v);
    if (t.e2 != 0.0) v = fma(t.e2, t.h2[e], v);
    if (t.e3 != 0.0) v = fma(t.e3, t.h3[e], v);
    if (t.e4 != 0.0) v = fma(t.e4, t.h4[e], v);
    t.out[e] = v;
}

// arguments of the six-node resident step (step_kernels.hpp)
struct TailCells {
    int ncells, nblocks;                // nblocks == 0: no cell part
    const int *cellmap;
    const double *glam, *area;
    TabRef dbctab;
    double *cellvals;
    // row-partitioned stepper: only the cells sel[0 .. nsel) (those that touch
    // the rank's rows); nullptr: all of them
    const int *sel;
    int nsel;
};

struct Tail6 {
    const double *x0;                   // warm start (not written here)
    double *xout;                       // new solution
    const double *r0;                   // residual the cycle started from
    const double *W;                    // K z_j, column j at W + j ld
    double *rnew;                       // velocity rows of the new residual
    int nv;
};

// Rows a kernel launch works on: local row i of the (possibly sliced) CSR
// arrays is global row `row0 + i` for i < len1, `row2 + i - len1` beyond -- one
// GPU: {0, n, 0, 0}; row-partitioned: this rank's velocity rows, then its
// pressure rows (K = [[F, JT], [J, 0]] sliced by rows).
struct RowMap {
    int row0, len1, row2, len2;
};
__device__ __forceinline__ int map_row(const RowMap &m, int i) {
    return i < m.len1 ? m.row0 + i : m.row2 + (i - m.len1);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int W>
__device__ __forceinline__ double subwave_sum(double v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum, result valid in every thread; `red` holds >= 4 doubles
__device__ __forceinline__ double block_sum(double v, double *red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                 // protect `red` from the previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
    return s;
}

// out[s] = sum_p part[s*pstride + p], p < nparts, for s < nscal; every thread
// of the block must call; `out` is LDS; ends with a barrier.
__device__ __forceinline__ void reduce_partials(const double *__restrict__ part,
                                                int nparts, int pstride,
                                                int nscal, double *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    if (2 * nscal <= nw) {
        // few scalars (the norms at the head of a cycle, one-column Arnoldi
        // steps): several waves share a scalar, so that its ~1e3 partials are
        // fetched in ONE batch of independent loads instead of three
        __shared__ double rp_tmp[16];
        const int wps = nw / nscal;              // waves per scalar
        const int s = wave / wps, q = wave % wps;
        if (s < nscal) {
            const double *ps = part + (size_t)s * pstride;
            const int step = 64 * wps;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            double a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
            for (int p = lane + 64 * q; p < nparts; p += 8 * step) {
                a0 += ps[p];
                a1 += (p + step < nparts) ? ps[p + step] : 0.0;
                a2 += (p + 2 * step < nparts) ? ps[p + 2 * step] : 0.0;
                a3 += (p + 3 * step < nparts) ? ps[p + 3 * step] : 0.0;
                a4 += (p + 4 * step < nparts) ? ps[p + 4 * step] : 0.0;
                a5 += (p + 5 * step < nparts) ? ps[p + 5 * step] : 0.0;
                a6 += (p + 6 * step < nparts) ? ps[p + 6 * step] : 0.0;
                a7 += (p + 7 * step < nparts) ? ps[p + 7 * step] : 0.0;
            }
            double a = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
            a = wave_sum(a);
            if (lane == 0) rp_tmp[wave] = a;
        }
        __syncthreads();
        if ((int)threadIdx.x < nscal) {
            double a = 0.0;
            for (int k = 0; k < wps; ++k) a += rp_tmp[threadIdx.x * wps + k];
            out[threadIdx.x] = a;
        }
        __syncthreads();
        return;
    }
    for (int s = wave; s < nscal; s += nw) {
        // 8 independent loads in flight per lane: the partials sit in L2 /
        // Infinity Cache and a dependent chain of loads would cost a round
        // trip each (fixed summation order -> deterministic)
        const double *ps = part + (size_t)s * pstride;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        double a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
        int p = lane;
        for (; p + 448 < nparts; p += 512) {
            a0 += ps[p];
            a1 += ps[p + 64];
            a2 += ps[p + 128];
            a3 += ps[p + 192];
            a4 += ps[p + 256];
            a5 += ps[p + 320];
            a6 += ps[p + 384];
            a7 += ps[p + 448];
        }
        const double t0 = (p < nparts) ? ps[p] : 0.0;
        const double t1 = (p + 64 < nparts) ? ps[p + 64] : 0.0;
        const double t2 = (p + 128 < nparts) ? ps[p + 128] : 0.0;
        const double t3 = (p + 192 < nparts) ? ps[p + 192] : 0.0;
        const double t4 = (p + 256 < nparts) ? ps[p + 256] : 0.0;
        const double t5 = (p + 320 < nparts) ? ps[p + 320] : 0.0;
        const double t6 = (p + 384 < nparts) ? ps[p + 384] : 0.0;
        double a = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
        a += ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + t6);
        a = wave_sum(a);
        if (lane == 0) out[s] = a;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// CSR SpMV, "vector" form: LPR lanes cooperate on one row, consecutive
// sub-waves take consecutive rows, so a wavefront reads a contiguous run of
// vals/colidx; shuffle reduction inside the sub-wave.
// ---------------------------------------------------------------------------
// the same with the row's pointers already in registers
template <int LPR>
__device__ __forceinline__ double csr_row_dot_k(const int *__restrict__ colidx,
                                                const double *__restrict__ vals,
                                                const double *__restrict__ x,
                                                int k0, int k1, int sublane) {
    int k = k0 + sublane;
    double s0 = 0.0, s1 = 0.0;
    for (; k + LPR < k1; k += 2 * LPR) {
        const int c0 = colidx[k], c1 = colidx[k + LPR];
        const double v0 = vals[k], v1 = vals[k + LPR];
        s0 = fma(v0, x[c0], s0);
        s1 = fma(v1, x[c1], s1);
    }
    if (k < k1) s0 = fma(vals[k], x[colidx[k]], s0);
    return subwave_sum<LPR>(s0 + s1);
}

template <int LPR>
__device__ __forceinline__ double csr_row_dot(const int *__restrict__ rowptr,
                                              const int *__restrict__ colidx,
                                              const double *__restrict__ vals,
                                              const double *__restrict__ x,
                                              int row, int sublane) {
    const int k1 = rowptr[row + 1];
    int k = rowptr[row] + sublane;
    double s0 = 0.0, s1 = 0.0;
    // two (col,val) pairs and then two gathers in flight per lane
    for (; k + LPR < k1; k += 2 * LPR) {
        const int c0 = colidx[k], c1 = colidx[k + LPR];
        const double v0 = vals[k], v1 = vals[k + LPR];
        s0 = fma(v0, x[c0], s0);
        s1 = fma(v1, x[c1], s1);
    }
    if (k < k1) s0 = fma(vals[k], x[colidx[k]], s0);
    return subwave_sum<LPR>(s0 + s1);
}

// y = alpha*A*x + beta*b   (b may be null -> beta ignored)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_spmv_vec(int nrows, const int *__restrict__ rowptr,
           const int *__restrict__ colidx, const double *__restrict__ vals,
           const double *__restrict__ x, double *__restrict__ y, double alpha,
           double beta, const double *__restrict__ b,
           const int *__restrict__ guard = nullptr) {
    if (guard && *guard) return;       // (e.g. ctl->done: solve has finished)
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int row = sub; row < nrows; row += nsub) {
        double s = csr_row_dot<LPR>(rowptr, colidx, vals, x, row, sublane);
        if (sublane == 0)
            y[row] = b ? fma(alpha, s, beta * b[row]) : alpha * s;
    }
}

// ---------------------------------------------------------------------------
// CSR SpMV, "stream" form (CSR-adaptive style): one workgroup owns a block of
// consecutive rows whose non-zeros fit an LDS tile; phase 1 streams
// val*x[col] products into LDS with perfectly coalesced reads of the CSR row
// block irrespective of the row lengths; phase 2 reduces each row out of LDS
// with G lanes per row.
// ---------------------------------------------------------------------------
constexpr int kStreamNnz = 2048;     // products per LDS tile (16 KiB)
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef int v2i32 __attribute__((ext_vector_type(2)));

template <int G, int TILE = kStreamNnz, int UNR = 4, int RPS = 0>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream(int nblocks, const int *__restrict__ rowblocks,
              const int *__restrict__ rowptr, const int *__restrict__ colidx,
              const double *__restrict__ vals, const double *__restrict__ x,
              double *__restrict__ y, double alpha, double beta,
              const double *__restrict__ b) {
    __shared__ double prod[TILE];
    __shared__ double red[4];
    __shared__ int rps[kBlock + 1];   // RPS: the block's row pointers
    // XCD-aware start: workgroups b, b+8, b+16, ... land on the same XCD
    // (round-robin dispatch over the 8 XCDs), so give each residue class a
    // CONTIGUOUS range of row blocks: the x entries a class gathers then form
    // one window of ~n/8 entries that fits that XCD's private 4 MiB L2
    // (speed only -- any placement computes the same result).
    const int gq = gridDim.x / 8, gr = gridDim.x % 8;
    const int cls = blockIdx.x % 8;
    const int vb = cls * gq + (cls < gr ? cls : gr) + blockIdx.x / 8;
    for (int blk = vb; blk < nblocks; blk += gridDim.x) {
        const int r0 = rowblocks[blk], r1 = rowblocks[blk + 1];
        const int k0 = rowptr[r0], k1 = rowptr[r1];
        const int nn = k1 - k0;
        if (nn > TILE) {
            // a single long row: block-wide strided dot product
            double s = 0.0;
            for (int k = k0 + threadIdx.x; k < k1; k += kBlock)
                s = fma(vals[k], x[colidx[k]], s);
            s = block_sum(s, red);
            if (threadIdx.x == 0)
                y[r0] = b ? fma(alpha, s, beta * b[r0]) : alpha * s;
            continue;
        }
        __syncthreads();             // previous tile fully consumed
        const int nr = r1 - r0;
        // the block's row pointers ride along with the value stream (one
        // coalesced load) so that phase 2 starts from LDS, not from a
        // dependent global load
        if (RPS) {
            if (threadIdx.x < nr)
                rps[threadIdx.x] = rowptr[r0 + threadIdx.x] - k0;
            if (threadIdx.x == 0) rps[nr] = nn;
        }
        // lane-contiguous 8-byte value / 4-byte index loads (measured faster
        // here than 16-byte pair loads, whose LDS writes conflict 2-way, and
        // than non-temporal loads, which lose the Infinity Cache at mid sizes)
        if (UNR == 0 && nn > 0) {
            // branch-free full-tile loads (see k_spmv_stream16)
            constexpr int NI = TILE / kBlock;
            double v[NI], xv[NI];
            int c[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int kk = k0 + min((int)threadIdx.x + i * kBlock, nn - 1);
                v[i] = vals[kk];
                c[i] = colidx[kk];
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) xv[i] = x[c[i]];
#pragma unroll
            for (int i = 0; i < NI; ++i)
                prod[threadIdx.x + i * kBlock] = v[i] * xv[i];
        } else {
#pragma unroll 2
            for (int k = threadIdx.x; k < nn; k += kBlock)
                prod[k] = vals[k0 + k] * x[colidx[k0 + k]];
        }
        __syncthreads();
        const int g = threadIdx.x % G, rsub = threadIdx.x / G;
        for (int r = rsub; r < nr; r += kBlock / G) {
            const int a0 = RPS ? rps[r] : rowptr[r0 + r] - k0;
            const int a1 = RPS ? rps[r + 1] : rowptr[r0 + r + 1] - k0;
            double s = 0.0;
            for (int k = a0 + g; k < a1; k += G) s += prod[k];
            s = subwave_sum<G>(s);
            if (g == 0) {
                const int row = r0 + r;
                y[row] = b ? fma(alpha, s, beta * b[row]) : alpha * s;
            }
        }
    }
}

// The same kernel reading 16-bit column offsets (CsrDev::c16): per row block
// two window bases, bit 15 of an entry selects the window.  HBM traffic per
// non-zero drops from 12 to 10 bytes; blocks whose columns do not fit the two
// windows (bases < 0) read the plain 32-bit indices.
// DIAG != 0 are diagnostic variants (scripts/spmv_c16.py; wrong results by
// design): 1 = x read at the stream position instead of the column (no
// gather), 2 = additionally no LDS reduction phase
// PAIR: every lane loads TWO consecutive non-zeros per instruction (16 bytes of
// values, 4 bytes of offsets) from the even-aligned stream -- half the load
// instructions for the same bytes
typedef unsigned short dns_ushort2 __attribute__((ext_vector_type(2)));
typedef double dns_double2 __attribute__((ext_vector_type(2)));

template <int G, int DIAG = 0, typename VT = double, int PAIR = 0>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream16(int nblocks, const int *__restrict__ rowblocks,
                const int *__restrict__ rowptr, const int *__restrict__ colidx,
                const unsigned short *__restrict__ c16,
                const int *__restrict__ c16base,
                const VT *__restrict__ vals, const double *__restrict__ x,
                double *__restrict__ y, double alpha, double beta,
                const double *__restrict__ b,
                const int *__restrict__ guard = nullptr) {
    if (guard && *guard) return;     // (ctl->done inside a Krylov cycle)
    constexpr int TILE = kStreamNnz;
    __shared__ double prod[TILE];
    __shared__ double red[4];
    __shared__ int rps[kBlock + 1];
    const int gq = gridDim.x / 8, gr = gridDim.x % 8;
    const int cls = blockIdx.x % 8;
    const int vb = cls * gq + (cls < gr ? cls : gr) + blockIdx.x / 8;
    for (int blk = vb; blk < nblocks; blk += gridDim.x) {
        // ONE 32-byte record per row block (r0, nr, k0, nn, blo, bhi): two
        // scalar loads in front of the stream loads instead of a chain of
        // three dependent ones (rowblocks -> rowptr -> ...)
        const int4 m0 = reinterpret_cast<const int4 *>(rowblocks)[2 * blk];
        const int4 m1 = reinterpret_cast<const int4 *>(rowblocks)[2 * blk + 1];
        const int r0 = m0.x, r1 = m0.x + m0.y;
        const int k0 = m0.z, nn = m0.w, k1 = k0 + nn;
        const int blo = m1.x, bhi = m1.y;
        if (nn > TILE) {
            double s = 0.0;
            for (int k = k0 + threadIdx.x; k < k1; k += kBlock)
                s = fma((double)vals[k], x[colidx[k]], s);
            s = block_sum(s, red);
            if (threadIdx.x == 0)
                y[r0] = b ? fma(alpha, s, beta * b[r0]) : alpha * s;
            continue;
        }
        __syncthreads();             // previous tile fully consumed
        const int nr = r1 - r0;
        if (threadIdx.x < nr) rps[threadIdx.x] = rowptr[r0 + threadIdx.x] - k0;
        if (threadIdx.x == 0) rps[nr] = nn;
        if (PAIR && blo >= 0 && nn > 0 && sizeof(VT) == 8 &&
            (nn + (k0 & 1)) <= TILE) {
            // pairs from the even-aligned stream: position q of the window
            // [ke, ke + TILE), ke = k0 rounded down to even; entries outside
            // [k0, k1) belong to the neighbours (or to the two spare entries
            // behind the arrays) and are not stored
            constexpr int NP2 = TILE / kBlock / 2;
            const int ke = k0 & ~1;
            const int last = (k1 - 1 - ke) >> 1;        // last pair with data
            dns_double2 v[NP2];
            dns_ushort2 e[NP2];
            const dns_double2 *v2 =
                reinterpret_cast<const dns_double2 *>(vals + ke);
            const dns_ushort2 *e2 =
                reinterpret_cast<const dns_ushort2 *>(c16 + ke);
#pragma unroll
            for (int i = 0; i < NP2; ++i) {
                const int q = min((int)threadIdx.x + i * kBlock, last);
                v[i] = v2[q];
                e[i] = e2[q];
            }
            double xa[NP2], xb[NP2];
#pragma unroll
            for (int i = 0; i < NP2; ++i) {
                const int ca = ((e[i].x & 0x8000) ? bhi : blo) + (e[i].x & 0x7fff);
                const int cb = ((e[i].y & 0x8000) ? bhi : blo) + (e[i].y & 0x7fff);
                // (a neighbour's entry decodes against this block's bases: keep
                // the address inside x)
                const int q = min((int)threadIdx.x + i * kBlock, last);
                const bool oka = ke + 2 * q >= k0, okb = ke + 2 * q + 1 < k1;
                xa[i] = oka ? x[ca] : 0.0;
                xb[i] = okb ? x[cb] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < NP2; ++i) {
                const int q = (int)threadIdx.x + i * kBlock;
                const int pa = 2 * q - (k0 & 1);
                if (q <= last) {
                    if (pa >= 0) prod[pa] = v[i].x * xa[i];
                    if (pa + 1 < nn) prod[pa + 1] = v[i].y * xb[i];
                }
            }
        } else if (blo >= 0 && nn > 0) {
            // branch-free: every lane issues its TILE/kBlock value and index
            // loads back to back (positions past the end are clamped to the
            // last entry; their products land in LDS slots nobody reads), then
            // the gathers, then the LDS stores -- the whole 20 KB tile is in
            // flight at once instead of two loads per lane
            constexpr int NI = TILE / kBlock;
            double v[NI];
            int e[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int kk = k0 + min((int)threadIdx.x + i * kBlock, nn - 1);
                v[i] = (double)vals[kk];
                e[i] = c16[kk];
            }
            double xv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int col = ((e[i] & 0x8000) ? bhi : blo) + (e[i] & 0x7fff);
                if (DIAG == 3) col &= 1023;       // gather from an 8 KB range
                else if (DIAG) col = r0 + (i & 7);
                xv[i] = x[col];
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
                prod[threadIdx.x + i * kBlock] = v[i] * xv[i];
        } else {
#pragma unroll 2
            for (int k = threadIdx.x; k < nn; k += kBlock)
                prod[k] = (double)vals[k0 + k] * x[colidx[k0 + k]];
        }
        __syncthreads();
        const int g = threadIdx.x % G, rsub = threadIdx.x / G;
        for (int r = rsub; r < nr; r += kBlock / G) {
            const int a0 = rps[r], a1 = (DIAG == 2) ? a0 + 1 : rps[r + 1];
            double s = 0.0;
            for (int k = a0 + g; k < a1; k += G) s += prod[k];
            s = subwave_sum<G>(s);
            if (g == 0) {
                const int row = r0 + r;
                y[row] = b ? fma(alpha, s, beta * b[row]) : alpha * s;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The streaming kernel with the epilogues the solver needs in the bandwidth
// regime (n >~ 1e5), so that its large applies all run at the streaming rate:
//   * plain           y = alpha A x + beta b
//   * split x         the input is [x ; x2] in two buffers (Gc [V_j,v ; zp]:
//                     no packing kernel in front)
//   * Jacobi sweep    y = xin + omega dinv .* (b - A xin)   (multigrid smoother)
//   * fused dots      part[i][wg] = <V_i, y> (i < nvec) and <y, y>: the
//                     Gram-Schmidt dots of the Arnoldi step ride on the K apply
//                     (what k_spmv_multidot does for the sub-wave kernel)
// Any grid size works (workgroups stride over the row blocks), so a fixed grid
// bounds the number of partials the consumers re-reduce.
// ---------------------------------------------------------------------------
struct StreamEpi {
    double alpha, beta;
    const double *b;
    const double *x2;            // second input buffer (columns >= nsplit)
    int nsplit;
    const double *dinv, *xin;    // Jacobi sweep (dinv != nullptr)
    double omega;
    int add_cb;                  // y = alpha A x + omega dinv .* b (the fused
                                 // multigrid operators; dinv, b, omega as in
                                 // the sweep)
    const double *V;             // fused dots (part != nullptr)
    size_t ld;
    int nvec, with_ww;
    double *part;
    int nparts;
    double *part_bb;             // != nullptr: partials of <b, b> as well (the
                                 // residual r = b - A x of a solve with both
                                 // of its norms in ONE launch)
    int map_on;                  // row-partitioned solve: the CSR arrays hold
    RowMap rm;                   // a rank's row block; local row -> global row
                                 // (y, b, V are indexed by global row)
};

inline StreamEpi stream_epi_plain(double alpha, double beta, const double *b) {
    StreamEpi e;
    memset(&e, 0, sizeof(e));
    e.alpha = alpha;
    e.beta = beta;
    e.b = b;
    return e;
}

constexpr int kStreamDots = 8;   // fused dots: at most this many basis vectors
constexpr int kStreamGrid = 1024;   // workgroups (= partials) of such a launch

// DOTS: instantiation with the fused-dots epilogue; the others keep the dot
// accumulators out of the register file (58 instead of 80 VGPRs: 8 instead of
// 5-6 waves per SIMD, which these latency-hiding-by-occupancy kernels turn
// into bandwidth)
template <int G, typename VT, bool DOTS = true>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream16x(int nblocks, const int *__restrict__ rowblocks,
                 const int *__restrict__ rowptr,
                 const int *__restrict__ colidx,
                 const unsigned short *__restrict__ c16,
                 const int *__restrict__ c16base, const VT *__restrict__ vals,
                 const double *__restrict__ x, double *__restrict__ y,
                 StreamEpi ep, const int *__restrict__ guard) {
    if (guard && *guard) return;
    constexpr int TILE = kStreamNnz;
    __shared__ double prod[TILE];
    __shared__ double red[4];
    __shared__ int rps[kBlock + 1];
    __shared__ double srow[kBlock];      // fused dots: the tile's row results
    const int gq = gridDim.x / 8, gr = gridDim.x % 8;
    const int cls = blockIdx.x % 8;
    const int vb = cls * gq + (cls < gr ? cls : gr) + blockIdx.x / 8;
    const bool split = ep.x2 != nullptr;
    const bool dots = DOTS && ep.part != nullptr;
    double acc[DOTS ? kStreamDots + 1 : 1];
    double accb = 0.0;
#pragma unroll
    for (int i = 0; i < (DOTS ? kStreamDots + 1 : 1); ++i) acc[i] = 0.0;
    for (int blk = vb; blk < nblocks; blk += gridDim.x) {
        // ONE 32-byte record per row block (r0, nr, k0, nn, blo, bhi): two
        // scalar loads in front of the stream loads instead of a chain of
        // three dependent ones (rowblocks -> rowptr -> ...)
        const int4 m0 = reinterpret_cast<const int4 *>(rowblocks)[2 * blk];
        const int4 m1 = reinterpret_cast<const int4 *>(rowblocks)[2 * blk + 1];
        const int r0 = m0.x, r1 = m0.x + m0.y;
        const int k0 = m0.z, nn = m0.w, k1 = k0 + nn;
        const int blo = m1.x, bhi = m1.y;
        __syncthreads();             // previous tile fully consumed
        const int nr = r1 - r0;
        bool single = nn > TILE;     // a single long row
        if (single) {
            double s = 0.0;
            for (int k = k0 + threadIdx.x; k < k1; k += kBlock) {
                const int c = colidx[k];
                const double xv = (split && c >= ep.nsplit)
                                      ? ep.x2[c - ep.nsplit] : x[c];
                s = fma((double)vals[k], xv, s);
            }
            s = block_sum(s, red);
            if (threadIdx.x == 0) prod[0] = s;
            if (threadIdx.x == 0) {
                rps[0] = 0;
                rps[1] = 1;
            }
        } else {
            if (threadIdx.x < nr)
                rps[threadIdx.x] = rowptr[r0 + threadIdx.x] - k0;
            if (threadIdx.x == 0) rps[nr] = nn;
            if (blo >= 0 && nn > 0) {
                constexpr int NI = TILE / kBlock;
                double v[NI];
                int e[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int kk =
                        k0 + min((int)threadIdx.x + i * kBlock, nn - 1);
                    v[i] = (double)vals[kk];
                    e[i] = c16[kk];
                }
                double xv[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int col =
                        ((e[i] & 0x8000) ? bhi : blo) + (e[i] & 0x7fff);
                    xv[i] = (split && col >= ep.nsplit)
                                ? ep.x2[col - ep.nsplit] : x[col];
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    prod[threadIdx.x + i * kBlock] = v[i] * xv[i];
            } else {
#pragma unroll 2
                for (int k = threadIdx.x; k < nn; k += kBlock) {
                    const int c = colidx[k0 + k];
                    const double xv = (split && c >= ep.nsplit)
                                          ? ep.x2[c - ep.nsplit] : x[c];
                    prod[k] = (double)vals[k0 + k] * xv;
                }
            }
        }
        __syncthreads();
        const int g = threadIdx.x % G, rsub = threadIdx.x / G;
        for (int r = rsub; r < nr; r += kBlock / G) {
            const int a0 = rps[r], a1 = rps[r + 1];
            double s = 0.0;
            for (int k = a0 + g; k < a1; k += G) s += prod[k];
            s = subwave_sum<G>(s);
            if (g == 0) {
                const int row = ep.map_on ? map_row(ep.rm, r0 + r) : r0 + r;
                double out;
                if (ep.dinv && ep.add_cb) {
                    out = fma(ep.alpha, s,
                              ep.omega * ep.dinv[row] * ep.b[row]);
                } else if (ep.dinv) {
                    const double xi = ep.xin ? ep.xin[row] : 0.0;
                    out = ep.alpha *
                          (xi + ep.omega * ep.dinv[row] * (ep.b[row] - s));
                } else {
                    out = ep.b ? fma(ep.alpha, s, ep.beta * ep.b[row])
                               : ep.alpha * s;
                }
                y[row] = out;
                if (DOTS && dots) srow[r] = out;
            }
        }
        if (DOTS && dots) {
            // thread t takes row t of the tile: basis loads are coalesced and
            // every lane works (the row results sit in 1 of G lanes)
            __syncthreads();
            if ((int)threadIdx.x < nr) {
                const double out = srow[threadIdx.x];
                const size_t row =
                    ep.map_on ? (size_t)map_row(ep.rm, r0 + (int)threadIdx.x)
                              : (size_t)r0 + threadIdx.x;
#pragma unroll
                for (int i = 0; i < kStreamDots; ++i)
                    if (i < ep.nvec)
                        acc[DOTS ? i : 0] = fma(ep.V[(size_t)i * ep.ld + row],
                                                out, acc[DOTS ? i : 0]);
                acc[DOTS ? kStreamDots : 0] =
                    fma(out, out, acc[DOTS ? kStreamDots : 0]);
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
            for (int q = 0; q < kStreamDots; ++q)
                if (q == i) a = acc[DOTS ? q : 0];
            a = block_sum(a, red);
            if (threadIdx.x == 0)
                ep.part[(size_t)i * ep.nparts + blockIdx.x] = a;
        }
        if (ep.with_ww) {
            const double a = block_sum(acc[DOTS ? kStreamDots : 0], red);
            if (threadIdx.x == 0)
                ep.part[(size_t)ep.nvec * ep.nparts + blockIdx.x] = a;
        }
        if (ep.part_bb) {
            const double a = block_sum(accb, red);
            if (threadIdx.x == 0) ep.part_bb[blockIdx.x] = a;
        }
    }
}

// ---------------------------------------------------------------------------
// vector kernels
// ---------------------------------------------------------------------------
// y = a*x + b*y
__global__ void __launch_bounds__(kBlock)
k_axpby(int64_t n, double a, const double *__restrict__ x, double b,
        double *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kBlock)
        y[i] = (b == 0.0) ? a * x[i] : fma(a, x[i], b * y[i]);
}

__global__ void __launch_bounds__(kBlock)
k_scale_by(int n, const double *__restrict__ d, double *__restrict__ y) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        y[i] *= d[i];
}

// partials[blockIdx] = sum x[i]*y[i] over this block's grid-stride share
__global__ void __launch_bounds__(kBlock)
k_dot_partials(int64_t n, const double *__restrict__ x,
               const double *__restrict__ y, double *__restrict__ partials) {
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kBlock)
        s = fma(x[i], y[i], s);
    s = block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// out[0] = sum of partials (single block)
__global__ void __launch_bounds__(kBlock)
k_sum_partials(const double *__restrict__ partials, int nparts,
               double *__restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += kBlock) s += partials[p];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

// two such sums in ONE launch (workgroup 0: a -> out[0], 1: b -> out[1])
__global__ void __launch_bounds__(kBlock)
k_sum_partials2(const double *__restrict__ pa, const double *__restrict__ pb,
                int nparts, double *__restrict__ out) {
    __shared__ double red[4];
    const double *__restrict__ p = blockIdx.x == 0 ? pa : pb;
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += kBlock) s += p[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kBlock)
k_diag_inv(int nrows, const int *__restrict__ rowptr,
           const int *__restrict__ colidx, const double *__restrict__ vals,
           double *__restrict__ dinv) {
    for (int row = blockIdx.x * kBlock + threadIdx.x; row < nrows;
         row += gridDim.x * kBlock) {
        double d = 0.0;
        for (int k = rowptr[row]; k < rowptr[row + 1]; ++k)
            if (colidx[k] == row) d += vals[k];
        dinv[row] = (d != 0.0) ? 1.0 / d : 1.0;
    }
}

// deterministic start vectors of the power iterations (as tests/krylov_model.py)
__global__ void __launch_bounds__(kBlock)
k_fill_wave(int n, double *__restrict__ x, double freq, double phase,
            int use_cos) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock) {
        const double a = freq * (double)i + phase;
        x[i] = 1.0 + 0.5 * (use_cos ? cos(a) : sin(a));
    }
}

// K.vals <- F.vals for the velocity rows (same pattern, new values)
__global__ void __launch_bounds__(kBlock)
k_scatter_fvals(int nv, const int *__restrict__ f_rowptr,
                const int *__restrict__ k_rowptr,
                const double *__restrict__ fvals, double *__restrict__ kvals) {
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / 8;
    const int sl = threadIdx.x % 8;
    const int nsub = gridDim.x * (kBlock / 8);
    for (int row = sub; row < nv; row += nsub) {
        const int f0 = f_rowptr[row], f1 = f_rowptr[row + 1];
        const int k0 = k_rowptr[row];
        for (int k = f0 + sl; k < f1; k += 8) kvals[k0 + (k - f0)] = fvals[k];
    }
}

// the same for a row block of K: local row i of `k_rowptr` is global row
// row0 + i of F
__global__ void __launch_bounds__(kBlock)
k_scatter_fvals_rows(int row0, int nrows, const int *__restrict__ f_rowptr,
                     const int *__restrict__ k_rowptr,
                     const double *__restrict__ fvals,
                     double *__restrict__ kvals) {
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / 8;
    const int sl = threadIdx.x % 8;
    const int nsub = gridDim.x * (kBlock / 8);
    for (int i = sub; i < nrows; i += nsub) {
        const int f0 = f_rowptr[row0 + i], f1 = f_rowptr[row0 + i + 1];
        const int k0 = k_rowptr[i];
        for (int k = f0 + sl; k < f1; k += 8) kvals[k0 + (k - f0)] = fvals[k];
    }
}

// ---------------------------------------------------------------------------
// dense Schur block:  y = alpha * A x,  A is n x n row-major; one wavefront
// per row, 16-byte loads where the row start is 16-byte aligned.
// `xsel`/`ysel`: vector selectors (see vec_at)
// ---------------------------------------------------------------------------
template <typename VT>
__global__ void __launch_bounds__(kBlock)
k_gemv_rows(int n, const VT *__restrict__ a, const double *__restrict__ x,
            double *__restrict__ y, double alpha, const DnsCtl *ctl) {
    if (ctl && ctl->done) return;
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * kBlock) >> 6;
    for (int row = wave; row < n; row += nwaves) {
        const VT *ar = a + (size_t)row * n;
        // four independent loads in flight per lane
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = lane;
        for (; c + 192 < n; c += 256) {
            s0 = fma((double)ar[c], x[c], s0);
            s1 = fma((double)ar[c + 64], x[c + 64], s1);
            s2 = fma((double)ar[c + 128], x[c + 128], s2);
            s3 = fma((double)ar[c + 192], x[c + 192], s3);
        }
        for (; c < n; c += 64) s0 = fma((double)ar[c], x[c], s0);
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) y[row] = alpha * s;
    }
}

// in-place Gauss-Jordan inversion of an SPD matrix, step k, part 1:
// save the scaled pivot row and the pivot column
__global__ void __launch_bounds__(kBlock)
k_gj_pivot(int n, int k, const double *__restrict__ a,
           double *__restrict__ prow, double *__restrict__ pcol,
           int *__restrict__ flag) {
    const double piv = a[(size_t)k * n + k];
    if (!(fabs(piv) > 0.0) || isnan(piv)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 1;
        return;
    }
    const double p = 1.0 / piv;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock) {
        prow[i] = a[(size_t)k * n + i] * p;
        pcol[i] = a[(size_t)i * n + k];
    }
}

// part 2: rank-one update of everything outside row/column k, then the
// pivot row / column themselves
__global__ void __launch_bounds__(kBlock)
k_gj_update(int n, int k, double *__restrict__ a,
            const double *__restrict__ prow, const double *__restrict__ pcol,
            const int *__restrict__ flag) {
    if (*flag) return;
    const double p = 1.0 / pcol[k];
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * kBlock) {
        const int i = (int)(e / n), j = (int)(e % n);
        double v;
        if (i == k)
            v = (j == k) ? p : prow[j];
        else if (j == k)
            v = -pcol[i] * p;
        else
            v = a[e] - pcol[i] * prow[j];
        a[e] = v;
    }
}

// ---------------------------------------------------------------------------
// Jacobi-Chebyshev approximation of F^-1 (block preconditioner)
// ---------------------------------------------------------------------------
// start:  t = rv - JT zp;  r = dinv*t;  d = r/theta   (x = d if `single`)
// rv = rsel[j*ld .. ] velocity part of the selected input vector
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_cheb_init(int nv, const int *__restrict__ rowptr,
            const int *__restrict__ colidx, const double *__restrict__ vals,
            const double *__restrict__ zp, const double *__restrict__ rbase,
            size_t ld, const int *__restrict__ jsel,
            const double *__restrict__ dinv, double inv_theta,
            double *__restrict__ r, double *__restrict__ d,
            const int *__restrict__ guard, int row0, int row1) {
    if (*guard) return;
    (void)nv;
    const double *rv = rbase + (size_t)(*jsel) * ld;
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int row = row0 + sub; row < row1; row += nsub) {
        double s = csr_row_dot<LPR>(rowptr, colidx, vals, zp, row, sublane);
        if (sublane == 0) {
            const double rr = dinv[row] * (rv[row] - s);
            r[row] = rr;
            d[row] = rr * inv_theta;
        }
    }
}

// one Chebyshev step:
//   r -= dinv*(F d_in);  x (+)= d_in;  d_new = c1*d_in + c2*r
//   last: x += d_new  else d_out = d_new
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_cheb_step(int nv, const int *__restrict__ rowptr,
            const int *__restrict__ colidx, const double *__restrict__ vals,
            const double *__restrict__ dinv, const double *__restrict__ d_in,
            double *__restrict__ d_out, double *__restrict__ r,
            double *__restrict__ x, double c1, double c2, int first, int last,
            const int *__restrict__ guard, int row0, int row1) {
    if (*guard) return;
    (void)nv;
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int row = row0 + sub; row < row1; row += nsub) {
        double s = csr_row_dot<LPR>(rowptr, colidx, vals, d_in, row, sublane);
        if (sublane == 0) {
            const double di = d_in[row];
            const double rn = r[row] - dinv[row] * s;
            const double dn = c1 * di + c2 * rn;
            const double xo = (first ? 0.0 : x[row]) + di;
            r[row] = rn;
            if (last) {
                x[row] = xo + dn;
            } else {
                x[row] = xo;
                d_out[row] = dn;
            }
        }
    }
}

// explicit approximate inverse: zv = Gc [rv; zp], Gc = [G, -G JT] (NV x (NV+NP))
// with G = p(D^-1 F) D^-1 the SAME Chebyshev polynomial as k_cheb_*, formed once
// on the host: one SpMV instead of `degree` dependent launches (latency-bound
// sizes).  Columns < nv read the selected input vector, the rest read zp.
template <int LPR, typename VT>
__global__ void __launch_bounds__(kBlock)
k_spmv_split(int nv, const int *__restrict__ rowptr,
             const int *__restrict__ colidx, const VT *__restrict__ vals,
             const double *__restrict__ rbase, size_t ld,
             const int *__restrict__ jsel, const double *__restrict__ zp,
             double *__restrict__ zv, const int *__restrict__ guard,
             double *__restrict__ xacc, int row0, int row1) {
    // the first row's pointers are asked for BEFORE the guard is looked at:
    // written guard-first, the kernel starts with a chain of dependent scalar
    // loads (kernel arguments -> *guard -> more arguments -> *jsel) in front of
    // its first vector load -- 1-2 us of a 5 us kernel
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    const int rfirst = row0 + sub;
    int kf0 = 0, kf1 = 0;
    if (rfirst < row1) {
        kf0 = rowptr[rfirst];
        kf1 = rowptr[rfirst + 1];
    }
    const int gdone = *guard;
    const int jcol = *jsel;
    if (gdone) return;
    const double *rv = rbase + (size_t)jcol * ld;
    for (int row = rfirst; row < row1; row += nsub) {
        const int k1 = (row == rfirst) ? kf1 : rowptr[row + 1];
        int k = ((row == rfirst) ? kf0 : rowptr[row]) + sublane;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (; k + 3 * LPR < k1; k += 4 * LPR) {
            const int c0 = colidx[k], c1 = colidx[k + LPR];
            const int c2 = colidx[k + 2 * LPR], c3 = colidx[k + 3 * LPR];
            const double v0 = (double)vals[k], v1 = (double)vals[k + LPR];
            const double v2 = (double)vals[k + 2 * LPR];
            const double v3 = (double)vals[k + 3 * LPR];
            s0 = fma(v0, c0 < nv ? rv[c0] : zp[c0 - nv], s0);
            s1 = fma(v1, c1 < nv ? rv[c1] : zp[c1 - nv], s1);
            s2 = fma(v2, c2 < nv ? rv[c2] : zp[c2 - nv], s2);
            s3 = fma(v3, c3 < nv ? rv[c3] : zp[c3 - nv], s3);
        }
        for (; k < k1; k += LPR) {
            const int c = colidx[k];
            s0 = fma((double)vals[k], c < nv ? rv[c] : zp[c - nv], s0);
        }
        const double s = subwave_sum<LPR>((s0 + s1) + (s2 + s3));
        if (sublane == 0) {
            zv[row] = s;
            if (xacc) xacc[row] += s;      // fused correction x_v += z_v
        }
    }
}

// Jacobi Schur fallback: zp = -sdinv * rp   (xacc: x_p += z_p)
__global__ void __launch_bounds__(kBlock)
k_schur_jacobi(int np, const double *__restrict__ sdinv,
               const double *__restrict__ rbase, size_t ld,
               const int *__restrict__ jsel, int nv, double *__restrict__ zp,
               const int *__restrict__ guard, double *__restrict__ xacc,
               int row0, int row1) {
    if (*guard) return;
    (void)np;
    const double *rp = rbase + (size_t)(*jsel) * ld + nv;
    for (int i = row0 + blockIdx.x * kBlock + threadIdx.x; i < row1;
         i += gridDim.x * kBlock) {
        const double v = -sdinv[i] * rp[i];
        zp[i] = v;
        if (xacc) xacc[i] += v;
    }
}

// out[row] = scale * <A[row,:], x> for the rows of this workgroup (row =
// blockIdx.x, += gridDim.x): the WHOLE workgroup works on one dense row so
// that every load of the row is in flight at once (latency, not bandwidth,
// bounds the n ~ 1e3 Schur block); `red` holds >= 4 doubles of LDS
template <typename VT>
__device__ __forceinline__ void dense_rows_block(const VT *__restrict__ a,
                                                 const double *__restrict__ x,
                                                 int n, double scale,
                                                 double *__restrict__ out,
                                                 double *__restrict__ xacc,
                                                 double *red, int row0 = 0,
                                                 int row1 = -1, int lda = 0) {
    if (row1 < 0) row1 = n;
    if (lda <= 0) lda = n;
    for (int row = row0 + blockIdx.x; row < row1; row += gridDim.x) {
        const VT *ar = a + (size_t)row * lda;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = threadIdx.x;
        for (; c + 3 * kBlock < n; c += 4 * kBlock) {
            const double a0 = (double)ar[c], a1 = (double)ar[c + kBlock];
            const double a2 = (double)ar[c + 2 * kBlock];
            const double a3 = (double)ar[c + 3 * kBlock];
            s0 = fma(a0, x[c], s0);
            s1 = fma(a1, x[c + kBlock], s1);
            s2 = fma(a2, x[c + 2 * kBlock], s2);
            s3 = fma(a3, x[c + 3 * kBlock], s3);
        }
        const double t0 = (c < n) ? (double)ar[c] * x[c] : 0.0;
        const double t1 =
            (c + kBlock < n) ? (double)ar[c + kBlock] * x[c + kBlock] : 0.0;
        const double t2 = (c + 2 * kBlock < n)
                              ? (double)ar[c + 2 * kBlock] * x[c + 2 * kBlock]
                              : 0.0;
        double s = ((s0 + s1) + (s2 + s3)) + ((t0 + t1) + t2);
        s = block_sum(s, red);
        if (threadIdx.x == 0) {
            const double v = scale * s;
            out[row] = v;
            if (xacc) xacc[row] += v;
        }
    }
}

// one WAVE per dense row (the Arnoldi head kernels: four rows per workgroup, so
// that a workgroup's reduction of the dot-product partials is shared by four
// rows and the kernel has np/4 workgroups instead of np): the whole row is in
// flight before the first use; returns the wave-reduced dot product
typedef float dns_float4 __attribute__((ext_vector_type(4)));

// fp32 row whose start is 16-byte aligned and whose length is padded to a
// multiple of four (zeros): 16-byte loads, a quarter of the load instructions.
// `x` has n entries (nothing beyond them is read)
__device__ __forceinline__ double dense_row_wave_f4(const float *__restrict__ ar,
                                                    const double *__restrict__ x,
                                                    int n, int lane) {
    const dns_float4 *a4 = reinterpret_cast<const dns_float4 *>(ar);
    const int n4 = n >> 2, rem = n & 3;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (rem && lane == 63) {
        // the last, partial group of four
        const dns_float4 v = a4[n4];
        const double *xq = x + 4 * n4;
        s0 = (double)v.x * xq[0];
        if (rem > 1) s1 = (double)v.y * xq[1];
        if (rem > 2) s2 = (double)v.z * xq[2];
    }
    int q = lane;
    for (; q + 192 < n4; q += 256) {
        const dns_float4 v0 = a4[q], v1 = a4[q + 64], v2 = a4[q + 128],
                         v3 = a4[q + 192];
        const double *x0 = x + 4 * q, *x1 = x0 + 256, *x2 = x0 + 512,
                     *x3 = x0 + 768;
        s0 = fma((double)v0.x, x0[0], s0);
        s1 = fma((double)v0.y, x0[1], s1);
        s2 = fma((double)v0.z, x0[2], s2);
        s3 = fma((double)v0.w, x0[3], s3);
        s0 = fma((double)v1.x, x1[0], s0);
        s1 = fma((double)v1.y, x1[1], s1);
        s2 = fma((double)v1.z, x1[2], s2);
        s3 = fma((double)v1.w, x1[3], s3);
        s0 = fma((double)v2.x, x2[0], s0);
        s1 = fma((double)v2.y, x2[1], s1);
        s2 = fma((double)v2.z, x2[2], s2);
        s3 = fma((double)v2.w, x2[3], s3);
        s0 = fma((double)v3.x, x3[0], s0);
        s1 = fma((double)v3.y, x3[1], s1);
        s2 = fma((double)v3.z, x3[2], s2);
        s3 = fma((double)v3.w, x3[3], s3);
    }
    // up to three more 16-byte loads per lane, all issued before their use
    const bool t0 = q < n4, t1 = q + 64 < n4, t2 = q + 128 < n4;
    const dns_float4 z4 = {0.f, 0.f, 0.f, 0.f};
    const dns_float4 v0 = t0 ? a4[q] : z4, v1 = t1 ? a4[q + 64] : z4,
                     v2 = t2 ? a4[q + 128] : z4;
    if (t0) {
        const double *xq = x + 4 * q;
        s0 = fma((double)v0.x, xq[0], s0);
        s1 = fma((double)v0.y, xq[1], s1);
        s2 = fma((double)v0.z, xq[2], s2);
        s3 = fma((double)v0.w, xq[3], s3);
    }
    if (t1) {
        const double *xq = x + 4 * (q + 64);
        s0 = fma((double)v1.x, xq[0], s0);
        s1 = fma((double)v1.y, xq[1], s1);
        s2 = fma((double)v1.z, xq[2], s2);
        s3 = fma((double)v1.w, xq[3], s3);
    }
    if (t2) {
        const double *xq = x + 4 * (q + 128);
        s0 = fma((double)v2.x, xq[0], s0);
        s1 = fma((double)v2.y, xq[1], s1);
        s2 = fma((double)v2.z, xq[2], s2);
        s3 = fma((double)v2.w, xq[3], s3);
    }
    return wave_sum((s0 + s1) + (s2 + s3));
}

template <typename VT>
__device__ __forceinline__ double dense_row_wave(const VT *__restrict__ ar,
                                                 const double *__restrict__ x,
                                                 int n, int lane) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = lane;
    for (; c + 448 < n; c += 512) {
        const double a0 = (double)ar[c], a1 = (double)ar[c + 64];
        const double a2 = (double)ar[c + 128], a3 = (double)ar[c + 192];
        const double a4 = (double)ar[c + 256], a5 = (double)ar[c + 320];
        const double a6 = (double)ar[c + 384], a7 = (double)ar[c + 448];
        s0 = fma(a0, x[c], s0);
        s1 = fma(a1, x[c + 64], s1);
        s2 = fma(a2, x[c + 128], s2);
        s3 = fma(a3, x[c + 192], s3);
        s0 = fma(a4, x[c + 256], s0);
        s1 = fma(a5, x[c + 320], s1);
        s2 = fma(a6, x[c + 384], s2);
        s3 = fma(a7, x[c + 448], s3);
    }
    for (; c < n; c += 64) s0 = fma((double)ar[c], x[c], s0);
    return wave_sum((s0 + s1) + (s2 + s3));
}

// dense Schur: zp = -Sinv rp  (rp = selected vector's pressure part)
template <typename VT>
__global__ void __launch_bounds__(kBlock)
k_schur_dense(int np, const VT *__restrict__ sinv,
              const double *__restrict__ rbase, size_t ld,
              const int *__restrict__ jsel, int nv, double *__restrict__ zp,
              const int *__restrict__ guard, double *__restrict__ xacc,
              int row0, int row1, int lda) {
    if (*guard) return;
    __shared__ double red[4];
    const double *rp = rbase + (size_t)(*jsel) * ld + nv;
    dense_rows_block<VT>(sinv, rp, np, -1.0, zp, xacc, red, row0, row1, lda);
}

// plain streaming kernels: what the HBM delivers to a kernel of this library
// (dns_hbm_probe); 16-byte accesses, grid-stride, fully coalesced

__global__ void __launch_bounds__(kBlock)
k_stream_read(int64_t n2, const dns_double2 *__restrict__ a,
              double *__restrict__ part) {
    __shared__ double red[4];
    double s0 = 0.0, s1 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + stride < n2; i += 2 * stride) {
        const dns_double2 u = a[i], v = a[i + stride];
        s0 += u.x + u.y;
        s1 += v.x + v.y;
    }
    if (i < n2) {
        const dns_double2 u = a[i];
        s0 += u.x + u.y;
    }
    const double s = block_sum(s0 + s1, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// the same with eight 16-byte loads in flight per lane
__global__ void __launch_bounds__(kBlock)
k_stream_read8(int64_t n2, const dns_double2 *__restrict__ a,
               double *__restrict__ part) {
    __shared__ double red[4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 7 * stride < n2; i += 8 * stride) {
        const dns_double2 u0 = a[i], u1 = a[i + stride];
        const dns_double2 u2 = a[i + 2 * stride], u3 = a[i + 3 * stride];
        const dns_double2 u4 = a[i + 4 * stride], u5 = a[i + 5 * stride];
        const dns_double2 u6 = a[i + 6 * stride], u7 = a[i + 7 * stride];
        s0 += (u0.x + u0.y) + (u4.x + u4.y);
        s1 += (u1.x + u1.y) + (u5.x + u5.y);
        s2 += (u2.x + u2.y) + (u6.x + u6.y);
        s3 += (u3.x + u3.y) + (u7.x + u7.y);
    }
    for (; i < n2; i += stride) {
        const dns_double2 u = a[i];
        s0 += u.x + u.y;
    }
    const double s = block_sum((s0 + s1) + (s2 + s3), red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// 8-byte lane loads (what a CSR value stream uses), eight in flight
__global__ void __launch_bounds__(kBlock)
k_stream_read_b64(int64_t n, const double *__restrict__ a,
                  double *__restrict__ part) {
    __shared__ double red[4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        s0 += a[i] + a[i + 4 * stride];
        s1 += a[i + stride] + a[i + 5 * stride];
        s2 += a[i + 2 * stride] + a[i + 6 * stride];
        s3 += a[i + 3 * stride] + a[i + 7 * stride];
    }
    for (; i < n; i += stride) s0 += a[i];
    const double s = block_sum((s0 + s1) + (s2 + s3), red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// block-contiguous 8-byte lane loads: workgroup b reads tiles b, b+grid, ...
// of 2048 doubles (the access pattern of the CSR stream kernels)
__global__ void __launch_bounds__(kBlock)
k_stream_read_tiles(int64_t n, const double *__restrict__ a,
                    double *__restrict__ part) {
    __shared__ double red[4];
    double s0 = 0.0, s1 = 0.0;
    const int64_t ntiles = n / 2048;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const double *p = a + t * 2048;
#pragma unroll 2
        for (int k = threadIdx.x; k < 2048; k += 2 * kBlock) {
            s0 += p[k];
            s1 += p[k + kBlock];
        }
    }
    const double s = block_sum(s0 + s1, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kBlock)
k_stream_copy(int64_t n2, const dns_double2 *__restrict__ a,
              dns_double2 *__restrict__ b) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride)
        b[i] = a[i];
}

__global__ void __launch_bounds__(kBlock)
k_stream_triad(int64_t n2, const dns_double2 *__restrict__ a,
               const dns_double2 *__restrict__ b, double sc,
               dns_double2 *__restrict__ c) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
         i += stride) {
        const dns_double2 u = a[i], v = b[i];
        dns_double2 r;
        r.x = fma(sc, v.x, u.x);
        r.y = fma(sc, v.y, u.y);
        c[i] = r;
    }
}

// zp = -x (and xacc += zp) unless the guard is raised
__global__ void __launch_bounds__(kBlock)
k_mg_jacobi_guard(int n, const double *__restrict__ x, double *__restrict__ zp,
                  double *__restrict__ xacc, const int *__restrict__ guard) {
    if (*guard) return;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock) {
        const double v = -x[i];
        zp[i] = v;
        if (xacc) xacc[i] += v;
    }
}

// one damped-Jacobi sweep  xout = xin + omega * dinv .* (b - S xin)
// (xin == nullptr: from zero, xout = omega * dinv .* b)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_mg_sweep(int n, const int *__restrict__ rowptr,
           const int *__restrict__ colidx, const double *__restrict__ vals,
           const double *__restrict__ dinv, double omega,
           const double *__restrict__ b, const double *__restrict__ xin,
           double *__restrict__ xout, const int *__restrict__ guard,
           double oscale = 1.0, double *__restrict__ xacc = nullptr,
           int row0 = 0) {
    // oscale, xacc: the last sweep of a V-cycle writes zp = -x itself (and
    // x_p += zp)
    // row0: the CSR arrays hold the n rows that start at global row `row0`
    // (row-partitioned cycle; vectors are indexed by global row)
    if (guard && *guard) return;
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int li = sub; li < n; li += nsub) {
        const int row = row0 + li;
        double s = 0.0;
        if (xin) s = csr_row_dot<LPR>(rowptr, colidx, vals, xin, li, sublane);
        if (sublane == 0) {
            const double v = oscale * ((xin ? xin[row] : 0.0) +
                                       omega * dinv[row] * (b[row] - s));
            xout[row] = v;
            if (xacc) xacc[row] += v;
        }
    }
}

// One operator of the fused V(2,2) cycle (dns_saddle::build_mg_schur):
//   out = A [xa ; xb] (+ omega dinv .* b)
// columns < nsplit read xa, the others xb (xb == nullptr: a plain product)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_mg_split(int nrows, const int *__restrict__ rowptr,
           const int *__restrict__ colidx, const double *__restrict__ vals,
           const double *__restrict__ xa, int nsplit,
           const double *__restrict__ xb, const double *__restrict__ dinv,
           const double *__restrict__ b, double omega,
           double *__restrict__ out, const int *__restrict__ guard,
           int row0 = 0) {
    // row0: as in k_mg_sweep
    if (guard && *guard) return;
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int li = sub; li < nrows; li += nsub) {
        const int row = row0 + li;
        const int k1 = rowptr[li + 1];
        int k = rowptr[li] + sublane;
        // (the row's own addend does not depend on the gather chain)
        const double add =
            (dinv && sublane == 0) ? omega * dinv[row] * b[row] : 0.0;
        double s0 = 0.0, s1 = 0.0;
        for (; k + LPR < k1; k += 2 * LPR) {
            const int c0 = colidx[k], c1 = colidx[k + LPR];
            const double v0 = vals[k], v1 = vals[k + LPR];
            s0 = fma(v0, (xb && c0 >= nsplit) ? xb[c0 - nsplit] : xa[c0], s0);
            s1 = fma(v1, (xb && c1 >= nsplit) ? xb[c1 - nsplit] : xa[c1], s1);
        }
        if (k < k1) {
            const int c = colidx[k];
            s0 = fma(vals[k], (xb && c >= nsplit) ? xb[c - nsplit] : xa[c], s0);
        }
        const double s = subwave_sum<LPR>(s0 + s1);
        if (sublane == 0) out[row] = s + add;
    }
}

__global__ void __launch_bounds__(kBlock)
k_to_f32_rows(int nrows, int ncols, int ldo, const double *__restrict__ in,
              float *__restrict__ out) {
    // rows padded with zeros to `ldo` entries
    const int64_t tot = (int64_t)nrows * ldo;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < tot;
         i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / ldo;
        const int c = (int)(i - r * ldo);
        out[i] = (c < ncols) ? (float)in[r * ncols + c] : 0.0f;
    }
}

__global__ void __launch_bounds__(kBlock)
k_to_f32(int64_t n, const double *__restrict__ in, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kBlock)
        out[i] = (float)in[i];
}

}  // namespace dns
