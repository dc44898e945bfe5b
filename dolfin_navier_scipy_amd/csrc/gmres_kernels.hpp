// Fused Arnoldi kernels of right-preconditioned GMRES (gfx950, fp64).
//
// The reference-size systems (n ~ 1e4) are launch/latency bound: a kernel costs
// 3-4 us whatever it does, so the cycle is cut into as few dependent launches
// as the data dependencies allow.  Per Arnoldi step j:
//
//   A  k_arn_head      close column j-1 (Givens, residual, flags) +
//                      V[j] = w/||w|| + Schur block  zp = -Sh^-1 V[j]_p
//   B  k_spmv_split    zv = Gc [V[j]_v; zp]        (or the Chebyshev kernels)
//   C  k_spmv_multidot w = K z  fused with  h_i = <V_i, w>, i <= j
//   D  k_orth          w -= V h  (+ ||w||^2, or the second-pass dots of CGS2)
//  (E  k_orth          second Gram-Schmidt pass + ||w||^2)
//
// and per cycle: k_resid_norm (r = b - K x, ||r||^2, ||b||^2) in front,
// k_arn_tail (last column, y = R^-1 g, history) and the correction behind:
// x += Z y with the kept Z_j = P^-1 V_j on one GPU (k_basis_combine_acc), or
// k_basis_combine + x += P^-1 (V y) in the row-partitioned solve.  `j` is a plain kernel argument: a cycle
// is captured as ONE hipGraph, every node has its own arguments, and the
// device flag `ctl->done` turns the nodes after convergence into no-ops.
//
// Reductions: producers write one partial per workgroup, consumers sum them
// (fixed order, no atomics); workgroup 0 alone writes the control block.
#pragma once
#include "kernels.hpp"

namespace dns {

// r = b - K x with partials of ||r||^2 and ||b||^2 (one per workgroup)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_resid_norm(int nrows, const int *__restrict__ rowptr,
             const int *__restrict__ colidx, const double *__restrict__ vals,
             const double *__restrict__ x, const double *__restrict__ b,
             double *__restrict__ r, double *__restrict__ part_rr,
             double *__restrict__ part_bb, RowMap rm) {
    // the whole matrix on one GPU; this rank's row slice (local CSR arrays,
    // global columns) in the row-partitioned solve
    __shared__ double red[4];
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    double arr = 0.0, abb = 0.0;
    (void)nrows;
    for (int li = sub; li < rm.len1 + rm.len2; li += nsub) {
        const int row = map_row(rm, li);
        double s = csr_row_dot<LPR>(rowptr, colidx, vals, x, li, sublane);
        if (sublane == 0) {
            const double bv = b[row];
            const double v = bv - s;
            r[row] = v;
            arr = fma(v, v, arr);
            abb = fma(bv, bv, abb);
        }
    }
    arr = block_sum(arr, red);
    abb = block_sum(abb, red);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = arr;
        if (part_bb) part_bb[blockIdx.x] = abb;
    }
}

// out[s] = sum of the partials of scalar s (one workgroup): what is all-reduced
// across ranks in the row-partitioned solve
__global__ void __launch_bounds__(kBlock)
k_sum_partials_n(const double *__restrict__ part, int nparts, int nscal,
                 double *__restrict__ out) {
    __shared__ double sums[kMaxRestart + 1];
    reduce_partials(part, nparts, nparts, nscal, sums);
    if (threadIdx.x < nscal) out[threadIdx.x] = sums[threadIdx.x];
}

// the residual norm at which a solve raises `done` (oversolve: below `tol`)
__device__ __forceinline__ double stop_tol(const DnsCtl *ctl) {
    return ctl->stop_frac > 0.0 ? ctl->tol * ctl->stop_frac : ctl->tol;
}

// Givens update of column j (thread 0 of workgroup 0 only)
__device__ inline void givens_close(DnsCtl *ctl, int j, double hn,
                                    int maxiter) {
    double *Rc = ctl->R + (size_t)j * (kMaxRestart + 1);
    for (int i = 0; i <= j; ++i) Rc[i] = ctl->hcol[i];
    for (int i = 0; i < j; ++i) {
        const double t = ctl->cs[i] * Rc[i] + ctl->sn[i] * Rc[i + 1];
        Rc[i + 1] = -ctl->sn[i] * Rc[i] + ctl->cs[i] * Rc[i + 1];
        Rc[i] = t;
    }
    const double den = hypot(Rc[j], hn);
    int status = DNS_OK;
    double c = 1.0, s = 0.0;
    if (den > 0.0) {
        c = Rc[j] / den;
        s = hn / den;
    } else {
        status = DNS_BREAKDOWN;
    }
    ctl->cs[j] = c;
    ctl->sn[j] = s;
    Rc[j] = den;
    ctl->g[j + 1] = -s * ctl->g[j];
    ctl->g[j] = c * ctl->g[j];
    const double res = fabs(ctl->g[j + 1]);
    ctl->resnorm = res;
    ctl->hist[j + 1] = res;
    ctl->jdone = j + 1;
    ctl->total_it += 1;
    // (a NaN is never "converged": it ends the solve as a breakdown)
    const bool conv = res <= ctl->tol;
    if (isnan(res) && status == DNS_OK) status = DNS_BREAKDOWN;
    if (status != DNS_OK) ctl->status = status;
    if (conv && !ctl->conv) ctl->need_it = ctl->total_it;
    if (conv) ctl->conv = 1;
    // (oversolve: the cycle goes on below the tolerance, see DnsCtl)
    const bool stop = res <= stop_tol(ctl);
    if (stop || status != DNS_OK || !(hn > 0.0) || isnan(res) ||
        ctl->total_it >= maxiter)
        ctl->done = 1;
}

// head of Arnoldi step j.  `src` is the un-normalised new basis vector (the
// residual r for j == 0, else w of step j-1) with its ||.||^2 partials.
//   SK 1/2: zp = -Sinv * src_p / ||src||  (Sinv dense fp64 / fp32)
//   SK 0  : zp = -sdinv .* src_p / ||src||
//   SK 3  : no Schur block here (multigrid V-cycle launched behind the head)
template <int SK>
__global__ void __launch_bounds__(kBlock)
k_arn_head(int n, int nv, int np, int j, const double *__restrict__ src,
           const double *__restrict__ src_part, int src_nparts,
           double *__restrict__ V, size_t ld, const void *__restrict__ sinv,
           double *__restrict__ zp, DnsCtl *ctl, double rtol, double atol,
           const double *__restrict__ bb_part, int bb_nparts, int maxiter,
           int prow0, int prow1, int first,
           const double *__restrict__ sp_in, int *stepctr = nullptr,
           int sld = 0, int scol0 = 0, int sncols = -1) {
    // sld: row stride of the dense inverse (SK == 2: a multiple of four, rows
    // padded with zeros and 16-byte aligned) -- 0: rows of np entries
    // [scol0, scol0 + sncols): the COLUMNS of the inverse this rank holds
    // (row-partitioned solve: every rank forms its share of all rows of
    // zp = -Sinv tau from its own entries of tau; the shares are all-reduced);
    // sncols < 0: all columns
    // stepctr (first cycle of a time step's solve only): the device step
    // counter behind the rhs / boundary-value tables; every kernel of the step
    // that reads a table row has run before this one
    // sp_in (full block factorisation): tau(src) = src_p - J Fh^-1 src_v
    // replaces src_p as the input of the Schur block
    // `first` (only with j == 0): 1 = first cycle of a solve -- the counters of
    // the previous solve still sit in the control block and are reset here;
    // 2 = the solve goes on after a Gram-Schmidt fallback: only the status is
    // cleared
    // [prow0, prow1): the Schur rows this rank computes (all of them on one
    // GPU); normalisation and the control block are done by every rank alike
    if (j > 0 && ctl->done) return;
    __shared__ double sc[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gwave = blockIdx.x * (kBlock / 64) + wave;
    const int nwaves = gridDim.x * (kBlock / 64);
    const double *sp = (sp_in ? sp_in : src + nv) + scol0;
    const int ncl = sncols >= 0 ? sncols : np;
    const size_t rst = sld ? (size_t)sld : (size_t)np;
    // dense Schur rows: one wave per row; the first row's loads are issued
    // BEFORE the reduction of the norm partials (they do not depend on it)
    double s_first = 0.0;
    const int row_first = prow0 + gwave;
    if ((SK == 1 || SK == 2) && row_first < prow1) {
        if (SK == 1)
            s_first = dense_row_wave<double>(
                (const double *)sinv + row_first * rst, sp, ncl, lane);
        else if (sld)
            s_first = dense_row_wave_f4(
                (const float *)sinv + row_first * rst, sp, ncl, lane);
        else
            s_first = dense_row_wave<float>(
                (const float *)sinv + row_first * rst, sp, ncl, lane);
    }
    // (likewise this thread's first entry of the vector to be normalised)
    const int ef = blockIdx.x * kBlock + threadIdx.x;
    const double src_first = (ef < n) ? src[ef] : 0.0;
    reduce_partials(src_part, src_nparts, src_nparts, 1, sc);
    if (j == 0) reduce_partials(bb_part, bb_nparts, bb_nparts, 1, sc + 1);
    const double hn = sqrt(sc[0]);
    bool stop = false;
    double tol = 0.0, bn = 0.0;
    if (j == 0) {
        bn = sqrt(sc[1]);
        tol = fmax(rtol * bn, atol);
        const double stol = ctl->stop_frac > 0.0 ? tol * ctl->stop_frac : tol;
        stop = !(hn > stol) || isnan(hn) ||
               (first != 1 && ctl->total_it >= maxiter) ||
               (first == 0 && ctl->status != DNS_OK);
    }
    if (!stop && hn > 0.0) {
        const double scale = 1.0 / hn;
        double *vj = V + (size_t)j * ld;
        if (ef < n) vj[ef] = src_first * scale;
        for (int e = ef + gridDim.x * kBlock; e < n; e += gridDim.x * kBlock)
            vj[e] = src[e] * scale;
        if (SK == 1 || SK == 2) {
            if (row_first < prow1 && lane == 0)
                zp[row_first] = -scale * s_first;
            for (int row = row_first + nwaves; row < prow1; row += nwaves) {
                const double s =
                    (SK == 1)
                        ? dense_row_wave<double>(
                              (const double *)sinv + row * rst, sp, ncl, lane)
                        : (sld ? dense_row_wave_f4(
                                     (const float *)sinv + row * rst, sp, ncl,
                                     lane)
                               : dense_row_wave<float>(
                                     (const float *)sinv + row * rst, sp, ncl,
                                     lane));
                if (lane == 0) zp[row] = -scale * s;
            }
        } else if (SK == 0) {
            const double *sd = (const double *)sinv;
            for (int i = prow0 + blockIdx.x * kBlock + threadIdx.x; i < prow1;
                 i += gridDim.x * kBlock)
                zp[i] = -sd[i] * sp[i] * scale;     // (scol0 == 0 here)
        }   // SK == 3: the Schur block is a multi-kernel V-cycle behind this
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (j == 0) {
            if (first == 1) {
                ctl->total_it = 0;
                ctl->hist_len = 0;
                ctl->conv = 0;
                if (stepctr) *stepctr += 1;
            }
            if (first) ctl->status = DNS_OK;
            ctl->predone = 0;
            ctl->jdone = 0;
            ctl->zero = 0;
            ctl->beta = hn;
            ctl->resnorm = hn;
            ctl->bnorm = bn;
            ctl->tol = tol;
            ctl->g[0] = hn;
            ctl->hist[0] = hn;
            if (hn <= tol) ctl->conv = 1;
            ctl->need_it = (hn <= tol) ? 0 : -1;
            if (isnan(hn) || isnan(tol)) ctl->status = DNS_BREAKDOWN;
            ctl->done = stop ? 1 : 0;
        } else {
            givens_close(ctl, j - 1, hn, maxiter);
        }
    }
}

// w = K z fused with the Gram-Schmidt dots  part[i*nparts + wg] = <V_i, w>
// (i <= j).  After the row sum is known in all LPR lanes of the sub-wave,
// lane l accumulates the dots with V_i, i = l (mod LPR).
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_spmv_multidot(int nrows, const int *__restrict__ rowptr,
                const int *__restrict__ colidx,
                const double *__restrict__ vals, const double *__restrict__ z,
                double *__restrict__ w, const double *__restrict__ V, size_t ld,
                int j, double *__restrict__ part, int nparts,
                const DnsCtl *ctl, RowMap rm, int with_ww) {
    // with_ww: one more scalar, part[(j+1)*nparts + wg] = <w, w> (the norm of
    // the orthogonalised vector then follows from Pythagoras, k_arn_head_f)
    (void)nrows;
    constexpr int NQ = (kMaxRestart + 1 + LPR) / LPR;   // dots per lane
    __shared__ double wred[kBlock / 64][NQ * LPR];
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    const int nvec = j + 1;
    // the first row's pointers are asked for BEFORE the guard is looked at
    // (guard-first the kernel starts with a chain of dependent scalar loads in
    // front of its first vector load)
    int kf0 = 0, kf1 = 0;
    if (sub < rm.len1 + rm.len2) {
        kf0 = rowptr[sub];
        kf1 = rowptr[sub + 1];
    }
    const int gdone = ctl->done;
    if (gdone) return;
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
    for (int li = sub; li < rm.len1 + rm.len2; li += nsub) {
        const int row = map_row(rm, li);
        // basis entries first: their loads overlap the row's own load chain
        double vq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = q * LPR + sublane;
            vq[q] = (i < nvec) ? V[(size_t)i * ld + row] : 0.0;
        }
        const int k0 = (li == sub) ? kf0 : rowptr[li];
        const int k1 = (li == sub) ? kf1 : rowptr[li + 1];
        const double s = csr_row_dot_k<LPR>(colidx, vals, z, k0, k1, sublane);
        if (sublane == 0) w[row] = s;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double vv =
                (with_ww && q * LPR + sublane == nvec) ? s : vq[q];
            acc[q] = fma(vv, s, acc[q]);
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double v = acc[q];
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        if (lane < LPR) wred[wave][q * LPR + lane] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nvec + (with_ww ? 1 : 0); i += kBlock) {
        double s = 0.0;
        for (int ww = 0; ww < kBlock / 64; ++ww) s += wred[ww][i];
        part[(size_t)i * nparts + blockIdx.x] = s;
    }
}

// h_{j,j-1} of classical Gram-Schmidt without forming w - V h first:
// ||w - V h||^2 = <w,w> - sum h_i^2 (V orthonormal).  hs = [h_0..h_{j-1}, <w,w>]
// Cancellation: the relative error of the result is ~ eps <w,w> / (2 d), so
// below d = 1e-8 <w,w> (error 1e-8) the value is not trusted (returns -1) and
// the solve falls back to the explicit Gram-Schmidt kernel (kGsFallback); a
// basis that has lost its orthogonality shows up the same way (d <= 0).
__device__ __forceinline__ double pythagoras_norm(const double *hs, int j) {
    double ss = 0.0;
    for (int i = 0; i < j; ++i) ss = fma(hs[i], hs[i], ss);
    const double d = hs[j] - ss;
    if (!(d > 1e-8 * hs[j])) return -1.0;
    return sqrt(d);
}

// head of Arnoldi step j >= 1 with the Gram-Schmidt update folded in ("fused"
// orthogonalisation, one launch less per step): `w` is the RAW K z_{j-1},
// `hpart` holds the partials of h_i = <V_i, w> (i < j) and of <w, w>.
//   V_j  = (w - sum h_i V_i) / hn
//   zp_j = -Sinv V_j,p = (-Sinv w_p - sum h_i zp_i) / hn    (zp_i kept in Z_i)
template <int SK>
__global__ void __launch_bounds__(kBlock)
k_arn_head_f(int n, int nv, int np, int j, const double *__restrict__ w,
             const double *__restrict__ hpart, int hnparts,
             double *__restrict__ V, size_t ld, double *__restrict__ Z,
             const void *__restrict__ sinv, DnsCtl *ctl, int maxiter,
             const double *__restrict__ sp_in, int prow0 = 0, int prow1 = -1,
             int sld = 0, int scol0 = 0, int sncols = -1, int add_corr = 1) {
    // sld, scol0, sncols: as in k_arn_head; add_corr: this rank adds the
    // Gram-Schmidt correction - sum h_i zp_i to its share (exactly one rank
    // does when the shares are all-reduced)
    // sp_in: tau(w) = w_p - J Fh^-1 w_v (full block factorisation); the
    // linearity argument below holds for tau as it does for the pressure part
    // [prow0, prow1): the Schur rows this rank computes (all on one GPU)
    if (prow1 < 0) prow1 = np;
    if (ctl->done) return;
    __shared__ double h[kMaxRestart + 2];
    // the tau kernel in front may have found that column j-1 converges: then
    // V_j and zp_j are never used and only the column is closed
    const bool pre = ctl->predone != 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gwave = blockIdx.x * (kBlock / 64) + wave;
    const int nwaves = gridDim.x * (kBlock / 64);
    const double *wp = (sp_in ? sp_in : w + nv) + scol0;
    const int ncl = sncols >= 0 ? sncols : np;
    const size_t rst = sld ? (size_t)sld : (size_t)np;
    // dense Schur rows by waves; the first row's loads go out before the
    // reduction of the Gram-Schmidt partials
    double s_first = 0.0;
    const int row_first = prow0 + gwave;
    if ((SK == 1 || SK == 2) && row_first < prow1 && !pre) {
        if (SK == 1)
            s_first = dense_row_wave<double>(
                (const double *)sinv + row_first * rst, wp, ncl, lane);
        else if (sld)
            s_first = dense_row_wave_f4(
                (const float *)sinv + row_first * rst, wp, ncl, lane);
        else
            s_first = dense_row_wave<float>(
                (const float *)sinv + row_first * rst, wp, ncl, lane);
    }
    reduce_partials(hpart, hnparts, hnparts, j + 1, h);
    const double hn = pythagoras_norm(h, j);
    if (hn < 0.0) {
        // column j-1 stays open: the cycle ends with the columns before it
        if (blockIdx.x == 0 && threadIdx.x == 0) {
#ifdef DNS_TRACE_GS
            double ss = 0.0;
            for (int i = 0; i < j; ++i) ss += h[i] * h[i];
            printf("[gs] j=%d FALLBACK d/ww=%.3e\n", j, (h[j] - ss) / h[j]);
#endif
            ctl->status = kGsFallback;
            ctl->done = 1;
        }
        return;
    }
    if (hn > 0.0 && !pre) {
        const double scale = 1.0 / hn;
        double *vj = V + (size_t)j * ld;
        for (int e = blockIdx.x * kBlock + threadIdx.x; e < n;
             e += gridDim.x * kBlock) {
            double we = w[e];
            for (int i = 0; i < j; ++i)
                we = fma(-h[i], V[(size_t)i * ld + e], we);
            vj[e] = we * scale;
        }
        double *zp = Z + (size_t)j * ld + nv;
        if (SK == 3) {
            // multigrid Schur block: applied to V_j,p by the kernels behind
        } else if (SK == 0) {
            // zp_j = (-sd .* wp - sum h_i zp_i) / hn  (zp_i kept in Z_i)
            const double *sd = (const double *)sinv;
            for (int r = prow0 + blockIdx.x * kBlock + threadIdx.x; r < prow1;
                 r += gridDim.x * kBlock) {
                double acc = -sd[r] * wp[r];
                for (int i = 0; i < j; ++i)
                    acc = fma(-h[i], Z[(size_t)i * ld + nv + r], acc);
                zp[r] = acc * scale;
            }
        } else {
            for (int row = row_first; row < prow1; row += nwaves) {
                double s = s_first;
                if (row != row_first)
                    s = (SK == 1)
                            ? dense_row_wave<double>(
                                  (const double *)sinv + row * rst, wp, ncl,
                                  lane)
                            : (sld ? dense_row_wave_f4(
                                         (const float *)sinv + row * rst, wp,
                                         ncl, lane)
                                   : dense_row_wave<float>(
                                         (const float *)sinv + row * rst, wp,
                                         ncl, lane));
                if (lane == 0) {
                    double corr = 0.0;
                    if (add_corr)
                        for (int i = 0; i < j; ++i)
                            corr = fma(h[i], Z[(size_t)i * ld + nv + row],
                                       corr);
                    zp[row] = (-s - corr) * scale;
                }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int i = 0; i < j; ++i) ctl->hcol[i] = h[i];
        givens_close(ctl, j - 1, hn, maxiter);
        if (pre) ctl->done = 1;      // (V_j was not formed)
#ifdef DNS_TRACE_GS
        printf("[gs] j=%d hn/|w|=%.3e res=%.3e\n", j, hn / sqrt(h[j]),
               ctl->resnorm);
#endif
    }
}

// tau = src_p - (J Fh^-1) src_v, the Schur input of the full block
// factorisation.  With j >= 1 (fused Gram-Schmidt) every workgroup first
// repeats, read-only, the arithmetic with which the next head kernel will close
// column j-1; if that column converges nobody needs tau (nor V_j): the verdict
// goes to ctl->predone and the rows are skipped.
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_tau_guard(int np, int nv, const int *__restrict__ rowptr,
            const int *__restrict__ colidx, const double *__restrict__ vals,
            const double *__restrict__ src, double *__restrict__ tau,
            const double *__restrict__ hpart, int hnparts, int j, DnsCtl *ctl,
            int row0, int row1) {
    // rows [row0, row1): all of them on one GPU, this rank's Schur rows in
    // the row-partitioned solve.  j < 0: no verdict, but skip a finished solve
    (void)np;
    if (j < 0 && ctl->done) return;
    if (j > 0) {
        if (ctl->done) return;
        __shared__ double h[kMaxRestart + 2];
        __shared__ int verdict;
        reduce_partials(hpart, hnparts, hnparts, j + 1, h);
        if (threadIdx.x == 0) {
            int v = 0;
            const double hn = pythagoras_norm(h, j);
            if (hn >= 0.0) {
                // column j-1 under the rotations 0..j-2 (as givens_close)
                double lo = h[0];
                for (int i = 0; i + 1 < j; ++i)
                    lo = -ctl->sn[i] * lo + ctl->cs[i] * h[i + 1];
                const double den = hypot(lo, hn);
                const double sn = den > 0.0 ? hn / den : 0.0;
                v = fabs(sn * ctl->g[j - 1]) <= stop_tol(ctl);
            }
            verdict = v;
            if (blockIdx.x == 0) ctl->predone = v;
        }
        __syncthreads();
        if (verdict) return;
    }
    if (LPR == 64) {
        // long rows (J Fh^-1: ~270 entries): 128 lanes per row, so that a row
        // is two rounds of loads instead of four or five dependent ones
        __shared__ double half[kBlock / 64];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int pair = threadIdx.x >> 7, l128 = threadIdx.x & 127;
        for (int base = row0 + blockIdx.x * 2; base < row1;
             base += gridDim.x * 2) {
            const int row = base + pair;
            double s0 = 0.0, s1 = 0.0;
            if (row < row1) {
                const int k1 = rowptr[row + 1];
                int k = rowptr[row] + l128;
                for (; k + 128 < k1; k += 256) {
                    const int c0 = colidx[k], c1 = colidx[k + 128];
                    const double v0 = vals[k], v1 = vals[k + 128];
                    s0 = fma(v0, src[c0], s0);
                    s1 = fma(v1, src[c1], s1);
                }
                if (k < k1) s0 = fma(vals[k], src[colidx[k]], s0);
            }
            const double s = wave_sum(s0 + s1);
            __syncthreads();
            if (lane == 0) half[wave] = s;
            __syncthreads();
            if (l128 == 0 && row < row1)
                tau[row] = src[nv + row] - (half[2 * pair] + half[2 * pair + 1]);
        }
        return;
    }
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int row = row0 + sub; row < row1; row += nsub) {
        const double s = csr_row_dot<LPR>(rowptr, colidx, vals, src, row,
                                          sublane);
        if (sublane == 0) tau[row] = src[nv + row] - s;
    }
}

// plain w = K z (guarded), used when the dots run as their own kernel
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_spmv_guard(int nrows, const int *__restrict__ rowptr,
             const int *__restrict__ colidx, const double *__restrict__ vals,
             const double *__restrict__ x, double *__restrict__ w,
             const DnsCtl *ctl, int row0, int row1) {
    if (ctl->done) return;
    (void)nrows;
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int row = row0 + sub; row < row1; row += nsub) {
        double s = csr_row_dot<LPR>(rowptr, colidx, vals, x, row, sublane);
        if (sublane == 0) w[row] = s;
    }
}

// part[i*nparts + wg] = <V_i, w>, i <= j  (streaming form for large n)
__global__ void __launch_bounds__(kBlock)
k_multidot(int n, const double *__restrict__ V, size_t ld,
           const double *__restrict__ w, double *__restrict__ part, int nparts,
           int j, const DnsCtl *ctl) {
    if (ctl->done) return;
    const int nvec = j + 1;
    __shared__ double wred[kBlock / 64][8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < nvec; i0 += 8) {
        double acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.0;
        for (int e = blockIdx.x * kBlock + threadIdx.x; e < n;
             e += gridDim.x * kBlock) {
            const double we = w[e];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i0 + q < nvec)
                    acc[q] = fma(V[(size_t)(i0 + q) * ld + e], we, acc[q]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = wave_sum(acc[q]);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) wred[wave][q] = acc[q];
        }
        __syncthreads();
        if (threadIdx.x < 8 && i0 + threadIdx.x < nvec) {
            double s = 0.0;
            for (int ww = 0; ww < kBlock / 64; ++ww) s += wred[ww][threadIdx.x];
            part[(size_t)(i0 + threadIdx.x) * nparts + blockIdx.x] = s;
        }
    }
}

// Gram-Schmidt update  w -= sum_i h_i V_i,  h_i = sum of `part_in`.
//   pass 0: hcol = h      pass 1: hcol += h   (workgroup 0)
//   MODE 0: partial of ||w||^2 -> out[wg]
//   MODE 1: second-pass dots   -> out[i*out_nparts + wg] = <V_i, w_new>
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_orth(int n, const double *__restrict__ V, size_t ld, double *__restrict__ w,
       const double *__restrict__ part_in, int in_nparts, int j, int pass,
       double *__restrict__ out, int out_nparts, DnsCtl *ctl,
       RowMap rm = RowMap{0, -1, 0, 0}) {
    // rm (MODE 0 only): the entries to update and to take the norm over -- all
    // of them on one GPU (len1 < 0), this rank's own rows when partitioned
    if (ctl->done) return;
    if (rm.len1 < 0) rm = RowMap{0, n, 0, 0};
    const int nvec = j + 1;
    __shared__ double h[kMaxRestart + 1];
    __shared__ double red[4];
    __shared__ double wred[kBlock / 64][8];
    reduce_partials(part_in, in_nparts, in_nparts, nvec, h);
    if (blockIdx.x == 0 && threadIdx.x < nvec)
        ctl->hcol[threadIdx.x] =
            (pass == 0 ? 0.0 : ctl->hcol[threadIdx.x]) + h[threadIdx.x];
    double acc = 0.0;
    for (int li = blockIdx.x * kBlock + threadIdx.x; li < rm.len1 + rm.len2;
         li += gridDim.x * kBlock) {
        const int e = map_row(rm, li);
        double we = w[e];
        for (int i = 0; i < nvec; ++i)
            we = fma(-h[i], V[(size_t)i * ld + e], we);
        w[e] = we;
        acc = fma(we, we, acc);
    }
    if (MODE == 0) {
        acc = block_sum(acc, red);
        if (threadIdx.x == 0) out[blockIdx.x] = acc;
        return;
    }
    // MODE 1: dots of the updated w with every V_i (own elements re-read)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < nvec; i0 += 8) {
        double a8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] = 0.0;
        for (int e = blockIdx.x * kBlock + threadIdx.x; e < n;
             e += gridDim.x * kBlock) {
            const double we = w[e];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i0 + q < nvec)
                    a8[q] = fma(V[(size_t)(i0 + q) * ld + e], we, a8[q]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) a8[q] = wave_sum(a8[q]);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) wred[wave][q] = a8[q];
        }
        __syncthreads();
        if (threadIdx.x < 8 && i0 + threadIdx.x < nvec) {
            double s = 0.0;
            for (int ww = 0; ww < kBlock / 64; ++ww) s += wred[ww][threadIdx.x];
            out[(size_t)(i0 + threadIdx.x) * out_nparts + blockIdx.x] = s;
        }
    }
}

// batch accumulators of the oversolve policy: columns the solve needed to meet
// the tolerance, residual / tolerance in front of its last column
__device__ __forceinline__ void acc_need_prev(DnsCtl *ctl, int need,
                                              double prev_res) {
    if (need > ctl->acc_maxneed) ctl->acc_maxneed = need;
    ctl->acc_sumneed += need;
    if (ctl->tol > 0.0) {
        const double rel = prev_res / ctl->tol;
        if (rel > ctl->acc_maxprev) ctl->acc_maxprev = rel;
    }
}

// tail of a cycle of `c` steps (ONE workgroup): close the last column unless
// the cycle already stopped, y = R^-1 g, append the cycle's residual norms to
// the solve's history
__global__ void __launch_bounds__(kBlock)
k_arn_tail(int c, const double *__restrict__ norm_part, int nparts,
           DnsCtl *ctl, double *__restrict__ histbuf, int hist_cap,
           int maxiter, int fused) {
    // fused: `norm_part` holds the partials of h_0..h_{c-1} and <w, w> of the
    // last step (no Gram-Schmidt kernel ran), else those of ||w||^2
    __shared__ double sc[kMaxRestart + 2];
    // (the cycle usually stopped earlier: then there is nothing to reduce)
    const bool open_col = !ctl->done && c > 0;
    if (open_col)
        reduce_partials(norm_part, nparts, nparts, fused ? c + 1 : 1, sc);
    if (threadIdx.x != 0) return;
    if (open_col) {
        double hn;
        if (fused) {
            hn = pythagoras_norm(sc, c);
            for (int i = 0; i < c; ++i) ctl->hcol[i] = sc[i];
        } else {
            hn = sqrt(sc[0]);
        }
        if (hn < 0.0) {
            ctl->status = kGsFallback;
            ctl->done = 1;
        } else {
            givens_close(ctl, c - 1, hn, maxiter);
        }
    }
    const int jcols = ctl->jdone;
    for (int i = jcols - 1; i >= 0; --i) {
        double s = ctl->g[i];
        for (int k = i + 1; k < jcols; ++k)
            s -= ctl->R[(size_t)k * (kMaxRestart + 1) + i] * ctl->y[k];
        const double d = ctl->R[(size_t)i * (kMaxRestart + 1) + i];
        ctl->y[i] = (d != 0.0) ? s / d : 0.0;
    }
    int hl = ctl->hist_len;
    for (int i = (hl > 0 ? 1 : 0); i <= jcols && hl < hist_cap; ++i)
        histbuf[hl++] = ctl->hist[i];
    ctl->hist_len = hl;
    // bookkeeping for pipelined time stepping (one cycle per solve there; the
    // host looks at these once per batch of steps instead of once per step)
    ctl->acc_solves += 1;
    ctl->acc_iters += ctl->total_it;
    if (ctl->total_it > ctl->acc_maxit) ctl->acc_maxit = ctl->total_it;
    if (ctl->conv && ctl->tol > 0.0) {
        const double rel = ctl->resnorm / ctl->tol;
        if (rel > ctl->acc_maxrel) ctl->acc_maxrel = rel;
    }
    if (!ctl->conv) ctl->acc_fail += 1;
    acc_need_prev(ctl, ctl->conv ? ctl->need_it : ctl->total_it + 1,
                  jcols > 0 ? ctl->hist[jcols - 1] : ctl->resnorm);
}

// Tail of a cycle AND the correction x += Z y in ONE launch (one GPU): every
// workgroup repeats the tiny tail computation -- close the last column if it
// is still open, y = R^-1 g -- in its own LDS from read-only device state, then
// updates its share of x.  Workgroup 0 alone commits the solve's bookkeeping
// (residual, flags, history, batch accumulators); it writes NONE of the
// fields the other workgroups read in this launch (done, jdone, R, g, cs, sn),
// which nobody needs after the cycle's end: the next cycle's head resets them.
__global__ void __launch_bounds__(kBlock)
k_arn_tail_acc(int c, int n, const double *__restrict__ norm_part, int nparts,
               DnsCtl *ctl, double *__restrict__ histbuf, int hist_cap,
               int maxiter, const double *__restrict__ Z, size_t ld,
               double *__restrict__ x, TailExtrap te) {
    __shared__ double sc[kMaxRestart + 2];
    __shared__ double yl[kMaxRestart];
    __shared__ int jl;
    const bool open_col = !ctl->done && c > 0;
    // this thread's first element: everything it needs from memory is asked
    // for BEFORE the reduction and the serial close of the column (none of it
    // depends on them) -- with n <= gridDim.x * kBlock that is all there is
    const int ef = blockIdx.x * kBlock + threadIdx.x;
    double pfx = 0.0, pz0 = 0.0, pz1 = 0.0, ph1 = 0.0, ph2 = 0.0, ph3 = 0.0,
           ph4 = 0.0;
    if (ef < n) {
        pfx = x[ef];
        if (c > 0) pz0 = Z[ef];
        if (c > 1) pz1 = Z[ld + ef];
        if (te.out) {
            if (te.e1 != 0.0) ph1 = te.h1[ef];
            if (te.e2 != 0.0) ph2 = te.h2[ef];
            if (te.e3 != 0.0) ph3 = te.h3[ef];
            if (te.e4 != 0.0) ph4 = te.h4[ef];
        }
    }
    if (open_col) reduce_partials(norm_part, nparts, nparts, c + 1, sc);
    if (threadIdx.x == 0) {
        // local copies of what closing column c-1 changes
        // (the column being closed lives in LDS: as a local array indexed
        // at run time it went to scratch memory, 528 bytes per lane of every
        // wave of the launch for one thread's use)
        __shared__ double rcol[kMaxRestart + 1];
        double gl[2] = {0.0, 0.0};
        int jcols = ctl->jdone, status = DNS_OK, conv = 0, tot = ctl->total_it;
        double res = ctl->resnorm;
        bool closed = false;
        if (open_col) {
            const int j = c - 1;
            const double hn = pythagoras_norm(sc, c);
            if (hn < 0.0) {
                status = kGsFallback;
            } else {
                for (int i = 0; i <= j; ++i) rcol[i] = sc[i];
                for (int i = 0; i < j; ++i) {
                    const double t = ctl->cs[i] * rcol[i] +
                                     ctl->sn[i] * rcol[i + 1];
                    rcol[i + 1] = -ctl->sn[i] * rcol[i] +
                                  ctl->cs[i] * rcol[i + 1];
                    rcol[i] = t;
                }
                const double den = hypot(rcol[j], hn);
                double cc = 1.0, ss = 0.0;
                if (den > 0.0) {
                    cc = rcol[j] / den;
                    ss = hn / den;
                } else {
                    status = DNS_BREAKDOWN;
                }
                rcol[j] = den;
                gl[1] = -ss * ctl->g[j];
                gl[0] = cc * ctl->g[j];
                res = fabs(gl[1]);
                jcols = j + 1;
                tot += 1;
                conv = res <= ctl->tol;
                if (isnan(res)) status = DNS_BREAKDOWN;
                closed = true;
            }
        }
        // y = R^-1 g with the (possibly) locally closed last column
        for (int i = jcols - 1; i >= 0; --i) {
            double s = (closed && i == jcols - 1) ? gl[0] : ctl->g[i];
            for (int k = i + 1; k < jcols; ++k) {
                const double rik = (closed && k == jcols - 1)
                                       ? rcol[i]
                                       : ctl->R[(size_t)k * (kMaxRestart + 1) + i];
                s -= rik * yl[k];
            }
            const double d = (closed && i == jcols - 1)
                                 ? rcol[i]
                                 : ctl->R[(size_t)i * (kMaxRestart + 1) + i];
            yl[i] = (d != 0.0) ? s / d : 0.0;
        }
        jl = jcols;
        if (blockIdx.x == 0) {
            // commit (fields nobody else reads in this launch)
            const int was_conv = ctl->conv;
            if (closed) {
                ctl->resnorm = res;
                ctl->hist[jcols] = res;
                ctl->total_it = tot;
                if (status != DNS_OK) ctl->status = status;
                if (conv) ctl->conv = 1;
            } else if (status != DNS_OK) {
                ctl->status = status;
            }
            const int nowconv = was_conv || conv;
            int hl = ctl->hist_len;
            for (int i = (hl > 0 ? 1 : 0); i <= jcols && hl < hist_cap; ++i)
                histbuf[hl++] = (closed && i == jcols) ? res : ctl->hist[i];
            ctl->hist_len = hl;
            ctl->acc_solves += 1;
            ctl->acc_iters += tot;
            if (tot > ctl->acc_maxit) ctl->acc_maxit = tot;
            if (nowconv && ctl->tol > 0.0) {
                const double rel = res / ctl->tol;
                if (rel > ctl->acc_maxrel) ctl->acc_maxrel = rel;
            }
            if (!nowconv) ctl->acc_fail += 1;
            if (conv && !was_conv) ctl->need_it = tot;
            acc_need_prev(ctl, nowconv ? ctl->need_it : tot + 1,
                          jcols > 0 ? ctl->hist[jcols - 1] : res);
        }
    }
    __syncthreads();
    const int jcols = jl;
    if (jcols == 0 && !te.out) return;
    if (ef < n) {
        double s = pfx;
        if (jcols > 0) s = fma(yl[0], pz0, s);
        if (jcols > 1) s = fma(yl[1], pz1, s);
        for (int i = 2; i < jcols; ++i)
            s = fma(yl[i], Z[(size_t)i * ld + ef], s);
        if (jcols > 0) x[ef] = s;
        if (te.out) {
            double v = te.e0 * s;
            v = fma(te.e1, ph1, v);
            v = fma(te.e2, ph2, v);
            v = fma(te.e3, ph3, v);
            v = fma(te.e4, ph4, v);
            te.out[ef] = v;
        }
    }
    for (int e = ef + gridDim.x * kBlock; e < n; e += gridDim.x * kBlock) {
        double s = x[e];
        for (int i = 0; i < jcols; ++i)
            s = fma(yl[i], Z[(size_t)i * ld + e], s);
        if (jcols > 0) x[e] = s;
        if (te.out) tail_extrapolate(te, e, s);
    }
}

// ---- one-step cycle of the row-partitioned solve without the all-reduce of
// the residual norm.  The first basis vector stays UN-normalised (V_0 = r),
// z = P^-1 r, w = K z, and the four sums <w, r>, <w, w>, <r, r>, <b, b> are
// all-reduced TOGETHER behind the step: one collective per time step less.
// In the scaled basis v = r / rho (rho = ||r||) this is the Arnoldi step the
// other kernels take:  h = <w, r> / rho^2,  hn^2 = <w, w> / rho^2 - h^2,
// x += (<w, r> / <w, w>) z,  ||r_new|| = rho sqrt(1 - <w, r>^2 / (rr ww)).
//
// out[0..1] = sums of the first two scalars of `parta` (nparta partials each),
// out[2] = sum of partr, out[3] = sum of partb (nparts each): ONE workgroup
__global__ void __launch_bounds__(kBlock)
k_sum_lazy4(const double *__restrict__ parta, int nparta,
            const double *__restrict__ partr, const double *__restrict__ partb,
            int nparts, double *__restrict__ out) {
    // all four sums side by side: independent loads, ONE exchange through LDS
    __shared__ double red[4][kBlock / 64];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = threadIdx.x; i < nparta; i += kBlock) {
        s0 += parta[i];
        s1 += parta[(size_t)nparta + i];
    }
    for (int i = threadIdx.x; i < nparts; i += kBlock) {
        s2 += partr[i];
        s3 += partb[i];
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[0][wave] = s0;
        red[1][wave] = s1;
        red[2][wave] = s2;
        red[3][wave] = s3;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += red[threadIdx.x][w];
        out[threadIdx.x] = t;
    }
}

// u = sum_{i<jdone} y_i V_i
__global__ void __launch_bounds__(kBlock)
k_basis_combine(int n, const double *__restrict__ V, size_t ld,
                const DnsCtl *ctl, double *__restrict__ u) {
    __shared__ double y[kMaxRestart];
    const int jcols = ctl->jdone;
    if (threadIdx.x < jcols) y[threadIdx.x] = ctl->y[threadIdx.x];
    __syncthreads();
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < n;
         e += gridDim.x * kBlock) {
        double s = 0.0;
        for (int i = 0; i < jcols; ++i)
            s = fma(y[i], V[(size_t)i * ld + e], s);
        u[e] = s;
    }
}

// x += sum_{i<jdone} y_i Z_i  with Z_i = P^-1 V_i kept from the Arnoldi steps
// (the "flexible" form of the correction: no preconditioner apply behind the
// cycle)
__global__ void __launch_bounds__(kBlock)
k_basis_combine_acc(int n, const double *__restrict__ Z, size_t ld,
                    const DnsCtl *ctl, double *__restrict__ x, TailExtrap te) {
    __shared__ double y[kMaxRestart];
    const int jcols = ctl->jdone;
    if (jcols == 0 && !te.out) return;
    if (threadIdx.x < jcols) y[threadIdx.x] = ctl->y[threadIdx.x];
    __syncthreads();
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < n;
         e += gridDim.x * kBlock) {
        double s = x[e];
        for (int i = 0; i < jcols; ++i)
            s = fma(y[i], Z[(size_t)i * ld + e], s);
        if (jcols > 0) x[e] = s;
        if (te.out) tail_extrapolate(te, e, s);
    }
}

}  // namespace dns
