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
// `jv[par]` is the Arnoldi column of the iteration with parity `par`; the
// closing kernel writes `jv[par ^ 1]`, so no kernel reads a slot that is
// written in the same launch.
// ---------------------------------------------------------------------------
struct DnsCtl {
    int jv[2];
    int done;        // != 0: all later kernels of this solve return at once
    int status;      // DNS_OK / DNS_BREAKDOWN
    int zero;        // constant 0 (vector selector / never-raised guard)
    int total_it;
    int hist_len;    // entries appended to the solve's residual history
    int conv;        // 1 once ||r|| <= tol was observed
    double beta, tol, resnorm, bnorm;
    // BiCGStab scalars
    double rho, alpha, omega;
    double pad2;
    double hcol[kMaxRestart + 1];
    double cs[kMaxRestart], sn[kMaxRestart], g[kMaxRestart + 1];
    double y[kMaxRestart];
    double hist[kMaxRestart + 1];                       // per-cycle residuals
    double R[(kMaxRestart + 1) * kMaxRestart];          // column major
};

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
    for (int s = wave; s < nscal; s += nw) {
        double a = 0.0;
        for (int p = lane; p < nparts; p += 64)
            a += part[(size_t)s * pstride + p];
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
template <int LPR>
__device__ __forceinline__ double csr_row_dot(const int *__restrict__ rowptr,
                                              const int *__restrict__ colidx,
                                              const double *__restrict__ vals,
                                              const double *__restrict__ x,
                                              int row, int sublane) {
    const int k1 = rowptr[row + 1];
    double s = 0.0;
    for (int k = rowptr[row] + sublane; k < k1; k += LPR)
        s = fma(vals[k], x[colidx[k]], s);
    return subwave_sum<LPR>(s);
}

// y = alpha*A*x + beta*b   (b may be null -> beta ignored)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_spmv_vec(int nrows, const int *__restrict__ rowptr,
           const int *__restrict__ colidx, const double *__restrict__ vals,
           const double *__restrict__ x, double *__restrict__ y, double alpha,
           double beta, const double *__restrict__ b) {
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

template <int G>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream(int nblocks, const int *__restrict__ rowblocks,
              const int *__restrict__ rowptr, const int *__restrict__ colidx,
              const double *__restrict__ vals, const double *__restrict__ x,
              double *__restrict__ y, double alpha, double beta,
              const double *__restrict__ b) {
    __shared__ double prod[kStreamNnz];
    __shared__ double red[4];
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int r0 = rowblocks[blk], r1 = rowblocks[blk + 1];
        const int k0 = rowptr[r0], k1 = rowptr[r1];
        const int nn = k1 - k0;
        if (nn > kStreamNnz) {
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
#pragma unroll 4
        for (int k = threadIdx.x; k < nn; k += kBlock)
            prod[k] = vals[k0 + k] * x[colidx[k0 + k]];
        __syncthreads();
        const int nr = r1 - r0;
        const int g = threadIdx.x % G, rsub = threadIdx.x / G;
        for (int r = rsub; r < nr; r += kBlock / G) {
            const int a0 = rowptr[r0 + r] - k0, a1 = rowptr[r0 + r + 1] - k0;
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

// ---------------------------------------------------------------------------
// dense Schur block:  y = alpha * A x,  A is n x n row-major; one wavefront
// per row, 16-byte loads where the row start is 16-byte aligned.
// `xsel`/`ysel`: vector selectors (see vec_at)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_gemv_rows(int n, const double *__restrict__ a, const double *__restrict__ x,
            double *__restrict__ y, double alpha, const DnsCtl *ctl) {
    if (ctl && ctl->done) return;
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * kBlock) >> 6;
    for (int row = wave; row < n; row += nwaves) {
        const double *ar = a + (size_t)row * n;
        double s = 0.0;
        for (int c = lane; c < n; c += 64) s = fma(ar[c], x[c], s);
        s = wave_sum(s);
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
            const int *__restrict__ guard) {
    if (*guard) return;
    const double *rv = rbase + (size_t)(*jsel) * ld;
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int row = sub; row < nv; row += nsub) {
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
            const int *__restrict__ guard) {
    if (*guard) return;
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int row = sub; row < nv; row += nsub) {
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

// Jacobi Schur fallback: zp = -sdinv * rp
__global__ void __launch_bounds__(kBlock)
k_schur_jacobi(int np, const double *__restrict__ sdinv,
               const double *__restrict__ rbase, size_t ld,
               const int *__restrict__ jsel, int nv, double *__restrict__ zp,
               const int *__restrict__ guard) {
    if (*guard) return;
    const double *rp = rbase + (size_t)(*jsel) * ld + nv;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < np;
         i += gridDim.x * kBlock)
        zp[i] = -sdinv[i] * rp[i];
}

// dense Schur: zp = -Sinv rp  (rp = selected vector's pressure part)
__global__ void __launch_bounds__(kBlock)
k_schur_dense(int np, const double *__restrict__ sinv,
              const double *__restrict__ rbase, size_t ld,
              const int *__restrict__ jsel, int nv, double *__restrict__ zp,
              const int *__restrict__ guard) {
    if (*guard) return;
    const double *rp = rbase + (size_t)(*jsel) * ld + nv;
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * kBlock) >> 6;
    for (int row = wave; row < np; row += nwaves) {
        const double *ar = sinv + (size_t)row * np;
        double s = 0.0;
        for (int c = lane; c < np; c += 64) s = fma(ar[c], rp[c], s);
        s = wave_sum(s);
        if (lane == 0) zp[row] = -s;
    }
}

// ---------------------------------------------------------------------------
// GMRES kernels
// ---------------------------------------------------------------------------
// w = alpha*K*x + beta*b with a fused partial of ||w||^2 (one per workgroup)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_spmv_norm(int nrows, const int *__restrict__ rowptr,
            const int *__restrict__ colidx, const double *__restrict__ vals,
            const double *__restrict__ x, double *__restrict__ w, double alpha,
            double beta, const double *__restrict__ b,
            double *__restrict__ partials, const DnsCtl *ctl) {
    if (ctl && ctl->done) return;
    __shared__ double red[4];
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    double acc = 0.0;
    for (int row = sub; row < nrows; row += nsub) {
        double s = csr_row_dot<LPR>(rowptr, colidx, vals, x, row, sublane);
        if (sublane == 0) {
            const double v = b ? fma(alpha, s, beta * b[row]) : alpha * s;
            w[row] = v;
            acc = fma(v, v, acc);
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// plain w = K z inside the Arnoldi loop (guarded by ctl->done)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_spmv_guard(int nrows, const int *__restrict__ rowptr,
             const int *__restrict__ colidx, const double *__restrict__ vals,
             const double *__restrict__ x, double *__restrict__ w,
             const DnsCtl *ctl) {
    if (ctl->done) return;
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int row = sub; row < nrows; row += nsub) {
        double s = csr_row_dot<LPR>(rowptr, colidx, vals, x, row, sublane);
        if (sublane == 0) w[row] = s;
    }
}

// cycle start: beta = sqrt(sum partials); V[0] = r/beta; per-cycle reset of
// the control block.  Identical for every cycle of a solve (the host zeroes
// `total_it/hist_len/conv/status` once before the first), so one captured
// graph serves all cycles.
__global__ void __launch_bounds__(kBlock)
k_gmres_start(int n, const double *__restrict__ r, double *__restrict__ v0,
              const double *__restrict__ partials, int nparts, DnsCtl *ctl,
              double rtol, double atol,
              const double *__restrict__ bnorm_partials, int maxiter) {
    __shared__ double sc[2];
    reduce_partials(partials, nparts, nparts, 1, sc);
    reduce_partials(bnorm_partials, nparts, nparts, 1, sc + 1);
    const double beta = sqrt(sc[0]);
    const double bn = sqrt(sc[1]);
    const double tol = fmax(rtol * bn, atol);
    const bool conv = !(beta > tol);
    const bool stop = conv || isnan(beta) || ctl->total_it >= maxiter ||
                      ctl->status != DNS_OK;
    if (!stop) {
        const double ib = 1.0 / beta;
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
             i += gridDim.x * kBlock)
            v0[i] = r[i] * ib;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->jv[0] = 0;
        ctl->jv[1] = 0;
        ctl->zero = 0;
        ctl->beta = beta;
        ctl->resnorm = beta;
        ctl->bnorm = bn;
        ctl->tol = tol;
        ctl->g[0] = beta;
        ctl->hist[0] = beta;
        if (conv) ctl->conv = 1;
        ctl->done = stop ? 1 : 0;
    }
}

// partials[i*nparts + block] = <V_i, w>,  i <= j
__global__ void __launch_bounds__(kBlock)
k_multidot(int n, const double *__restrict__ V, size_t ld,
           const double *__restrict__ w, double *__restrict__ partials,
           int nparts, const DnsCtl *ctl, int par) {
    if (ctl->done) return;
    const int nvec = ctl->jv[par] + 1;
    __shared__ double wred[4][8];
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
            for (int ww = 0; ww < (kBlock >> 6); ++ww)
                s += wred[ww][threadIdx.x];
            partials[(size_t)(i0 + threadIdx.x) * nparts + blockIdx.x] = s;
        }
    }
}

// w -= sum_i h_i V_i with h_i = sum of the partials of k_multidot;
// workgroup 0 accumulates the Hessenberg column; optional partial of ||w||^2
__global__ void __launch_bounds__(kBlock)
k_orth_update(int n, const double *__restrict__ V, size_t ld,
              double *__restrict__ w, const double *__restrict__ partials,
              int nparts, DnsCtl *ctl, int par, int pass,
              double *__restrict__ norm_partials) {
    if (ctl->done) return;
    const int nvec = ctl->jv[par] + 1;
    __shared__ double h[kMaxRestart + 1];
    __shared__ double red[4];
    reduce_partials(partials, nparts, nparts, nvec, h);
    if (blockIdx.x == 0 && threadIdx.x < nvec)
        ctl->hcol[threadIdx.x] =
            (pass == 0 ? 0.0 : ctl->hcol[threadIdx.x]) + h[threadIdx.x];
    double acc = 0.0;
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < n;
         e += gridDim.x * kBlock) {
        double we = w[e];
        for (int i = 0; i < nvec; ++i)
            we = fma(-h[i], V[(size_t)i * ld + e], we);
        w[e] = we;
        acc = fma(we, we, acc);
    }
    if (norm_partials) {
        acc = block_sum(acc, red);
        if (threadIdx.x == 0) norm_partials[blockIdx.x] = acc;
    }
}

// V[j+1] = w/||w||; workgroup 0: Givens rotations, residual estimate, flags
__global__ void __launch_bounds__(kBlock)
k_gmres_close(int n, const double *__restrict__ w, double *__restrict__ V,
              size_t ld, const double *__restrict__ norm_partials, int nparts,
              DnsCtl *ctl, int par, int mcycle, int maxiter) {
    if (ctl->done) return;
    const int j = ctl->jv[par];
    __shared__ double sc[1];
    reduce_partials(norm_partials, nparts, nparts, 1, sc);
    const double hn = sqrt(sc[0]);
    if (hn > 0.0) {
        const double ih = 1.0 / hn;
        double *vn = V + (size_t)(j + 1) * ld;
        for (int e = blockIdx.x * kBlock + threadIdx.x; e < n;
             e += gridDim.x * kBlock)
            vn[e] = w[e] * ih;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double *Rc = ctl->R + (size_t)j * (kMaxRestart + 1);
        for (int i = 0; i <= j; ++i) Rc[i] = ctl->hcol[i];
        double hj1 = hn;
        for (int i = 0; i < j; ++i) {
            const double t = ctl->cs[i] * Rc[i] + ctl->sn[i] * Rc[i + 1];
            Rc[i + 1] = -ctl->sn[i] * Rc[i] + ctl->cs[i] * Rc[i + 1];
            Rc[i] = t;
        }
        const double den = hypot(Rc[j], hj1);
        int status = DNS_OK;
        double c = 1.0, s = 0.0;
        if (den > 0.0) {
            c = Rc[j] / den;
            s = hj1 / den;
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
        ctl->jv[par ^ 1] = j + 1;
        ctl->total_it += 1;
        const bool conv = !(res > ctl->tol);
        if (status != DNS_OK) ctl->status = status;
        if (conv) ctl->conv = 1;
        if (conv || status != DNS_OK || !(hn > 0.0) || isnan(res) ||
            ctl->total_it >= maxiter)
            ctl->done = 1;
        (void)mcycle;
    }
}

// end of a cycle (single thread): y = R^-1 g for the completed columns and
// the cycle's residual norms appended to the solve's history
__global__ void k_gmres_finish_y(DnsCtl *ctl, double *__restrict__ histbuf,
                                 int hist_cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int jcols = max(ctl->jv[0], ctl->jv[1]);
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
}

// u = sum_{i<jcols} y_i V_i   (jcols read from the control block)
__global__ void __launch_bounds__(kBlock)
k_basis_combine(int n, const double *__restrict__ V, size_t ld,
                const DnsCtl *ctl, double *__restrict__ u) {
    __shared__ double y[kMaxRestart];
    const int jcols = max(ctl->jv[0], ctl->jv[1]);
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

}  // namespace dns
