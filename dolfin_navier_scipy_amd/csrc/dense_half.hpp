// Dense inverse of a multigrid level in half precision.
//
// The coarse end of the Schur multigrid is launch bound: a level of 5000 rows
// costs five dependent launches of ~5 us per cycle (pre-smoother, restriction,
// the dense inverse below it, prolongation + post-smoother, second sweep) for
// a few hundred kilobytes of work.  Its inverse as ONE matrix-vector product
// is a single launch -- bandwidth bound, so it pays when the matrix is small
// in BYTES: 5000^2 entries are 100 MB in fp32 (25 us: as slow as the launches
// it replaces, measured in round 3), 50 MB in fp16 (11-12 us).
//
// Storage: row major, leading dimension a multiple of 8 (16-byte loads of
// eight entries per lane), entries divided by `scale` (max |a_ij| -> 1024:
// head room above, 2^-24 / 1024 of the largest entry below before a value
// flushes to zero).  The rounding (2^-11 relative per entry) acts like a
// random perturbation of the coarse solve of relative size
// ~2^-11 / sqrt(n) in the spectral norm; the level sits inside a V-cycle
// that is a preconditioner, and the Krylov-step counts of the parity tests
// (tests/test_gpu_saddle.py) hold it to what the sparse levels gave.
#pragma once
#include <hip/hip_fp16.h>
#include "kernels.hpp"

namespace dns {

// max |a_i| over count entries -> out[0] (one workgroup per partial, then a
// second launch with one workgroup)
__global__ void __launch_bounds__(kBlock)
k_absmax(int64_t count, const double *__restrict__ a, double *__restrict__ part) {
    __shared__ double red[4];
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * kBlock)
        m = fmax(m, fabs(a[i]));
    // (block_sum adds; a maximum needs its own reduction)
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        part[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// h[r * ldh + c] = a[r * n + c] * inv_scale (zero in the padding columns)
__global__ void __launch_bounds__(kBlock)
k_to_half_rows(int n, int ldh, const double *__restrict__ a,
               const double *__restrict__ amax, int nmax, double target,
               __half *__restrict__ h, double *__restrict__ scale_out) {
    double m = 0.0;
    for (int i = 0; i < nmax; ++i) m = fmax(m, amax[i]);
    const double scale = (m > 0.0) ? m / target : 1.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = scale;
    const double inv = 1.0 / scale;
    const int64_t total = (int64_t)n * ldh;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * kBlock) {
        const int r = (int)(i / ldh), c = (int)(i % ldh);
        h[i] = __float2half(c < n ? (float)(a[(size_t)r * n + c] * inv) : 0.f);
    }
}

__device__ __forceinline__ float dot8(const uint4 q, const float *xs) {
    const __half2 *hp = reinterpret_cast<const __half2 *>(&q);
    const float4 x0 = *reinterpret_cast<const float4 *>(xs);
    const float4 x1 = *reinterpret_cast<const float4 *>(xs + 4);
    const float2 a0 = __half22float2(hp[0]), a1 = __half22float2(hp[1]);
    const float2 a2 = __half22float2(hp[2]), a3 = __half22float2(hp[3]);
    float s = a0.x * x0.x;
    s = fmaf(a0.y, x0.y, s);
    s = fmaf(a1.x, x0.z, s);
    s = fmaf(a1.y, x0.w, s);
    s = fmaf(a2.x, x1.x, s);
    s = fmaf(a2.y, x1.y, s);
    s = fmaf(a3.x, x1.z, s);
    s = fmaf(a3.y, x1.w, s);
    return s;
}

// y = alpha * scale * H x : one wavefront per row, x staged in LDS as fp32
// (ldh floats of dynamic shared memory), four 16-byte loads in flight per
// lane, fp32 partial sums per lane, the row's sum in fp64
__global__ void __launch_bounds__(kBlock)
k_gemv_half(int n, int ldh, const __half *__restrict__ h,
            const double *__restrict__ scale, const double *__restrict__ x,
            double *__restrict__ y, double alpha, const DnsCtl *ctl) {
    extern __shared__ float xs[];
    if (ctl && ctl->done) return;
    for (int i = threadIdx.x; i < ldh; i += kBlock)
        xs[i] = (i < n) ? (float)x[i] : 0.f;
    __syncthreads();
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * kBlock) >> 6;
    const int nq = ldh >> 3;                    // 16-byte packets per row
    const double f = alpha * scale[0];
    for (int row = wave; row < n; row += nwaves) {
        const uint4 *hr = reinterpret_cast<const uint4 *>(h + (size_t)row * ldh);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int q = lane;
        for (; q + 192 < nq; q += 256) {
            const uint4 a0 = hr[q], a1 = hr[q + 64], a2 = hr[q + 128],
                        a3 = hr[q + 192];
            s0 += dot8(a0, xs + 8 * q);
            s1 += dot8(a1, xs + 8 * (q + 64));
            s2 += dot8(a2, xs + 8 * (q + 128));
            s3 += dot8(a3, xs + 8 * (q + 192));
        }
        for (; q < nq; q += 64) s0 += dot8(hr[q], xs + 8 * q);
        const double s = wave_sum((double)s0 + (double)s1 + (double)s2 +
                                  (double)s3);
        if (lane == 0) y[row] = f * s;
    }
}

}  // namespace dns
