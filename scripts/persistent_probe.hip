// Measured answer to "would a persistent kernel beat two graph nodes?" for the
// two phases of the N = 2 step whose join is a pressure-sized vector:
//   phase A ("tau"):  tau = r_p - (J Fh^-1) r_v   -- 1289 sparse rows of ~270
//                     entries (4 MB of fp64 values + int32 indices)
//   phase B ("head"): zp = -Sh^-1 tau             -- dense 1289 x 1292 fp32
//                     (6.6 MB), one workgroup per group of rows
// Variant 1: two kernels, dependent, in a replayed hipGraph (what the library
//            does: ~1.7 us per graph edge).
// Variant 2: ONE cooperative kernel, phase A, grid barrier (arrive counter +
//            agent-scope release / acquire fences), phase B -- with G = 32 ...
//            256 workgroups, optionally confined to the 32 CUs of one XCD by a
//            CU mask (one L2, no cross-XCD coherence traffic).
// Prints microseconds per (A, B) pair for each variant.
//   hipcc --offload-arch=gfx950 -O3 -o persistent_probe persistent_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                              \
    do {                                                                   \
        hipError_t e = (x);                                                \
        if (e != hipSuccess) {                                             \
            printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

constexpr int kBlock = 256;
constexpr int NP = 1289, NV = 9356, SLD = 1292, ROWLEN = 270;

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// phase A for the rows [r0, r1): one wave per row
__device__ __forceinline__ void phase_a(int r0, int r1, int wave_global,
                                        int nwaves, const int *rp,
                                        const int *ci, const double *va,
                                        const double *rv, const double *rpv,
                                        double *tau) {
    const int lane = threadIdx.x & 63;
    for (int row = r0 + wave_global; row < r1; row += nwaves) {
        double s = 0.0;
        for (int k = rp[row] + lane; k < rp[row + 1]; k += 64)
            s = fma(va[k], rv[ci[k]], s);
        s = wave_sum(s);
        if (lane == 0) tau[row] = rpv[row] - s;
    }
}

// phase B: one wave per row of the dense inverse
__device__ __forceinline__ void phase_b(int wave_global, int nwaves,
                                        const float *sinv, const double *tau,
                                        double *zp) {
    const int lane = threadIdx.x & 63;
    for (int row = wave_global; row < NP; row += nwaves) {
        const float *a = sinv + (size_t)row * SLD;
        double s = 0.0;
        for (int k = lane; k < NP; k += 64) s = fma((double)a[k], tau[k], s);
        s = wave_sum(s);
        if (lane == 0) zp[row] = -s;
    }
}

__global__ void __launch_bounds__(kBlock)
k_a(const int *rp, const int *ci, const double *va, const double *rv,
    const double *rpv, double *tau) {
    const int w = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    phase_a(0, NP, w, gridDim.x * (kBlock / 64), rp, ci, va, rv, rpv, tau);
}

__global__ void __launch_bounds__(kBlock)
k_b(const float *sinv, const double *tau, double *zp) {
    const int w = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    phase_b(w, gridDim.x * (kBlock / 64), sinv, tau, zp);
}

// both phases, `reps` times, with a grid barrier between the phases (and one
// behind B: the next A of a real step depends on zp through two more phases)
__global__ void __launch_bounds__(kBlock)
k_persistent(const int *rp, const int *ci, const double *va, const double *rv,
             const double *rpv, double *tau, const float *sinv, double *zp,
             unsigned *counter, int reps) {
    const int w = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int nw = gridDim.x * (kBlock / 64);
    unsigned target = 0;
    auto barrier = [&]() {
        target += gridDim.x;
        __syncthreads();
        if (threadIdx.x == 0) {
            // release: this workgroup's stores before the arrive
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            // (bounded: a grid that is not co-resident must still drain)
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) < target &&
                   ++spins < (1 << 20))
                __builtin_amdgcn_s_sleep(1);
            // acquire: the other workgroups' stores behind the barrier
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    };
    for (int it = 0; it < reps; ++it) {
        phase_a(0, NP, w, nw, rp, ci, va, rv, rpv, tau);
        barrier();
        phase_b(w, nw, sinv, tau, zp);
        barrier();
    }
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    std::vector<int> rp(NP + 1), ci((size_t)NP * ROWLEN);
    std::vector<double> va((size_t)NP * ROWLEN), rv(NV), rpv(NP);
    std::vector<float> sinv((size_t)NP * SLD);
    unsigned seed = 7u;
    auto rnd = [&]() {
        seed = seed * 1664525u + 1013904223u;
        return (seed >> 8) & 0xffffff;
    };
    for (int i = 0; i <= NP; ++i) rp[i] = i * ROWLEN;
    for (int i = 0; i < NP; ++i)
        for (int k = 0; k < ROWLEN; ++k) {
            // a window of velocity dofs around the row's patch
            const int base = (int)((long)i * (NV - 2000) / NP);
            ci[(size_t)i * ROWLEN + k] = base + (int)(rnd() % 2000);
            va[(size_t)i * ROWLEN + k] = 1e-3 * ((int)(rnd() % 2001) - 1000);
        }
    for (auto &v : rv) v = 1e-3 * ((int)(rnd() % 2001) - 1000);
    for (auto &v : rpv) v = 1e-3 * ((int)(rnd() % 2001) - 1000);
    for (auto &v : sinv) v = 1e-3f * ((int)(rnd() % 2001) - 1000);
    int *d_rp, *d_ci;
    double *d_va, *d_rv, *d_rpv, *d_tau, *d_zp;
    float *d_sinv;
    unsigned *d_cnt;
    CK(hipMalloc(&d_rp, rp.size() * 4));
    CK(hipMalloc(&d_ci, ci.size() * 4));
    CK(hipMalloc(&d_va, va.size() * 8));
    CK(hipMalloc(&d_rv, rv.size() * 8));
    CK(hipMalloc(&d_rpv, rpv.size() * 8));
    CK(hipMalloc(&d_tau, NP * 8));
    CK(hipMalloc(&d_zp, NP * 8));
    CK(hipMalloc(&d_sinv, sinv.size() * 4));
    CK(hipMalloc(&d_cnt, 4));
    CK(hipMemcpy(d_rp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ci, ci.data(), ci.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_va, va.data(), va.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rv, rv.data(), rv.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rpv, rpv.data(), rpv.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sinv, sinv.data(), sinv.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> ref(NP), got(NP);

    // ---- variant 1: two kernels per pair, replayed graph
    {
        hipStream_t s;
        CK(hipStreamCreate(&s));
        const int ga = (NP + 3) / 4, gb = (NP + 3) / 4;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int it = 0; it < reps; ++it) {
            hipLaunchKernelGGL(k_a, ga, kBlock, 0, s, d_rp, d_ci, d_va, d_rv,
                               d_rpv, d_tau);
            hipLaunchKernelGGL(k_b, gb, kBlock, 0, s, d_sinv, d_tau, d_zp);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("two graph nodes per pair (grids %d / %d)      : %7.2f us per "
               "pair\n", ga, gb, 1e3 * ms / (5.0 * reps));
        CK(hipMemcpy(ref.data(), d_zp, NP * 8, hipMemcpyDeviceToHost));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        CK(hipStreamDestroy(s));
    }

    // ---- variant 2: one persistent kernel, grid barrier between the phases
    for (int masked = 0; masked < 2; ++masked) {
        hipStream_t s;
        if (masked) {
            // the first 32 CUs = one XCD (MI355X: 8 XCDs x 32 CUs)
            const uint32_t mask[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0};
            if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) {
                printf("no CU-masked stream on this device\n");
                continue;
            }
        } else {
            CK(hipStreamCreate(&s));
        }
        for (int G : {32, 64, 128, 256}) {
            if (masked && G > 64) continue;   // (resident on 32 CUs)
            CK(hipMemsetAsync(d_cnt, 0, 4, s));
            CK(hipMemsetAsync(d_zp, 0, NP * 8, s));
            int r2 = reps;
            void *args[] = {&d_rp,  &d_ci,   &d_va, &d_rv,  &d_rpv,
                            &d_tau, &d_sinv, &d_zp, &d_cnt, &r2};
            // warm-up launch, then the timed one
            CK(hipLaunchCooperativeKernel((void *)k_persistent, dim3(G),
                                          dim3(kBlock), args, 0, s));
            CK(hipStreamSynchronize(s));
            CK(hipMemsetAsync(d_cnt, 0, 4, s));
            CK(hipEventRecord(e0, s));
            CK(hipLaunchCooperativeKernel((void *)k_persistent, dim3(G),
                                          dim3(kBlock), args, 0, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(got.data(), d_zp, NP * 8, hipMemcpyDeviceToHost));
            double err = 0.0, nrm = 0.0;
            for (int i = 0; i < NP; ++i) {
                err += (got[i] - ref[i]) * (got[i] - ref[i]);
                nrm += ref[i] * ref[i];
            }
            printf("persistent, %3d workgroups%s: %7.2f us per pair (two "
                   "barriers), result %s\n", G,
                   masked ? ", one XCD (CU mask)" : "                    ",
                   1e3 * ms / reps, err <= 1e-20 * nrm ? "equal" : "DIFFERS");
        }
        CK(hipStreamDestroy(s));
    }
    return 0;
}
