// Micro-benchmark: what does a small DEPENDENT kernel cost on MI355X?
// hipcc --offload-arch=gfx950 -O3 -o latency_probe latency_probe.hip
// Chains of N identical kernels in one stream, captured as a hipGraph and
// replayed; reports microseconds per kernel for several kernel bodies.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                              \
    do {                                                                   \
        hipError_t e = (x);                                                \
        if (e != hipSuccess) {                                             \
            printf("%s: %s\n", #x, hipGetErrorString(e));                  \
            return 1;                                                      \
        }                                                                  \
    } while (0)

__global__ void k_empty() {}

__global__ void k_flag(const int *f) {
    if (*f) return;
}

// out[i] = a*in[i] : one load, one store per thread
__global__ void k_scale(int n, const double *in, double *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.0000001 * in[i];
}

// dependent chain of `depth` loads per thread, then one store
__global__ void k_chain(int n, const int *idx, const double *in, double *out,
                        int depth) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = i;
    for (int d = 0; d < depth; ++d) p = idx[p];
    out[i] = in[p];
}

// read-only: every thread loads, one thread per block stores a reduced value
__global__ void k_reduce(int n, const double *in, double *out) {
    __shared__ double red[4];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = (i < n) ? in[i] : 0.0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// stream `bytes` (read) with all CUs, tiny output
__global__ void k_stream(size_t n8, const double *in, double *out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8;
         i += (size_t)gridDim.x * blockDim.x)
        s += in[i];
    if (s == 12345.678) out[0] = s;
}

template <typename F>
double time_chain(hipStream_t s, int reps, int chain, F enqueue) {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < chain; ++k) enqueue(k);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    return 1e3 * ms / (reps * (double)chain);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 10645;
    const size_t big = (size_t)32 << 20;   // doubles: 256 MiB
    double *a, *b, *c, *bigbuf;
    int *idx, *flag;
    CK(hipMalloc(&a, n * 8 * 2));
    CK(hipMalloc(&b, n * 8 * 2));
    CK(hipMalloc(&c, n * 8 * 2));
    CK(hipMalloc(&bigbuf, big * 8));
    CK(hipMalloc(&idx, n * 4));
    CK(hipMalloc(&flag, 4));
    CK(hipMemset(a, 0, n * 16));
    CK(hipMemset(b, 0, n * 16));
    CK(hipMemset(bigbuf, 0, big * 8));
    CK(hipMemset(flag, 0, 4));
    std::vector<int> hidx(n);
    for (int i = 0; i < n; ++i) hidx[i] = (int)(((long)i * 7919 + 13) % n);
    CK(hipMemcpy(idx, hidx.data(), n * 4, hipMemcpyHostToDevice));
    const int chain = 64, reps = 50;
    const int g = (n + 255) / 256;
    printf("us per dependent kernel (graph replay, chain of %d):\n", chain);
    printf("  empty, 1 block            : %.2f\n",
           time_chain(s, reps, chain, [&](int) {
               hipLaunchKernelGGL(k_empty, 1, 64, 0, s);
           }));
    printf("  empty, 256 blocks         : %.2f\n",
           time_chain(s, reps, chain, [&](int) {
               hipLaunchKernelGGL(k_empty, 256, 256, 0, s);
           }));
    printf("  flag read, 256 blocks     : %.2f\n",
           time_chain(s, reps, chain, [&](int) {
               hipLaunchKernelGGL(k_flag, 256, 256, 0, s, flag);
           }));
    printf("  scale n=10645 (42 blocks) : %.2f\n",
           time_chain(s, reps, chain, [&](int k) {
               hipLaunchKernelGGL(k_scale, g, 256, 0, s, n, (k & 1) ? b : a,
                                  (k & 1) ? a : b);
           }));
    printf("  reduce n=10645 -> 42      : %.2f\n",
           time_chain(s, reps, chain, [&](int k) {
               hipLaunchKernelGGL(k_reduce, g, 256, 0, s, n, a, c);
           }));
    for (int depth = 1; depth <= 4; ++depth)
        printf("  chain depth %d + load      : %.2f\n", depth,
               time_chain(s, reps, chain, [&](int k) {
                   hipLaunchKernelGGL(k_chain, g, 256, 0, s, n, idx,
                                      (k & 1) ? b : a, (k & 1) ? a : b, depth);
               }));
    for (size_t mb : {1, 4, 8, 16, 32, 64}) {
        const size_t n8 = mb * 1024 * 1024 / 8;
        printf("  stream %3zu MB read        : %.2f  (%.0f GB/s)\n", mb,
               time_chain(s, reps, 16,
                          [&](int) {
                              hipLaunchKernelGGL(k_stream, 1024, 256, 0, s, n8,
                                                 bigbuf, c);
                          }),
               0.0);
    }
    // producer and consumer map an element to DIFFERENT workgroups (hence
    // different XCDs): 256-thread blocks alternate with 64-thread blocks
    printf("  scale, alternating 256/64-thread blocks : %.2f\n",
           time_chain(s, reps, chain, [&](int k) {
               if (k & 1)
                   hipLaunchKernelGGL(k_scale, g, 256, 0, s, n, b, a);
               else
                   hipLaunchKernelGGL(k_scale, (n + 63) / 64, 64, 0, s, n, a,
                                      b);
           }));
    printf("  scale, alternating 256/1024-thread blocks: %.2f\n",
           time_chain(s, reps, chain, [&](int k) {
               if (k & 1)
                   hipLaunchKernelGGL(k_scale, g, 256, 0, s, n, b, a);
               else
                   hipLaunchKernelGGL(k_scale, (n + 1023) / 1024, 1024, 0, s, n,
                                      a, b);
           }));
    for (int nb : {1, 42, 256, 1024}) {
        const int nn = nb * 256;
        (void)nn;
        printf("  scale, %4d blocks x256    : %.2f\n", nb,
               time_chain(s, reps, chain, [&](int k) {
                   hipLaunchKernelGGL(k_scale, nb, 256, 0, s,
                                      nb * 256 > 2 * n ? 2 * n : nb * 256,
                                      (k & 1) ? b : a, (k & 1) ? a : b);
               }));
    }
    return 0;
}
