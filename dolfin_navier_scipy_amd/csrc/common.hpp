// Shared host-side plumbing for the gfx950 saddle-point library.
#pragma once
#include <exception>
#include <new>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dns_amd.h"

namespace dns {

constexpr int kBlock = 256;      // 4 wavefronts of 64 lanes
constexpr int kWave = 64;
constexpr int kMaxRestart = 64;  // GMRES cycle length bound (DnsCtl arrays)

inline thread_local std::string g_last_error;

inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// the C-ABI is an exception barrier: `body` (a lambda returning a status) runs
// inside try/catch, a C++ exception becomes DNS_ERR_HOST with its message
template <typename Body>
inline int guarded(Body body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(DNS_ERR_HOST, "out of host memory");
    } catch (const std::exception &e) {
        return fail(DNS_ERR_HOST, "host-side exception: %s", e.what());
    } catch (...) {
        return fail(DNS_ERR_HOST, "host-side exception");
    }
}

#define DNS_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess)                                               \
            return dns::fail(DNS_ERR_HIP, "%s failed: %s (%s:%d)", #call,    \
                             hipGetErrorString(e__), __FILE__, __LINE__);    \
    } while (0)

#define DNS_TRY(call)                                                        \
    do {                                                                     \
        int s__ = (call);                                                    \
        if (s__ != DNS_OK) return s__;                                       \
    } while (0)

// DNS_DEBUG_UPLOADS=1: every copy between HOST memory and the device says
// its host range [ptr, ptr + bytes) on stderr before it is enqueued, so that
// the address of a "Memory access fault by GPU ... on address" report can be
// tied to a buffer and an offset (round 4: a fault at a host heap address
// whose record held no ranges)
inline bool debug_uploads() {
    static const bool on = [] {
        const char *e = getenv("DNS_DEBUG_UPLOADS");
        return e && e[0] != '0' && e[0] != 0;
    }();
    return on;
}
inline void log_host_copy(const char *what, const void *host, const void *dev,
                          size_t bytes) {
    if (!debug_uploads()) return;
    fprintf(stderr, "[dns copy] %-10s host [%p, %p) dev %p bytes %zu\n", what,
            host, (const void *)((const char *)host + bytes), dev, bytes);
    fflush(stderr);
}

// ---------------------------------------------------------------------------
// Host <-> device copies never hand PAGEABLE memory to the DMA engine.
// Twice now (rounds 4 and 5) a run of the GPU suite died with "Memory access
// fault by GPU ... on address <page-aligned HOST heap address>" inside a call
// whose only touch of host memory was a hipMemcpyAsync out of pageable memory
// (a caller's NumPy block, a std::vector of the set-up) -- intermittently, at
// different places, every kernel argument accounted for.  For pageable
// memory the runtime pins the pages in place and lets a blit kernel read
// them; what exactly goes wrong there is not ours to find (DNS_DEBUG_UPLOADS
// prints the ranges for whoever looks).  So every copy goes through a
// page-locked bounce buffer of this thread: the CPU copies between the
// caller's memory and the bounce buffer, the DMA engine only ever sees
// hipHostMalloc memory.  4 MiB chunks, each synchronised (a copy IS complete
// when the call returns, as DevBuf::upload promises anyway): PCIe rate minus
// a few per cent.
// ---------------------------------------------------------------------------
struct BounceBuffer {
    void *p = nullptr;
    size_t bytes = 0;
    ~BounceBuffer() {
        if (p) (void)hipHostFree(p);
    }
    int reserve(size_t want) {
        if (want <= bytes) return DNS_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(DNS_ERR_HIP, "hipHostMalloc(%zu) failed: %s", want,
                        hipGetErrorString(e));
        }
        bytes = want;
        return DNS_OK;
    }
};
inline thread_local BounceBuffer g_bounce;
constexpr size_t kBounceChunk = (size_t)4 << 20;

inline int staged_h2d(void *dev, const void *host, size_t bytes,
                      hipStream_t s) {
    if (bytes == 0) return DNS_OK;
    DNS_TRY(g_bounce.reserve(std::min(bytes, kBounceChunk)));
    const char *src = static_cast<const char *>(host);
    char *dst = static_cast<char *>(dev);
    for (size_t off = 0; off < bytes; off += kBounceChunk) {
        const size_t len = std::min(kBounceChunk, bytes - off);
        memcpy(g_bounce.p, src + off, len);
        DNS_HIP(hipMemcpyAsync(dst + off, g_bounce.p, len,
                               hipMemcpyHostToDevice, s));
        DNS_HIP(hipStreamSynchronize(s));
    }
    return DNS_OK;
}

inline int staged_d2h(void *host, const void *dev, size_t bytes,
                      hipStream_t s) {
    if (bytes == 0) return DNS_OK;
    DNS_TRY(g_bounce.reserve(std::min(bytes, kBounceChunk)));
    char *dst = static_cast<char *>(host);
    const char *src = static_cast<const char *>(dev);
    for (size_t off = 0; off < bytes; off += kBounceChunk) {
        const size_t len = std::min(kBounceChunk, bytes - off);
        DNS_HIP(hipMemcpyAsync(g_bounce.p, src + off, len,
                               hipMemcpyDeviceToHost, s));
        DNS_HIP(hipStreamSynchronize(s));
        memcpy(dst + off, g_bounce.p, len);
    }
    return DNS_OK;
}

// device buffer with explicit lifetime (no exceptions across the C-ABI)
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        DNS_HIP(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
        n = count;
        return DNS_OK;
    }
    // Host -> device, SAFE BY CONSTRUCTION: when the call returns the copy
    // has left `host`, whose lifetime is therefore nobody's concern (a local
    // std::vector, a caller's borrowed array).  An asynchronous copy out of
    // pageable memory that is freed -- and possibly unmapped -- before the
    // DMA engine gets to it is a GPU memory fault (round 3: intermittent
    // aborts of the test session).  Ordered on `s` like any other work.
    int upload(const T *host, size_t count, hipStream_t s) {
        return upload_async(host, count, s);
    }
    // ... and the variant that only enqueues: the CALLER guarantees that
    // `host` stays valid and unmodified until it has synchronised `s` on
    // EVERY path out of its scope (error returns included -- `SyncOnExit`)
    int upload_async(const T *host, size_t count, hipStream_t s) {
        if (count > n) return fail(DNS_ERR_BAD_ARGUMENT, "upload overflow");
        if (count == 0) return DNS_OK;
        log_host_copy("upload", host, p, count * sizeof(T));
        // (through the page-locked bounce buffer: complete on return)
        return staged_h2d(p, host, count * sizeof(T), s);
    }
    int download(T *host, size_t count, hipStream_t s) const {
        if (count > n) return fail(DNS_ERR_BAD_ARGUMENT, "download overflow");
        if (count == 0) return DNS_OK;
        log_host_copy("download", host, p, count * sizeof(T));
        // (through the page-locked bounce buffer: complete on return)
        return staged_d2h(host, p, count * sizeof(T), s);
    }
    int zero(hipStream_t s) {
        DNS_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
        return DNS_OK;
    }
};

// host -> device into the middle of a buffer, with the guarantee of
// DevBuf::upload: the copy has left `host` when the call returns
template <typename T>
inline int upload_to(T *dev, const T *host, size_t count, hipStream_t s) {
    if (count == 0) return DNS_OK;
    log_host_copy("upload_to", host, dev, count * sizeof(T));
    return staged_h2d(dev, host, count * sizeof(T), s);
}

// page-locked host staging (hipHostMalloc): copies between it and the device
// never pin or unpin caller memory
template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
    int reserve(size_t count) {
        if (count <= n) return DNS_OK;
        release();
        DNS_HIP(hipHostMalloc(reinterpret_cast<void **>(&p),
                              count * sizeof(T), hipHostMallocDefault));
        n = count;
        return DNS_OK;
    }
};

// synchronises the stream when the scope is left, whichever way: what makes a
// group of `upload_async` calls from buffers of that scope safe
struct SyncOnExit {
    hipStream_t s;
    explicit SyncOnExit(hipStream_t s_) : s(s_) {}
    SyncOnExit(const SyncOnExit &) = delete;
    SyncOnExit &operator=(const SyncOnExit &) = delete;
    ~SyncOnExit() { (void)hipStreamSynchronize(s); }
};

// CSR matrix resident in HBM
struct CsrDev {
    int nrows = 0, ncols = 0;
    int64_t nnz = 0;
    int lpr = 16;                 // lanes per row of the vector kernel
    DevBuf<int> rowptr, colidx;
    DevBuf<double> vals;
    // fp32 copy of the values (operators of the PRECONDITIONER in the
    // bandwidth regime: 6 instead of 10 bytes per non-zero cross the HBM)
    DevBuf<float> vals32;
    // row-block table of the LDS-streaming kernel (built on the host)
    DevBuf<int> rowblocks_t[3];   // tiles of 1024 / 2048 / 4096 non-zeros
    int nrowblocks_t[3] = {0, 0, 0};
    // 16-bit column indices for the 2048 tile: entry = 15-bit offset from one
    // of two bases of its row block (bit 15 selects); c16base[2b] < 0 marks a
    // block whose columns do not fit two 32768-wide windows (read raw)
    DevBuf<unsigned short> c16;
    DevBuf<int> c16base;
    // per row block of the 2048 tile: r0, nr, k0, nn, blo, bhi, 0, 0 -- what
    // the 16-bit streaming kernels read INSTEAD of rowblocks / rowptr / c16base
    // at the head of a tile (passed in the `rowblocks` argument)
    DevBuf<int> meta16;
    int c16_rawblocks = 0;

    int upload(const dns_csr *a, hipStream_t s);
    void release_all() {
        rowptr.release();
        colidx.release();
        vals.release();
        vals32.release();
        for (int t = 0; t < 3; ++t) {
            rowblocks_t[t].release();
            nrowblocks_t[t] = 0;
        }
        c16.release();
        c16base.release();
        meta16.release();
        nrows = ncols = 0;
        nnz = 0;
    }
};

inline int pick_lpr(double avg_nnz_per_row) {
    // smallest power of two that covers an average row in ONE pass of the
    // sub-wave: the reference-size systems are latency bound and every extra
    // pass is another dependent (col,val) -> x[col] round trip
    int lpr = 2;
    while (lpr < 64 && lpr < avg_nnz_per_row) lpr *= 2;
    return lpr;
}

inline int check_csr(const dns_csr *a, const char *name) {
    if (!a || !a->rowptr || (a->nnz > 0 && (!a->colidx || !a->vals)))
        return fail(DNS_ERR_BAD_ARGUMENT, "%s: null CSR arrays", name);
    if (a->nrows < 0 || a->ncols < 0 || a->nnz < 0)
        return fail(DNS_ERR_BAD_ARGUMENT, "%s: negative sizes", name);
    if (a->rowptr[0] != 0 || a->rowptr[a->nrows] != a->nnz)
        return fail(DNS_ERR_BAD_ARGUMENT, "%s: rowptr does not span nnz",
                    name);
    for (int i = 0; i < a->nrows; ++i)
        if (a->rowptr[i + 1] < a->rowptr[i])
            return fail(DNS_ERR_BAD_ARGUMENT, "%s: rowptr not monotone", name);
    for (int64_t k = 0; k < a->nnz; ++k)
        if (a->colidx[k] < 0 || a->colidx[k] >= a->ncols)
            return fail(DNS_ERR_BAD_ARGUMENT,
                        "%s: column index %d out of range at %lld", name,
                        a->colidx[k], (long long)k);
    return DNS_OK;
}

}  // namespace dns
