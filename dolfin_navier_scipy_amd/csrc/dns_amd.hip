// C-ABI implementation of include/dns_amd.h -- single translation unit for
// hipcc --offload-arch=gfx950.
#include "imex.hpp"
#include "solver.hpp"

using namespace dns;

namespace dns {

__global__ void __launch_bounds__(kBlock)
k_set_unit(int n, int idx, double val, double *__restrict__ x) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        x[i] = (i == idx) ? val : 0.0;
}

// sdinv[i] = 1 / sum_k J[i,k]^2 * dinv[k]   (diagonal of J D^-1 JT)
__global__ void __launch_bounds__(kBlock)
k_schur_diag(int np, const int *__restrict__ rowptr,
             const int *__restrict__ colidx, const double *__restrict__ vals,
             const double *__restrict__ dinv, double *__restrict__ sdinv) {
    for (int row = blockIdx.x * kBlock + threadIdx.x; row < np;
         row += gridDim.x * kBlock) {
        double s = 0.0;
        for (int k = rowptr[row]; k < rowptr[row + 1]; ++k)
            s = fma(vals[k] * vals[k], dinv[colidx[k]], s);
        sdinv[row] = (s > 0.0) ? 1.0 / s : 1.0;
    }
}

static void host_transpose(const dns_csr *a, std::vector<int> &rp,
                           std::vector<int> &ci, std::vector<double> &va) {
    rp.assign((size_t)a->ncols + 1, 0);
    ci.resize((size_t)a->nnz);
    va.resize((size_t)a->nnz);
    for (int64_t k = 0; k < a->nnz; ++k) rp[a->colidx[k] + 1]++;
    for (int c = 0; c < a->ncols; ++c) rp[c + 1] += rp[c];
    std::vector<int> pos(rp.begin(), rp.end() - 1);
    for (int r = 0; r < a->nrows; ++r)
        for (int k = a->rowptr[r]; k < a->rowptr[r + 1]; ++k) {
            const int c = a->colidx[k];
            ci[pos[c]] = r;
            va[pos[c]] = a->vals[k];
            pos[c]++;
        }
}

}  // namespace dns

struct dns_dist_data;
static void free_dist_data(dns_dist_data *d);

dns_saddle::~dns_saddle() {
    drop_graphs();
    free_dist_data(dd);
    if (hdr_host) (void)hipHostFree(hdr_host);
    if (scal_host) (void)hipHostFree(scal_host);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
}

// device, stream, knobs read once, pinned scratch (nv, np are set)
int dns_saddle::init_device(int dev) {
    device = dev;
    n = nv + np;
    ld = ((size_t)n + 63) / 64 * 64;
    DNS_HIP(hipSetDevice(device));
    // DNS_CU_MASK (experiment knob, read when the handle is created)
    // restricts the handle's stream to a subset of the CUs: "8=k" = the CUs i
    // with i % 8 == k, "8<k" = i % 8 < k.  Measured at N=2: a time step runs as
    // fast on the 32 CUs of "8=0" as on all 256 (it is latency bound), so
    // several simulations can share one GPU (scripts/xcd_ensemble.py: 1 / 2 /
    // 4 concurrent simulations advance 16.0k / 26.5k / 37.7k steps/s in total;
    // 8 host threads fall back to 22k -- launch-side contention).
    if (const char *cm = getenv("DNS_CU_MASK")) {
        int mod = 0, arg = 0;
        char op = 0;
        if (sscanf(cm, "%d%c%d", &mod, &op, &arg) != 3 || mod < 1 ||
            (op != '=' && op != '<') || arg < 0)
            return fail(DNS_ERR_BAD_ARGUMENT, "bad DNS_CU_MASK '%s'", cm);
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int count = 0;
        for (int i = 0; i < 256; ++i) {
            const bool on = (op == '=') ? (i % mod) == arg : (i % mod) < arg;
            if (on) {
                mask[i / 32] |= (1u << (i % 32));
                ++count;
            }
        }
        if (count == 0) return fail(DNS_ERR_BAD_ARGUMENT, "empty DNS_CU_MASK");
        DNS_HIP(hipExtStreamCreateWithCUMask(&stream, 8, mask));
    } else {
        DNS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }
    if (const char *sn = getenv("DNS_STREAM_NNZ")) stream_nnz = atoll(sn);
    if (const char *sn = getenv("DNS_MG_STREAM_NNZ")) mg_stream_nnz = atoll(sn);
    if (const char *sn = getenv("DNS_STREAM_GRID"))
        sgrid = std::max(64, std::min(atoi(sn), 16384));
    if (const char *sn = getenv("DNS_MG_DENSE_MAX")) mg_dense_max = atoi(sn);
    if (const char *sn = getenv("DNS_MG_DENSE_HALF_MAX"))
        mg_dense_half_max = atoi(sn);
    if (const char *sn = getenv("DNS_OVERSOLVE")) oversolve_env = sn[0] != '0';
    if (const char *sn = getenv("DNS_OVERSOLVE_CMIN"))
        oversolve_cmin_env = std::max(1, atoi(sn));
    if (const char *sn = getenv("DNS_OVERSOLVE_FRAC"))
        oversolve_frac = atof(sn);
    if (const char *sn = getenv("DNS_OVERSOLVE_RAISE"))
        oversolve_raise = atof(sn);
    if (const char *sn = getenv("DNS_OVERSOLVE_LOWER"))
        oversolve_lower = atof(sn);
    if (const char *sn = getenv("DNS_MG_FUSED")) mg_fused_knob = sn[0] != '0';
    if (const char *sn = getenv("DNS_MG_PART_MIN")) mg_part_min = atoi(sn);
    if (const char *sn = getenv("DNS_PAIR")) pair_knob = sn[0] != '0';
    if (const char *sn = getenv("DNS_PART_SETUP")) part_setup = sn[0] != '0';
    if (const char *sn = getenv("DNS_MG_ROWS")) mg_rows_knob = sn[0] != '0';
    if (const char *sn = getenv("DNS_DIST_GRAPH")) dist_graph_ok = sn[0] != '0';
    if (const char *sn = getenv("DNS_MG_CHEB")) mg_cheb = sn[0] != '0';
    if (const char *sn = getenv("DNS_MG_CYCLES"))
        mg_cycles_knob = std::max(0, std::min(2, atoi(sn)));
    if (const char *sn = getenv("DNS_MG_RHO"))
        mg_rho = std::max(0.01, std::min(0.95, atof(sn)));
    if (const char *sn = getenv("DNS_MG_CHEB_ALPHA"))
        mg_cheb_alpha = std::max(1.5, atof(sn));
    DNS_HIP(hipEventCreate(&ev0));
    DNS_HIP(hipEventCreate(&ev1));
    DNS_HIP(hipHostMalloc(reinterpret_cast<void **>(&hdr_host),
                          sizeof(CtlHeaderAcc)));
    DNS_HIP(hipHostMalloc(reinterpret_cast<void **>(&scal_host),
                          16 * sizeof(double)));
    return DNS_OK;
}

int dns_saddle::init(int dev, const dns_csr *f, const dns_csr *j,
                     const dns_csr *jt) {
    DNS_TRY(check_csr(f, "F"));
    DNS_TRY(check_csr(j, "J"));
    if (f->nrows != f->ncols)
        return fail(DNS_ERR_BAD_ARGUMENT, "F must be square");
    if (j->ncols != f->nrows)
        return fail(DNS_ERR_BAD_ARGUMENT, "J has %d columns, F has %d rows",
                    j->ncols, f->nrows);
    nv = f->nrows;
    np = j->nrows;
    DNS_TRY(init_device(dev));
    // J^T: given or formed here
    std::vector<int> trp, tci;
    std::vector<double> tva;
    dns_csr jth;
    if (jt) {
        DNS_TRY(check_csr(jt, "JT"));
        if (jt->nrows != nv || jt->ncols != np || jt->nnz != j->nnz)
            return fail(DNS_ERR_BAD_ARGUMENT, "JT is not shaped like J^T");
        jth = *jt;
    } else {
        host_transpose(j, trp, tci, tva);
        jth.nrows = nv;
        jth.ncols = np;
        jth.nnz = j->nnz;
        jth.rowptr = trp.data();
        jth.colidx = tci.data();
        jth.vals = tva.data();
    }
    DNS_TRY(F.upload(f, stream));
    DNS_TRY(J.upload(j, stream));
    DNS_TRY(JT.upload(&jth, stream));
    Fh = host_copy(f);
    Jh = host_copy(j);
    JTh = host_copy(&jth);
    // assembled K = [[F, JT], [J, 0]]
    {
        std::vector<int> krp((size_t)n + 1), kci;
        std::vector<double> kva;
        const int64_t knnz = f->nnz + 2 * j->nnz;
        if (knnz > 0x7fffffffLL)
            return fail(DNS_ERR_BAD_ARGUMENT, "K exceeds int32 indexing");
        kci.resize((size_t)knnz);
        kva.resize((size_t)knnz);
        int64_t pos = 0;
        krp[0] = 0;
        for (int r = 0; r < nv; ++r) {
            for (int k = f->rowptr[r]; k < f->rowptr[r + 1]; ++k) {
                kci[pos] = f->colidx[k];
                kva[pos++] = f->vals[k];
            }
            for (int k = jth.rowptr[r]; k < jth.rowptr[r + 1]; ++k) {
                kci[pos] = nv + jth.colidx[k];
                kva[pos++] = jth.vals[k];
            }
            krp[r + 1] = (int)pos;
        }
        for (int r = 0; r < np; ++r) {
            for (int k = j->rowptr[r]; k < j->rowptr[r + 1]; ++k) {
                kci[pos] = j->colidx[k];
                kva[pos++] = j->vals[k];
            }
            krp[nv + r + 1] = (int)pos;
        }
        dns_csr kh;
        kh.nrows = kh.ncols = n;
        kh.nnz = knnz;
        kh.rowptr = krp.data();
        kh.colidx = kci.data();
        kh.vals = kva.data();
        DNS_TRY(K.upload(&kh, stream));
    }
    DNS_TRY(dinv.alloc((size_t)nv));
    hipLaunchKernelGGL(k_diag_inv, grid_for_elems(nv), kBlock, 0, stream, nv,
                       F.rowptr.p, F.colidx.p, F.vals.p, dinv.p);
    DNS_HIP(hipGetLastError());
    return init_workspace();
}

// control blocks, Krylov vectors, partial sums (K.lpr is set)
int dns_saddle::init_workspace() {
    nred = (int)std::min<int64_t>(1024, std::max<int64_t>(64, (n + 255) / 256));
    DNS_TRY(ctl.alloc(1));
    DNS_TRY(ctl.zero(stream));
    DNS_TRY(bctl.alloc(1));
    DNS_TRY(bctl.zero(stream));
    DNS_TRY(scal.alloc(16));
    DNS_TRY(dsum.alloc(2 * kMaxRestart + 16));
    DNS_TRY(cheb_r.alloc((size_t)nv));
    DNS_TRY(cheb_d0.alloc((size_t)nv));
    DNS_TRY(cheb_d1.alloc((size_t)nv));
    DNS_TRY(w.alloc(ld));
    DNS_TRY(z.alloc(ld));
    DNS_TRY(u.alloc(ld));
    DNS_TRY(r.alloc(ld));
    DNS_TRY(xdev.alloc(ld));
    DNS_TRY(bdev.alloc(ld));
    // GMRES grids: SpMV-shaped kernels (one partial per workgroup, summed by
    // the consumer) stay at <= 512 workgroups; vector kernels use one thread
    // per entry up to 1024 workgroups
    gridS = gridC = std::max(1, std::min(grid_for_rows(n, K.lpr), 2048));
    gridD = (int)std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock,
                                                       1024));
    const size_t pmax =
        (size_t)std::max(std::max(std::max(gridS, gridD), nred), sgrid);
    DNS_TRY(partA.alloc((size_t)(kMaxRestart + 2) * pmax));
    DNS_TRY(partE.alloc((size_t)(kMaxRestart + 1) * pmax));
    DNS_TRY(partN.alloc(pmax));
    DNS_TRY(partR.alloc(pmax));
    DNS_TRY(partB.alloc(pmax));
    DNS_TRY(partC.alloc(pmax));
    DNS_TRY(u.zero(stream));
    DNS_HIP(hipStreamSynchronize(stream));
    dns_default_precond_opts(&popts);
    return DNS_OK;
}

// the pair format of K for the K applies of the bandwidth regime (one GPU);
// silently absent for odd sizes
int dns_saddle::build_pair() {
    if (comm) return DNS_OK;         // (the row block's: setup_dist)
    Kp.release_all();
    if (!pair_knob || !streams(K)) return DNS_OK;
    if (fh_stale) return DNS_OK;     // (host values not current)
    const HostCsr kf = host_k_slice(Fh, JTh, Jh, nv, 0, nv, 0, np);
    HostPair P;
    if (!host_pair_from_k(kf, nv, P)) return DNS_OK;
    DNS_TRY(Kp.upload(P, stream));
    return DNS_OK;
}

int dns_saddle::update_values(const double *fvals) {
    DNS_HIP(hipSetDevice(device));
    if (dist_sliced || rank_local) {
        // the device holds this rank's rows of K only: new values through the
        // host copy (created from rows: `fvals` are the values of the own
        // rows, which is all Fh holds)
        Fh.vals.assign(fvals, fvals + Fh.vals.size());
        fh_stale = false;
        drop_graphs();
        if (!dist_sliced) return DNS_OK;      // (before the first set-up)
        return update_values_dist();
    }
    Fh.vals.assign(fvals, fvals + F.nnz);
    fh_stale = false;
    if (Kp.ready) drop_graphs();     // (they launch the pair kernel)
    Kp.release_all();                // copies of the values: next set-up
    DNS_TRY(F.vals.upload(fvals, (size_t)F.nnz, stream));
    hipLaunchKernelGGL(k_scatter_fvals, grid_for_rows(nv, 8), kBlock, 0,
                       stream, nv, F.rowptr.p, K.rowptr.p, F.vals.p, K.vals.p);
    hipLaunchKernelGGL(k_diag_inv, grid_for_elems(nv), kBlock, 0, stream, nv,
                       F.rowptr.p, F.colidx.p, F.vals.p, dinv.p);
    DNS_HIP(hipGetLastError());
    DNS_HIP(hipStreamSynchronize(stream));   // fvals is borrowed
    return DNS_OK;
}

// F.vals was re-valued by a device kernel (trapezoidal stepper): bring K and
// D^-1 along; the host copy is fetched lazily before the next set-up
int dns_saddle::ensure_F_device() {
    if (rank_local)
        return fail(DNS_ERR_BAD_ARGUMENT,
                    "not available on a handle created from rows (the whole F "
                    "is nowhere)");
    if (F.nnz > 0 || !dist_sliced) return DNS_OK;
    // (released when the handle was sliced; the trapezoidal stepper assembles
    // F = M + dt/2 (A + N) in full on every rank and scatters its rows into
    // the rank's row block of K)
    const dns_csr fv = Fh.view();
    DNS_TRY(F.upload(&fv, stream));
    return DNS_OK;
}

int dns_saddle::device_values_changed(bool k_current, bool own_rows_only) {
    fvals_own_rows_only = own_rows_only;
    if (dist_sliced) {
        if (!dd || F.nnz == 0)
            return fail(DNS_ERR_NOT_READY, "sliced handle without its F block");
        const int v0 = dist_v0(), v1 = dist_v1();
        if (Kp.ready) drop_graphs();     // (they launch the pair kernel)
        Kp.release_all();
        if (v1 > v0)
            hipLaunchKernelGGL(k_scatter_fvals_rows, grid_for_rows(v1 - v0, 8),
                               kBlock, 0, stream, v0, v1 - v0, F.rowptr.p,
                               K.rowptr.p, F.vals.p, K.vals.p);
        // (1/diag(F) is read by the recurrence form of Fh^-1 and by the
        // set-up only: with the explicit polynomial it is refreshed lazily --
        // the kernel hunts the diagonal of every row, 19.5 us at n = 1e4)
        if (fhat_explicit && precond_ready)
            dinv_stale = true;
        else
            hipLaunchKernelGGL(k_diag_inv, grid_for_elems(nv), kBlock, 0,
                               stream, nv, F.rowptr.p, F.colidx.p, F.vals.p,
                               dinv.p);
        DNS_HIP(hipGetLastError());
        fh_stale = true;
        return DNS_OK;
    }
    if (Kp.ready) drop_graphs();
    Kp.release_all();
    if (!k_current)
        hipLaunchKernelGGL(k_scatter_fvals, grid_for_rows(nv, 8), kBlock, 0,
                           stream, nv, F.rowptr.p, K.rowptr.p, F.vals.p,
                           K.vals.p);
    if (fhat_explicit && precond_ready)
        dinv_stale = true;
    else
        hipLaunchKernelGGL(k_diag_inv, grid_for_elems(nv), kBlock, 0, stream,
                           nv, F.rowptr.p, F.colidx.p, F.vals.p, dinv.p);
    DNS_HIP(hipGetLastError());
    fh_stale = true;
    return DNS_OK;
}

int dns_saddle::dot_host(int64_t len, const double *x, const double *y,
                         double *out) {
    hipLaunchKernelGGL(k_dot_partials, nred, kBlock, 0, stream, len, x, y,
                       partC.p);
    hipLaunchKernelGGL(k_sum_partials, 1, kBlock, 0, stream, partC.p, nred,
                       scal.p);
    DNS_HIP(hipGetLastError());
    DNS_HIP(hipMemcpyAsync(scal_host, scal.p, sizeof(double),
                           hipMemcpyDeviceToHost, stream));
    DNS_HIP(hipStreamSynchronize(stream));
    *out = scal_host[0];
    return DNS_OK;
}

int dns_saddle::estimate_bounds() {
    // power iterations on D^-1 F; same sequence as tests/krylov_model.py
    double *x = cheb_d0.p, *y = cheb_d1.p;
    const int g = grid_for_elems(nv);
    double nrm2 = 0, lam = 0;
    hipLaunchKernelGGL(k_fill_wave, g, kBlock, 0, stream, nv, x, 0.37, 7.0, 0);
    for (int it = 0; it < 30; ++it) {
        DNS_TRY(launch_spmv(F, x, y, 1.0, 0.0, nullptr, DNS_SPMV_VECTOR,
                            stream));
        hipLaunchKernelGGL(k_scale_by, g, kBlock, 0, stream, nv, dinv.p, y);
        double xx, yy;
        DNS_TRY(dot_host(nv, x, x, &xx));
        DNS_TRY(dot_host(nv, y, y, &yy));
        if (!(yy > 0.0) || !(xx > 0.0))
            return fail(DNS_BREAKDOWN, "power iteration broke down");
        lam = std::sqrt(yy / xx);
        hipLaunchKernelGGL(k_axpby, g, kBlock, 0, stream, (int64_t)nv,
                           1.0 / std::sqrt(yy), y, 0.0, x);
    }
    const double lmax = lam;
    const double shift = 1.05 * lmax;
    hipLaunchKernelGGL(k_fill_wave, g, kBlock, 0, stream, nv, x, 0.61, 7.0, 1);
    double mu = 0;
    for (int it = 0; it < 50; ++it) {
        DNS_TRY(launch_spmv(F, x, y, 1.0, 0.0, nullptr, DNS_SPMV_VECTOR,
                            stream));
        hipLaunchKernelGGL(k_scale_by, g, kBlock, 0, stream, nv, dinv.p, y);
        // y = shift*x - y
        hipLaunchKernelGGL(k_axpby, g, kBlock, 0, stream, (int64_t)nv, shift,
                           x, -1.0, y);
        double xx, yy;
        DNS_TRY(dot_host(nv, x, x, &xx));
        DNS_TRY(dot_host(nv, y, y, &yy));
        if (!(yy > 0.0) || !(xx > 0.0)) break;
        mu = std::sqrt(yy / xx);
        hipLaunchKernelGGL(k_axpby, g, kBlock, 0, stream, (int64_t)nv,
                           1.0 / std::sqrt(yy), y, 0.0, x);
    }
    (void)nrm2;
    double lmin = shift - mu;
    if (!(lmin > 0.0)) lmin = lmax / 100.0;
    lam_lo = popts.eig_lo_safety * lmin;
    lam_hi = popts.eig_hi_safety * lmax;
    return DNS_OK;
}

// Chebyshev recurrence for zv = Fh^-1 (rv - JT zp)  (one GPU; the
// row-partitioned solve uses the explicit polynomial matrix)
static int apply_fhat(dns_saddle *h, const double *rbase, size_t ldr,
                      const int *jsel, const double *zp, double *zv,
                      const int *guard) {
    const int deg = h->popts.cheb_degree;
    const int v0 = 0, v1 = h->nv;
    const int gj = grid_for_rows(v1 - v0, h->JT.lpr);
    double *d0 = (deg == 1) ? zv : h->cheb_d0.p;
    DNS_LPR_SWITCH(h->JT.lpr,
                   hipLaunchKernelGGL(k_cheb_init<L>, gj, kBlock, 0, h->stream,
                                      h->nv, h->JT.rowptr.p, h->JT.colidx.p,
                                      h->JT.vals.p, zp, rbase, ldr, jsel,
                                      h->dinv.p, 1.0 / h->theta, h->cheb_r.p,
                                      d0, guard, v0, v1));
    const int gf = grid_for_rows(v1 - v0, h->F.lpr);
    double *dbuf[2] = {h->cheb_d0.p, h->cheb_d1.p};
    for (int s = 0; s < deg - 1; ++s) {
        const int first = (s == 0), last = (s == deg - 2);
        DNS_LPR_SWITCH(
            h->F.lpr,
            hipLaunchKernelGGL(k_cheb_step<L>, gf, kBlock, 0, h->stream, h->nv,
                               h->F.rowptr.p, h->F.colidx.p, h->F.vals.p,
                               h->dinv.p, dbuf[s & 1], dbuf[(s + 1) & 1],
                               h->cheb_r.p, zv, h->c1[s], h->c2[s], first,
                               last, guard, v0, v1));
        h->spmv_count++;
    }
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

// velocity part of the preconditioner:  zv = Fh^-1 (rv - JT zp)
// (`xacc`: also x_v += zv, the fused correction at the end of a cycle)
int dns_saddle::apply_fhat_part(const double *rvec, const double *zp,
                                double *zv, const int *guard, double *xacc) {
    bool xacc_fused = false;
    if (fhat_explicit && streams(Gc)) {
        // bandwidth regime: the LDS-streaming kernel with 16-bit column
        // offsets (6 instead of 8 bytes per non-zero with fp32 values); its
        // input [rv ; zp] is read from the two buffers directly
        StreamEpi ep = stream_epi_plain(1.0, 0.0, nullptr);
        ep.x2 = zp;
        ep.nsplit = nv;
        if (fp32_store)
            DNS_TRY(launch_stream16x<float>(Gc, gc32.p, rvec, zv, ep, stream,
                                            guard));
        else
            DNS_TRY(launch_stream16x<double>(Gc, Gc.vals.p, rvec, zv, ep,
                                             stream, guard));
    } else if (fhat_explicit) {
        const int r0 = 0, r1 = nv;
        const int g = grid_for_rows(r1 - r0, Gc.lpr);
        double *xa = xacc;
        xacc_fused = xa != nullptr;
        if (fp32_store) {
            DNS_LPR_SWITCH(
                Gc.lpr,
                hipLaunchKernelGGL((k_spmv_split<L, float>), g, kBlock, 0,
                                   stream, nv, Gc.rowptr.p, Gc.colidx.p,
                                   gc32.p, rvec, (size_t)0, zero_ptr(), zp, zv,
                                   guard, xa, r0, r1));
        } else {
            DNS_LPR_SWITCH(
                Gc.lpr,
                hipLaunchKernelGGL((k_spmv_split<L, double>), g, kBlock, 0,
                                   stream, nv, Gc.rowptr.p, Gc.colidx.p,
                                   Gc.vals.p, rvec, (size_t)0, zero_ptr(), zp,
                                   zv, guard, xa, r0, r1));
        }
        DNS_HIP(hipGetLastError());
        if (dist()) DNS_TRY(comm->allgatherv(zv, st_v, stream));
    } else {
        DNS_TRY(apply_fhat(this, rvec, 0, zero_ptr(), zp, zv, guard));
    }
    if (xacc && !xacc_fused) {
        hipLaunchKernelGGL(k_axpby, grid_for_elems(nv), kBlock, 0, stream,
                           (int64_t)nv, 1.0, zv, 1.0, xacc);
        DNS_HIP(hipGetLastError());
    }
    return DNS_OK;
}

// z = P^-1 r for one n-vector r (device pointers); xacc != null: x += z
int dns_saddle::apply_precond(const double *rvec, double *zout,
                              const int *guard, double *xacc) {
    // (one GPU: BiCGStab, the parity entry and the probes; the row-partitioned
    // solve applies the preconditioner inside enqueue_cycle_dist)
    if (dist_sliced)
        return fail(DNS_ERR_BAD_ARGUMENT,
                    "not available on a row-partitioned handle");
    double *zp = zout + nv;
    double *xp = xacc ? xacc + nv : nullptr;
    double *xpf = xp;                        // fused x_p += z_p
    const int r0 = 0, r1 = np;
    const double *rvec_schur = rvec;
    if (have_jg) {
        // full block factorisation: tau = r_p - (J Fh^-1) r_v is what the
        // Schur block sees; its kernels read `rbase + nv`
        DNS_LPR_SWITCH(
            JG.lpr,
            hipLaunchKernelGGL(k_tau_guard<L>,
                               grid_for_rows(r1 - r0, JG.lpr == 64 ? 128 : JG.lpr),
                               kBlock, 0, stream, np, nv, JG.rowptr.p,
                               JG.colidx.p, JG.vals.p, rvec, tau.p,
                               (const double *)nullptr, 0, 0, ctl.p, r0, r1));
        rvec_schur = tau.p - nv;
    }
    if (popts.schur == DNS_SCHUR_MG) {
        DNS_TRY(schur_mg_apply(rvec_schur + nv, zp, xp, guard));
        return apply_fhat_part(rvec, zp, zout, guard, xacc);
    }
    if (popts.schur == DNS_SCHUR_DENSE) {
        const int g = std::max(1, std::min(r1 - r0, 2048));
        if (fp32_store)
            hipLaunchKernelGGL(k_schur_dense<float>, g, kBlock, 0, stream, np,
                               sinv32.p, rvec_schur, (size_t)0, zero_ptr(), nv, zp,
                               guard, xpf, r0, r1, sld);
        else
            hipLaunchKernelGGL(k_schur_dense<double>, g, kBlock, 0, stream, np,
                               sinv.p, rvec_schur, (size_t)0, zero_ptr(), nv, zp,
                               guard, xpf, r0, r1, np);
    } else {
        hipLaunchKernelGGL(k_schur_jacobi, grid_for_elems(r1 - r0), kBlock, 0,
                           stream, np, sinv.p, rvec_schur, (size_t)0, zero_ptr(), nv,
                           zp, guard, xpf, r0, r1);
    }
    return apply_fhat_part(rvec, zp, zout, guard, xacc);
}

// the dense Schur inverse as fp32 with padded, 16-byte aligned rows
int dns_saddle::schur_to_f32() {
    sld = (np + 3) & ~3;
    DNS_TRY(sinv32.alloc((size_t)np * sld));
    hipLaunchKernelGGL(k_to_f32_rows, grid_for_elems((int64_t)np * sld), kBlock,
                       0, stream, np, np, sld, sinv.p, sinv32.p);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

int dns_saddle::to_f32(const double *in, DevBuf<float> &out, size_t count) {
    DNS_TRY(out.alloc(count));
    hipLaunchKernelGGL(k_to_f32, grid_for_elems((int64_t)count), kBlock, 0,
                       stream, (int64_t)count, in, out.p);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

// Explicit polynomial approximate inverse, formed on the host at set-up:
//   G = p(D^-1 F) D^-1 with p the degree-(k-1) Chebyshev polynomial of
//   k_cheb_init/k_cheb_step (so G b == the recurrence applied to b),
//   Gc = [G, -G JT],  S = J G JT (dense Schur complement of the SAME Fh).
static double wall_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int dns_saddle::build_explicit(bool dense_schur) {
    const bool dbg = getenv("DNS_DEBUG") != nullptr;
    double t0 = wall_now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const double t1 = wall_now();
        fprintf(stderr, "[dns] setup %-28s %8.1f ms\n", what, 1e3 * (t1 - t0));
        t0 = t1;
    };
    if (rank_local) {
        // the rows of F and J^T this rank's rows of the polynomial reach
        HostCsr DF, JTx;
        std::vector<double> dvx;
        DNS_TRY(extend_rows_for_setup(DF, JTx, dvx));
        lap("ghost rows of F, JT");
        // convection-dominated F (setup_precond): refused here, by a bound of
        // the skew radius from the own rows and their first ring
        DNS_TRY(skew_bound_rows(DF, &skew_eta));
        if (skew_eta * skew_eta > 0.25 * lam_lo * lam_hi)
            return fail(DNS_ERR_BAD_ARGUMENT,
                        "convection-dominated F (bound of the skew radius "
                        "%.2e against Chebyshev bounds [%.2e, %.2e]): the "
                        "polynomial of the symmetric part needs a handle "
                        "created from whole matrices",
                        skew_eta, lam_lo, lam_hi);
        host_scale_rows(dvx, DF);
        tmp_presliced = false;
        return build_explicit_part(dense_schur, DF, dvx, JTx);
    }
    const HostCsr &Fsrc = pc_sym ? Fpc_h : Fh;
    std::vector<double> dv((size_t)nv, 1.0);
    for (int i = 0; i < nv; ++i) {
        double d = 0.0;
        for (int k = Fsrc.rowptr[i]; k < Fsrc.rowptr[i + 1]; ++k)
            if (Fsrc.colidx[k] == i) d += Fsrc.vals[k];
        dv[i] = (d != 0.0) ? 1.0 / d : 1.0;
    }
    HostCsr DF = Fsrc;
    host_scale_rows(dv, DF);
    tmp_presliced = false;
    if (comm && part_setup)
        return build_explicit_part(dense_schur, DF, dv, JTh);
    // (with drop_tol == 0 the zero entries of the pattern of F^(k-1) stay)
    HostCsr G = host_cheb_poly(DF, dv, theta, c1, c2, popts.drop_tol);
    lap("polynomial G (row recurrences)");
    HostCsr GJT = host_spgemm(G, JTh);
    lap("G*JT");
    HostCsr negGJT = GJT;
    for (auto &v : negGJT.vals) v = -v;
    HostCsr Gch = host_hstack(G, negGJT);
    if (!comm) {
        dns_csr gv = Gch.view();
        DNS_TRY(Gc.upload(&gv, stream));
        if (fp32_store) DNS_TRY(to_f32(Gc.vals.p, gc32, (size_t)Gc.nnz));
    }
    lap("hstack + upload Gc");
    HostCsr JGh;
    if (popts.factorization == DNS_FACT_FULL) {
        JGh = host_spgemm(Jh, G);
        if (!comm) {
            dns_csr jv = JGh.view();
            DNS_TRY(JG.upload(&jv, stream));
            if (fp32_store && streams(JG))
                DNS_TRY(to_f32(JG.vals.p, JG.vals32, (size_t)JG.nnz + 2));
        }
        DNS_TRY(tau.alloc((size_t)std::max(1, np)));
        have_jg = true;
        lap("J*G (full factorisation)");
        if (dbg)
            fprintf(stderr, "[dns] setup nnz(JG) = %ld\n",
                    (long)JGh.vals.size());
    }
    if (dbg)
        fprintf(stderr, "[dns] setup nnz(G) = %ld, nnz(Gc) = %ld\n",
                (long)G.vals.size(), (long)Gc.nnz);
    if (dense_schur) {
        HostCsr S = host_spgemm(Jh, GJT);
        lap("S = J*(G*JT)");
        std::vector<double> sd((size_t)np * np, 0.0);
        for (int i = 0; i < np; ++i)
            for (int k = S.rowptr[i]; k < S.rowptr[i + 1]; ++k)
                sd[(size_t)i * np + S.colidx[k]] = S.vals[k];
        DNS_TRY(sinv.alloc((size_t)np * np));
        DNS_TRY(sinv.upload(sd.data(), sd.size(), stream));
        DNS_HIP(hipStreamSynchronize(stream));
        lap("densify + upload S");
        DNS_TRY(invert_dense(sinv.p, np));
        lap("Gauss-Jordan inverse");
    } else if (popts.schur == DNS_SCHUR_MG) {
        HostCsr S = host_spgemm(Jh, GJT);
        lap("S = J*(G*JT)");
        DNS_TRY(build_mg_schur(S));
        lap("multigrid hierarchy");
    }
    if (comm) {
        // row-partitioned: setup_precond slices these once the rest of the
        // set-up (which works on whole matrices) is through
        tmp_Gch = std::move(Gch);
        tmp_JGh = std::move(JGh);
    }
    return DNS_OK;
}

int dns_saddle::invert_dense(double *a, int nn) {
    DevBuf<double> prow, pcol;
    DevBuf<int> flag;
    DNS_TRY(prow.alloc((size_t)nn));
    DNS_TRY(pcol.alloc((size_t)nn));
    DNS_TRY(flag.alloc(1));
    DNS_TRY(flag.zero(stream));
    const int g1 = grid_for_elems(nn);
    const int g2 = grid_for_elems((int64_t)nn * nn);
    for (int k = 0; k < nn; ++k) {
        hipLaunchKernelGGL(k_gj_pivot, g1, kBlock, 0, stream, nn, k, a, prow.p,
                           pcol.p, flag.p);
        hipLaunchKernelGGL(k_gj_update, g2, kBlock, 0, stream, nn, k, a,
                           prow.p, pcol.p, flag.p);
    }
    DNS_HIP(hipGetLastError());
    int hflag = 0;
    DNS_HIP(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost,
                           stream));
    DNS_HIP(hipStreamSynchronize(stream));
    if (hflag)
        return fail(DNS_BREAKDOWN, "zero pivot in the dense Schur inverse");
    return DNS_OK;
}

int dns_saddle::build_dense_schur() {
    // S[c, :] = J Fh^-1 JT e_c  (S is symmetric for symmetric F)
    DNS_TRY(sinv.alloc((size_t)np * np));
    DNS_TRY(u.zero(stream));
    DNS_TRY(ctl.zero(stream));
    double *zp = z.p + nv;
    for (int c = 0; c < np; ++c) {
        hipLaunchKernelGGL(k_set_unit, grid_for_elems(np), kBlock, 0, stream,
                           np, c, -1.0, zp);
        DNS_TRY(apply_fhat(this, u.p, 0, zero_ptr(), zp, z.p, zero_ptr()));
        DNS_TRY(launch_spmv(J, z.p, sinv.p + (size_t)c * np, 1.0, 0.0, nullptr,
                            DNS_SPMV_VECTOR, stream));
    }
    DNS_HIP(hipGetLastError());
    return invert_dense(sinv.p, np);
}

// Multigrid hierarchy of the Schur block: S_0 given (sparse), Galerkin coarse
// operators through the prolongations handed over by dns_saddle_set_schur_mg,
// dense inverse on the coarsest level.
int dns_saddle::mg_prepare(int n0) {
    mg_ready = false;
    if (!mg_set)
        return fail(DNS_ERR_NOT_READY,
                    "DNS_SCHUR_MG needs dns_saddle_set_schur_mg first");
    if (!mg_prol_h.empty() && mg_prol_h[0].nrows != np)
        return fail(DNS_ERR_BAD_ARGUMENT,
                    "finest prolongation has %d rows, NP = %d",
                    mg_prol_h[0].nrows, np);
    // the coarsest level gets a dense inverse: stop at the FIRST level small
    // enough for one.  (Measured: a 5000-dof level as a 100 MB fp32 inverse
    // instead of two more sparse levels does not pay -- refine 2: 2145 vs
    // 2172 steps/s, refine 3: 568 vs 702 with 2.6 instead of 2.45 Krylov steps
    // per time step -- hence the small default, DNS_MG_DENSE_MAX.)
    const int dense_max = std::max(mg_dense_max,
                                   std::min(mg_dense_half_max, 16384));
    int L = (int)mg_prol_h.size() + 1;
    {
        int nl = n0;
        for (int l = 0; l + 1 < L; ++l) {
            if (nl <= dense_max) {
                L = l + 1;
                break;
            }
            nl = mg_prol_h[l].ncols;
        }
    }
    mg.clear();
    for (int l = 0; l < L; ++l) mg.emplace_back();
    // two sweeps each way: the cycle runs on fused operators (9 launches for
    // three levels instead of 16); DNS_MG_FUSED=0 keeps the plain cycle
    mg_fused = mg_nu == 2 && mg_fused_knob;
    mg_fused11 = mg_nu == 1 && mg_fused_knob && !comm && mg_cycles < 2;
    mg_rows_parts = 0;
    return DNS_OK;
}

int dns_saddle::build_mg_schur(const HostCsr &S0) {
    DNS_TRY(mg_prepare(S0.nrows));
    return build_mg_levels(0, S0);
}

// levels l0 .. of the hierarchy from the WHOLE operator of level l0
int dns_saddle::build_mg_levels(int l0, HostCsr Sl) {
    const int L = (int)mg.size();
    for (int l = l0; l < L; ++l) {
        MgLevel &lv = mg[l];
        lv.n = Sl.nrows;
        DNS_TRY(lv.x.alloc((size_t)lv.n));
        DNS_TRY(lv.b.alloc((size_t)lv.n));
        DNS_TRY(lv.r.alloc((size_t)lv.n));
        DNS_TRY(lv.x2.alloc((size_t)lv.n));
        if (l == L - 1) {
            // coarsest: dense inverse
            std::vector<double> sd((size_t)lv.n * lv.n, 0.0);
            for (int i = 0; i < lv.n; ++i)
                for (int k = Sl.rowptr[i]; k < Sl.rowptr[i + 1]; ++k)
                    sd[(size_t)i * lv.n + Sl.colidx[k]] += Sl.vals[k];
            if ((int64_t)lv.n * lv.n * 8 > (int64_t)8 << 30)
                return fail(DNS_ERR_BAD_ARGUMENT,
                            "coarsest multigrid level too large (%d)", lv.n);
            DNS_TRY(mg_cinv.alloc(sd.size()));
            DNS_TRY(mg_cinv.upload(sd.data(), sd.size(), stream));
            DNS_HIP(hipStreamSynchronize(stream));
            DNS_TRY(invert_dense(mg_cinv.p, lv.n));
            mg_cinv16.release();
            mg_cld16 = 0;
            if (mg_dense_half_max > 0 && lv.n > mg_dense_max &&
                lv.n <= mg_dense_half_max) {
                // (a level this size is dense only in half precision)
                const int ldh = (lv.n + 7) / 8 * 8;
                const int ga = 1024;
                DevBuf<double> amax;
                DNS_TRY(amax.alloc((size_t)ga));
                DNS_TRY(mg_cscale.alloc(1));
                DNS_TRY(mg_cinv16.alloc((size_t)lv.n * ldh));
                hipLaunchKernelGGL(k_absmax, ga, kBlock, 0, stream,
                                   (int64_t)lv.n * lv.n, mg_cinv.p, amax.p);
                hipLaunchKernelGGL(k_to_half_rows,
                                   grid_for_elems((int64_t)lv.n * ldh), kBlock,
                                   0, stream, lv.n, ldh, mg_cinv.p, amax.p, ga,
                                   1024.0, mg_cinv16.p, mg_cscale.p);
                DNS_HIP(hipGetLastError());
                DNS_HIP(hipStreamSynchronize(stream));
                mg_cld16 = ldh;
                mg_cinv.release();
                mg_cinv32.release();
            } else if (fp32_store) {
                DNS_TRY(to_f32(mg_cinv.p, mg_cinv32, sd.size()));
            }
            break;
        }
        const HostCsr &P = mg_prol_h[l];
        if (P.nrows != lv.n)
            return fail(DNS_ERR_BAD_ARGUMENT,
                        "prolongation %d has %d rows, level has %d", l, P.nrows,
                        lv.n);
        std::vector<double> dv((size_t)lv.n, 1.0);
        for (int i = 0; i < lv.n; ++i) {
            double d = 0.0;
            for (int k = Sl.rowptr[i]; k < Sl.rowptr[i + 1]; ++k)
                if (Sl.colidx[k] == i) d += Sl.vals[k];
            dv[i] = (d != 0.0) ? 1.0 / d : 1.0;
        }
        {
            const double lmax = std::max(1e-300, host_jacobi_lmax(Sl));
            lv.omega = lv.omega2 = 4.0 / (3.0 * lmax);
            if (mg_cheb && mg_nu == 2) {
                // two Richardson steps at the Chebyshev roots of
                // [lmax / alpha, 1.05 lmax]
                const double hi = 1.05 * lmax, lo = lmax / mg_cheb_alpha;
                const double mid = 0.5 * (hi + lo), rad = 0.5 * (hi - lo);
                const double c = 0.70710678118654752;    // cos(pi / 4)
                lv.omega = 1.0 / (mid + rad * c);
                lv.omega2 = 1.0 / (mid - rad * c);
            }
        }
        HostCsr PT = host_transpose(P);
        dns_csr sv = Sl.view(), pv = P.view(), tv = PT.view();
        DNS_TRY(lv.S.upload(&sv, stream));
        DNS_TRY(lv.P.upload(&pv, stream));
        DNS_TRY(lv.PT.upload(&tv, stream));
        DNS_TRY(lv.dinv.alloc((size_t)lv.n));
        DNS_TRY(lv.dinv.upload(dv.data(), dv.size(), stream));
        DNS_HIP(hipStreamSynchronize(stream));
        const HostCsr SP = host_spgemm(Sl, P);
        if (mg_fused11) {
            // (solver.hpp, mg_fused11) one sweep each way, w = 4 / (3 lmax)
            lv.omega = lv.omega2 = 4.0 / (3.0 * std::max(
                                              1e-300, host_jacobi_lmax(Sl)));
            std::vector<double> wd((size_t)lv.n);
            for (int i = 0; i < lv.n; ++i) wd[i] = lv.omega * dv[i];
            const HostCsr I = host_diag(std::vector<double>((size_t)lv.n, 1.0));
            HostCsr WS = Sl;
            host_scale_rows(wd, WS);                       // w D^-1 S
            // (I + T) w D^-1 = (2 I - w D^-1 S) w D^-1
            HostCsr Ap = host_add(2.0, I, -1.0, WS);
            for (size_t k = 0; k < Ap.vals.size(); ++k)
                Ap.vals[k] *= wd[Ap.colidx[k]];
            // Rd = P^T - (P^T S) w D^-1
            HostCsr PTSw = host_spgemm(PT, Sl);
            for (size_t k = 0; k < PTSw.vals.size(); ++k)
                PTSw.vals[k] *= wd[PTSw.colidx[k]];
            const HostCsr Rd = host_add(1.0, PT, -1.0, PTSw);
            HostCsr WSP = SP;
            host_scale_rows(wd, WSP);
            HostCsr U = host_hstack(Ap, host_add(1.0, P, -1.0, WSP));
            if (l == 0)                                    // zp = -x
                for (double &v : U.vals) v = -v;
            dns_csr rv = Rd.view(), qv = U.view();
            DNS_TRY(lv.Rr.upload(&rv, stream));
            DNS_TRY(lv.Qq.upload(&qv, stream));
            if (fp32_store)
                for (CsrDev *op : {&lv.Rr, &lv.Qq})
                    if (streams_mg(*op))
                        DNS_TRY(to_f32(op->vals.p, op->vals32,
                                       (size_t)op->nnz + 2));
            DNS_HIP(hipStreamSynchronize(stream));
            lv.nnz_P = lv.P.nnz;
            lv.nnz_S = lv.S.nnz;
            lv.P.release_all();
            lv.PT.release_all();
        }
        if (mg_fused) {
            // the operators of the fused cycle (solver.hpp, MgLevel)
            // first sweep of a pair with w1 = omega, second with w2 = omega2:
            //   T   = I - w1 D^-1 S
            //   pre = ((w1 + w2) I - w2 (w1 D^-1 S)) D^-1   (two sweeps from 0)
            std::vector<double> wd((size_t)lv.n);
            for (int i = 0; i < lv.n; ++i) wd[i] = lv.omega * dv[i];
            const HostCsr I = host_diag(std::vector<double>((size_t)lv.n, 1.0));
            HostCsr WS = Sl;
            host_scale_rows(wd, WS);                       // w1 D^-1 S
            const HostCsr T = host_add(1.0, I, -1.0, WS);
            HostCsr Ap = host_add(lv.omega + lv.omega2, I, -lv.omega2, WS);
            for (size_t k = 0; k < Ap.vals.size(); ++k)    // ... D^-1
                Ap.vals[k] *= dv[Ap.colidx[k]];
            HostCsr mPTS = host_spgemm(PT, Sl);
            for (double &v : mPTS.vals) v = -v;
            const HostCsr Rr = host_hstack(PT, mPTS);
            HostCsr WSP = SP;
            host_scale_rows(wd, WSP);
            const HostCsr Qq = host_hstack(T, host_add(1.0, P, -1.0, WSP));
            dns_csr av = Ap.view(), rv = Rr.view(), qv = Qq.view();
            DNS_TRY(lv.Apre.upload(&av, stream));
            DNS_TRY(lv.Rr.upload(&rv, stream));
            DNS_TRY(lv.Qq.upload(&qv, stream));
            if (fp32_store) {
                // the cycle is part of the preconditioner: its operators cross
                // the HBM as fp32 where they are streamed (6 B per non-zero;
                // a level that is row-partitioned later uploads its row blocks,
                // which drops the copy: those levels stream fp64)
                for (CsrDev *op : {&lv.Apre, &lv.Rr, &lv.Qq, &lv.S})
                    if (streams_mg(*op))
                        DNS_TRY(to_f32(op->vals.p, op->vals32,
                                       (size_t)op->nnz + 2));
            }
            DNS_HIP(hipStreamSynchronize(stream));
            // the prolongation itself is not applied by the fused cycle
            lv.nnz_P = lv.P.nnz;
            lv.nnz_S = lv.S.nnz;
            lv.P.release_all();
            lv.PT.release_all();
            if (comm) {
                lv.Sh = Sl;
                lv.Apreh = std::move(Ap);
                lv.Rrh = Rr;
                lv.Qqh = Qq;
            }
        }
        Sl = host_spgemm(PT, SP);
    }
    mg_ready = true;
    return DNS_OK;
}

// zp = -MG(in): one V(nu, nu) cycle with damped Jacobi; xacc: x_p += zp;
// guard: device flag (solve finished) that turns every launch into a no-op
int dns_saddle::schur_mg_apply(const double *in, double *zp, double *xacc,
                               const int *guard) {
    const int L = (int)mg.size();
    if (mg_fused && L > 1) return schur_mg_apply_fused(in, zp, xacc, guard);
    if (mg_fused11 && L > 1 && !dist())
        return mg_cycle_fused11(in, zp, xacc, guard);
    // sweeps ping-pong between lv.x and lv.x2; returns where the result is
    auto smooth = [&](MgLevel &lv, const double *b, bool from_zero,
                      double *cur) -> double * {
        for (int sweep = 0; sweep < mg_nu; ++sweep) {
            double *nxt = (cur == lv.x.p) ? lv.x2.p : lv.x.p;
            const double *xin = (from_zero && sweep == 0) ? nullptr : cur;
            if (xin && !dist() && streams_mg(lv.S)) {
                // bandwidth regime: the sweep as an epilogue of the streaming
                // kernel
                StreamEpi ep = stream_epi_plain(1.0, 0.0, b);
                ep.dinv = lv.dinv.p;
                ep.xin = xin;
                ep.omega = (sweep & 1) ? lv.omega2 : lv.omega;
                if (launch_stream16x<double>(lv.S, lv.S.vals.p, xin, nxt, ep,
                                             stream, guard) != DNS_OK)
                    return (double *)nullptr;
            } else {
                DNS_LPR_SWITCH(
                    lv.S.lpr,
                    hipLaunchKernelGGL(k_mg_sweep<L>,
                                       grid_for_rows(lv.n, lv.S.lpr), kBlock, 0,
                                       stream, lv.n, lv.S.rowptr.p,
                                       lv.S.colidx.p, lv.S.vals.p, lv.dinv.p,
                                       (sweep & 1) ? lv.omega2 : lv.omega, b,
                                       xin, nxt, guard));
            }
            cur = nxt;
        }
        return cur;
    };
    std::vector<double *> xat((size_t)L, nullptr);
    for (int l = 0; l + 1 < L; ++l) {
        MgLevel &lv = mg[l];
        const double *b = (l == 0) ? in : lv.b.p;
        xat[l] = smooth(lv, b, true, lv.x2.p);
        if (!xat[l]) return DNS_ERR_HIP;
        const bool bigS = !dist() && streams_mg(lv.S);
        DNS_TRY(launch_spmv(lv.S, xat[l], lv.r.p, -1.0, 1.0, b,
                            bigS ? DNS_SPMV_STREAM16 : DNS_SPMV_VECTOR, stream,
                            guard));
        DNS_TRY(launch_spmv(lv.PT, lv.r.p, mg[l + 1].b.p, 1.0, 0.0, nullptr,
                            (!dist() && streams(lv.PT)) ? DNS_SPMV_STREAM16
                                                        : DNS_SPMV_VECTOR,
                            stream, guard));
    }
    {
        MgLevel &lc = mg[L - 1];
        const double *b = (L == 1) ? in : lc.b.p;
        // (the dense kernel's own guard is the solve's control block)
        const DnsCtl *gctl = (guard && guard == done_ptr())
                                 ? (const DnsCtl *)ctl.p
                                 : (const DnsCtl *)nullptr;
        DNS_TRY(mg_coarse_apply(b, lc.x.p, gctl));
        xat[L - 1] = lc.x.p;
    }
    for (int l = L - 2; l >= 0; --l) {
        MgLevel &lv = mg[l];
        const double *b = (l == 0) ? in : lv.b.p;
        DNS_TRY(launch_spmv(lv.P, xat[l + 1], xat[l], 1.0, 1.0, xat[l],
                            (!dist() && streams(lv.P)) ? DNS_SPMV_STREAM16
                                                       : DNS_SPMV_VECTOR,
                            stream, guard));
        xat[l] = smooth(lv, b, false, xat[l]);
        if (!xat[l]) return DNS_ERR_HIP;
    }
    if (!guard) {
        hipLaunchKernelGGL(k_axpby, grid_for_elems(np), kBlock, 0, stream,
                           (int64_t)np, -1.0, xat[0], 0.0, zp);
        if (xacc)
            hipLaunchKernelGGL(k_axpby, grid_for_elems(np), kBlock, 0, stream,
                               (int64_t)np, 1.0, zp, 1.0, xacc);
    } else {
        // guarded form: zp = -x through the (guarded) Jacobi kernel
        hipLaunchKernelGGL(k_mg_jacobi_guard, grid_for_elems(np), kBlock, 0,
                           stream, np, xat[0], zp, xacc, guard);
    }
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

// out = A [xa ; xb] (+ omega dinv .* b): sub-wave kernel, or the streaming
// kernel in the bandwidth regime
int dns_saddle::mg_op(const CsrDev &A, const double *xa, int nsplit,
                      const double *xb, const MgLevel *add, const double *b,
                      double *out, const int *guard) {
    // (called with whole operators only: the levels every rank runs in full)
    if (streams_mg(A)) {
        StreamEpi ep = stream_epi_plain(1.0, 0.0, nullptr);
        ep.x2 = xb;
        ep.nsplit = nsplit;
        if (add) {
            ep.dinv = add->dinv.p;
            ep.b = b;
            ep.omega = add->omega;
            ep.add_cb = 1;
        }
        if (A.vals32.p)
            return launch_stream16x<float>(A, A.vals32.p, xa, out, ep, stream,
                                           guard);
        return launch_stream16x<double>(A, A.vals.p, xa, out, ep, stream, guard);
    }
    DNS_LPR_SWITCH(
        A.lpr,
        hipLaunchKernelGGL(k_mg_split<L>, grid_for_rows(A.nrows, A.lpr), kBlock,
                           0, stream, A.nrows, A.rowptr.p, A.colidx.p, A.vals.p,
                           xa, nsplit, xb,
                           add ? add->dinv.p : (const double *)nullptr, b,
                           add ? add->omega : 0.0, out, guard));
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

// the V(2,2) cycle on the fused operators: per level two launches down
// (x_pre = Apre b; b_c = Rr [b; x_pre]) and two up (x' = Qq [x_pre; e] + c(b);
// one more sweep -- the finest level's writes zp = -x'' itself)
int dns_saddle::mg_coarse_apply(const double *b, double *x,
                                const DnsCtl *gctl) {
    const int nc = mg.back().n;
    const int gg = std::max(1, std::min((nc + 3) / 4, 2048));
    if (mg_cinv16.p)
        hipLaunchKernelGGL(k_gemv_half, gg, kBlock,
                           (size_t)mg_cld16 * sizeof(float), stream, nc,
                           mg_cld16, mg_cinv16.p, mg_cscale.p, b, x, 1.0, gctl);
    else if (fp32_store && mg_cinv32.p)
        hipLaunchKernelGGL(k_gemv_rows<float>, gg, kBlock, 0, stream, nc,
                           mg_cinv32.p, b, x, 1.0, gctl);
    else
        hipLaunchKernelGGL(k_gemv_rows<double>, gg, kBlock, 0, stream, nc,
                           mg_cinv.p, b, x, 1.0, gctl);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

int dns_saddle::mg_cycle_fused(const double *in, double *zp, double *xacc,
                                const int *guard, double scale) {
    // zp = -scale * V(in)  (and xacc += zp)
    const int L = (int)mg.size();
    for (int l = 0; l + 1 < L; ++l) {
        MgLevel &lv = mg[l];
        const double *b = (l == 0) ? in : lv.b.p;
        DNS_TRY(mg_op(lv.Apre, b, lv.n, nullptr, nullptr, nullptr, lv.x.p,
                      guard));
        DNS_TRY(mg_op(lv.Rr, b, lv.n, lv.x.p, nullptr, nullptr, mg[l + 1].b.p,
                      guard));
    }
    {
        MgLevel &lc = mg[L - 1];
        const DnsCtl *gctl = (guard && guard == done_ptr())
                                 ? (const DnsCtl *)ctl.p
                                 : (const DnsCtl *)nullptr;
        DNS_TRY(mg_coarse_apply(lc.b.p, lc.x.p, gctl));
    }
    for (int l = L - 2; l >= 0; --l) {
        MgLevel &lv = mg[l];
        const double *b = (l == 0) ? in : lv.b.p;
        DNS_TRY(mg_op(lv.Qq, lv.x.p, lv.n, mg[l + 1].x.p, &lv, b, lv.x2.p,
                      guard));
        // the second post-sweep; on the finest level it delivers zp = -x''
        double *out = (l == 0) ? zp : lv.x.p;
        const double osc = (l == 0) ? -scale : 1.0;
        // (whole operators: one GPU, or the levels every rank runs in full)
        if (streams_mg(lv.S)) {
            StreamEpi ep = stream_epi_plain(osc, 0.0, b);
            ep.dinv = lv.dinv.p;
            ep.xin = lv.x2.p;
            ep.omega = lv.omega2;
            if (lv.S.vals32.p)
                DNS_TRY(launch_stream16x<float>(lv.S, lv.S.vals32.p, lv.x2.p,
                                                out, ep, stream, guard));
            else
                DNS_TRY(launch_stream16x<double>(lv.S, lv.S.vals.p, lv.x2.p,
                                                 out, ep, stream, guard));
            if (l == 0 && xacc)
                hipLaunchKernelGGL(k_axpby, grid_for_elems(np), kBlock, 0,
                                   stream, (int64_t)np, 1.0, zp, 1.0, xacc);
        } else {
            DNS_LPR_SWITCH(
                lv.S.lpr,
                hipLaunchKernelGGL(k_mg_sweep<L>, grid_for_rows(lv.n, lv.S.lpr),
                                   kBlock, 0, stream, lv.n, lv.S.rowptr.p,
                                   lv.S.colidx.p, lv.S.vals.p, lv.dinv.p,
                                   lv.omega2, b, lv.x2.p, out, guard, osc,
                                   (l == 0) ? xacc : (double *)nullptr));
        }
    }
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

// the V(1,1) cycle on two fused operators per level (solver.hpp, mg_fused11):
// zp = -V(in), xacc += zp
int dns_saddle::mg_cycle_fused11(const double *in, double *zp, double *xacc,
                                 const int *guard) {
    const int L = (int)mg.size();
    for (int l = 0; l + 1 < L; ++l) {
        MgLevel &lv = mg[l];
        const double *b = (l == 0) ? in : lv.b.p;
        DNS_TRY(mg_op(lv.Rr, b, lv.n, nullptr, nullptr, nullptr,
                      mg[l + 1].b.p, guard));
    }
    {
        MgLevel &lc = mg[L - 1];
        const DnsCtl *gctl = (guard && guard == done_ptr())
                                 ? (const DnsCtl *)ctl.p
                                 : (const DnsCtl *)nullptr;
        DNS_TRY(mg_coarse_apply(lc.b.p, lc.x.p, gctl));
    }
    for (int l = L - 2; l >= 0; --l) {
        MgLevel &lv = mg[l];
        const double *b = (l == 0) ? in : lv.b.p;
        DNS_TRY(mg_op(lv.Qq, b, lv.n, mg[l + 1].x.p, nullptr, nullptr,
                      (l == 0) ? zp : lv.x.p, guard));
    }
    if (xacc) {
        if (guard)
            return fail(DNS_ERR_BAD_ARGUMENT,
                        "fused V(1,1) cycle: no guarded accumulate");
        hipLaunchKernelGGL(k_axpby, grid_for_elems(np), kBlock, 0, stream,
                           (int64_t)np, 1.0, zp, 1.0, xacc);
    }
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

// one cycle, or two as Richardson steps with Chebyshev weights:
//   x1 = a1 V(in);  x2 = x1 + a2 V(in - S x1);  zp = -x2
int dns_saddle::schur_mg_apply_fused(const double *in, double *zp, double *xacc,
                                     const int *guard) {
    if (!mg_two_now)
        return mg_cycle_fused(in, zp, xacc, guard, 1.0);
    const double mid = 1.0 - 0.5 * mg_rho, rad = 0.5 * mg_rho;
    const double c = 0.70710678118654752;
    const double a1 = 1.0 / (mid + rad * c), a2 = 1.0 / (mid - rad * c);
    MgLevel &l0 = mg[0];
    if (mg_r2.n < (size_t)l0.n) {
        // (first use: outside any capture -- ensure_solver_buffers sizes them)
        DNS_TRY(mg_r2.alloc((size_t)l0.n));
        DNS_TRY(mg_z2.alloc((size_t)l0.n));
    }
    DNS_TRY(mg_cycle_fused(in, zp, nullptr, guard, a1));
    // r = in - S x1 = in + S zp
    DNS_TRY(launch_spmv(l0.S, zp, mg_r2.p, 1.0, 1.0, in,
                        streams(l0.S) ? DNS_SPMV_STREAM16 : DNS_SPMV_VECTOR,
                        stream, guard));
    // zp += -a2 V(r)
    DNS_TRY(mg_cycle_fused(mg_r2.p, mg_z2.p, zp, guard, a2));
    if (xacc) {
        if (guard)
            return fail(DNS_ERR_BAD_ARGUMENT,
                        "two multigrid cycles: no guarded accumulate");
        hipLaunchKernelGGL(k_axpby, grid_for_elems(np), kBlock, 0, stream,
                           (int64_t)np, 1.0, zp, 1.0, xacc);
    }
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

int dns_saddle::build_jacobi_schur() {
    DNS_TRY(sinv.alloc((size_t)np));
    hipLaunchKernelGGL(k_schur_diag, grid_for_elems(np), kBlock, 0, stream, np,
                       J.rowptr.p, J.colidx.p, J.vals.p, dinv.p, sinv.p);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

int dns_saddle::set_stop_frac(double f) {
    // (k_batch_begin with nothing to copy: the accumulators and the fraction)
    CopyList none;
    none.count = 0;
    hipLaunchKernelGGL(k_batch_begin, 1, kBlock, 0, stream, none, ctl.p, f);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

int dns_saddle::setup_precond(const dns_precond_opts *o) {
    DNS_HIP(hipSetDevice(device));
    if (o) popts = *o;
    // (on by default where the cycle is launch bound: there a no-op node
    // costs what a working node costs.  Beyond ~1.5e6 unknowns the nodes are
    // bandwidth bound, a skipped one is nearly free, and the two-column
    // cycle is marginal -- residual levels between 0.2 and 0.9 x tol at
    // n = 2.8M: 589 steps/s with the slack-column policy against 436-616
    // depending on where a measuring window falls, profiles/r05_oversolve)
    oversolve = oversolve_env >= 0 ? oversolve_env != 0
                                   : (popts.schur == DNS_SCHUR_MG &&
                                      n < 1500000);
    // (two cycles per application of the multigrid block where whole Krylov
    // cycles are run: solver.hpp, mg_cycles; one column then suffices)
    mg_cycles = mg_cycles_knob > 0
                    ? mg_cycles_knob
                    : ((oversolve && popts.schur == DNS_SCHUR_MG) ? 2 : 1);
    // (the shortest cycle of the batches: oversolve_cmin_eff(), once the
    // partition of the hierarchy is known)
    oversolve_cmin = oversolve_cmin_eff();
    mg_two_now = mg_two_for(1000);    // (outside a Krylov cycle: as the option says)
    if (popts.cheb_degree < 1 || popts.cheb_degree > 256)
        return fail(DNS_ERR_BAD_ARGUMENT, "cheb_degree out of range");
    precond_ready = false;
    drop_graphs();   // captured kernel arguments (coefficients, buffers)
    // (values a device kernel wrote -- trapezoidal stepper -- first: a sliced
    // handle gets its matrices back from the HOST copies)
    if (rank_local) {
        if (popts.fhat == DNS_FHAT_CHEB || popts.cheb_degree < 2 ||
            popts.cheb_degree > 12)
            return fail(DNS_ERR_BAD_ARGUMENT,
                        "a handle created from rows needs the explicit Fh^-1 "
                        "(degree 2..12)");
        if (popts.schur == DNS_SCHUR_JACOBI)
            return fail(DNS_ERR_BAD_ARGUMENT,
                        "a handle created from rows needs the dense or the "
                        "multigrid Schur block");
    }
    if (fh_stale && F.nnz > 0) {
        if (fvals_own_rows_only && comm && comm->nranks > 1) {
            // a row-partitioned assembly wrote this rank's rows of F only;
            // the rings of rows the set-up reads beyond them belong to other
            // ranks (stale here: M + dt/2 A, or the last refresh): every rank
            // must set up from the SAME, whole matrix -- bounds, the
            // symmetric-part branch and its error return are decisions all
            // ranks take alike, and the collectives behind them are entered
            // by all or by none.  Blocks of non-zeros by velocity-row owner.
            std::vector<int> zs((size_t)comm->nranks + 1, 0);
            for (int q = 0; q <= comm->nranks; ++q)
                zs[q] = Fh.rowptr[st_v[q]];
            DNS_TRY(comm->allgatherv(F.vals.p, zs, stream));
        }
        DNS_TRY(F.vals.download(Fh.vals.data(), (size_t)F.nnz, stream));
        DNS_HIP(hipStreamSynchronize(stream));
    }
    fh_stale = false;
    fvals_own_rows_only = false;
    DNS_TRY(restore_full_device());   // (row-partitioned handle set up again)
    if (dinv_stale && F.nnz > 0) {
        hipLaunchKernelGGL(k_diag_inv, grid_for_elems(nv), kBlock, 0, stream,
                           nv, F.rowptr.p, F.colidx.p, F.vals.p, dinv.p);
        DNS_HIP(hipGetLastError());
        dinv_stale = false;
    }
    // the set-up is done redundantly and in full by every rank (identical
    // preconditioners without any communication); applies are partitioned
    dist_active = false;
    pc_sym = false;
    skew_eta = pc_shift = 0.0;
    if (popts.eig_lo > 0.0 && popts.eig_hi > popts.eig_lo) {
        lam_lo = popts.eig_lo;
        lam_hi = popts.eig_hi;
    } else if (rank_local) {
        DNS_TRY(estimate_bounds_rows());
    } else {
        DNS_TRY(estimate_bounds());
        // Nonsymmetric F: the eigenvalues of D^-1 F leave the real axis by up
        // to eta (Bendixson).  The Chebyshev residual polynomial is bounded by
        // one only inside the ellipse with foci lam_lo, lam_hi through the
        // origin, whose minor semi-axis is sqrt(lam_lo lam_hi): beyond it the
        // "approximate inverse" AMPLIFIES (a mode at the low end with
        // Im = 0.05 is multiplied by ~10 at degree 12) and GMRES stagnates --
        // the steady Oseen / Newton systems A + N(v) of snu:458,497 at cell
        // Peclet numbers > 1 (eta ~ 2.4 at Re = 50 on cylinder_2).  Then the
        // polynomial is built for the symmetric part of F, shifted by sigma D
        // if that part is indefinite (Newton: the strain of the linearisation
        // point); K keeps the true F, GMRES deals with the skew part (a few
        // hundred Krylov steps at Re = 50, none of them can blow up).
        const HostCsr FTh = host_transpose(Fh);
        skew_eta = host_skew_radius(Fh, FTh);
        if (skew_eta * skew_eta > 0.25 * lam_lo * lam_hi) {
            if (nv > 1000000 || popts.cheb_degree < 2 || popts.cheb_degree > 12)
                return fail(DNS_ERR_BAD_ARGUMENT,
                            "convection-dominated F (skew radius %.2e against "
                            "Chebyshev bounds [%.2e, %.2e]): needs the "
                            "explicit polynomial (degree 2..12, NV <= 1e6)",
                            skew_eta, lam_lo, lam_hi);
            Fpc_h = host_add(0.5, Fh, 0.5, FTh);
            double lmin = 0.0, lmax = 1.0;
            host_jacobi_bounds(Fpc_h, &lmin, &lmax);
            const double floor_lo = 0.012 * lmax;
            if (lmin < floor_lo) {
                pc_shift = floor_lo - lmin;
                for (int i = 0; i < nv; ++i)
                    for (int k = Fpc_h.rowptr[i]; k < Fpc_h.rowptr[i + 1]; ++k)
                        if (Fpc_h.colidx[k] == i)
                            Fpc_h.vals[k] *= 1.0 + pc_shift;
                // (D changes with the shift: bounds of the shifted matrix)
                host_jacobi_bounds(Fpc_h, &lmin, &lmax);
            }
            lam_lo = popts.eig_lo_safety * lmin;
            lam_hi = popts.eig_hi_safety * lmax;
            pc_sym = true;
            if (getenv("DNS_DEBUG"))
                fprintf(stderr,
                        "[dns] setup: skew radius %.3e -> polynomial of the "
                        "symmetric part, shift %.3e, bounds [%.3e, %.3e]\n",
                        skew_eta, pc_shift, lam_lo, lam_hi);
        }
    }
    theta = 0.5 * (lam_hi + lam_lo);
    delta = 0.5 * (lam_hi - lam_lo);
    c1.clear();
    c2.clear();
    {
        const double sigma1 = theta / delta;
        double rho = 1.0 / sigma1;
        for (int s = 0; s < popts.cheb_degree - 1; ++s) {
            const double rho_new = 1.0 / (2.0 * sigma1 - rho);
            c1.push_back(rho_new * rho);
            c2.push_back(2.0 * rho_new / delta);
            rho = rho_new;
        }
    }
    if (popts.schur != DNS_SCHUR_DENSE && popts.schur != DNS_SCHUR_JACOBI &&
        popts.schur != DNS_SCHUR_MG)
        return fail(DNS_ERR_BAD_ARGUMENT, "unknown Schur option %d",
                    popts.schur);
    mg_ready = false;
    if (popts.schur == DNS_SCHUR_DENSE &&
        (int64_t)np * np * 8 > (int64_t)64 << 30)
        return fail(DNS_ERR_BAD_ARGUMENT,
                    "dense Schur inverse too large for NP=%d", np);
    // explicit polynomial matrix: worth it while the apply is launch-latency
    // bound (pattern of F^(k-1): more bytes, far fewer dependent launches)
    fhat_explicit = (popts.fhat == DNS_FHAT_EXPLICIT) || pc_sym || rank_local ||
                    (popts.fhat == DNS_FHAT_AUTO && nv <= 1000000 &&
                     popts.cheb_degree >= 2 && popts.cheb_degree <= 12);
    fp32_store = popts.fp32_store != 0;
    have_jg = false;
    if (popts.factorization != DNS_FACT_TRIANGULAR &&
        popts.factorization != DNS_FACT_FULL)
        return fail(DNS_ERR_BAD_ARGUMENT, "unknown factorization %d",
                    popts.factorization);
    if (comm && !fhat_explicit)
        return fail(DNS_ERR_BAD_ARGUMENT,
                    "the row-partitioned solve needs the explicit Fh^-1 (fhat "
                    "= explicit / auto, degree 2..12, NV <= 1e6)");
    if (comm && popts.schur == DNS_SCHUR_JACOBI &&
        popts.factorization == DNS_FACT_FULL)
        ;   // (fine: the diagonal Schur block is replicated like the others)
    if (popts.factorization == DNS_FACT_FULL && !fhat_explicit)
        return fail(DNS_ERR_BAD_ARGUMENT,
                    "the full block factorisation needs the explicit Fh^-1 "
                    "(fhat = explicit / auto within its size limits)");
    if (fhat_explicit) {
        DNS_TRY(build_explicit(popts.schur == DNS_SCHUR_DENSE));
        if (popts.schur == DNS_SCHUR_JACOBI) DNS_TRY(build_jacobi_schur());
        if (fp32_store && popts.schur == DNS_SCHUR_DENSE) {
            // (only the fp32 copy is applied from here on: the fp64 inverse,
            // 8 NP^2 bytes, goes back to the allocator)
            DNS_TRY(schur_to_f32());
            DNS_HIP(hipStreamSynchronize(stream));
            sinv.release();
        }
    } else if (popts.schur == DNS_SCHUR_DENSE) {
        DNS_TRY(build_dense_schur());
        if (fp32_store) {
            DNS_TRY(schur_to_f32());
            DNS_HIP(hipStreamSynchronize(stream));
            sinv.release();
        }
    } else if (popts.schur == DNS_SCHUR_JACOBI) {
        DNS_TRY(build_jacobi_schur());
    } else if (popts.schur == DNS_SCHUR_MG) {
        // recurrence form of Fh^-1: the multigrid works on J D^-1 JT
        HostCsr DJT = JTh;
        std::vector<double> dv((size_t)nv, 1.0);
        for (int i = 0; i < nv; ++i) {
            double d = 0.0;
            for (int k = Fh.rowptr[i]; k < Fh.rowptr[i + 1]; ++k)
                if (Fh.colidx[k] == i) d += Fh.vals[k];
            dv[i] = (d != 0.0) ? 1.0 / d : 1.0;
        }
        host_scale_rows(dv, DJT);
        DNS_TRY(build_mg_schur(host_spgemm(Jh, DJT)));
    } else {
        return fail(DNS_ERR_BAD_ARGUMENT, "unknown Schur option %d",
                    popts.schur);
    }
    DNS_HIP(hipStreamSynchronize(stream));
    if (comm) {
        // from here on this rank keeps ITS row blocks only
        DNS_TRY(setup_dist(tmp_Gch, have_jg ? &tmp_JGh : nullptr,
                           tmp_presliced));
        tmp_presliced = false;
        tmp_Gch = HostCsr();
        tmp_JGh = HostCsr();
    }
    DNS_TRY(build_pair());
    precond_ready = true;
    dist_active = (comm != nullptr);
    return DNS_OK;
}

int dns_saddle::ensure_workspace(int m) {
    const size_t need = (size_t)(m + 1) * ld;
    if (V.n < need) {
        DNS_HIP(hipStreamSynchronize(stream));
        drop_graphs();               // graphs hold the old buffer address
        DNS_TRY(V.alloc(need));
    }
    if (Z.n < (size_t)m * ld) {
        DNS_HIP(hipStreamSynchronize(stream));
        drop_graphs();
        DNS_TRY(Z.alloc((size_t)m * ld));
    }
    if (xcat.n < ld) DNS_TRY(xcat.alloc(ld));
    return DNS_OK;
}

int dns_saddle::read_header() {
    DNS_HIP(hipMemcpyAsync(hdr_host, ctl.p, sizeof(CtlHeaderAcc),
                           hipMemcpyDeviceToHost, stream));
    DNS_HIP(hipStreamSynchronize(stream));
    return DNS_OK;
}

int dns_saddle::true_residual(const double *b, const double *x, double *out) {
    // (row-partitioned: this rank's rows, then the sum over the ranks; `x` is
    // whole on every rank after a solve)
    const RowMap rm = dist_sliced ? dist_rowmap() : RowMap{0, n, 0, 0};
    DNS_LPR_SWITCH(K.lpr,
                   hipLaunchKernelGGL(k_resid_norm<L>, gridS, kBlock, 0, stream,
                                      n, K.rowptr.p, K.colidx.p, K.vals.p, x,
                                      b, r.p, partR.p, (double *)nullptr, rm));
    hipLaunchKernelGGL(k_sum_partials, 1, kBlock, 0, stream, partR.p, gridS,
                       scal.p);
    DNS_HIP(hipGetLastError());
    if (dist_sliced && comm) DNS_TRY(comm->allreduce(scal.p, 1, stream));
    DNS_HIP(hipMemcpyAsync(scal_host, scal.p, sizeof(double),
                           hipMemcpyDeviceToHost, stream));
    DNS_HIP(hipStreamSynchronize(stream));
    *out = std::sqrt(scal_host[0]);
    spmv_count++;
    return DNS_OK;
}

void dns_saddle::drop_graphs() {
    // replays may still be in flight on the stream
    if (!graphs.empty() && stream) (void)hipStreamSynchronize(stream);
    for (auto &g : graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    graphs.clear();
    graph_generation++;
}

// Run `body` (which only enqueues work on `stream`) either eagerly or as a
// hipGraph captured on first use and replayed afterwards.  Launch-bound loops
// of ~50 tiny kernels replay at the device's kernel-boundary rate instead of
// the host's launch rate.
template <typename Body>
int dns_saddle::run_cached(const std::vector<uint64_t> &key, bool use_graph,
                           Body body, bool launch) {
    // inside an enclosing capture (several time steps in one graph) the body
    // is simply enqueued into that capture
    if (!use_graph || capturing) return body();
    for (auto &g : graphs)
        if (g.key == key) {
            if (launch) {
                DNS_HIP(hipGraphLaunch(g.exec, stream));
                if (comm) {
                    comm->n_allreduce += g.d_allreduce;
                    comm->n_allgather += g.d_allgather;
                    comm->n_alltoall += g.d_alltoall;
                    comm->bytes_alltoall += g.d_bytes_a2a;
                    comm->bytes_allgather += g.d_bytes_ag;
                }
            }
            return DNS_OK;
        }
    GraphEntry ge;
    ge.key = key;
    const int64_t c0[5] = {comm ? comm->n_allreduce : 0,
                           comm ? comm->n_allgather : 0,
                           comm ? comm->n_alltoall : 0,
                           comm ? comm->bytes_alltoall : 0,
                           comm ? comm->bytes_allgather : 0};
    // (RCCL may register memory / start proxies lazily inside a call: the
    // relaxed mode lets it; nothing else of this thread runs meanwhile)
    DNS_HIP(hipStreamBeginCapture(stream, dist() ? hipStreamCaptureModeRelaxed
                                                 : hipStreamCaptureModeThreadLocal));
    capturing = true;
    int rc;
    try {
        rc = body();
    } catch (...) {
        // a host-side exception (std::bad_alloc in a builder, ...) must not
        // leave the stream capturing: every later body() would then be
        // recorded into a graph nobody launches
        capturing = false;
        hipGraph_t partial = nullptr;
        (void)hipStreamEndCapture(stream, &partial);
        if (partial) (void)hipGraphDestroy(partial);
        if (comm) {
            comm->n_allreduce = c0[0];
            comm->n_allgather = c0[1];
            comm->n_alltoall = c0[2];
            comm->bytes_alltoall = c0[3];
            comm->bytes_allgather = c0[4];
        }
        throw;                     // the C-ABI barrier reports DNS_ERR_HOST
    }
    capturing = false;
    hipError_t e = hipStreamEndCapture(stream, &ge.graph);
    if (comm) {
        ge.d_allreduce = comm->n_allreduce - c0[0];
        ge.d_allgather = comm->n_allgather - c0[1];
        ge.d_alltoall = comm->n_alltoall - c0[2];
        ge.d_bytes_a2a = comm->bytes_alltoall - c0[3];
        ge.d_bytes_ag = comm->bytes_allgather - c0[4];
        if (!launch) {           // captured only: nothing was communicated
            comm->n_allreduce = c0[0];
            comm->n_allgather = c0[1];
            comm->n_alltoall = c0[2];
            comm->bytes_alltoall = c0[3];
            comm->bytes_allgather = c0[4];
        }
    }
    if (rc != DNS_OK) {
        if (ge.graph) (void)hipGraphDestroy(ge.graph);
        return rc;
    }
    if (e != hipSuccess)
        return fail(DNS_ERR_HIP, "hipStreamEndCapture: %s",
                    hipGetErrorString(e));
    DNS_HIP(hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
    graph_captures++;
    if (graphs.size() > 256) drop_graphs();
    graphs.push_back(ge);
    if (launch) DNS_HIP(hipGraphLaunch(ge.exec, stream));
    return DNS_OK;
}

static inline uint64_t bits_of(double v) {
    uint64_t u;
    memcpy(&u, &v, sizeof(u));
    return u;
}

// One restart cycle of right-preconditioned GMRES with `c` Arnoldi steps:
// residual, start, c x (precondition, K apply, Gram-Schmidt, close), then the
// correction x += P^-1 (V y).  Kernels after convergence return at once
// (ctl->done), so `c` may overshoot.  Nothing here synchronises or allocates.
int dns_saddle::enqueue_cycle(const double *b, double *x, int c,
                              const dns_solve_opts *o, int first,
                              bool have_resid) {
    // (cycles per application of the multigrid block in THIS Krylov cycle:
    // part of what a captured graph of length c holds)
    mg_two_now = mg_two_for(c);
    // `first`: 1 = the head kernel of this cycle resets the solve's counters,
    // 2 = only its status (going on after a Gram-Schmidt fallback)
    // (one GPU; the row-partitioned solve has its own: enqueue_cycle_dist)
    // r = b - K x, ||r||^2, ||b||^2 (unless the caller's prologue kernel has
    // produced r and the partials already)
    int resid_np = gridS;
    if (!have_resid && streams(K)) {
        // bandwidth regime: r = b - K x at the streaming rate with the
        // partials of both norms from the same launch
        StreamEpi ep = stream_epi_plain(-1.0, 1.0, b);
        ep.part = partR.p;
        ep.nvec = 0;
        ep.with_ww = 1;
        resid_np = Kp.ready ? pair_grid(Kp, sgrid)
                            : stream_grid(K, sgrid);
        ep.nparts = resid_np;
        ep.part_bb = partB.p;
        if (Kp.ready)
            DNS_TRY(launch_pair16x(Kp, x, r.p, ep, stream, nullptr,
                                   sgrid));
        else
            DNS_TRY(launch_stream16x<double>(K, K.vals.p, x, r.p, ep, stream,
                                             nullptr, sgrid));
    } else if (!have_resid) {
        DNS_LPR_SWITCH(
            K.lpr, hipLaunchKernelGGL(k_resid_norm<L>, gridS, kBlock, 0, stream,
                                      n, K.rowptr.p, K.colidx.p, K.vals.p, x, b,
                                      r.p, partR.p, partB.p,
                                      RowMap{0, n, 0, 0}));
    }
    // the consumers of a reduction read the per-workgroup partials
    const double *rr_part = partR.p, *bb_part = partB.p;
    const int rr_np =
        (have_resid && prologue_nparts > 0) ? prologue_nparts : resid_np;
    const bool dense = popts.schur == DNS_SCHUR_DENSE;
    const int q0 = 0, q1 = np;
    // (dense Schur rows: one wave each, four per workgroup)
    const int gridA =
        dense ? std::max(gridD, std::min((q1 - q0 + 3) / 4, 2048)) : gridD;
    // reorth == 2: Gram-Schmidt folded into the next head kernel
    // bandwidth regime: K through the streaming kernel with the dots fused in
    // (while they fit its accumulators)
    const bool stream_k = streams(K);
    const bool fusedgs = o->reorth == 2 && (fuse_dots || stream_k);
    const int gridK = stream_k ? (Kp.ready ? pair_grid(Kp, sgrid)
                                           : stream_grid(K, sgrid))
                               : gridC;
    // partials the consumers of step j's dots read: written by the kernel
    // that applied K in step j
    auto kparts = [&](int jj) {
        return (stream_k && jj + 1 <= kStreamDots) ? gridK : gridC;
    };
    // six-node step: K z_j is kept per column (the tail forms the new
    // residual r0 - sum y_j K z_j from them)
    const bool s6 = step6.on && fusedgs && !stream_k && have_resid &&
                    Wcols.n >= (size_t)c * ld;
    auto wcol = [&](int jj) -> double * {
        return s6 ? Wcols.p + (size_t)jj * ld : w.p;
    };
    for (int j = 0; j < c; ++j) {
        // the preconditioned vectors are kept (Z_j) for the correction behind
        // the cycle
        double *zj = Z.p + (size_t)j * ld;
        double *zp = zj + nv;
        const double *src = (j == 0) ? r.p : wcol(j - 1);
        const double *spart = (j == 0) ? rr_part : partN.p;
        const int snp = (j == 0) ? rr_np : gridD;
        // full block factorisation: tau = src_p - (J Fh^-1) src_v feeds the
        // Schur block instead of src_p (unguarded: 3 us when already done)
        const double *tin = nullptr;
        const bool mgs = popts.schur == DNS_SCHUR_MG;
        if (s6 && j == 0 && step6.kx && have_jg && !mgs) {
            // six-node step: r = b - kx is formed HERE (extra workgroups),
            // the Schur rows gather b - kx themselves
            const int gt = std::max(1, std::min((np + 1) / 2, 4096));
            hipLaunchKernelGGL(k_tau_first, gt + gridD, kBlock, 0, stream, gt, n,
                               np, nv, JG.rowptr.p, JG.colidx.p, JG.vals.p, b,
                               step6.kx, tau.p, r.p, partR.p, partB.p);
            tin = tau.p;
        } else if (have_jg && !mgs) {
            const int jt = (fusedgs && j > 0) ? j : 0;
            DNS_LPR_SWITCH(
                JG.lpr,
                hipLaunchKernelGGL(k_tau_guard<L>,
                                   grid_for_rows(q1 - q0, JG.lpr == 64 ? 128 : JG.lpr),
                                   kBlock, 0,
                                   stream, np, nv, JG.rowptr.p, JG.colidx.p,
                                   JG.vals.p, src, tau.p, partA.p,
                                   jt > 0 ? kparts(jt - 1) : gridC, jt, ctl.p,
                                   q0, q1));
            tin = tau.p;
        }
        if (fusedgs && j > 0 && mgs) {
            hipLaunchKernelGGL(k_arn_head_f<3>, gridA, kBlock, 0, stream, n, nv,
                               np, j, wcol(j - 1), partA.p, kparts(j - 1), V.p, ld, Z.p,
                               (const void *)nullptr, ctl.p, o->maxiter,
                               (const double *)nullptr);
        } else if (!(fusedgs && j > 0) && mgs) {
            hipLaunchKernelGGL(k_arn_head<3>, gridA, kBlock, 0, stream, n, nv,
                               np, j, src, spart, snp, V.p, ld,
                               (const void *)nullptr, zp, ctl.p, o->rtol,
                               o->atol, bb_part, rr_np, o->maxiter, q0, q1,
                               (j == 0) ? first : 0, (const double *)nullptr,
                               (j == 0 && first == 1) ? step_counter : nullptr);
        } else if (fusedgs && j > 0) {
            if (dense && fp32_store)
                hipLaunchKernelGGL(k_arn_head_f<2>, gridA, kBlock, 0, stream, n,
                                   nv, np, j, wcol(j - 1), partA.p, kparts(j - 1), V.p,
                                   ld,
                                   Z.p, (const void *)sinv32.p, ctl.p,
                                   o->maxiter, tin, 0, np, sld);
            else if (dense)
                hipLaunchKernelGGL(k_arn_head_f<1>, gridA, kBlock, 0, stream, n,
                                   nv, np, j, wcol(j - 1), partA.p, kparts(j - 1), V.p,
                                   ld,
                                   Z.p, (const void *)sinv.p, ctl.p,
                                   o->maxiter, tin);
            else
                hipLaunchKernelGGL(k_arn_head_f<0>, gridA, kBlock, 0, stream, n,
                                   nv, np, j, wcol(j - 1), partA.p, kparts(j - 1), V.p,
                                   ld,
                                   Z.p, (const void *)sinv.p, ctl.p,
                                   o->maxiter, tin);
        } else if (dense && fp32_store)
            hipLaunchKernelGGL(k_arn_head<2>, gridA, kBlock, 0, stream, n, nv,
                               np, j, src, spart, snp, V.p, ld,
                               (const void *)sinv32.p, zp, ctl.p, o->rtol,
                               o->atol, bb_part, rr_np, o->maxiter, q0, q1,
                               (j == 0) ? first : 0, tin,
                               (j == 0 && first == 1) ? step_counter : nullptr,
                               sld);
        else if (dense)
            hipLaunchKernelGGL(k_arn_head<1>, gridA, kBlock, 0, stream, n, nv,
                               np, j, src, spart, snp, V.p, ld,
                               (const void *)sinv.p, zp, ctl.p, o->rtol,
                               o->atol, bb_part, rr_np, o->maxiter, q0, q1,
                               (j == 0) ? first : 0, tin,
                               (j == 0 && first == 1) ? step_counter : nullptr);
        else
            hipLaunchKernelGGL(k_arn_head<0>, gridA, kBlock, 0, stream, n, nv,
                               np, j, src, spart, snp, V.p, ld,
                               (const void *)sinv.p, zp, ctl.p, o->rtol,
                               o->atol, bb_part, rr_np, o->maxiter, q0, q1,
                               (j == 0) ? first : 0, tin,
                               (j == 0 && first == 1) ? step_counter : nullptr);
        if (mgs) {
            // Schur block = V-cycle on V_j,p (or tau(V_j))
            const double *sin = V.p + (size_t)j * ld + nv;
            if (have_jg && streams(JG)) {
                // tau = V_j,p - JG V_j,v through the streaming kernel
                const double *vj = V.p + (size_t)j * ld;
                if (JG.vals32.p)
                    DNS_TRY(launch_stream16x<float>(
                        JG, JG.vals32.p, vj, tau.p,
                        stream_epi_plain(-1.0, 1.0, vj + nv), stream,
                        done_ptr()));
                else
                    DNS_TRY(launch_spmv(JG, vj, tau.p, -1.0, 1.0, vj + nv,
                                        DNS_SPMV_STREAM16, stream,
                                        done_ptr()));
                sin = tau.p;
            } else if (have_jg) {
                DNS_LPR_SWITCH(
                    JG.lpr,
                    hipLaunchKernelGGL(k_tau_guard<L>,
                                       grid_for_rows(np, JG.lpr == 64 ? 128 : JG.lpr),
                                       kBlock, 0,
                                       stream, np, nv, JG.rowptr.p,
                                       JG.colidx.p, JG.vals.p,
                                       V.p + (size_t)j * ld, tau.p,
                                       (const double *)nullptr, 0, -1, ctl.p,
                                       0, np));
                sin = tau.p;
            }
            DNS_TRY(schur_mg_apply(sin, zp, nullptr, done_ptr()));
        }
        DNS_TRY(apply_fhat_part(V.p + (size_t)j * ld, zp, zj, done_ptr(),
                                nullptr));
        if (stream_k && fusedgs && j + 1 <= kStreamDots) {
            StreamEpi ep = stream_epi_plain(1.0, 0.0, nullptr);
            ep.V = V.p;
            ep.ld = ld;
            ep.nvec = j + 1;
            ep.with_ww = 1;
            ep.part = partA.p;
            ep.nparts = gridK;
            if (Kp.ready)
                DNS_TRY(launch_pair16x(Kp, zj, w.p, ep, stream, done_ptr(),
                                       sgrid));
            else
                DNS_TRY(launch_stream16x<double>(K, K.vals.p, zj, w.p, ep,
                                                 stream, done_ptr(),
                                                 sgrid));
            continue;                    // no Gram-Schmidt kernel
        } else if (fuse_dots || fusedgs) {
            DNS_LPR_SWITCH(
                K.lpr,
                hipLaunchKernelGGL(k_spmv_multidot<L>, gridC, kBlock, 0,
                                   stream, n, K.rowptr.p, K.colidx.p,
                                   K.vals.p, zj, wcol(j), V.p, ld, j, partA.p,
                                   gridC, ctl.p, RowMap{0, n, 0, 0},
                                   fusedgs ? 1 : 0));
            if (fusedgs) continue;       // no Gram-Schmidt kernel
        } else {
            if (K.c16.p) {
                // bandwidth regime: the LDS-streaming kernel with 16-bit
                // column offsets (the roofline kernel of bench.py)
                DNS_TRY(launch_spmv(K, zj, w.p, 1.0, 0.0, nullptr,
                                    DNS_SPMV_STREAM16, stream, done_ptr()));
            } else {
                DNS_LPR_SWITCH(
                    K.lpr,
                    hipLaunchKernelGGL(k_spmv_guard<L>,
                                       grid_for_rows(n, K.lpr), kBlock, 0,
                                       stream, n, K.rowptr.p, K.colidx.p,
                                       K.vals.p, zj, w.p, ctl.p, 0, n));
            }
            hipLaunchKernelGGL(k_multidot, gridC, kBlock, 0, stream, n, V.p,
                               ld, w.p, partA.p, gridC, j, ctl.p);
        }
        const double *hpart = partA.p;
        const int hnp = gridC;
        if (o->reorth == 1) {
            hipLaunchKernelGGL(k_orth<1>, gridD, kBlock, 0, stream, n, V.p, ld,
                               w.p, hpart, hnp, j, 0, partE.p, gridD, ctl.p);
            hipLaunchKernelGGL(k_orth<0>, gridD, kBlock, 0, stream, n, V.p, ld,
                               w.p, partE.p, gridD, j, 1, partN.p, gridD,
                               ctl.p);
        } else {
            hipLaunchKernelGGL(k_orth<0>, gridD, kBlock, 0, stream, n, V.p, ld,
                               w.p, hpart, hnp, j, 0, partN.p, gridD, ctl.p);
        }
    }
    if (s6) {
        // six-node step: out-of-place tail + new residual + convection cells
        Tail6 t6 = step6.t6;
        t6.r0 = r.p;
        t6.W = Wcols.p;
        t6.nv = nv;
        hipLaunchKernelGGL(k_arn_tail6, gridD + step6.cells.nblocks, kBlock, 0,
                           stream, c, n, gridD, partA.p, kparts(c - 1), ctl.p,
                           histdev.p, (int)hist_cap, o->maxiter, Z.p, ld, t6,
                           tail_extrap, step6.cells);
        DNS_HIP(hipGetLastError());
        return DNS_OK;
    }
    if (fusedgs) {
        // one GPU, fused Gram-Schmidt: tail and correction in ONE launch
        hipLaunchKernelGGL(k_arn_tail_acc, gridD, kBlock, 0, stream, c, n,
                           partA.p, kparts(c - 1), ctl.p, histdev.p,
                           (int)hist_cap, o->maxiter, Z.p, ld, x, tail_extrap);
        DNS_HIP(hipGetLastError());
        return DNS_OK;
    }
    hipLaunchKernelGGL(k_arn_tail, 1, kBlock, 0, stream, c, partN.p, gridD,
                       ctl.p, histdev.p, (int)hist_cap, o->maxiter, 0);
    hipLaunchKernelGGL(k_basis_combine_acc, gridD, kBlock, 0, stream, n, Z.p,
                       ld, ctl.p, x, tail_extrap);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

int dns_saddle::ensure_solver_buffers(const dns_solve_opts *o) {
    const int m = std::max(1, std::min(o->restart, kMaxRestart));
    DNS_TRY(ensure_workspace(m));
    if (mg_cycles > 1 && !mg.empty() && mg_r2.n < (size_t)mg[0].n) {
        DNS_TRY(mg_r2.alloc((size_t)mg[0].n));
        DNS_TRY(mg_z2.alloc((size_t)mg[0].n));
    }
    const size_t need_hist = (size_t)o->maxiter + 2 * kMaxRestart + 8;
    if (histdev.n < need_hist) {
        DNS_HIP(hipStreamSynchronize(stream));
        drop_graphs();               // graphs hold the old buffer address
        DNS_TRY(histdev.alloc(need_hist));
    }
    hist_cap = histdev.n;
    return DNS_OK;
}

int dns_saddle::gmres(const double *b, double *x, const dns_solve_opts *o,
                      dns_solve_stats *st, const std::function<int()> &prologue,
                      uint64_t prologue_key, bool prologue_has_resid) {
    const int m = std::max(1, std::min(o->restart, kMaxRestart));
    DNS_TRY(ensure_solver_buffers(o));
    // (RCCL calls are captured with the kernels; host-callback communicators
    // need plain launches)
    const bool graph = o->use_graph != 0 && graph_capable();
    // first cycle length: what the previous solve needed plus slack (time
    // stepping repeats itself), rounded to the polling granularity
    const int gran = std::max(1, o->check_every);
    int c = (last_iters >= 0) ? last_iters + 1 : std::min(m, 16);
    c = std::min(m, std::max(gran, (c + gran - 1) / gran * gran));
    if (cycle_first > 0) c = std::min(m, cycle_first);
    // a host-callback communicator cannot pipeline (its collectives wait for
    // the stream): a solve the previous one predicts as a single step starts
    // with the ONE-step cycle the pipelined RCCL path runs (k_arn_tail_lazy1:
    // norms with the dots, the step always taken) -- the two-ranks-on-one-GPU
    // rehearsal runs the algorithm the ranks of a node run
    else if (dist() && !graph_capable() && last_iters >= 0 && last_iters <= 1)
        c = 1;
    if (pipeline_c > 0) c = std::min(m, pipeline_c);
    // dots fused into the K apply while the system is launch-latency bound
    fuse_dots = n <= 400000;
    int restarts = 0;
    int dist_stalls = 0;
    bool first = true;          // prologue + first cycle of the solve
    int reset = 1;              // what the head kernel resets (k_arn_head)
    dns_solve_opts oo = *o;
    // (eight solves in a row that fell back: stop trying on this system)
    if (gs_fallbacks >= 8 && oo.reorth == 2) oo.reorth = 0;
    // the tail kernel of EVERY cycle writes the next time step's warm start
    // (tail_extrap): its coefficients and pointers are baked into the captured
    // node, so they are part of every cycle's key -- a second cycle captured
    // while the history was still filling (cubic coefficients) must not be
    // replayed once it is full (the carry-over front kernel relies on x0
    // being exactly the combination it computes K x0 for)
    uint64_t tek = 0x7e;
    {
        const double te_c[5] = {tail_extrap.e0, tail_extrap.e1, tail_extrap.e2,
                                tail_extrap.e3, tail_extrap.e4};
        for (int i = 0; i < 5; ++i) {
            uint64_t u;
            memcpy(&u, &te_c[i], sizeof(u));
            tek ^= u + 0x9e3779b97f4a7c15ULL + (tek << 6) + (tek >> 2);
        }
        tek ^= (uint64_t)(uintptr_t)tail_extrap.out + 0x9e3779b97f4a7c15ULL +
               (tek << 6) + (tek >> 2);
    }
    while (true) {
        std::vector<uint64_t> key = {
            1u, tek, (uint64_t)(uintptr_t)b, (uint64_t)(uintptr_t)x, (uint64_t)c,
            (uint64_t)oo.reorth, (uint64_t)oo.maxiter, bits_of(oo.rtol),
            bits_of(oo.atol), first ? prologue_key : 0u,
            (uint64_t)popts.cheb_degree, (uint64_t)popts.schur,
            (uint64_t)first + 2u * (uint64_t)reset, (uint64_t)fhat_explicit,
            (uint64_t)fuse_dots, (uint64_t)(uintptr_t)z_plan_override,
            (uint64_t)dist_x0_exchange + 2u * (uint64_t)dist_lazy1,
            (uint64_t)(uintptr_t)dist_rnew};
        DNS_TRY(run_cached(key, graph, [&]() -> int {
            if (first && prologue) DNS_TRY(prologue());
            if (dist())
                return enqueue_cycle_dist(b, x, c, &oo, reset,
                                          first && prologue_has_resid);
            return enqueue_cycle(b, x, c, &oo, reset,
                                 first && prologue_has_resid);
        }));
        first = false;
        reset = 0;
        if (pipeline_c > 0) {
            // pipelined time stepping: one cycle, nobody waits; the device
            // accumulates iterations / failures for the batch (k_arn_tail)
            st->iters = -1;
            st->status = DNS_OK;
            return DNS_OK;
        }
        DNS_TRY(read_header());
        if (hdr_host->status == kGsFallback) {
            // the fused Gram-Schmidt gave up on its norm: go on from the
            // current iterate with the explicit kernel (row-partitioned: with
            // a new fused cycle -- it restarts from a fresh, orthonormal basis)
            (void)dist_stalls;
            gs_fallbacks++;
            oo.reorth = 0;
            reset = 2;
            restarts++;
            continue;
        }
        if (hdr_host->status != DNS_OK || hdr_host->conv ||
            hdr_host->total_it >= oo.maxiter || std::isnan(hdr_host->resnorm))
            break;
        restarts++;
        c = std::min(m, std::max(2 * c, 8));
    }
    if (o->reorth == 2 && oo.reorth == 2) gs_fallbacks = 0;   // clean fused solve
    const int total = hdr_host->total_it;
    last_iters = total;
    history.assign((size_t)std::max(1, hdr_host->hist_len), hdr_host->resnorm);
    if (hdr_host->hist_len > 0 && want_history) {
        DNS_TRY(dns::staged_d2h(history.data(), histdev.p,
                                history.size() * sizeof(double), stream));
    }
    st->iters = total;
    st->restarts = restarts;
    st->bnorm = hdr_host->bnorm;
    st->est_relres = hdr_host->bnorm > 0 ? hdr_host->resnorm / hdr_host->bnorm
                                         : hdr_host->resnorm;
    if (hdr_host->status != DNS_OK)
        st->status = hdr_host->status;
    else if (hdr_host->conv)
        st->status = DNS_OK;
    else
        st->status = DNS_NOT_CONVERGED;
    // K applies + F applies: per iteration 1 + (deg-1); per cycle one residual
    // and one correction
    spmv_count += (int64_t)total * popts.cheb_degree +
                  (int64_t)(restarts + 1) * (1 + popts.cheb_degree - 1);
    return DNS_OK;
}

int dns_saddle::bicgstab(const double *b, double *x, const dns_solve_opts *o,
                         dns_solve_stats *st) {
    const int check = std::max(1, o->check_every);
    const size_t hcap = (size_t)o->maxiter + 2;
    if (bi_rhat.n < ld) {
        DNS_TRY(bi_rhat.alloc(ld));
        DNS_TRY(bi_p.alloc(ld));
        DNS_TRY(bi_v.alloc(ld));
        DNS_TRY(bi_s.alloc(ld));
        DNS_TRY(bi_t.alloc(ld));
        DNS_TRY(bi_y.alloc(ld));
    }
    if (histdev.n < hcap) {
        DNS_HIP(hipStreamSynchronize(stream));
        drop_graphs();
        DNS_TRY(histdev.alloc(hcap));
    }
    DNS_LPR_SWITCH(K.lpr,
                   hipLaunchKernelGGL(k_resid_norm<L>, nred, kBlock, 0, stream,
                                      n, K.rowptr.p, K.colidx.p, K.vals.p, x,
                                      b, r.p, partR.p, partB.p,
                                      RowMap{0, n, 0, 0}));
    spmv_count++;
    // <rhat, r> = <r, r> at the start: partR doubles as part_rr and part_nn
    hipLaunchKernelGGL(k_bicg_start, nred, kBlock, 0, stream, n, r.p,
                       bi_rhat.p, bi_p.p, bi_v.p, partR.p, partB.p, nred,
                       ctl.p, bctl.p, o->rtol, o->atol, histdev.p);
    hipLaunchKernelGGL(k_axpby, 1, kBlock, 0, stream, (int64_t)nred, 1.0,
                       partR.p, 0.0, partN.p);
    int it = 0;
    bool done = false;
    while (!done) {
        const int chunk = std::min(check, o->maxiter + 1 - it);
        for (int c = 0; c < chunk; ++c, ++it) {
            const int par = it & 1;
            hipLaunchKernelGGL(k_bicg_p, nred, kBlock, 0, stream, n, r.p,
                               bi_p.p, bi_v.p, partR.p, partN.p, nred, ctl.p,
                               bctl.p, par, o->maxiter, histdev.p);
            hipLaunchKernelGGL(k_bicg_flag, 1, 64, 0, stream, ctl.p,
                               o->maxiter);
            DNS_TRY(apply_precond(bi_p.p, bi_y.p, done_ptr(), nullptr));
            DNS_LPR_SWITCH(
                K.lpr,
                hipLaunchKernelGGL(k_spmv_dot2<L>, nred, kBlock, 0, stream, n,
                                   K.rowptr.p, K.colidx.p, K.vals.p, bi_y.p,
                                   bi_v.p, bi_rhat.p, (const double *)nullptr,
                                   1, partA.p, partA.p + nred, ctl.p));
            hipLaunchKernelGGL(k_bicg_s, nred, kBlock, 0, stream, n, r.p,
                               bi_v.p, bi_s.p, partA.p, nred, ctl.p, bctl.p);
            DNS_TRY(apply_precond(bi_s.p, z.p, done_ptr(), nullptr));
            DNS_LPR_SWITCH(
                K.lpr,
                hipLaunchKernelGGL(k_spmv_dot2<L>, nred, kBlock, 0, stream, n,
                                   K.rowptr.p, K.colidx.p, K.vals.p, z.p,
                                   bi_t.p, bi_s.p, (const double *)nullptr, 2,
                                   partA.p, partA.p + nred, ctl.p));
            spmv_count += 2;
            hipLaunchKernelGGL(k_bicg_x, nred, kBlock, 0, stream, n, x, r.p,
                               bi_rhat.p, bi_y.p, z.p, bi_s.p, bi_t.p, partA.p,
                               partA.p + nred, nred, partR.p, partN.p, ctl.p,
                               bctl.p, par);
        }
        DNS_HIP(hipGetLastError());
        DNS_TRY(read_header());
        done = hdr_host->done != 0 || it > o->maxiter;
    }
    const int total = hdr_host->total_it;
    history.assign((size_t)total + 1, 0.0);
    DNS_TRY(dns::staged_d2h(history.data(), histdev.p,
                            history.size() * sizeof(double), stream));
    st->iters = total;
    st->restarts = 0;
    st->bnorm = hdr_host->bnorm;
    st->est_relres = hdr_host->bnorm > 0 ? hdr_host->resnorm / hdr_host->bnorm
                                         : hdr_host->resnorm;
    const bool conv = !(hdr_host->resnorm > hdr_host->tol);
    st->status = conv ? DNS_OK
                      : (hdr_host->status != DNS_OK ? hdr_host->status
                                                    : DNS_NOT_CONVERGED);
    return DNS_OK;
}

int dns_saddle::solve_device(const double *b, double *x,
                             const dns_solve_opts *o, dns_solve_stats *st) {
    if (!precond_ready)
        return fail(DNS_ERR_NOT_READY,
                    "dns_saddle_setup_precond must run before a solve");
    if (o->maxiter < 1 || o->restart < 1)
        return fail(DNS_ERR_BAD_ARGUMENT, "maxiter/restart must be positive");
    const int64_t spmv0 = spmv_count;
    DNS_HIP(hipEventRecord(ev0, stream));
    if (o->method == DNS_METHOD_GMRES)
        DNS_TRY(gmres(b, x, o, st));
    else if (o->method == DNS_METHOD_BICGSTAB && dist())
        return fail(DNS_ERR_BAD_ARGUMENT,
                    "the row-partitioned solve runs GMRES only");
    else if (o->method == DNS_METHOD_BICGSTAB)
        DNS_TRY(bicgstab(b, x, o, st));
    else
        return fail(DNS_ERR_BAD_ARGUMENT, "unknown method %d", o->method);
    // row-partitioned: every rank gets the whole velocity once per solve
    if (dist()) DNS_TRY(comm->allgatherv(x, st_v, stream));
    DNS_HIP(hipEventRecord(ev1, stream));
    DNS_HIP(hipEventSynchronize(ev1));
    float ms = 0.f;
    DNS_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    st->device_seconds = 1e-3 * ms;
    double tr = 0.0;
    DNS_TRY(true_residual(b, x, &tr));
    st->true_relres = st->bnorm > 0 ? tr / st->bnorm : tr;
    st->spmv_count = (int32_t)(spmv_count - spmv0);
    return DNS_OK;
}

// ===========================================================================
// extern "C"
// ===========================================================================
extern "C" {

int dns_version(void) { return 100; }

const char *dns_status_string(int status) {
    switch (status) {
        case DNS_OK: return "ok";
        case DNS_NOT_CONVERGED: return "not converged";
        case DNS_BREAKDOWN: return "breakdown";
        case DNS_ERR_HIP: return "HIP error";
        case DNS_ERR_BAD_ARGUMENT: return "bad argument";
        case DNS_ERR_NOT_READY: return "not ready";
        case DNS_ERR_COMM: return "communication error";
        case DNS_ERR_HOST: return "host-side failure";
        default: return "unknown status";
    }
}

const char *dns_last_error(void) { return g_last_error.c_str(); }

static int dns_device_count_impl(int *count) {
    if (!count) return fail(DNS_ERR_BAD_ARGUMENT, "null count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(DNS_ERR_HIP, "hipGetDeviceCount: %s",
                    hipGetErrorString(e));
    }
    *count = c;
    return DNS_OK;
}

int dns_device_count(int *count) {
    return dns::guarded([&]() -> int { return dns_device_count_impl(count); });
}

static int dns_device_name_impl(int device, char *buf, size_t buflen) {
    if (!buf || buflen == 0) return fail(DNS_ERR_BAD_ARGUMENT, "null buffer");
    hipDeviceProp_t prop;
    DNS_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
             prop.multiProcessorCount);
    return DNS_OK;
}

int dns_device_name(int device, char *buf, size_t buflen) {
    return dns::guarded([&]() -> int { return dns_device_name_impl(device, buf, buflen); });
}

static int dns_device_synchronize_impl(int device) {
    DNS_HIP(hipSetDevice(device));
    DNS_HIP(hipDeviceSynchronize());
    return DNS_OK;
}

int dns_device_synchronize(int device) {
    return dns::guarded([&]() -> int { return dns_device_synchronize_impl(device); });
}

void dns_default_precond_opts(dns_precond_opts *o) {
    o->cheb_degree = 4;
    o->schur = DNS_SCHUR_DENSE;
    o->fhat = DNS_FHAT_AUTO;
    o->fp32_store = 1;
    o->drop_tol = 3e-3;
    o->eig_lo_safety = 0.9;
    o->eig_hi_safety = 1.05;
    o->eig_lo = 0.0;
    o->eig_hi = 0.0;
    o->factorization = DNS_FACT_TRIANGULAR;
    o->pad = 0;
}

void dns_default_solve_opts(dns_solve_opts *o) {
    o->method = DNS_METHOD_GMRES;
    o->restart = 60;
    o->maxiter = 400;
    o->reorth = 1;
    o->rtol = 1e-10;
    o->atol = 0.0;
    o->check_every = 4;
    o->use_graph = 0;
}

static int dns_saddle_create_impl(int device, const dns_csr *f, const dns_csr *j,
                      const dns_csr *jt, dns_saddle **out) {
    if (!out) return fail(DNS_ERR_BAD_ARGUMENT, "null output handle");
    *out = nullptr;
    dns_saddle *h = new (std::nothrow) dns_saddle();
    if (!h) return fail(DNS_ERR_BAD_ARGUMENT, "out of host memory");
    const int s = h->init(device, f, j, jt);
    if (s != DNS_OK) {
        delete h;
        return s;
    }
    *out = h;
    return DNS_OK;
}

int dns_saddle_create(int device, const dns_csr *f, const dns_csr *j,
                      const dns_csr *jt, dns_saddle **out) {
    return dns::guarded([&]() -> int { return dns_saddle_create_impl(device, f, j, jt, out); });
}

void dns_saddle_destroy(dns_saddle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    delete h;
}

static int dns_saddle_update_values_impl(dns_saddle *h, const double *f_vals) {
    if (!h || !f_vals) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    return h->update_values(f_vals);
}

int dns_saddle_update_values(dns_saddle *h, const double *f_vals) {
    return dns::guarded([&]() -> int { return dns_saddle_update_values_impl(h, f_vals); });
}

static int dns_saddle_set_schur_mg_impl(dns_saddle *h, int32_t nprol, const dns_csr *prol,
                            int32_t smooth_steps) {
    if (!h || nprol < 0 || (nprol > 0 && !prol) || smooth_steps < 1 ||
        smooth_steps > 8)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    std::vector<HostCsr> ps;
    int rows = h->np;
    for (int l = 0; l < nprol; ++l) {
        DNS_TRY(check_csr(&prol[l], "prolongation"));
        if (prol[l].nrows != rows || prol[l].ncols < 1 ||
            prol[l].ncols > rows)
            return fail(DNS_ERR_BAD_ARGUMENT,
                        "prolongation %d is %d x %d, expected %d rows and at "
                        "most as many columns", l, prol[l].nrows,
                        prol[l].ncols, rows);
        ps.push_back(host_copy(&prol[l]));
        rows = prol[l].ncols;
    }
    h->mg_prol_h.swap(ps);
    h->mg_set = true;
    h->mg_nu = smooth_steps;
    h->mg_ready = false;
    if (h->popts.schur == DNS_SCHUR_MG) h->precond_ready = false;
    return DNS_OK;
}

int dns_saddle_set_schur_mg(dns_saddle *h, int32_t nprol, const dns_csr *prol,
                            int32_t smooth_steps) {
    return dns::guarded([&]() -> int { return dns_saddle_set_schur_mg_impl(h, nprol, prol, smooth_steps); });
}

static int dns_saddle_set_option_impl(dns_saddle *h, const char *name, double value) {
    if (!h || !name) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    const std::string k(name);
    if (k == "stream_nnz") h->stream_nnz = (int64_t)value;
    else if (k == "mg_stream_nnz") h->mg_stream_nnz = (int64_t)value;
    else if (k == "pair") h->pair_knob = value != 0.0;
    else if (k == "part_setup") h->part_setup = value != 0.0;
    else if (k == "mg_rows") h->mg_rows_knob = value != 0.0;
    else if (k == "dist_x0_exchange") h->dist_x0_exchange = value != 0.0;
    else if (k == "dist_lazy1") h->dist_lazy1 = value != 0.0;
    else if (k == "cycle_first") h->cycle_first = (int)value;
    else if (k == "mg_dense_max") h->mg_dense_max = (int)value;
    else if (k == "mg_dense_half_max") h->mg_dense_half_max = (int)value;
    else if (k == "mg_part_min") h->mg_part_min = (int)value;
    else if (k == "mg_fused") h->mg_fused_knob = value != 0.0;
    else if (k == "mg_cheb") h->mg_cheb = value != 0.0;
    else if (k == "mg_cheb_alpha") h->mg_cheb_alpha = std::max(1.5, value);
    else if (k == "mg_cycles")
        h->mg_cycles_knob = std::max(0, std::min(2, (int)value));
    else if (k == "mg_rho") h->mg_rho = std::max(0.01, std::min(0.95, value));
    else if (k == "dist_graph") h->dist_graph_ok = value != 0.0;
    else if (k == "oversolve") h->oversolve_env = value < 0.0 ? -1 : (value != 0.0);
    else if (k == "oversolve_cmin") h->oversolve_cmin_env = std::max(1, (int)value);
    else
        return fail(DNS_ERR_BAD_ARGUMENT, "unknown option '%s'", name);
    h->precond_ready = false;        // (set up again with the new setting)
    h->drop_graphs();
    return DNS_OK;
}

int dns_saddle_set_option(dns_saddle *h, const char *name, double value) {
    return dns::guarded([&]() -> int { return dns_saddle_set_option_impl(h, name, value); });
}

static int dns_saddle_setup_precond_impl(dns_saddle *h, const dns_precond_opts *opts) {
    if (!h) return fail(DNS_ERR_BAD_ARGUMENT, "null handle");
    return h->setup_precond(opts);
}

int dns_saddle_setup_precond(dns_saddle *h, const dns_precond_opts *opts) {
    return dns::guarded([&]() -> int { return dns_saddle_setup_precond_impl(h, opts); });
}

static int dns_saddle_solve_impl(dns_saddle *h, const double *rhs_v, const double *rhs_p,
                     const double *x0, double *out_vp,
                     const dns_solve_opts *opts, dns_solve_stats *stats) {
    if (!h || !rhs_v || !out_vp)
        return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    dns_solve_opts o;
    if (opts)
        o = *opts;
    else
        dns_default_solve_opts(&o);
    dns_solve_stats local;
    dns_solve_stats *st = stats ? stats : &local;
    memset(st, 0, sizeof(*st));
    DNS_HIP(hipSetDevice(h->device));
    DNS_TRY(h->bdev.upload(rhs_v, (size_t)h->nv, h->stream));
    if (rhs_p) {
        DNS_TRY(upload_to(h->bdev.p + h->nv, rhs_p, (size_t)h->np,
                          h->stream));
    } else {
        DNS_HIP(hipMemsetAsync(h->bdev.p + h->nv, 0,
                               (size_t)h->np * sizeof(double), h->stream));
    }
    if (x0)
        DNS_TRY(h->xdev.upload(x0, (size_t)h->n, h->stream));
    else
        DNS_TRY(h->xdev.zero(h->stream));
    DNS_TRY(h->solve_device(h->bdev.p, h->xdev.p, &o, st));
    DNS_TRY(h->xdev.download(out_vp, (size_t)h->n, h->stream));
    DNS_HIP(hipStreamSynchronize(h->stream));
    return DNS_OK;
}

int dns_saddle_solve(dns_saddle *h, const double *rhs_v, const double *rhs_p,
                     const double *x0, double *out_vp,
                     const dns_solve_opts *opts, dns_solve_stats *stats) {
    return dns::guarded([&]() -> int { return dns_saddle_solve_impl(h, rhs_v, rhs_p, x0, out_vp, opts, stats); });
}

// `ncols` right-hand sides in ONE call: the blocks travel to the device once,
// the solves run back to back on the resident system (same graphs, same
// preconditioner), the solutions come back once.  Column c of `rhs_v` is at
// rhs_v + c NV, of `rhs_p` at rhs_p + c NP (NULL: zero), of `out_vp` at
// out_vp + c (NV + NP); `x0` holds `x0_cols` start vectors (0: none, 1: the
// same for every column, else one per column).
static int dns_saddle_solve_multi_impl(dns_saddle *h, int32_t ncols,
                                       const double *rhs_v, const double *rhs_p,
                                       const double *x0, int32_t x0_cols,
                                       double *out_vp,
                                       const dns_solve_opts *opts,
                                       dns_solve_stats *stats) {
    if (!h || !rhs_v || !out_vp || ncols < 1 || x0_cols < 0 ||
        (x0_cols > 1 && x0_cols != ncols) || (x0_cols > 0 && !x0))
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    dns_solve_opts o;
    if (opts)
        o = *opts;
    else
        dns_default_solve_opts(&o);
    DNS_HIP(hipSetDevice(h->device));
    const size_t nv = (size_t)h->nv, np = (size_t)h->np, n = (size_t)h->n,
                 ld = h->ld, k = (size_t)ncols;
    if (h->mrhs.n < k * ld) DNS_TRY(h->mrhs.alloc(k * ld));
    if (h->msol.n < k * ld) DNS_TRY(h->msol.alloc(k * ld));
    // The caller's blocks (pageable NumPy memory) never meet the DMA engine:
    // they are packed into a page-locked staging buffer by the CPU and travel
    // as ONE copy per direction.  Round 4 had a GPU memory fault inside this
    // function at a page-aligned HOST heap address (profiles/r04_final/
    // gpu_fault_solve_multi.txt); what this function did that no other upload
    // of the library does is keep SEVERAL asynchronous copies out of pageable
    // memory in flight at once -- first one pitched 2-D copy, then 3 k 1-D
    // copies whose ranges share their boundary pages (a column is NV doubles,
    // not a multiple of the page size).  The runtime pins pageable ranges in
    // place page by page and unpins them when a copy retires: a page shared
    // by a retired and a pending copy is the one mechanism that produces a
    // fault at exactly such an address.  HYPOTHESIS (the record holds no
    // ranges to prove it; DNS_DEBUG_UPLOADS=1 prints them from now on) --
    // the staging removes every concurrent pin of caller memory either way.
    log_host_copy("multi rhs_v", rhs_v, h->mrhs.p, k * nv * sizeof(double));
    if (rhs_p)
        log_host_copy("multi rhs_p", rhs_p, h->mrhs.p, k * np * sizeof(double));
    if (x0_cols > 0)
        log_host_copy("multi x0", x0, h->msol.p,
                      (x0_cols > 1 ? k : 1) * n * sizeof(double));
    log_host_copy("multi out", out_vp, h->msol.p, k * n * sizeof(double));
    DNS_TRY(h->mstage.reserve(2 * k * ld));
    {
        double *sr = h->mstage.p, *sx = h->mstage.p + k * ld;
        memset(sr, 0, 2 * k * ld * sizeof(double));
        for (size_t c = 0; c < k; ++c) {
            memcpy(sr + c * ld, rhs_v + c * nv, nv * sizeof(double));
            if (rhs_p && np > 0)
                memcpy(sr + c * ld + nv, rhs_p + c * np, np * sizeof(double));
            if (x0_cols > 0)
                memcpy(sx + c * ld, x0 + (x0_cols > 1 ? c * n : 0),
                       n * sizeof(double));
        }
        SyncOnExit arrived(h->stream);
        DNS_HIP(hipMemcpyAsync(h->mrhs.p, sr, k * ld * sizeof(double),
                               hipMemcpyHostToDevice, h->stream));
        DNS_HIP(hipMemcpyAsync(h->msol.p, sx, k * ld * sizeof(double),
                               hipMemcpyHostToDevice, h->stream));
    }
    h->col_history.assign(k, std::vector<double>());
    int worst = DNS_OK;
    for (size_t c = 0; c < k; ++c) {
        dns_solve_stats local;
        dns_solve_stats *st = stats ? stats + c : &local;
        memset(st, 0, sizeof(*st));
        DNS_TRY(h->solve_device(h->mrhs.p + c * ld, h->msol.p + c * ld, &o, st));
        h->col_history[c] = h->history;
        if (st->status != DNS_OK && worst == DNS_OK) worst = st->status;
    }
    DNS_HIP(hipMemcpyAsync(h->mstage.p, h->msol.p, k * ld * sizeof(double),
                           hipMemcpyDeviceToHost, h->stream));
    DNS_HIP(hipStreamSynchronize(h->stream));
    for (size_t c = 0; c < k; ++c)
        memcpy(out_vp + c * n, h->mstage.p + c * ld, n * sizeof(double));
    (void)worst;       // (per-column statuses are in `stats`, like the single solve)
    return DNS_OK;
}

int dns_saddle_solve_multi(dns_saddle *h, int32_t ncols, const double *rhs_v,
                           const double *rhs_p, const double *x0,
                           int32_t x0_cols, double *out_vp,
                           const dns_solve_opts *opts, dns_solve_stats *stats) {
    return dns::guarded([&]() -> int { return dns_saddle_solve_multi_impl(h, ncols, rhs_v, rhs_p, x0, x0_cols, out_vp, opts, stats); });
}

static int dns_saddle_residual_history_col_impl(dns_saddle *h, int32_t col,
                                                double *out, int32_t cap,
                                                int32_t *count) {
    if (!h || !count || col < 0 || (size_t)col >= h->col_history.size())
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    const std::vector<double> &hist = h->col_history[(size_t)col];
    *count = (int32_t)hist.size();
    if (out)
        for (int32_t i = 0; i < std::min(*count, cap); ++i) out[i] = hist[i];
    return DNS_OK;
}

int dns_saddle_residual_history_col(dns_saddle *h, int32_t col, double *out,
                                    int32_t cap, int32_t *count) {
    return dns::guarded([&]() -> int { return dns_saddle_residual_history_col_impl(h, col, out, cap, count); });
}

static int dns_saddle_residual_history_impl(dns_saddle *h, double *out, int32_t cap,
                                int32_t *count) {
    if (!h || !count) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    const int32_t nh = (int32_t)h->history.size();
    *count = nh;
    if (out)
        for (int32_t i = 0; i < std::min(nh, cap); ++i) out[i] = h->history[i];
    return DNS_OK;
}

int dns_saddle_residual_history(dns_saddle *h, double *out, int32_t cap,
                                int32_t *count) {
    return dns::guarded([&]() -> int { return dns_saddle_residual_history_impl(h, out, cap, count); });
}

static int dns_saddle_apply_impl(dns_saddle *h, const double *x, double *y) {
    if (!h || !x || !y) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    DNS_HIP(hipSetDevice(h->device));
    DNS_TRY(h->xdev.upload(x, (size_t)h->n, h->stream));
    DNS_TRY(launch_spmv(h->K, h->xdev.p, h->w.p, 1.0, 0.0, nullptr,
                        DNS_SPMV_VECTOR, h->stream));
    DNS_TRY(h->w.download(y, (size_t)h->n, h->stream));
    DNS_HIP(hipStreamSynchronize(h->stream));
    return DNS_OK;
}

int dns_saddle_apply(dns_saddle *h, const double *x, double *y) {
    return dns::guarded([&]() -> int { return dns_saddle_apply_impl(h, x, y); });
}

static int dns_saddle_apply_precond_impl(dns_saddle *h, const double *r, double *z) {
    if (!h || !r || !z) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    if (!h->precond_ready)
        return fail(DNS_ERR_NOT_READY, "preconditioner not set up");
    DNS_HIP(hipSetDevice(h->device));
    DNS_TRY(h->ctl.zero(h->stream));
    DNS_TRY(h->xdev.upload(r, (size_t)h->n, h->stream));
    DNS_TRY(h->apply_precond(h->xdev.p, h->z.p, h->zero_ptr(), nullptr));
    DNS_TRY(h->z.download(z, (size_t)h->n, h->stream));
    DNS_HIP(hipStreamSynchronize(h->stream));
    return DNS_OK;
}

int dns_saddle_apply_precond(dns_saddle *h, const double *r, double *z) {
    return dns::guarded([&]() -> int { return dns_saddle_apply_precond_impl(h, r, z); });
}

static int dns_comm_unique_id_impl(char *out) {
    if (!out) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    static_assert(sizeof(ncclUniqueId) <= DNS_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess)
        return fail(DNS_ERR_COMM, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memset(out, 0, DNS_UNIQUE_ID_BYTES);
    memcpy(out, &id, sizeof(id));
    return DNS_OK;
}

int dns_comm_unique_id(char *out) {
    return dns::guarded([&]() -> int { return dns_comm_unique_id_impl(out); });
}

static int dns_comm_create_rccl_impl(int device, int32_t nranks, int32_t rank,
                         const char *uid, dns_comm **out) {
    if (!out || !uid || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    *out = nullptr;
    DNS_HIP(hipSetDevice(device));
    dns_comm *c = new (std::nothrow) dns_comm();
    if (!c) return fail(DNS_ERR_BAD_ARGUMENT, "out of host memory");
    c->rank = rank;
    c->nranks = nranks;
    c->device = device;
    ncclUniqueId id;
    memcpy(&id, uid, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
        c->nccl = nullptr;
        delete c;
        return fail(DNS_ERR_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    if (const char *form = getenv("DNS_COMM_ALLGATHER"))
        c->ag_form = strcmp(form, "bcast") == 0 ? 1
                     : strcmp(form, "staged") == 0 ? 2 : 0;
    *out = c;
    return DNS_OK;
}

int dns_comm_create_rccl(int device, int32_t nranks, int32_t rank,
                         const char *uid, dns_comm **out) {
    return dns::guarded([&]() -> int { return dns_comm_create_rccl_impl(device, nranks, rank, uid, out); });
}

static int dns_comm_create_callbacks_impl(int device, int32_t nranks, int32_t rank,
                              dns_allreduce_cb allreduce,
                              dns_allgatherv_cb allgatherv, void *ctx,
                              dns_comm **out) {
    if (!out || !allreduce || !allgatherv || nranks < 1 || rank < 0 ||
        rank >= nranks)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    dns_comm *c = new (std::nothrow) dns_comm();
    if (!c) return fail(DNS_ERR_BAD_ARGUMENT, "out of host memory");
    c->rank = rank;
    c->nranks = nranks;
    c->device = device;
    c->ar_cb = allreduce;
    c->ag_cb = allgatherv;
    c->ctx = ctx;
    *out = c;
    return DNS_OK;
}

int dns_comm_create_callbacks(int device, int32_t nranks, int32_t rank,
                              dns_allreduce_cb allreduce,
                              dns_allgatherv_cb allgatherv, void *ctx,
                              dns_comm **out) {
    return dns::guarded([&]() -> int { return dns_comm_create_callbacks_impl(device, nranks, rank, allreduce, allgatherv, ctx, out); });
}

void dns_comm_destroy(dns_comm *c) { delete c; }

static int dns_comm_stats_impl(dns_comm *c, int64_t *n_allreduce, int64_t *n_allgather) {
    if (!c) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    if (n_allreduce) *n_allreduce = c->n_allreduce;
    if (n_allgather) *n_allgather = c->n_allgather;
    return DNS_OK;
}

int dns_comm_stats(dns_comm *c, int64_t *n_allreduce, int64_t *n_allgather) {
    return dns::guarded([&]() -> int { return dns_comm_stats_impl(c, n_allreduce, n_allgather); });
}

// ---- first-contact self-test of the communicator -------------------------
// ONE primitive per call, so that the caller can say which one did not come
// back: 0 all-reduce of `count` doubles, 1 one grouped Send/Recv exchange round
// the ring (to rank+1, from rank-1: the halo exchange's call pattern), 2
// all-gather of UNEQUAL blocks (count + q doubles on rank q: staged form), 3
// all-gather of equal blocks (in-place form).  `graph` = 0: plain launches; 1:
// the same calls captured in a hipGraph (relaxed mode, as run_cached captures
// a cycle) and replayed.  One call is checked entry by entry against what the
// ranks must have produced (*ok), then `reps` calls are timed with an event
// pair on the launch stream (*us_per_call; the wait for the slowest peer
// included).  Collective: every rank calls with the same arguments.
static int dns_comm_selftest_impl(dns_comm *c, int32_t which, int32_t graph,
                                  int32_t count, int32_t reps, int32_t *ok,
                                  double *us_per_call) {
    if (!c || !ok || !us_per_call || which < 0 || which > 3 || count < 1 ||
        reps < 1)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    *ok = 0;
    *us_per_call = 0.0;
    if (graph && !c->nccl)
        return fail(DNS_ERR_BAD_ARGUMENT, "a host-callback communicator "
                    "synchronises its stream: nothing to capture");
    DNS_HIP(hipSetDevice(c->device));
    const int n = c->nranks, me = c->rank;
    hipStream_t s = nullptr;
    DNS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Guard {
        hipStream_t s;
        hipEvent_t a = nullptr, b = nullptr;
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        ~Guard() {
            if (ge) (void)hipGraphExecDestroy(ge);
            if (g) (void)hipGraphDestroy(g);
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    } guard{s};
    DNS_HIP(hipEventCreate(&guard.a));
    DNS_HIP(hipEventCreate(&guard.b));
    // layouts
    std::vector<int> starts((size_t)n + 1, 0);
    for (int q = 0; q < n; ++q)
        starts[q + 1] = starts[q] + count + (which == 2 ? q : 0);
    const int total = which >= 2 ? starts[n] : 2 * count;
    DevBuf<double> buf;
    DNS_TRY(buf.alloc((size_t)total));
    if (which >= 2) DNS_TRY(c->reserve_gather((size_t)(count + n), s));
    std::vector<double> host((size_t)total, 0.0), back((size_t)total, 0.0);
    const int nxt = (me + 1) % n, prv = (me + n - 1) % n;
    std::vector<int> sc((size_t)n, 0), sd((size_t)n, 0), rc((size_t)n, 0),
        rd((size_t)n, 0);
    if (n > 1) {
        sc[nxt] = count;
        rc[prv] = count;
    }
    auto fill = [&](bool zero) {
        for (int i = 0; i < total; ++i) host[i] = 0.0;
        if (zero) return;
        if (which == 0)
            for (int i = 0; i < count; ++i) host[i] = (me + 1.0) * (i + 1.0);
        else if (which == 1)
            for (int i = 0; i < count; ++i) host[i] = 1000.0 * me + i;
        else
            for (int i = starts[me]; i < starts[me + 1]; ++i)
                host[i] = 1000.0 * me + (i - starts[me]);
    };
    auto one = [&]() -> int {
        if (which == 0) return c->allreduce(buf.p, count, s);
        if (which == 1)
            return c->alltoallv(buf.p, sc, sd, buf.p + count, rc, rd, s);
        return c->allgatherv(buf.p, starts, s);
    };
    auto run = [&](int calls, bool timed) -> int {
        if (!graph) {
            if (timed) DNS_HIP(hipEventRecord(guard.a, s));
            for (int r = 0; r < calls; ++r) DNS_TRY(one());
            if (timed) DNS_HIP(hipEventRecord(guard.b, s));
            return DNS_OK;
        }
        if (guard.ge) (void)hipGraphExecDestroy(guard.ge);
        if (guard.g) (void)hipGraphDestroy(guard.g);
        guard.ge = nullptr;
        guard.g = nullptr;
        DNS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        int rc2 = DNS_OK;
        for (int r = 0; r < calls && rc2 == DNS_OK; ++r) rc2 = one();
        hipError_t e = hipStreamEndCapture(s, &guard.g);
        DNS_TRY(rc2);
        if (e != hipSuccess)
            return fail(DNS_ERR_HIP, "hipStreamEndCapture: %s",
                        hipGetErrorString(e));
        DNS_HIP(hipGraphInstantiate(&guard.ge, guard.g, nullptr, nullptr, 0));
        if (timed) {                 // first replay untimed
            DNS_HIP(hipGraphLaunch(guard.ge, s));
            DNS_HIP(hipEventRecord(guard.a, s));
        }
        DNS_HIP(hipGraphLaunch(guard.ge, s));
        if (timed) DNS_HIP(hipEventRecord(guard.b, s));
        return DNS_OK;
    };
    // (1) one call, checked
    fill(false);
    DNS_TRY(buf.upload(host.data(), (size_t)total, s));
    DNS_TRY(run(1, false));
    DNS_TRY(buf.download(back.data(), (size_t)total, s));
    DNS_HIP(hipStreamSynchronize(s));
    bool good = true;
    if (which == 0) {
        for (int i = 0; i < count; ++i)
            good = good && back[i] == 0.5 * n * (n + 1.0) * (i + 1.0);
    } else if (which == 1) {
        for (int i = 0; i < count && n > 1; ++i)
            good = good && back[count + i] == 1000.0 * prv + i;
    } else {
        for (int q = 0; q < n; ++q)
            for (int i = starts[q]; i < starts[q + 1]; ++i)
                good = good && back[i] == 1000.0 * q + (i - starts[q]);
    }
    *ok = good ? 1 : 0;
    // (2) `reps` calls, timed (on zeros: repeated sums stay finite)
    fill(true);
    DNS_TRY(buf.upload(host.data(), (size_t)total, s));
    DNS_TRY(run(reps, true));
    DNS_HIP(hipEventSynchronize(guard.b));
    float ms = 0.f;
    DNS_HIP(hipEventElapsedTime(&ms, guard.a, guard.b));
    *us_per_call = 1e3 * (double)ms / reps;
    return DNS_OK;
}

int dns_comm_selftest(dns_comm *c, int32_t which, int32_t graph, int32_t count,
                      int32_t reps, int32_t *ok, double *us_per_call) {
    return dns::guarded([&]() -> int { return dns_comm_selftest_impl(c, which, graph, count, reps, ok, us_per_call); });
}

int dns_comm_set_gather_form(dns_comm *c, int32_t form) {
    if (!c || form < 0 || form > 2)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    c->ag_form = form;
    return DNS_OK;
}

// how the all-gathers of this communicator were issued: out[0..2] = in-place
// ncclAllGather, staged ncclAllGather, group of ncclBroadcasts
int dns_comm_gather_forms(dns_comm *c, int64_t *out3) {
    if (!c || !out3) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    out3[0] = c->n_ag_inplace;
    out3[1] = c->n_ag_staged;
    out3[2] = c->n_ag_bcast;
    return DNS_OK;
}

static int dns_saddle_set_comm_impl(dns_saddle *h, dns_comm *c) {
    if (!h) return fail(DNS_ERR_BAD_ARGUMENT, "null handle");
    if (h->rank_local && c != h->comm)
        return fail(DNS_ERR_BAD_ARGUMENT,
                    "a handle created from rows keeps the communicator it was "
                    "created on");
    h->drop_graphs();
    DNS_HIP(hipSetDevice(h->device));
    DNS_TRY(h->restore_full_device()); // (a sliced handle gets its matrices back)
    h->comm = c;
    h->dist_active = false;
    h->precond_ready = false;          // set up (again) after attaching
    if (c) {
        h->st_v = partition_starts(h->nv, c->nranks);
        h->st_p = partition_starts(h->np, c->nranks);
    }
    return DNS_OK;
}

int dns_saddle_set_comm(dns_saddle *h, dns_comm *c) {
    return dns::guarded([&]() -> int { return dns_saddle_set_comm_impl(h, c); });
}

static int dns_partition_range_impl(int32_t n, int32_t nranks, int32_t rank,
                        int32_t *start, int32_t *end) {
    if (!start || !end || n < 0 || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    const std::vector<int> st = partition_starts(n, nranks);
    *start = st[rank];
    *end = st[rank + 1];
    return DNS_OK;
}

int dns_partition_range(int32_t n, int32_t nranks, int32_t rank,
                        int32_t *start, int32_t *end) {
    return dns::guarded([&]() -> int { return dns_partition_range_impl(n, nranks, rank, start, end); });
}

static int dns_device_read_impl(int device, const void *dev, void *host, size_t bytes) {
    if (!dev || !host) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    DNS_HIP(hipSetDevice(device));
    DNS_HIP(hipDeviceSynchronize());
    dns::log_host_copy("device_read", host, dev, bytes);
    DNS_TRY(dns::staged_d2h(host, dev, bytes, nullptr));
    return DNS_OK;
}

int dns_device_read(int device, const void *dev, void *host, size_t bytes) {
    return dns::guarded([&]() -> int { return dns_device_read_impl(device, dev, host, bytes); });
}

static int dns_device_write_impl(int device, void *dev, const void *host, size_t bytes) {
    if (!dev || !host) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    DNS_HIP(hipSetDevice(device));
    DNS_HIP(hipDeviceSynchronize());
    dns::log_host_copy("device_write", host, dev, bytes);
    DNS_TRY(dns::staged_h2d(dev, host, bytes, nullptr));
    return DNS_OK;
}

int dns_device_write(int device, void *dev, const void *host, size_t bytes) {
    return dns::guarded([&]() -> int { return dns_device_write_impl(device, dev, host, bytes); });
}

static int dns_saddle_probe_impl(dns_saddle *h, int32_t which, int32_t chain,
                     int32_t reps, double *us_per_launch) {
    if (!h || !us_per_launch || chain < 1 || reps < 1)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    if (!h->precond_ready) return fail(DNS_ERR_NOT_READY, "no preconditioner");
    DNS_HIP(hipSetDevice(h->device));
    DNS_TRY(h->ensure_workspace(8));
    if (h->histdev.n < 1024) DNS_TRY(h->histdev.alloc(1024));
    h->hist_cap = h->histdev.n;
    hipStream_t s = h->stream;
    const int j = 3, n = h->n, nv = h->nv, np = h->np;
    const bool dense = h->popts.schur == DNS_SCHUR_DENSE;
    const int gridA =
        dense ? std::max(h->gridD, std::min((np + 3) / 4, 2048)) : h->gridD;
    double *zp = h->z.p + nv;
    auto body = [&]() -> int {
        // which == 8: the four kernels of one Arnoldi step in sequence;
        // which = 1000 + 10 a + b: the pair (a, b) in sequence
        int seq[4] = {which, 0, 0, 0};
        int nseq = 1;
        if (which == 8) {
            nseq = 4;
            for (int q = 0; q < 4; ++q) seq[q] = q;
        } else if (which >= 1000) {
            nseq = 2;
            seq[0] = (which - 1000) / 10;
            seq[1] = (which - 1000) % 10;
        }
        for (int k0 = 0; k0 < chain; ++k0) {
            for (int sub = 0; sub < nseq; ++sub)
            switch (seq[sub]) {
                case 0:
                    if (dense && h->fp32_store)
                        hipLaunchKernelGGL(k_arn_head<2>, gridA, kBlock, 0, s, n,
                                           nv, np, j, h->w.p, h->partN.p,
                                           h->gridD, h->V.p, h->ld,
                                           (const void *)h->sinv32.p, zp,
                                           h->ctl.p, 1e-10, 0.0, h->partB.p,
                                           h->gridS, 1 << 30, 0, np, 0,
                                           (const double *)nullptr,
                                           (int *)nullptr, h->sld);
                    else if (dense)
                        hipLaunchKernelGGL(k_arn_head<1>, gridA, kBlock, 0, s, n,
                                           nv, np, j, h->w.p, h->partN.p,
                                           h->gridD, h->V.p, h->ld,
                                           (const void *)h->sinv.p, zp,
                                           h->ctl.p, 1e-10, 0.0, h->partB.p,
                                           h->gridS, 1 << 30, 0, np, 0, (const double *)nullptr);
                    else if (h->popts.schur == DNS_SCHUR_JACOBI)
                        // (sinv is the DIAGONAL here: np entries)
                        hipLaunchKernelGGL(k_arn_head<0>, gridA, kBlock, 0, s, n,
                                           nv, np, j, h->w.p, h->partN.p,
                                           h->gridD, h->V.p, h->ld,
                                           (const void *)h->sinv.p, zp,
                                           h->ctl.p, 1e-10, 0.0, h->partB.p,
                                           h->gridS, 1 << 30, 0, np, 0, (const double *)nullptr);
                    else
                        hipLaunchKernelGGL(k_arn_head<3>, gridA, kBlock, 0, s, n,
                                           nv, np, j, h->w.p, h->partN.p,
                                           h->gridD, h->V.p, h->ld,
                                           (const void *)nullptr, zp,
                                           h->ctl.p, 1e-10, 0.0, h->partB.p,
                                           h->gridS, 1 << 30, 0, np, 0, (const double *)nullptr);
                    break;
                case 1:
                    DNS_TRY(h->apply_fhat_part(h->V.p + (size_t)j * h->ld, zp,
                                               h->z.p, h->zero_ptr(), nullptr));
                    break;
                case 2:
                    DNS_LPR_SWITCH(
                        h->K.lpr,
                        hipLaunchKernelGGL(k_spmv_multidot<L>, h->gridC, kBlock,
                                           0, s, n, h->K.rowptr.p,
                                           h->K.colidx.p, h->K.vals.p, h->z.p,
                                           h->w.p, h->V.p, h->ld, j,
                                           h->partA.p, h->gridC, h->ctl.p,
                                           RowMap{0, n, 0, 0}, 0));
                    break;
                case 3:
                    hipLaunchKernelGGL(k_orth<0>, h->gridD, kBlock, 0, s, n,
                                       h->V.p, h->ld, h->w.p, h->partA.p,
                                       h->gridC, j, 0, h->partN.p, h->gridD,
                                       h->ctl.p);
                    break;
                case 4:
                    DNS_LPR_SWITCH(
                        h->K.lpr,
                        hipLaunchKernelGGL(k_resid_norm<L>, h->gridS, kBlock, 0,
                                           s, n, h->K.rowptr.p, h->K.colidx.p,
                                           h->K.vals.p, h->xdev.p, h->bdev.p,
                                           h->r.p, h->partR.p, h->partB.p,
                                           RowMap{0, n, 0, 0}));
                    break;
                case 5:
                    hipLaunchKernelGGL(k_arn_tail, 1, kBlock, 0, s, 0,
                                       h->partN.p, h->gridD, h->ctl.p,
                                       h->histdev.p, 0, 1 << 30, 0);
                    break;
                case 6:
                    hipLaunchKernelGGL(k_basis_combine, h->gridD, kBlock, 0, s,
                                       n, h->V.p, h->ld, h->ctl.p, h->u.p);
                    break;
                default:
                    DNS_TRY(h->apply_precond(h->u.p, h->z.p, h->zero_ptr(),
                                             nullptr));
                    break;
            }
        }
        DNS_HIP(hipGetLastError());
        return DNS_OK;
    };
    // clear the stop flag so that the guarded kernels do their work, and make
    // the convergence test of the head kernel unreachable (tol = -1)
    {
        CtlHeader hd;
        memset(&hd, 0, sizeof(hd));
        hd.tol = -1.0;
        DNS_HIP(hipMemcpyAsync(h->ctl.p, &hd, sizeof(hd), hipMemcpyHostToDevice,
                               s));
        DNS_HIP(hipStreamSynchronize(s));
    }
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    DNS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = body();
    DNS_HIP(hipStreamEndCapture(s, &g));
    if (rc != DNS_OK) return rc;
    DNS_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    DNS_HIP(hipGraphLaunch(ge, s));
    DNS_HIP(hipStreamSynchronize(s));
    DNS_HIP(hipEventRecord(h->ev0, s));
    for (int r = 0; r < reps; ++r) DNS_HIP(hipGraphLaunch(ge, s));
    DNS_HIP(hipEventRecord(h->ev1, s));
    DNS_HIP(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    DNS_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    const int per =
        (which == 7) ? 2 : (which == 8 ? 4 : (which >= 1000 ? 2 : 1));
    *us_per_launch = 1e3 * ms / ((double)reps * chain * per);
    return DNS_OK;
}

int dns_saddle_probe(dns_saddle *h, int32_t which, int32_t chain,
                     int32_t reps, double *us_per_launch) {
    return dns::guarded([&]() -> int { return dns_saddle_probe_impl(h, which, chain, reps, us_per_launch); });
}

static int dns_saddle_cheb_bounds_impl(dns_saddle *h, double *lo, double *hi) {
    if (!h || !lo || !hi) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    *lo = h->lam_lo;
    *hi = h->lam_hi;
    return DNS_OK;
}

int dns_saddle_cheb_bounds(dns_saddle *h, double *lo, double *hi) {
    return dns::guarded([&]() -> int { return dns_saddle_cheb_bounds_impl(h, lo, hi); });
}

static int dns_saddle_precond_info_impl(dns_saddle *h, int32_t cap, int64_t *out,
                            int32_t *count) {
    if (!h || !count) return fail(DNS_ERR_BAD_ARGUMENT, "null argument");
    if (!h->precond_ready)
        return fail(DNS_ERR_NOT_READY, "preconditioner not set up");
    std::vector<int64_t> v = {h->K.nnz,
                              h->fhat_explicit ? h->Gc.nnz : 0,
                              h->have_jg ? h->JG.nnz : 0,
                              (int64_t)h->np,
                              (int64_t)h->fp32_store,
                              (int64_t)h->popts.cheb_degree,
                              (int64_t)h->popts.schur};
    const int L = (h->popts.schur == DNS_SCHUR_MG) ? (int)h->mg.size() : 0;
    v.push_back(L);
    v.push_back(h->mg_nu);
    for (int l = 0; l < L; ++l) {
        v.push_back(h->mg[l].n);
        // (the whole level's operator and prolongation, also when the level
        // is stored by row blocks / the prolongation has been folded away)
        v.push_back(l + 1 < L ? std::max<int64_t>(h->mg[l].S.nnz,
                                                  h->mg[l].nnz_S) : 0);
        v.push_back(l + 1 < L ? std::max<int64_t>(h->mg[l].P.nnz,
                                                  h->mg[l].nnz_P) : 0);
    }
    // (last: device bytes of the pair format of K, 0 = the CSR kernels apply K)
    v.push_back(h->Kp.ready ? h->Kp.bytes() : 0);
    // (bytes per entry of the coarsest level's dense inverse as applied)
    v.push_back(L == 0 ? 0
                       : (h->mg_cinv16.p ? 2
                                         : (h->fp32_store && h->mg_cinv32.p ? 4
                                                                            : 8)));
    // (cycles per application of the multigrid block as it runs)
    v.push_back(L == 0 ? 0 : h->mg_cycles_eff());
    // (... in Krylov cycles of at most this many columns: 1 by default, any
    // when the option says two)
    v.push_back(L == 0 ? 0 : (h->mg_cycles_knob == 2 ? 1000 : 1));
    *count = (int32_t)v.size();
    if (out)
        for (int32_t i = 0; i < std::min<int32_t>(cap, *count); ++i)
            out[i] = v[i];
    return DNS_OK;
}

int dns_saddle_precond_info(dns_saddle *h, int32_t cap, int64_t *out,
                            int32_t *count) {
    return dns::guarded([&]() -> int { return dns_saddle_precond_info_impl(h, cap, out, count); });
}

// ---- standalone kernels ----------------------------------------------------
struct ScopedStream {
    hipStream_t s = nullptr;
    ~ScopedStream() {
        if (s) (void)hipStreamDestroy(s);
    }
};

static int dns_spmv_impl(int device, const dns_csr *a, const double *x, double *y,
             double alpha, double beta, int32_t variant) {
    DNS_TRY(check_csr(a, "A"));
    if (!x || !y) return fail(DNS_ERR_BAD_ARGUMENT, "null vector");
    DNS_HIP(hipSetDevice(device));
    ScopedStream ss;
    DNS_HIP(hipStreamCreate(&ss.s));
    CsrDev A;
    DNS_TRY(A.upload(a, ss.s));
    DevBuf<double> dx, dy, db;
    DNS_TRY(dx.alloc((size_t)a->ncols));
    DNS_TRY(dy.alloc((size_t)a->nrows));
    DNS_TRY(db.alloc((size_t)a->nrows));
    DNS_TRY(dx.upload(x, (size_t)a->ncols, ss.s));
    DNS_TRY(db.upload(y, (size_t)a->nrows, ss.s));
    DNS_TRY(launch_spmv(A, dx.p, dy.p, alpha, beta,
                        beta != 0.0 ? db.p : nullptr, variant, ss.s));
    DNS_TRY(dy.download(y, (size_t)a->nrows, ss.s));
    DNS_HIP(hipStreamSynchronize(ss.s));
    return DNS_OK;
}

int dns_spmv(int device, const dns_csr *a, const double *x, double *y,
             double alpha, double beta, int32_t variant) {
    return dns::guarded([&]() -> int { return dns_spmv_impl(device, a, x, y, alpha, beta, variant); });
}

static int dns_dot_impl(int device, int64_t n, const double *x, const double *y,
            double *out) {
    if (!x || !y || !out || n < 0)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    DNS_HIP(hipSetDevice(device));
    ScopedStream ss;
    DNS_HIP(hipStreamCreate(&ss.s));
    DevBuf<double> dx, dy, part, res;
    const int g = grid_for_elems(n);
    DNS_TRY(dx.alloc((size_t)n));
    DNS_TRY(dy.alloc((size_t)n));
    DNS_TRY(part.alloc((size_t)g));
    DNS_TRY(res.alloc(1));
    DNS_TRY(dx.upload(x, (size_t)n, ss.s));
    DNS_TRY(dy.upload(y, (size_t)n, ss.s));
    hipLaunchKernelGGL(k_dot_partials, g, kBlock, 0, ss.s, n, dx.p, dy.p,
                       part.p);
    hipLaunchKernelGGL(k_sum_partials, 1, kBlock, 0, ss.s, part.p, g, res.p);
    DNS_HIP(hipGetLastError());
    DNS_TRY(res.download(out, 1, ss.s));
    DNS_HIP(hipStreamSynchronize(ss.s));
    return DNS_OK;
}

int dns_dot(int device, int64_t n, const double *x, const double *y,
            double *out) {
    return dns::guarded([&]() -> int { return dns_dot_impl(device, n, x, y, out); });
}

static int dns_axpy_impl(int device, int64_t n, double a, const double *x, double *y) {
    if (!x || !y || n < 0) return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    DNS_HIP(hipSetDevice(device));
    ScopedStream ss;
    DNS_HIP(hipStreamCreate(&ss.s));
    DevBuf<double> dx, dy;
    DNS_TRY(dx.alloc((size_t)n));
    DNS_TRY(dy.alloc((size_t)n));
    DNS_TRY(dx.upload(x, (size_t)n, ss.s));
    DNS_TRY(dy.upload(y, (size_t)n, ss.s));
    hipLaunchKernelGGL(k_axpby, grid_for_elems(n), kBlock, 0, ss.s, n, a, dx.p,
                       1.0, dy.p);
    DNS_HIP(hipGetLastError());
    DNS_TRY(dy.download(y, (size_t)n, ss.s));
    DNS_HIP(hipStreamSynchronize(ss.s));
    return DNS_OK;
}

int dns_axpy(int device, int64_t n, double a, const double *x, double *y) {
    return dns::guarded([&]() -> int { return dns_axpy_impl(device, n, a, x, y); });
}

static int dns_gemv_impl(int device, int32_t n, const double *a_rowmajor, const double *x,
             double *y, double alpha) {
    if (!a_rowmajor || !x || !y || n < 0)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    DNS_HIP(hipSetDevice(device));
    ScopedStream ss;
    DNS_HIP(hipStreamCreate(&ss.s));
    DevBuf<double> da, dx, dy;
    DNS_TRY(da.alloc((size_t)n * n));
    DNS_TRY(dx.alloc((size_t)n));
    DNS_TRY(dy.alloc((size_t)n));
    DNS_TRY(da.upload(a_rowmajor, (size_t)n * n, ss.s));
    DNS_TRY(dx.upload(x, (size_t)n, ss.s));
    hipLaunchKernelGGL(k_gemv_rows<double>,
                       std::max(1, std::min((n + 3) / 4, 2048)), kBlock, 0,
                       ss.s, n, da.p, dx.p, dy.p, alpha,
                       (const DnsCtl *)nullptr);
    DNS_HIP(hipGetLastError());
    DNS_TRY(dy.download(y, (size_t)n, ss.s));
    DNS_HIP(hipStreamSynchronize(ss.s));
    return DNS_OK;
}

int dns_gemv(int device, int32_t n, const double *a_rowmajor, const double *x,
             double *y, double alpha) {
    return dns::guarded([&]() -> int { return dns_gemv_impl(device, n, a_rowmajor, x, y, alpha); });
}

static int dns_dense_inverse_impl(int device, int32_t n, double *a_rowmajor) {
    if (!a_rowmajor || n < 1)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    DNS_HIP(hipSetDevice(device));
    dns_saddle tmp;   // only its stream and invert_dense are used
    tmp.device = device;
    DNS_HIP(hipStreamCreateWithFlags(&tmp.stream, hipStreamNonBlocking));
    DevBuf<double> da;
    DNS_TRY(da.alloc((size_t)n * n));
    DNS_TRY(da.upload(a_rowmajor, (size_t)n * n, tmp.stream));
    DNS_TRY(tmp.invert_dense(da.p, n));
    DNS_TRY(da.download(a_rowmajor, (size_t)n * n, tmp.stream));
    DNS_HIP(hipStreamSynchronize(tmp.stream));
    return DNS_OK;
}

int dns_dense_inverse(int device, int32_t n, double *a_rowmajor) {
    return dns::guarded([&]() -> int { return dns_dense_inverse_impl(device, n, a_rowmajor); });
}

static int dns_spmv_bench_impl(int device, const dns_csr *a, int32_t variant, int32_t reps,
                   int32_t warmup, double *avg_seconds, double *checksum) {
    DNS_TRY(check_csr(a, "A"));
    if (!avg_seconds || reps < 1)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    DNS_HIP(hipSetDevice(device));
    ScopedStream ss;
    DNS_HIP(hipStreamCreate(&ss.s));
    CsrDev A;
    DNS_TRY(A.upload(a, ss.s));
    DevBuf<double> dx, dy, part, res;
    DNS_TRY(dx.alloc((size_t)a->ncols));
    DNS_TRY(dy.alloc((size_t)a->nrows));
    hipLaunchKernelGGL(k_fill_wave, grid_for_elems(a->ncols), kBlock, 0, ss.s,
                       a->ncols, dx.p, 0.37, 1.0, 0);
    for (int i = 0; i < warmup; ++i)
        DNS_TRY(launch_spmv(A, dx.p, dy.p, 1.0, 0.0, nullptr, variant, ss.s));
    hipEvent_t e0, e1;
    DNS_HIP(hipEventCreate(&e0));
    DNS_HIP(hipEventCreate(&e1));
    DNS_HIP(hipEventRecord(e0, ss.s));
    for (int i = 0; i < reps; ++i)
        DNS_TRY(launch_spmv(A, dx.p, dy.p, 1.0, 0.0, nullptr, variant, ss.s));
    DNS_HIP(hipEventRecord(e1, ss.s));
    DNS_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DNS_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_seconds = 1e-3 * ms / reps;
    if (checksum) {
        const int g = grid_for_elems(a->nrows);
        DNS_TRY(part.alloc((size_t)g));
        DNS_TRY(res.alloc(1));
        hipLaunchKernelGGL(k_dot_partials, g, kBlock, 0, ss.s,
                           (int64_t)a->nrows, dy.p, dy.p, part.p);
        hipLaunchKernelGGL(k_sum_partials, 1, kBlock, 0, ss.s, part.p, g,
                           res.p);
        DNS_TRY(res.download(checksum, 1, ss.s));
        DNS_HIP(hipStreamSynchronize(ss.s));
    }
    return DNS_OK;
}

int dns_spmv_bench(int device, const dns_csr *a, int32_t variant, int32_t reps,
                   int32_t warmup, double *avg_seconds, double *checksum) {
    return dns::guarded([&]() -> int { return dns_spmv_bench_impl(device, a, variant, reps, warmup, avg_seconds, checksum); });
}

// y = K x through the pair format (pair.hpp); DNS_ERR_BAD_ARGUMENT for odd
// sizes.  reps > 0: timed like dns_spmv_bench
// (HIP events on the launch stream), *avg_seconds = seconds per launch
static int dns_spmv_pair_impl(int device, const dns_csr *k, int32_t nv, const double *x,
                  double *y, int32_t reps, int32_t warmup, double *avg_seconds,
                  int64_t *format_bytes) {
    DNS_TRY(check_csr(k, "K"));
    if (!x || !y) return fail(DNS_ERR_BAD_ARGUMENT, "null vector");
    if (k->nrows != k->ncols)
        return fail(DNS_ERR_BAD_ARGUMENT, "K must be square");
    DNS_HIP(hipSetDevice(device));
    ScopedStream ss;
    DNS_HIP(hipStreamCreate(&ss.s));
    HostPair P;
    {
        const HostCsr kh = host_copy(k);
        const char *why = "";
        if (!host_pair_from_k(kh, nv, P, &why))
            return fail(DNS_ERR_BAD_ARGUMENT, "no pair format: %s", why);
    }
    PairDev A;
    DNS_TRY(A.upload(P, ss.s));
    if (format_bytes) *format_bytes = A.bytes();
    DevBuf<double> dx, dy;
    DNS_TRY(dx.alloc((size_t)k->ncols));
    DNS_TRY(dy.alloc((size_t)k->nrows));
    DNS_TRY(dx.upload(x, (size_t)k->ncols, ss.s));
    const StreamEpi ep = stream_epi_plain(1.0, 0.0, nullptr);
    // (warmup < 0: the diagnostic variant without the gather of x -- timing
    // only, the result is wrong by design; scripts/spmv_c16.py)
    int diag = warmup < 0 ? 1 : 0;
    if (const char *ab = getenv("DNS_PAIR_AB")) diag = atoi(ab);   // A/B record
    int gcap = 65535;
    if (const char *gc = getenv("DNS_PAIR_GRID")) gcap = std::max(1, atoi(gc));
    for (int i = 0; i < std::max(1, std::abs((int)warmup)); ++i)
        DNS_TRY(launch_pair16x(A, dx.p, dy.p, ep, ss.s, nullptr, gcap, diag));
    if (reps > 0 && avg_seconds) {
        hipEvent_t e0, e1;
        DNS_HIP(hipEventCreate(&e0));
        DNS_HIP(hipEventCreate(&e1));
        DNS_HIP(hipEventRecord(e0, ss.s));
        for (int i = 0; i < reps; ++i)
            DNS_TRY(launch_pair16x(A, dx.p, dy.p, ep, ss.s, nullptr, gcap,
                                   diag));
        DNS_HIP(hipEventRecord(e1, ss.s));
        DNS_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        DNS_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_seconds = 1e-3 * ms / reps;
    }
    DNS_TRY(dy.download(y, (size_t)k->nrows, ss.s));
    DNS_HIP(hipStreamSynchronize(ss.s));
    return DNS_OK;
}

int dns_spmv_pair(int device, const dns_csr *k, int32_t nv, const double *x,
                  double *y, int32_t reps, int32_t warmup, double *avg_seconds,
                  int64_t *format_bytes) {
    return dns::guarded([&]() -> int { return dns_spmv_pair_impl(device, k, nv, x, y, reps, warmup, avg_seconds, format_bytes); });
}

static int dns_hbm_probe_impl(int device, int64_t bytes, int32_t kind, int32_t reps,
                  double *gbytes_per_s) {
    if (!gbytes_per_s || reps < 1 || bytes < 4096 || kind < 0 || kind > 8)
        return fail(DNS_ERR_BAD_ARGUMENT, "bad argument");
    DNS_HIP(hipSetDevice(device));
    ScopedStream ss;
    DNS_HIP(hipStreamCreate(&ss.s));
    // kinds 3..5: read with 8 loads in flight on grids of 8/16/32 blocks/CU
    // 6: 8-byte lane loads, 7/8: tile-contiguous 8-byte loads (8 wg/CU, 1 tile/wg)
    const int narr = (kind == 0 || kind >= 3) ? 1 : (kind == 1 ? 2 : 3);
    const int64_t n2 = bytes / (16 * narr);       // double2 elements per array
    DevBuf<double> a, b, c, part;
    DNS_TRY(a.alloc((size_t)(2 * n2)));
    if (narr > 1) DNS_TRY(b.alloc((size_t)(2 * n2)));
    if (narr > 2) DNS_TRY(c.alloc((size_t)(2 * n2)));
    int grid = 256 * (kind == 3 ? 8 : (kind == 5 ? 32 : 16));
    if (kind == 7) grid = 256 * 8;
    if (kind == 8) grid = (int)std::min<int64_t>(65535, 2 * n2 / 2048);
    DNS_TRY(part.alloc((size_t)grid));
    hipLaunchKernelGGL(k_fill_wave, 2048, kBlock, 0, ss.s, (int)std::min<int64_t>(2 * n2, 1 << 30), a.p, 0.37, 1.0, 0);
    if (narr > 1) DNS_TRY(b.zero(ss.s));
    if (narr > 2) DNS_TRY(c.zero(ss.s));
    auto launch = [&]() {
        if (kind == 6)
            hipLaunchKernelGGL(k_stream_read_b64, grid, kBlock, 0, ss.s,
                               2 * n2, a.p, part.p);
        else if (kind >= 7)
            hipLaunchKernelGGL(k_stream_read_tiles, grid, kBlock, 0, ss.s,
                               2 * n2, a.p, part.p);
        else if (kind >= 3)
            hipLaunchKernelGGL(k_stream_read8, grid, kBlock, 0, ss.s, n2,
                               (const dns_double2 *)a.p, part.p);
        else if (kind == 0)
            hipLaunchKernelGGL(k_stream_read, grid, kBlock, 0, ss.s, n2,
                               (const dns_double2 *)a.p, part.p);
        else if (kind == 1)
            hipLaunchKernelGGL(k_stream_copy, grid, kBlock, 0, ss.s, n2,
                               (const dns_double2 *)a.p, (dns_double2 *)b.p);
        else
            hipLaunchKernelGGL(k_stream_triad, grid, kBlock, 0, ss.s, n2,
                               (const dns_double2 *)a.p,
                               (const dns_double2 *)b.p, 0.5,
                               (dns_double2 *)c.p);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEvent_t e0, e1;
    DNS_HIP(hipEventCreate(&e0));
    DNS_HIP(hipEventCreate(&e1));
    DNS_HIP(hipEventRecord(e0, ss.s));
    for (int i = 0; i < reps; ++i) launch();
    DNS_HIP(hipEventRecord(e1, ss.s));
    DNS_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DNS_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    DNS_HIP(hipGetLastError());
    *gbytes_per_s = (double)(16 * n2 * narr) * reps / (1e-3 * ms) * 1e-9;
    return DNS_OK;
}

int dns_hbm_probe(int device, int64_t bytes, int32_t kind, int32_t reps,
                  double *gbytes_per_s) {
    return dns::guarded([&]() -> int { return dns_hbm_probe_impl(device, bytes, kind, reps, gbytes_per_s); });
}

}  // extern "C"

#include "dist_solve.inc"
#include "rank_local.inc"
#include "mg_rows.inc"
static void free_dist_data(dns_dist_data *d) { delete d; }
dns::RowMap dns_saddle::dist_rowmap() const { return dd->kmap; }
#include "imex_capi.inc"
#include "trap_capi.inc"
#include "ops_capi.inc"
#include "conv_capi.inc"
