// BiCGStab and IMEX right-hand-side kernels (gfx950, fp64).
#pragma once
#include "kernels.hpp"

namespace dns {

// scalars of one BiCGStab iteration, double buffered by iteration parity
struct BicgScal {
    double rho, alpha, omega, pad;
};

struct BicgCtl {
    BicgScal sc[2];
    double rho_new, alpha_new;   // stashes written by workgroup 0
    double pad[2];
};

// w = K x with up to two fused dot products: <w, a0> and <w, a1>
// (a1 == nullptr: <w, w>)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_spmv_dot2(int nrows, const int *__restrict__ rowptr,
            const int *__restrict__ colidx, const double *__restrict__ vals,
            const double *__restrict__ x, double *__restrict__ w,
            const double *__restrict__ a0, const double *__restrict__ a1,
            int ndots, double *__restrict__ part0, double *__restrict__ part1,
            const DnsCtl *ctl) {
    if (ctl->done) return;
    __shared__ double red[4];
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    double acc0 = 0.0, acc1 = 0.0;
    for (int row = sub; row < nrows; row += nsub) {
        double s = csr_row_dot<LPR>(rowptr, colidx, vals, x, row, sublane);
        if (sublane == 0) {
            w[row] = s;
            acc0 = fma(s, a0[row], acc0);
            if (ndots > 1) acc1 = fma(s, a1 ? a1[row] : s, acc1);
        }
    }
    acc0 = block_sum(acc0, red);
    if (threadIdx.x == 0) part0[blockIdx.x] = acc0;
    if (ndots > 1) {
        acc1 = block_sum(acc1, red);
        if (threadIdx.x == 0) part1[blockIdx.x] = acc1;
    }
}

// start: residual norm / tolerance / flags; rhat = r; p = v = 0
__global__ void __launch_bounds__(kBlock)
k_bicg_start(int n, const double *__restrict__ r, double *__restrict__ rhat,
             double *__restrict__ p, double *__restrict__ v,
             const double *__restrict__ part_rr,
             const double *__restrict__ part_bb, int nparts, DnsCtl *ctl,
             BicgCtl *bc, double rtol, double atol, double *__restrict__ hist) {
    __shared__ double sc[2];
    reduce_partials(part_rr, nparts, nparts, 1, sc);
    reduce_partials(part_bb, nparts, nparts, 1, sc + 1);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock) {
        rhat[i] = r[i];
        p[i] = 0.0;
        v[i] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double beta = sqrt(sc[0]), bn = sqrt(sc[1]);
        const double tol = fmax(rtol * bn, atol);
        ctl->jdone = 0;
        ctl->zero = 0;
        ctl->beta = beta;
        ctl->resnorm = beta;
        ctl->bnorm = bn;
        ctl->tol = tol;
        ctl->total_it = 0;
        ctl->status = DNS_OK;
        ctl->done = !(beta > tol) ? 1 : 0;
        ctl->conv = (beta <= tol) ? 1 : 0;
        if (isnan(beta) || isnan(tol)) ctl->status = DNS_BREAKDOWN;
        ctl->hist_len = 0;
        bc->sc[0].rho = bc->sc[0].alpha = bc->sc[0].omega = 1.0;
        bc->sc[1] = bc->sc[0];
        hist[0] = beta;
    }
}

// convergence test of the previous iteration + search direction update
//   rho_new = <rhat, r>;  beta = (rho_new/rho)(alpha/omega)
//   p = r + beta (p - omega v)
__global__ void __launch_bounds__(kBlock)
k_bicg_p(int n, const double *__restrict__ r, double *__restrict__ p,
         const double *__restrict__ v, const double *__restrict__ part_rr,
         const double *__restrict__ part_nn, int nparts, DnsCtl *ctl,
         BicgCtl *bc, int par, int maxiter, double *__restrict__ hist) {
    if (ctl->done) return;
    __shared__ double sc[2];
    reduce_partials(part_rr, nparts, nparts, 1, sc);
    reduce_partials(part_nn, nparts, nparts, 1, sc + 1);
    const double rho_new = sc[0];
    const double resn = sqrt(sc[1]);
    const BicgScal o = bc->sc[par];
    const int it = ctl->total_it;
    const bool conv = resn <= ctl->tol;
    const bool brk = (rho_new == 0.0) || isnan(rho_new) || isnan(resn);
    const bool stop = conv || brk || it >= maxiter;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->resnorm = resn;
        hist[it] = resn;
        bc->rho_new = rho_new;
        if (brk && !conv) ctl->status = DNS_BREAKDOWN;
    }
    if (stop) {
        // every workgroup takes the same decision from the same partials;
        // the flag is raised by the NEXT kernel's view through `k_bicg_flag`
        return;
    }
    const double beta = (rho_new / o.rho) * (o.alpha / o.omega);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        p[i] = fma(beta, p[i] - o.omega * v[i], r[i]);
}

// raises ctl->done if the test in k_bicg_p said stop (single thread; runs
// right after k_bicg_p so that no workgroup of k_bicg_p reads a flag written
// in its own launch)
__global__ void k_bicg_flag(DnsCtl *ctl, int maxiter) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || ctl->done) return;
    const bool conv = ctl->resnorm <= ctl->tol;
    if (conv || ctl->status != DNS_OK || ctl->total_it >= maxiter ||
        isnan(ctl->resnorm))
        ctl->done = 1;
}

// alpha = rho_new / <rhat, v>;  s = r - alpha v
__global__ void __launch_bounds__(kBlock)
k_bicg_s(int n, const double *__restrict__ r, const double *__restrict__ v,
         double *__restrict__ s, const double *__restrict__ part_rv,
         int nparts, DnsCtl *ctl, BicgCtl *bc) {
    if (ctl->done) return;
    __shared__ double sc[1];
    reduce_partials(part_rv, nparts, nparts, 1, sc);
    const double rv = sc[0];
    const double alpha = (rv != 0.0) ? bc->rho_new / rv : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        bc->alpha_new = alpha;
        if (rv == 0.0 || isnan(rv)) ctl->status = DNS_BREAKDOWN;
    }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        s[i] = fma(-alpha, v[i], r[i]);
}

// omega = <t,s>/<t,t>;  x += alpha y + omega z;  r = s - omega t;
// partials of <rhat, r> and <r, r>
__global__ void __launch_bounds__(kBlock)
k_bicg_x(int n, double *__restrict__ x, double *__restrict__ r,
         const double *__restrict__ rhat, const double *__restrict__ y,
         const double *__restrict__ z, const double *__restrict__ s,
         const double *__restrict__ t, const double *__restrict__ part_ts,
         const double *__restrict__ part_tt, int nparts,
         double *__restrict__ part_rr, double *__restrict__ part_nn,
         DnsCtl *ctl, BicgCtl *bc, int par) {
    if (ctl->done) return;
    __shared__ double sc[2];
    __shared__ double red[4];
    reduce_partials(part_ts, nparts, nparts, 1, sc);
    reduce_partials(part_tt, nparts, nparts, 1, sc + 1);
    const double omega = (sc[1] != 0.0) ? sc[0] / sc[1] : 0.0;
    const double alpha = bc->alpha_new;
    double arr = 0.0, ann = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock) {
        x[i] += alpha * y[i] + omega * z[i];
        const double rn = fma(-omega, t[i], s[i]);
        r[i] = rn;
        arr = fma(rhat[i], rn, arr);
        ann = fma(rn, rn, ann);
    }
    arr = block_sum(arr, red);
    ann = block_sum(ann, red);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = arr;
        part_nn[blockIdx.x] = ann;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        BicgScal nx;
        nx.rho = bc->rho_new;
        nx.alpha = alpha;
        nx.omega = omega;
        nx.pad = 0.0;
        bc->sc[par ^ 1] = nx;
        ctl->total_it += 1;
        if (omega == 0.0 || isnan(omega)) ctl->status = DNS_BREAKDOWN;
    }
}

// ---------------------------------------------------------------------------
// IMEX right-hand side (tiu:125-128, tiu:342-346):
//   b_v = R1 (a_c v_c + a_p v_p) + cn_c nfc_c + cn_o nfc_o + g
//   b_p = gp
// ---------------------------------------------------------------------------
// row0 / [prow0, prow1): the CSR arrays hold the `nv` velocity rows starting at
// global row `row0` (row-partitioned: this rank's block; else 0 and all rows)
// and the pressure rows [prow0, prow1) are written
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_imex_rhs(int nv, int np, const int *__restrict__ rowptr,
           const int *__restrict__ colidx, const double *__restrict__ vals,
           const double *__restrict__ v_c, const double *__restrict__ v_p,
           double a_c, double a_p, const double *__restrict__ nfc_c,
           const double *__restrict__ nfc_o, double cn_c, double cn_o,
           TabRef gtab, TabRef gptab, double *__restrict__ b, int row0,
           int nv_all, int prow0, int prow1) {
    const double *__restrict__ g = tab_row(gtab);
    const double *__restrict__ gp = tab_row(gptab);
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    for (int li = sub; li < nv; li += nsub) {
        const int row = row0 + li;
        const int k1 = rowptr[li + 1];
        double s = 0.0;
        if (a_p != 0.0) {
            for (int k = rowptr[li] + sublane; k < k1; k += LPR) {
                const int c = colidx[k];
                s = fma(vals[k], fma(a_c, v_c[c], a_p * v_p[c]), s);
            }
        } else {
            for (int k = rowptr[li] + sublane; k < k1; k += LPR)
                s = fma(vals[k], a_c * v_c[colidx[k]], s);
        }
        s = subwave_sum<LPR>(s);
        if (sublane == 0)
            b[row] = s + cn_c * nfc_c[row] + cn_o * nfc_o[row] + g[row];
    }
    (void)np;
    for (int i = prow0 + blockIdx.x * kBlock + threadIdx.x; i < prow1;
         i += gridDim.x * kBlock)
        b[nv_all + i] = gp[i];
}

// the vector part of the IMEX right-hand side (bandwidth regime: R1 v then goes
// through the streaming kernel with beta = 1):
//   b_v = cn_c nfc_c + cn_o nfc_o + g ;  b_p = gp ;  xin = a_c v_c + a_p v_p
// gptr != nullptr: the convection gather rides along (nfc_c = scale * sum of
// the cell values of the row's inverted index)
__global__ void __launch_bounds__(kBlock)
k_imex_bvec(int nv, int np, const double *__restrict__ v_c,
            const double *__restrict__ v_p, double a_c, double a_p,
            double *__restrict__ nfc_c, const double *__restrict__ nfc_o,
            double cn_c, double cn_o, TabRef gtab, TabRef gptab,
            double *__restrict__ b, double *__restrict__ xin,
            const int *__restrict__ gptr, const int *__restrict__ gidx,
            const double *__restrict__ cellvals, double conv_scale) {
    const double *__restrict__ g = tab_row(gtab);
    const double *__restrict__ gp = tab_row(gptab);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nv + np;
         i += gridDim.x * kBlock) {
        if (i < nv) {
            double nc;
            if (gptr) {
                double s = 0.0;
                const int k1 = gptr[i + 1];
                for (int k = gptr[i]; k < k1; ++k) s += cellvals[gidx[k]];
                nc = conv_scale * s;
                nfc_c[i] = nc;
            } else {
                nc = nfc_c[i];
            }
            b[i] = cn_c * nc + cn_o * nfc_o[i] + g[i];
            xin[i] = (a_p != 0.0) ? fma(a_c, v_c[i], a_p * v_p[i])
                                  : a_c * v_c[i];
        } else {
            b[i] = gp[i - nv];
        }
    }
}

// x0 = e_c * x_c + e_p * x_p   (warm start, `krylovini='upd'` snu:1496-1501)
__global__ void __launch_bounds__(kBlock)
k_lincomb2(int n, double e_c, const double *__restrict__ x_c, double e_p,
           const double *__restrict__ x_p, double *__restrict__ out) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        out[i] = fma(e_c, x_c[i], e_p * x_p[i]);
}

// x0 = 5 x_c - 10 x_p + 10 x_pp - 5 x_p3 + x_p4  (quartic extrapolation)
__global__ void __launch_bounds__(kBlock)
k_lincomb5(int n, const double *__restrict__ x_c,
           const double *__restrict__ x_p, const double *__restrict__ x_pp,
           const double *__restrict__ x_p3, const double *__restrict__ x_p4,
           double *__restrict__ out) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        out[i] = 5.0 * x_c[i] - 10.0 * x_p[i] + 10.0 * x_pp[i] -
                 5.0 * x_p3[i] + x_p4[i];
}

// x0 = 4 x_c - 6 x_p + 4 x_pp - x_p3  (cubic extrapolation of the last four)
__global__ void __launch_bounds__(kBlock)
k_lincomb4(int n, const double *__restrict__ x_c,
           const double *__restrict__ x_p, const double *__restrict__ x_pp,
           const double *__restrict__ x_p3, double *__restrict__ out) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        out[i] = 4.0 * x_c[i] - 6.0 * x_p[i] + 4.0 * x_pp[i] - x_p3[i];
}

// x0 = 3 x_c - 3 x_p + x_pp  (quadratic extrapolation of the last three steps)
__global__ void __launch_bounds__(kBlock)
k_lincomb3(int n, const double *__restrict__ x_c,
           const double *__restrict__ x_p, const double *__restrict__ x_pp,
           double *__restrict__ out) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        out[i] = 3.0 * (x_c[i] - x_p[i]) + x_pp[i];
}

}  // namespace dns
