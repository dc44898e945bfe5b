// The front of a resident IMEX time step in the latency regime (n ~ 1e4), cut
// along its data dependencies instead of along its operations:
//
//   k_step_front : [convection element kernel]  ||  [x0 = extrapolation,
//                  kx = K x0, rs = R1 (a_c v_c + a_p v_p)]      -- ONE launch:
//                  the first workgroups evaluate the cells, the others walk
//                  the rows; neither half needs the other (both read v_c only)
//   k_step_back  : nfc_c = scale * gather(cell values);  b = rs + cn_c nfc_c +
//                  cn_o nfc_o + g;  r = b - kx;  partials of ||r||^2, ||b||^2
//
// Before (k_conv_cells -> one prologue kernel) the three gather chains of the
// prologue -- 12.2 us of the ~53 us step -- waited for the 5.8 us element
// kernel although only the (short) convection gather depends on it.
#pragma once
#include "convection.hpp"
#include "kernels.hpp"

namespace dns {

// MODE 0: x0 = extrapolation of up to five ring vectors, gathered per entry
// MODE 1 (PRE): the warm start x0 has been written into `x0` by the previous
//   step's tail kernel (TailExtrap): the K chain gathers that ONE vector (x0
//   is not stored again)
// MODE 2 (PRE + residual carry-over): the K chain gathers the CURRENT solution
//   x_c instead; K x_c is kept per ring slot (`cr.kx_c`), so that
//     K x0 = e_c K x_c + e_p K x_p + ...          (row-local, no gather) and
//     r_c  = b_prev - K x_c                       (the TRUE residual of the
//                                                  previous step's solve)
//   cost nothing beyond a few row-local loads.  The velocity part of r_c
//   (weighted like the state it belongs to: a_c r_c + a_p r_p) is added to
//   this step's right-hand side.  A Krylov solve stopped after one step leaves
//   an error delta = K^-1 r of the same sign step after step, which the next
//   step propagates as R1 delta_v = r_v - JT delta_p - theta dt A delta_v:
//   carrying r_v cancels the leading term, what is left of a step's error in
//   the NEXT state is K^-1 [theta dt A delta_v; 0] (small for the smooth
//   components that would add up) -- the distance to the direct-solve
//   trajectory stops growing with the number of steps
//   (scripts/carry_over_model.py: 10-20x at 800 steps of the wake at Re = 100).
struct CarryRef {
    double *kx_c;                       // out: K x_c (ring slot of x_c)
    const double *kx_p, *kx_pp, *kx_p3, *kx_p4;
    double *rprev;                      // in: r of the step before, out: r_c
};

template <int LPR, int MODE>
__global__ void __launch_bounds__(kBlock)
k_step_front(int nconv_blocks,
             // --- convection half
             int ncells, const int *__restrict__ cellmap,
             const double *__restrict__ glam, const double *__restrict__ area,
             TabRef dbctab, double *__restrict__ cellvals,
             // --- row half
             int n, int nv, const int *__restrict__ k_rowptr,
             const int *__restrict__ k_colidx,
             const double *__restrict__ k_vals,
             const int *__restrict__ r_rowptr,
             const int *__restrict__ r_colidx,
             const double *__restrict__ r_vals,
             const double *__restrict__ x_c, const double *__restrict__ x_p,
             const double *__restrict__ x_pp, const double *__restrict__ x_p3,
             const double *__restrict__ x_p4, double e_c, double e_p,
             double e_pp, double e_p3, double e_p4, double a_c, double a_p,
             double *x0, double *__restrict__ kx, double *rs, CarryRef cr) {
    constexpr bool PRE = MODE == 1;
    constexpr bool CARRY = MODE == 2;
    if ((int)blockIdx.x < nconv_blocks) {
        if (ncells > 0)
            conv_cells_block(blockIdx.x, ncells, cellmap, glam, area, x_c,
                             dbctab, cellvals);
        return;
    }
    const int rb = blockIdx.x - nconv_blocks;
    const int nrb = gridDim.x - nconv_blocks;
    const int sub = (rb * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = nrb * (kBlock / LPR);
    for (int row = sub; row < n; row += nsub) {
        // two independent gather chains, first pass issued level by level
        const bool isv = row < nv;
        int kk = k_rowptr[row] + sublane;
        const int kend = k_rowptr[row + 1];
        int rk = 0, rend = 0;
        if (isv) {
            rk = r_rowptr[row] + sublane;
            rend = r_rowptr[row + 1];
        }
        const bool k_on = kk < kend, r_on = rk < rend;
        int kc = 0, rc = 0;
        double kval = 0.0, rval = 0.0;
        if (k_on) {
            kc = k_colidx[kk];
            kval = k_vals[kk];
        }
        if (r_on) {
            rc = r_colidx[rk];
            rval = r_vals[rk];
        }
        // carry-over: the row's own entries of the ring of K x products, of the
        // previous right-hand side and of the previous residual (none of them
        // depends on the gathers: they go out with the first pass)
        double h_p = 0.0, h_pp = 0.0, h_p3 = 0.0, h_p4 = 0.0, b_old = 0.0,
               r_old = 0.0;
        if (CARRY && sublane == 0) {
            if (e_p != 0.0) h_p = cr.kx_p[row];
            if (e_pp != 0.0) h_pp = cr.kx_pp[row];
            if (e_p3 != 0.0) h_p3 = cr.kx_p3[row];
            if (e_p4 != 0.0) h_p4 = cr.kx_p4[row];
            if (isv) b_old = rs[row];   // (the step before left its b here)
            if (isv && a_p != 0.0) r_old = cr.rprev[row];
        }
        double ks = 0.0, rsum = 0.0;
        if (k_on && PRE) {
            ks = kval * x0[kc];
        } else if (k_on && CARRY) {
            ks = kval * x_c[kc];
        } else if (k_on) {
            double xv = e_c * x_c[kc];
            if (e_p != 0.0) xv = fma(e_p, x_p[kc], xv);
            if (e_pp != 0.0) xv = fma(e_pp, x_pp[kc], xv);
            if (e_p3 != 0.0) xv = fma(e_p3, x_p3[kc], xv);
            if (e_p4 != 0.0) xv = fma(e_p4, x_p4[kc], xv);
            ks = kval * xv;
        }
        if (r_on) {
            double vv = a_c * x_c[rc];
            if (a_p != 0.0) vv = fma(a_p, x_p[rc], vv);
            rsum = rval * vv;
        }
        for (kk += LPR; kk < kend; kk += LPR) {
            const int c = k_colidx[kk];
            if (PRE) {
                ks = fma(k_vals[kk], x0[c], ks);
                continue;
            }
            if (CARRY) {
                ks = fma(k_vals[kk], x_c[c], ks);
                continue;
            }
            double xv = e_c * x_c[c];
            if (e_p != 0.0) xv = fma(e_p, x_p[c], xv);
            if (e_pp != 0.0) xv = fma(e_pp, x_pp[c], xv);
            if (e_p3 != 0.0) xv = fma(e_p3, x_p3[c], xv);
            if (e_p4 != 0.0) xv = fma(e_p4, x_p4[c], xv);
            ks = fma(k_vals[kk], xv, ks);
        }
        for (rk += LPR; rk < rend; rk += LPR) {
            const int c = r_colidx[rk];
            double vv = a_c * x_c[c];
            if (a_p != 0.0) vv = fma(a_p, x_p[c], vv);
            rsum = fma(r_vals[rk], vv, rsum);
        }
        ks = subwave_sum<LPR>(ks);
        rsum = subwave_sum<LPR>(rsum);
        if (sublane == 0) {
            if (MODE == 0) {
                double xv = e_c * x_c[row];
                if (e_p != 0.0) xv = fma(e_p, x_p[row], xv);
                if (e_pp != 0.0) xv = fma(e_pp, x_pp[row], xv);
                if (e_p3 != 0.0) xv = fma(e_p3, x_p3[row], xv);
                if (e_p4 != 0.0) xv = fma(e_p4, x_p4[row], xv);
                x0[row] = xv;
            }
            if (CARRY) {
                cr.kx_c[row] = ks;                   // K x_c, kept
                double k0 = e_c * ks;                // K x0 by linearity
                k0 = fma(e_p, h_p, k0);
                k0 = fma(e_pp, h_pp, k0);
                k0 = fma(e_p3, h_p3, k0);
                k0 = fma(e_p4, h_p4, k0);
                kx[row] = k0;
                if (isv) {
                    const double r_c = b_old - ks;   // true residual, step k-1
                    cr.rprev[row] = r_c;
                    rs[row] = rsum + fma(a_c, r_c, a_p * r_old);
                }
            } else {
                kx[row] = ks;
                if (isv) rs[row] = rsum;
            }
        }
    }
}

// LPR lanes per row (the convection list of a velocity dof has ~6-12 entries)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_step_back(int n, int nv, double *__restrict__ nfc_c,
            const double *__restrict__ nfc_o, double cn_c, double cn_o,
            TabRef gtab, TabRef gptab, const int *__restrict__ gptr,
            const int *__restrict__ gidx, const double *__restrict__ cellvals,
            double conv_scale, double *__restrict__ b /* in: rs */,
            double *__restrict__ r /* in: kx */, double *__restrict__ part_rr,
            double *__restrict__ part_bb) {
    __shared__ double red[4];
    const double *__restrict__ g = tab_row(gtab);
    const double *__restrict__ gp = tab_row(gptab);
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    double arr = 0.0, abb = 0.0;
    for (int row = sub; row < n; row += nsub) {
        double cvs = 0.0;
        const bool isv = row < nv;
        // the row's own entries go out before the (three loads deep) gather
        // chain of the convection: none of them depends on it
        double b_in = 0.0, r_in = 0.0, no_in = 0.0, g_in = 0.0, nc_in = 0.0;
        if (sublane == 0) {
            r_in = r[row];
            if (isv) {
                b_in = b[row];
                no_in = nfc_o[row];
                g_in = g[row];
                if (!gptr) nc_in = nfc_c[row];
            } else {
                g_in = gp[row - nv];
            }
        }
        if (isv && gptr) {
            const int gend = gptr[row + 1];
            for (int k = gptr[row] + sublane; k < gend; k += LPR)
                cvs += cellvals[gidx[k]];
        }
        cvs = subwave_sum<LPR>(cvs);
        if (sublane == 0) {
            double bv;
            if (isv) {
                const double nc = gptr ? conv_scale * cvs : nc_in;
                if (gptr) nfc_c[row] = nc;
                bv = b_in + cn_c * nc + cn_o * no_in + g_in;
            } else {
                bv = g_in;
            }
            const double rv = bv - r_in;
            b[row] = bv;
            r[row] = rv;
            arr = fma(rv, rv, arr);
            abb = fma(bv, bv, abb);
        }
    }
    arr = block_sum(arr, red);
    abb = block_sum(abb, red);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = arr;
        part_bb[blockIdx.x] = abb;
    }
}

// Start of a pipelined batch in ONE launch: checkpoint of the ring (and of
// the convection history) and reset of the batch accumulators of the control
// block -- eight hipMemcpyAsync + two hipMemsetAsync cost the host ~50 us per
// batch, which a 20-step window sees.
struct CopyList {
    const double *src[12];
    double *dst[12];
    int n[12];
    int count;
};

__global__ void __launch_bounds__(kBlock)
k_batch_begin(CopyList cl, DnsCtl *ctl) {
    for (int q = 0; q < cl.count; ++q) {
        const double *__restrict__ s = cl.src[q];
        double *__restrict__ d = cl.dst[q];
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < cl.n[q];
             i += gridDim.x * kBlock)
            d[i] = s[i];
    }
    if (ctl && blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->acc_solves = 0;
        ctl->acc_fail = 0;
        ctl->acc_iters = 0;
        ctl->acc_maxit = 0;
        ctl->acc_maxrel = 0.0;
    }
}

}  // namespace dns
