// Convection vector N(u)u = inner(grad(u)*u, v)*dx for P2 velocities on
// triangles, evaluated on the device (replaces the host
// FEniCS callback `f_vdp` -> `get_v_conv_conts(semi_explicit=True)` ->
// `dts.get_convvec`, reference snu:1136-1140, snu:103-107, dts:427-472).
//
// Two kernels, no atomics (bitwise reproducible):
//   k_conv_cells : eight lanes per cell (one per quadrature point); gathers the
//                  12 local velocity values
//                  (inner dofs from the state vector, Dirichlet dofs from the
//                  boundary-value table: `append_bcs_vec`, dts:49-64, fused),
//                  7-point degree-5 quadrature, writes 12 local contributions
//                  cell-contiguously (struct of arrays -> coalesced)
//   k_conv_gather: one thread per inner dof sums its contributions through an
//                  inverted index (dof -> list of (local slot, cell)) and
//                  applies the sign/scale of the caller (`-1`: goes to the rhs)
#pragma once
#include "kernels.hpp"

namespace dns {

struct ConvTables {
    double phi[7][6];       // P2 shape functions at the quadrature points
    double dphi[7][6][3];   // d(phi_a)/d(lambda_i)
    double qw[7];           // weights (sum to 1)
};

__constant__ ConvTables c_conv;

// eight lanes per cell: lane q < 7 evaluates quadrature point q, lane 7 idles;
// the twelve local sums are reduced over the eight lanes with xor shuffles and
// lane l stores slots l and l + 8.  (One thread per cell left the kernel with
// 18 workgroups and a serial chain of ~700 fp64 operations: ~10 us at N=2.)
// where the value of inner dof m comes from: a vector, or -- in the tail kernel
// of a resident time step, where the new solution is still being written by
// other workgroups -- its definition x0 + sum_j y_j Z_j evaluated per entry
struct ConvFromVec {
    const double *__restrict__ v;
    __device__ __forceinline__ double at(int m) const { return v[m]; }
};

struct ConvFromCombo {
    const double *__restrict__ x0;      // warm start of the solve
    const double *__restrict__ Z;       // kept preconditioned basis, column j
    size_t ld;
    const double *y;                    // coefficients (LDS)
    int ncols;
    __device__ __forceinline__ double at(int m) const {
        double s = x0[m];
        for (int j = 0; j < ncols; ++j) s = fma(y[j], Z[(size_t)j * ld + m], s);
        return s;
    }
};

// One quadrature point of a cell: ph[a] = phi_a(x_q), cv_i = w_q |T| (u . grad)
// u_i (x_q).  Every product and sum is spelled out (fma / __dmul_rn): the
// eight-lanes-per-cell kernel and the one-lane-per-cell kernel of the streaming
// regime call this with the same operands and get the same bits, whatever the
// compiler would contract on its own.
__device__ __forceinline__ void
conv_point(int q, const double (&gl)[3][2], const double (&ul)[6][2],
           double wq, double (&ph)[6], double &cv0, double &cv1) {
#pragma clang fp contract(off)
    double uq[2] = {0.0, 0.0};
    double g[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // g[i][d] = d_d u_i
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        ph[a] = c_conv.phi[q][a];
        const double d0 = c_conv.dphi[q][a][0], d1 = c_conv.dphi[q][a][1],
                     d2 = c_conv.dphi[q][a][2];
        const double gx =
            fma(d2, gl[2][0], fma(d1, gl[1][0], __dmul_rn(d0, gl[0][0])));
        const double gy =
            fma(d2, gl[2][1], fma(d1, gl[1][1], __dmul_rn(d0, gl[0][1])));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uq[i] = fma(ph[a], ul[a][i], uq[i]);
            g[i][0] = fma(gx, ul[a][i], g[i][0]);
            g[i][1] = fma(gy, ul[a][i], g[i][1]);
        }
    }
    cv0 = __dmul_rn(wq, fma(g[0][1], uq[1], __dmul_rn(g[0][0], uq[0])));
    cv1 = __dmul_rn(wq, fma(g[1][1], uq[1], __dmul_rn(g[1][0], uq[0])));
}

// The element computation on twelve local values already in registers
__device__ __forceinline__ void
conv_cells_compute(const double (&ul)[6][2], int ncells, int cc, int c, int q,
                   bool live, const double *__restrict__ glam,
                   const double *__restrict__ area,
                   double *__restrict__ cellvals) {
#pragma clang fp contract(off)
    const int qq = (q < 7) ? q : 0;
    double gl[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gl[k][0] = glam[(size_t)(2 * k) * ncells + cc];
        gl[k][1] = glam[(size_t)(2 * k + 1) * ncells + cc];
    }
    double ph[6];
    const double wq =
        (q < 7 && live) ? __dmul_rn(c_conv.qw[qq], area[cc]) : 0.0;
    double cv0, cv1;
    conv_point(qq, gl, ul, wq, ph, cv0, cv1);
    double mine = 0.0, mine8 = 0.0;
#pragma unroll
    for (int sl = 0; sl < 12; ++sl) {
        double v = __dmul_rn(ph[sl >> 1], (sl & 1) ? cv1 : cv0);
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        if (sl < 8) {
            if (q == sl) mine = v;
        } else {
            if (q == sl - 8) mine8 = v;
        }
    }
    if (live) {
        cellvals[(size_t)q * ncells + c] = mine;
        if (q < 4) cellvals[(size_t)(q + 8) * ncells + c] = mine8;
    }
}

template <typename VAL>
__device__ __forceinline__ void
conv_cells_block(int bid, int ncells, const int *__restrict__ cellmap,   // [12][ncells]
             const double *__restrict__ glam,               // [6][ncells]
             const double *__restrict__ area,
             const VAL vsrc, TabRef dbctab,
             double *__restrict__ cellvals,                 // [12][ncells]
             const int *__restrict__ sel = nullptr, int nsel = 0) {
#pragma clang fp contract(off)
    // sel: only the cells sel[0..nsel) (row-partitioned time steppers: the
    // cells that touch this rank's rows)
    const double *__restrict__ dbcvals = tab_row(dbctab);
    const int t = bid * kBlock + threadIdx.x;
    const int slot = t >> 3;
    const int q = t & 7;
    // whole 8-lane groups are live or not
    const bool live = sel ? slot < nsel : slot < ncells;
    const int c = (sel && live) ? sel[slot] : slot;
    const int cc = live ? c : 0;
    const int qq = (q < 7) ? q : 0;
    // ONE lane per slot asks the memory: lane q of a cell's eight fetches dof
    // slots q and q + 8 (map entry, then the value) and gradient entry q; the
    // group then hands the twelve values and six gradient entries round by
    // shuffles.  (Every lane fetching everything cost 31 load instructions
    // per wave for 8 cells, each returning 32-64 useful bytes: 117 us for
    // 1.2e6 cells, whatever the locality of the gathers.)
    const int lane0 = (threadIdx.x & 63) & ~7;       // first lane of the group
    const int m_a = cellmap[(size_t)q * ncells + cc];
    const int m_b = (q < 4) ? cellmap[(size_t)(q + 8) * ncells + cc] : -1;
    const double u_a = (m_a >= 0) ? vsrc.at(m_a) : dbcvals[-m_a - 1];
    const double u_b = (q < 4) ? ((m_b >= 0) ? vsrc.at(m_b) : dbcvals[-m_b - 1])
                               : 0.0;
    const double gl_q = (q < 6) ? glam[(size_t)q * ncells + cc] : 0.0;
    double ul[6][2];
#pragma unroll
    for (int k = 0; k < 8; ++k) ul[k >> 1][k & 1] = __shfl(u_a, lane0 + k, 64);
#pragma unroll
    for (int k = 8; k < 12; ++k)
        ul[k >> 1][k & 1] = __shfl(u_b, lane0 + k - 8, 64);
    double gl[3][2];
#pragma unroll
    for (int k = 0; k < 6; ++k) gl[k >> 1][k & 1] = __shfl(gl_q, lane0 + k, 64);
    double ph[6];
    const double wq =
        (q < 7 && live) ? __dmul_rn(c_conv.qw[qq], area[cc]) : 0.0;
    double cv0, cv1;
    conv_point(qq, gl, ul, wq, ph, cv0, cv1);
    double mine = 0.0, mine8 = 0.0;     // slots q and q + 8 after the reduction
#pragma unroll
    for (int sl = 0; sl < 12; ++sl) {
        double v = __dmul_rn(ph[sl >> 1], (sl & 1) ? cv1 : cv0);
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        if (sl < 8) {
            if (q == sl) mine = v;
        } else {
            if (q == sl - 8) mine8 = v;
        }
    }
    if (live) {
        cellvals[(size_t)q * ncells + c] = mine;
        if (q < 4) cellvals[(size_t)(q + 8) * ncells + c] = mine8;
    }
}

__global__ void __launch_bounds__(kBlock)
k_conv_cells(int ncells, const int *__restrict__ cellmap,
             const double *__restrict__ glam, const double *__restrict__ area,
             const double *__restrict__ v_inner, TabRef dbctab,
             double *__restrict__ cellvals, const int *__restrict__ sel,
             int nsel) {
    conv_cells_block(blockIdx.x, ncells, cellmap, glam, area,
                     ConvFromVec{v_inner}, dbctab, cellvals, sel, nsel);
}

// The same cell values with ONE LANE PER CELL (streaming regime, >= kConvLaneMin
// cells): every load and store instruction of a wave covers 64 consecutive
// cells of a slot-major array -- 256 / 512 contiguous bytes instead of eight
// 32 / 64-byte pieces --, 31 independent loads per lane are in flight, and the
// seven quadrature points are walked in registers.  The sums over the points
// follow the tree of the eight-lane kernel's shuffles, ((0+1)+(2+3)) +
// ((4+5)+(6+0)): the values are the same bit for bit.
constexpr int kConvLaneMin = 100000;   // (refine 3, 1.5e5 cells: 1402 -> 1434 steps/s; refine 2, 3.8e4: no difference)

__global__ void __launch_bounds__(kBlock)
k_conv_cells_lane(int ncells, const int *__restrict__ cellmap,
                  const double *__restrict__ glam,
                  const double *__restrict__ area,
                  const double *__restrict__ v_inner, TabRef dbctab,
                  double *__restrict__ cellvals, const int *__restrict__ sel,
                  int nsel) {
#pragma clang fp contract(off)
    const double *__restrict__ dbcvals = tab_row(dbctab);
    const int slot = blockIdx.x * kBlock + threadIdx.x;
    const bool live = sel ? slot < nsel : slot < ncells;
    if (!live) return;
    const int c = sel ? sel[slot] : slot;
    int m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = cellmap[(size_t)k * ncells + c];
    double gl[3][2];
#pragma unroll
    for (int k = 0; k < 6; ++k) gl[k >> 1][k & 1] = glam[(size_t)k * ncells + c];
    const double ar = area[c];
    double ul[6][2];
#pragma unroll
    for (int k = 0; k < 12; ++k)
        ul[k >> 1][k & 1] = (m[k] >= 0) ? v_inner[m[k]] : dbcvals[-m[k] - 1];
    double part[12][4];                 // tree of the sums over the points
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        double ph[6];
        double cv0, cv1;
        conv_point(q, gl, ul, __dmul_rn(c_conv.qw[q], ar), ph, cv0, cv1);
#pragma unroll
        for (int sl = 0; sl < 12; ++sl) {
            // (a product of its own, as in the eight-lane kernel: not to be
            // contracted into the sum behind it)
            const double v = __dmul_rn(ph[sl >> 1], (sl & 1) ? cv1 : cv0);
            // pairs (0,1) (2,3) (4,5) (6,7): v_q + v_{q^1}, the eighth lane's
            // value is zero
            if (q & 1) part[sl][q >> 1] += v;
            else part[sl][q >> 1] = v;
        }
    }
#pragma unroll
    for (int sl = 0; sl < 12; ++sl) {
        part[sl][3] += 0.0;             // v_6 + v_7, v_7 = 0
        const double s = (part[sl][0] + part[sl][1]) +
                         (part[sl][2] + part[sl][3]);
        cellvals[(size_t)sl * ncells + c] = s;
    }
}

__global__ void __launch_bounds__(kBlock)
k_conv_gather(int row0, int row1, const int *__restrict__ gptr,
              const int *__restrict__ gidx,
              const double *__restrict__ cellvals, double scale,
              double *__restrict__ out) {
    for (int r = row0 + blockIdx.x * kBlock + threadIdx.x; r < row1;
         r += gridDim.x * kBlock) {
        double s = 0.0;
        for (int k = gptr[r]; k < gptr[r + 1]; ++k) s += cellvals[gidx[k]];
        out[r] = scale * s;
    }
}

}  // namespace dns

namespace dns {

// the right-hand side of the trapezoidal step formed by the row part of the
// assembly launch (k_trap_rhs in a launch of its own before): per velocity row
//   b = (M - tdt A)[row, :] v_c + tdt (fvn + fvn_c),   b_p = fp
// (M, A in the pattern the assembly is bound to).  With it the row part works
// with SIXTEEN lanes per row -- the Dirichlet-column list, the N(u)u list and
// the row of the pattern are walked side by side -- instead of one thread
// per row (one thread walking the 23 entries of a row: 9.8k -> 8.5k steps/s).
struct TrapRhs {
    const int *rowptr, *colidx;
    const double *v_c, *fvn_c, *fp;
    double *b;
    int nv, np;
    // cells_c != nullptr: fvn_c = fv_c - gather of the cell values of the
    // current velocity (k_conv_step_cells), formed here instead of being read
    const int *cgptr, *cgidx;
    const double *cells_c, *fv_c;
    // x0 != nullptr (round 5): the start residual of the step's solve is
    // formed HERE -- r = [b_v - F x0_v - JT x0_p; fp - J x0_v] with the
    // partials of ||r||^2 and ||b||^2, one pair per workgroup of this family
    // -- instead of by a launch of its own in front of the Krylov cycle.  The
    // new values of F are being written by the OTHER workgroups of this
    // launch, so the row does not read them: F x0 = (M + tdt A) x0 from the
    // values it walks anyway + tdt x the gather of the cell vectors
    // L_cell x0_cell the element kernel has left (`cx0`, the lists of the
    // convection vector).  (Re-forming N's entries per non-zero in the row --
    // three dependent loads each -- made the launch longer than the residual
    // kernel it saves: measured, 11.3k -> 10.8k steps/s.)
    const double *x0;
    const double *cx0;       // cell values of N(v_lin) x0_v (k_conv_step_cells)
    const int *jt_rp, *jt_ci;
    const double *jt_v;
    const int *j_rp, *j_ci;
    const double *j_v;
    double *r, *partR, *partB;
    // (with x0) workgroups at the END of the row family that take the
    // pressure rows -- beside the velocity rows, not behind them
    int pblocks;
};

// The update norm of a trapezoidal step (trap.hpp, k_trap_updnorm:
// dt ||v_n - v_lin||_M^2 and the new velocity into its trajectory slot) as a
// JOB the element launch of the NEXT step carries in extra workgroups: what it
// reads (the new velocity, its linearisation point, M) is complete and stays
// untouched until that step's solve ends -- one launch per step less.
struct UpdJob {
    const int *rowptr, *colidx;
    const double *mvals, *x, *y;
    double *part, *copy_out;
    int nv, nblk, lpr;
    double scale;
};

template <int LPR>
__device__ __forceinline__ void updnorm_block(int bid, const UpdJob &u,
                                              double *red) {
    const int sub = (bid * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = u.nblk * (kBlock / LPR);
    double acc = 0.0;
    for (int row = sub; row < u.nv; row += nsub) {
        double s = 0.0;
        const int k1 = u.rowptr[row + 1];
        for (int k = u.rowptr[row] + sublane; k < k1; k += LPR) {
            const int c = u.colidx[k];
            s = fma(u.mvals[k], u.x[c] - u.y[c], s);
        }
        s = subwave_sum<LPR>(s);
        if (sublane == 0) {
            const double xr = u.x[row];
            acc = fma(xr - u.y[row], s, acc);
            if (u.copy_out) u.copy_out[row] = xr;
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) u.part[bid] = u.scale * acc;
}

}  // namespace dns

struct dns_conv {
    int device = 0;
    int ncells = 0, nv_inner = 0, ndbc = 0;
    dns::DevBuf<int> cellmap, gptr, gidx;
    dns::DevBuf<double> glam, area, dbcvals, cellvals;
    // Dirichlet values that change from step to step (moving / controlled
    // boundaries): a table of `dbc_rows` value sets; the row is the device
    // step counter of the stepper the operator is attached to (`dbc_ctr`) or a
    // row the host names (`dbc_row`, trapezoidal sweeps: the slot)
    dns::DevBuf<double> dbc_tab;
    int dbc_rows = 0, dbc_row = 0;
    const int *dbc_ctr = nullptr;
    uint64_t dbc_gen = 0;     // bumped whenever the values / the table change
    dns::TabRef dbc_ref(int row_shift = 0) const {
        if (dbc_rows <= 0) return {dbcvals.p, nullptr, 0, 1};
        if (dbc_ctr) return {dbc_tab.p, dbc_ctr, std::max(1, ndbc), dbc_rows};
        const int r = std::min(std::max(dbc_row + row_shift, 0), dbc_rows - 1);
        return {dbc_tab.p + (size_t)r * std::max(1, ndbc), nullptr, 0, 1};
    }
    int lane_min = dns::kConvLaneMin;      // DNS_CONV_LANE_MIN (read at create)
    dns::DevBuf<double> cellvals_c;        // cell values of a second velocity
    dns::DevBuf<double> cellvals_x0;       // ... of N(v_lin) x0 (trapezoidal
                                           // stepper: residual in the gather)
                                           // (trapezoidal step: the current one)
    int enqueue_step_cells(const double *v_lin, int newton, const double *x_c,
                           int row_shift_c, hipStream_t s, const int *sel,
                           int nsel,
                           const double *x0 = nullptr,
                           const dns::UpdJob *upd = nullptr);
    std::vector<int> cmap_host;            // [12][ncells], as on the device
    std::vector<int> gptr_host, gidx_host; // the inverted index, as on the device
    struct dns_conv_mat *mat = nullptr;    // bound matrix pattern (optional)
    ~dns_conv();
    // local matrices of N1(u) (+ N2(u)) -> mat->L
    int enqueue_mat_cells(const double *v_dev, int newton, hipStream_t s,
                          const int *sel = nullptr, int nsel = 0);
    // non-zeros [z0, z1) / rows [r0, r1) of the bound pattern (one rank's)
    struct RowRange {
        int z0, z1, r0, r1;
    };
    // nvals (and F = M + tdt (A + N) if fvals) in the bound pattern
    int enqueue_mat_gather(double *nvals, const double *mvals,
                           const double *avals, double tdt, double *fvals,
                           hipStream_t s);
    int enqueue_bc_gather(double *rhsbc, hipStream_t s);
    // both in one launch (+ fvn = fv + rhsbc (+ rhscon) if fvn)
    int enqueue_mat_bc_gather(double *nvals, const double *mvals,
                              const double *avals, double tdt, double *fvals,
                              double *rhsbc, const double *fv,
                              const double *rhscon, double *fvn,
                              hipStream_t s, const int *kpos = nullptr,
                              double *kvals = nullptr,
                              bool rhscon_from_cells = false,
                              const RowRange *rr = nullptr,
                              const dns::TrapRhs *rhs = nullptr);
    // element kernel alone (the gather is fused into the step prologue)
    int enqueue_cells(const double *v_dev, hipStream_t s,
                      const int *sel = nullptr, int nsel = 0) {
        const int live = sel ? nsel : ncells;
        if (live <= 0) return DNS_OK;
        if (live >= lane_min) {
            // streaming regime: one lane per cell (same values, bit for bit)
            hipLaunchKernelGGL(dns::k_conv_cells_lane,
                               (live + dns::kBlock - 1) / dns::kBlock,
                               dns::kBlock, 0, s, ncells, cellmap.p, glam.p,
                               area.p, v_dev, dbc_ref(), cellvals.p, sel, nsel);
            DNS_HIP(hipGetLastError());
            return DNS_OK;
        }
        const int g1 = (8 * live + dns::kBlock - 1) / dns::kBlock;
        hipLaunchKernelGGL(dns::k_conv_cells, g1, dns::kBlock, 0, s, ncells,
                           cellmap.p, glam.p, area.p, v_dev, dbc_ref(),
                           cellvals.p, sel, nsel);
        DNS_HIP(hipGetLastError());
        return DNS_OK;
    }
    // the rows [row0, row1) of scale * N(u)u from the cells sel[0..nsel) (the
    // cells that touch those rows: dns_imex::ensure_partition)
    int enqueue_rows(const double *v_dev, double scale, double *out_dev,
                     int row0, int row1, const int *sel, int nsel,
                     hipStream_t s) {
        DNS_TRY(enqueue_cells(v_dev, s, sel, nsel));
        if (row1 <= row0) return DNS_OK;
        const int g2 = std::max(1, std::min((row1 - row0 + dns::kBlock - 1) /
                                                dns::kBlock, 2048));
        hipLaunchKernelGGL(dns::k_conv_gather, g2, dns::kBlock, 0, s, row0,
                           row1, gptr.p, gidx.p, cellvals.p, scale, out_dev);
        DNS_HIP(hipGetLastError());
        return DNS_OK;
    }
    // enqueue out_dev = scale * N(u)u[inner] for the inner velocity `v_dev`
    int enqueue(const double *v_dev, double scale, double *out_dev,
                hipStream_t s) {
        DNS_TRY(enqueue_cells(v_dev, s));
        const int g2 = std::max(1, std::min((nv_inner + dns::kBlock - 1) /
                                                dns::kBlock, 2048));
        hipLaunchKernelGGL(dns::k_conv_gather, g2, dns::kBlock, 0, s, 0,
                           nv_inner, gptr.p, gidx.p, cellvals.p, scale,
                           out_dev);
        DNS_HIP(hipGetLastError());
        return DNS_OK;
    }
};

// ---------------------------------------------------------------------------
// Linearised convection MATRICES on the device (what
// `get_v_conv_conts` -> `dts.get_convmats` + `condense_velmatsbybcs` assemble
// with FEniCS for every time step of the Newton/Picard sweeps, snu:109-133,
// snu:1443-1448, snu:1529-1534, dts:325-376, dts:610-642):
//   N1(u)[(a,i),(b,k)] = delta_ik int (u . grad phi_b) phi_a        (Picard)
//   N2(u)[(a,i),(b,k)] = int phi_b d_k u_i phi_a                    (+ Newton)
// k_conv_mat_cells : local 12x12 matrices, 8 lanes per cell, lane a < 6 owns
//                    the rows of test function a; slot-major layout
//                    L[(a,i)*12 + (b,k)][cell] (coalesced stores)
// k_conv_mat_gather: one thread per non-zero of the bound CSR pattern sums its
//                    contributions through an inverted index (fixed order, no
//                    atomics) and, fused, forms  F = M + tdt (A + N)
// k_conv_bc_gather : rhsbc = -N[:, Dirichlet columns] * Dirichlet values
// ---------------------------------------------------------------------------
namespace dns {

// waves per SIMD the element-matrix kernels are compiled for: left alone the
// compiler takes 256 VGPRs + 74 AGPRs (ONE wave per SIMD: 256 workgroups on
// the whole chip, the 342 of a Newton step run in two rounds)
#ifndef DNS_CELLS_WAVES
#define DNS_CELLS_WAVES 1
#endif
#ifndef DNS_CELLS_QUNROLL
#define DNS_CELLS_QUNROLL 1
#endif
constexpr int kCellsWaves = DNS_CELLS_WAVES;

__device__ __forceinline__ void
conv_mat_cells_block(int bid, int ncells, const int *__restrict__ cellmap,
                     const double *__restrict__ glam,
                     const double *__restrict__ area,
                     const double *__restrict__ v_inner, TabRef dbctab,
                     int newton, double *__restrict__ L,    // [ncells][144]
                     const int *__restrict__ sel, int nsel,
                     const double *__restrict__ x0 = nullptr,
                     double *__restrict__ cx0 = nullptr,     // [12][ncells]
                     double *__restrict__ clin = nullptr) {  // [12][ncells]
    // clin: the cell values of N(v_lin) v_lin = N1_cell u_cell ride along as
    // well (Newton's right-hand side, snu:126-133: a family of workgroups of
    // its own before, 114 of the 598 of a launch that has 512 places)
    // x0 / cx0: the rows (a, i) of L_cell x0_cell ride along (x0 an inner
    // vector: Dirichlet dofs contribute nothing) -- the cell values of
    // N(v_lin) x0 for the residual formed in the gather launch (TrapRhs)
    // sel: only the cells sel[0..nsel) (row-partitioned trapezoidal stepper:
    // the cells that touch this rank's rows)
    const double *__restrict__ dbcvals = tab_row(dbctab);
    const int t = bid * kBlock + threadIdx.x;
    const int slot = t >> 3;
    const int a = t & 7;
    if (slot >= (sel ? nsel : ncells) || a >= 6) return;
    const int c = sel ? sel[slot] : slot;
    double ul[6][2], xl[6][2];
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = cellmap[(size_t)(2 * b + i) * ncells + c];
            ul[b][i] = (m >= 0) ? v_inner[m] : dbcvals[-m - 1];
            xl[b][i] = (x0 && m >= 0) ? x0[m] : 0.0;
        }
    double gl[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gl[k][0] = glam[(size_t)(2 * k) * ncells + c];
        gl[k][1] = glam[(size_t)(2 * k + 1) * ncells + c];
    }
    const double ar = area[c];
    double n1[6];
    double n2[6][2][2];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        n1[b] = 0.0;
        n2[b][0][0] = n2[b][0][1] = n2[b][1][0] = n2[b][1][1] = 0.0;
    }
    // (the quadrature loop is NOT unrolled: unrolled, the seven points' basis
    // tables and temporaries take the kernel to 256 VGPRs + 74 AGPRs -- one
    // wave per SIMD, 256 workgroups on the whole chip)
#pragma unroll DNS_CELLS_QUNROLL
    for (int q = 0; q < 7; ++q) {
        double uq[2] = {0.0, 0.0};
        double g[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // g[i][k] = d_k u_i
        double gp[6][2];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const double d0 = c_conv.dphi[q][b][0], d1 = c_conv.dphi[q][b][1],
                         d2 = c_conv.dphi[q][b][2];
            gp[b][0] = d0 * gl[0][0] + d1 * gl[1][0] + d2 * gl[2][0];
            gp[b][1] = d0 * gl[0][1] + d1 * gl[1][1] + d2 * gl[2][1];
            const double ph = c_conv.phi[q][b];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uq[i] = fma(ph, ul[b][i], uq[i]);
                g[i][0] = fma(gp[b][0], ul[b][i], g[i][0]);
                g[i][1] = fma(gp[b][1], ul[b][i], g[i][1]);
            }
        }
        const double wa = c_conv.qw[q] * ar * c_conv.phi[q][a];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            n1[b] = fma(wa, uq[0] * gp[b][0] + uq[1] * gp[b][1], n1[b]);
            if (newton) {
                const double wb = wa * c_conv.phi[q][b];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    n2[b][i][0] = fma(wb, g[i][0], n2[b][i][0]);
                    n2[b][i][1] = fma(wb, g[i][1], n2[b][i][1]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int slot = (2 * a + i) * 12 + (2 * b + k);
                const double v =
                    ((i == k) ? n1[b] : 0.0) + (newton ? n2[b][i][k] : 0.0);
                L[(size_t)c * 144 + slot] = v;
            }
    if (clin) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double y = 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) y = fma(n1[b], ul[b][i], y);
            clin[(size_t)(2 * a + i) * ncells + c] = y;
        }
    }
    if (cx0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double y = 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                y = fma(n1[b], xl[b][i], y);
                if (newton) {
                    y = fma(n2[b][i][0], xl[b][0], y);
                    y = fma(n2[b][i][1], xl[b][1], y);
                }
            }
            cx0[(size_t)(2 * a + i) * ncells + c] = y;
        }
    }
}

__global__ void __launch_bounds__(kBlock, kCellsWaves)
k_conv_mat_cells(int ncells, const int *__restrict__ cellmap,
                 const double *__restrict__ glam,
                 const double *__restrict__ area,
                 const double *__restrict__ v_inner, TabRef dbctab,
                 int newton, double *__restrict__ L,
                 const int *__restrict__ sel = nullptr, int nsel = 0) {
    conv_mat_cells_block(blockIdx.x, ncells, cellmap, glam, area, v_inner,
                         dbctab, newton, L, sel, nsel);
}

// The element work of a trapezoidal step in ONE launch, three independent
// families of workgroups: [0, ga) the local matrices of N1 (+ N2) about the
// linearisation point, [ga, ga + gb) the cell values of N(x_c) x_c for the
// CURRENT velocity (f_c - N_c v_c of the right-hand side; its own boundary
// values and output array), the rest -- Newton -- the cell values of
// N(v_lin) v_lin.  (Three launches before, two of them at the end of the
// step before.)
__global__ void __launch_bounds__(kBlock, kCellsWaves)
k_conv_step_cells(int ga, int gb, int ncells, const int *__restrict__ cellmap,
                  const double *__restrict__ glam,
                  const double *__restrict__ area,
                  const double *__restrict__ v_lin, TabRef dbc_lin, int newton,
                  double *__restrict__ L, const double *__restrict__ x_c,
                  TabRef dbc_c, double *__restrict__ cells_c,
                  double *__restrict__ cells_lin,
                  const int *__restrict__ sel, int nsel,
                  const double *__restrict__ x0, double *__restrict__ cx0,
                  int gc, UpdJob upd) {
    const int bid = blockIdx.x;
    if (bid < ga) {
        // (gc == 0 with Newton: N(v_lin) v_lin comes out of this family)
        conv_mat_cells_block(bid, ncells, cellmap, glam, area, v_lin, dbc_lin,
                             newton, L, sel, nsel, x0, cx0,
                             (newton && gc == 0) ? cells_lin
                                                 : (double *)nullptr);
    } else if (bid < ga + gb) {
        conv_cells_block(bid - ga, ncells, cellmap, glam, area,
                         ConvFromVec{x_c}, dbc_c, cells_c, sel, nsel);
    } else if (bid < ga + gb + gc) {
        conv_cells_block(bid - ga - gb, ncells, cellmap, glam, area,
                         ConvFromVec{v_lin}, dbc_lin, cells_lin, sel, nsel);
    } else {
        // (the update norm of the step before, UpdJob)
        __shared__ double red[4];
        const int ub = bid - ga - gb - gc;
        if (upd.lpr <= 8)
            updnorm_block<8>(ub, upd, red);
        else if (upd.lpr <= 16)
            updnorm_block<16>(ub, upd, red);
        else if (upd.lpr <= 32)
            updnorm_block<32>(ub, upd, red);
        else
            updnorm_block<64>(ub, upd, red);
    }
}

// nvals[z] = sum of the local contributions of non-zero z;  if fvals:
// fvals[z] = mvals[z] + tdt * (avals[z] + nvals[z])
__global__ void __launch_bounds__(kBlock)
k_conv_mat_gather(int nnz, const int *__restrict__ mptr,
                  const int *__restrict__ midx, const double *__restrict__ L,
                  double *__restrict__ nvals, const double *__restrict__ mvals,
                  const double *__restrict__ avals, double tdt,
                  double *__restrict__ fvals) {
    for (int z = blockIdx.x * kBlock + threadIdx.x; z < nnz;
         z += gridDim.x * kBlock) {
        double s = 0.0;
        for (int k = mptr[z]; k < mptr[z + 1]; ++k) s += L[midx[k]];
        nvals[z] = s;
        if (fvals) fvals[z] = mvals[z] + tdt * (avals[z] + s);
    }
}

// rhsbc[r] = - sum over (local entry, Dirichlet value) pairs of row r
__global__ void __launch_bounds__(kBlock)
k_conv_bc_gather(int nrows, const int *__restrict__ bptr,
                 const int *__restrict__ bidx, const int *__restrict__ bbc,
                 const double *__restrict__ L, TabRef dbctab,
                 double *__restrict__ rhsbc) {
    const double *__restrict__ dbcvals = tab_row(dbctab);
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < nrows;
         r += gridDim.x * kBlock) {
        double s = 0.0;
        for (int k = bptr[r]; k < bptr[r + 1]; ++k)
            s = fma(L[bidx[k]], dbcvals[bbc[k]], s);
        rhsbc[r] = -s;
    }
}

// The row part of k_conv_mat_bc_gather for the trapezoidal stepper
// (`rhs.b != nullptr`): Dirichlet-column right-hand side, momentum right-hand
// side, and -- WITHR -- the start residual with its norms.  What is present
// (HASG: cell values of N(u)u gathered here, Newton; WITHR: x0; CELLS: the
// current velocity's cell values) is a template parameter: with run-time
// tests every optional load sat in a basic block of its own that ended with
// a wait.
template <bool HASG, bool WITHR, bool CELLS>
__device__ __forceinline__ void
trap_rows(int rb, int nrb, int r0, int nrows, const int *__restrict__ bptr,
          const int *__restrict__ bidx, const int *__restrict__ bbc,
          const double *__restrict__ dbcvals, const double *__restrict__ L,
          const double *__restrict__ mvals, const double *__restrict__ avals,
          double tdt, double *__restrict__ rhsbc, const double *__restrict__ fv,
          const double *__restrict__ rhscon, double *__restrict__ fvn,
          const int *__restrict__ gptr, const int *__restrict__ gidx,
          const double *__restrict__ cellvals, const TrapRhs &rhs) {
    constexpr int LR = 16;
    constexpr bool withr = WITHR;
    // (the last `pblocks` workgroups: pressure rows only)
    const int pbl = withr ? min(rhs.pblocks, nrb - 1) : 0;
    const int nrv = nrb - pbl;
    const bool prows = rb >= nrv;
    const int sub = ((prows ? rb - nrv : rb) * kBlock + threadIdx.x) / LR,
              sl = threadIdx.x % LR;
    const int nsub = (prows ? pbl : nrv) * (kBlock / LR);
    __shared__ double red[4];
    double sr = 0.0, sb = 0.0;       // this thread's share of the norms
    for (int r = r0 + sub; r < (prows ? 0 : nrows); r += nsub) {
        double s = 0.0, rc = 0.0, acc = 0.0, kx = 0.0, nx = 0.0, cc = 0.0;
        // The lists of a row -- boundary pairs, cell values of N(u)u,
        // the row of M / A, cell vectors of x0 and of the current
        // velocity, the row of J^T -- are walked in THREE memory round
        // trips for the common case (a lane's first entry of each list,
        // two of the matrix row): all list bounds, then all indices and
        // matrix values, then all gathers.  One list after the other it
        // was two round trips per list.  Longer lists finish in the
        // loops behind; every sum keeps the order it had.
        constexpr bool wcx = WITHR || CELLS;
        const int pb = bptr[r] + sl, pb1 = bptr[r + 1];
        const int pg = HASG ? gptr[r] + sl : 0, pg1 = HASG ? gptr[r + 1] : 0;
        const int pf = rhs.rowptr[r] + sl, pf1 = rhs.rowptr[r + 1];
        const int pc = wcx ? rhs.cgptr[r] + sl : 0,
                  pc1 = wcx ? rhs.cgptr[r + 1] : 0;
        const int pj = withr ? rhs.jt_rp[r] + sl : 0,
                  pj1 = withr ? rhs.jt_rp[r + 1] : 0;
        const bool vb = pb < pb1, vg = pg < pg1, vf = pf < pf1,
                   vf2 = pf + LR < pf1, vc = pc < pc1, vj = pj < pj1;
        // (a list that is empty for this lane reads entry 0 of its
        // arrays -- they are never empty -- and drops the value)
        const int ib = bidx[vb ? pb : 0], ibc = bbc[vb ? pb : 0];
        const int ig = HASG ? gidx[vg ? pg : 0] : 0;
        const int qf = vf ? pf : 0, qf2 = vf2 ? pf + LR : 0;
        const int icf = rhs.colidx[qf], icf2 = rhs.colidx[qf2];
        const double mk = mvals[qf], ak = avals[qf];
        const double mk2 = mvals[qf2], ak2 = avals[qf2];
        const int ic = wcx ? rhs.cgidx[vc ? pc : 0] : 0;
        const int qj = vj ? pj : 0;
        const int ij = withr ? rhs.jt_ci[qj] : 0;
        const double jv = withr ? rhs.jt_v[qj] : 0.0;
        // (an index read for an empty list is not used as one: a table
        // without entries holds one uninitialised element)
        const double lb = L[vb ? ib : 0], db = dbcvals[vb ? ibc : 0];
        const double gcv = HASG ? cellvals[vg ? ig : 0] : 0.0;
        const int jf = vf ? icf : 0, jf2 = vf2 ? icf2 : 0;
        const double vc1 = rhs.v_c[jf], vc2 = rhs.v_c[jf2];
        const double x01 = withr ? rhs.x0[jf] : 0.0,
                     x02 = withr ? rhs.x0[jf2] : 0.0;
        const int jc = vc ? ic : 0;
        const double cxv = withr ? rhs.cx0[jc] : 0.0;
        const double ccv = CELLS ? rhs.cells_c[jc] : 0.0;
        const double xj = withr ? rhs.x0[vj ? rhs.nv + ij : 0] : 0.0;
        if (vb) s = fma(lb, db, s);
        for (int k = pb + LR; k < pb1; k += LR)
            s = fma(L[bidx[k]], dbcvals[bbc[k]], s);
        if (vg) rc += gcv;
        for (int k = pg + LR; k < pg1; k += LR) rc += cellvals[gidx[k]];
        if (vf) {
            acc = fma(mk - tdt * ak, vc1, acc);
            if (withr) kx = fma(fma(tdt, ak, mk), x01, kx);
        }
        if (vf2) {
            acc = fma(mk2 - tdt * ak2, vc2, acc);
            if (withr) kx = fma(fma(tdt, ak2, mk2), x02, kx);
        }
        for (int k = pf + 2 * LR; k < pf1; k += LR) {
            const int c = rhs.colidx[k];
            const double mkk = mvals[k], akk = avals[k];
            acc = fma(mkk - tdt * akk, rhs.v_c[c], acc);
            if (withr) kx = fma(fma(tdt, akk, mkk), rhs.x0[c], kx);
        }
        if (withr) {
            if (vc) nx += cxv;
            for (int k = pc + LR; k < pc1; k += LR)
                nx += rhs.cx0[rhs.cgidx[k]];
            kx = fma(tdt, nx, kx);
            if (vj) kx = fma(jv, xj, kx);
            for (int k = pj + LR; k < pj1; k += LR)
                kx = fma(rhs.jt_v[k], rhs.x0[rhs.nv + rhs.jt_ci[k]], kx);
        }
        if (rhs.cells_c) {
            if (vc) cc += ccv;
            for (int k = pc + LR; k < pc1; k += LR)
                cc += rhs.cells_c[rhs.cgidx[k]];
        }
        s = subwave_sum<LR>(s);
        rc = subwave_sum<LR>(rc);
        acc = subwave_sum<LR>(acc);
        cc = subwave_sum<LR>(cc);
        if (withr) kx = subwave_sum<LR>(kx);
        if (sl == 0) {
            if (!HASG && rhscon) rc = rhscon[r];
            rhsbc[r] = -s;
            const double fn = fv[r] - s + rc;
            if (fvn) fvn[r] = fn;
            const double fc = CELLS ? rhs.fv_c[r] - cc
                                         : rhs.fvn_c[r];
            const double bv = acc + tdt * (fn + fc);
            rhs.b[r] = bv;
            if (withr) {
                const double rv = bv - kx;
                rhs.r[r] = rv;
                sr = fma(rv, rv, sr);
                sb = fma(bv, bv, sb);
            }
        }
    }
    if (!withr) {
        for (int i = rb * kBlock + threadIdx.x; i < rhs.np;
             i += nrb * kBlock)
            rhs.b[rhs.nv + i] = rhs.fp[i];
        return;
    }
    // pressure rows: b_p = fp, r_p = fp - J x0_v (sixteen lanes per row)
    for (int i = sub; i < ((prows || pbl == 0) ? rhs.np : 0); i += nsub) {
        double jx = 0.0;
        for (int k = rhs.j_rp[i] + sl; k < rhs.j_rp[i + 1]; k += LR)
            jx = fma(rhs.j_v[k], rhs.x0[rhs.j_ci[k]], jx);
        jx = subwave_sum<LR>(jx);
        if (sl == 0) {
            const double bp = rhs.fp[i], rp = bp - jx;
            rhs.b[rhs.nv + i] = bp;
            rhs.r[rhs.nv + i] = rp;
            sr = fma(rp, rp, sr);
            sb = fma(bp, bp, sb);
        }
    }
    sr = block_sum(sr, red);
    __syncthreads();
    sb = block_sum(sb, red);
    if (threadIdx.x == 0) {
        rhs.partR[rb] = sr;
        rhs.partB[rb] = sb;
    }
    return;
}

// both gathers of an assembly in ONE launch (they read the same local
// matrices and do not need each other): workgroups [0, gm) the matrix values,
// the rest the Dirichlet-column right-hand side and -- fvn != nullptr -- the
// momentum right-hand side of the trapezoidal stepper behind it,
// fvn = fv + rhsbc (+ rhscon)
__global__ void __launch_bounds__(kBlock)
k_conv_mat_bc_gather(int gm, int nnz, const int *__restrict__ mptr,
                     const int *__restrict__ midx, const double *__restrict__ L,
                     double *__restrict__ nvals,
                     const double *__restrict__ mvals,
                     const double *__restrict__ avals, double tdt,
                     double *__restrict__ fvals, int nrows,
                     const int *__restrict__ bptr, const int *__restrict__ bidx,
                     const int *__restrict__ bbc, TabRef dbctab,
                     double *__restrict__ rhsbc, const double *__restrict__ fv,
                     const double *__restrict__ rhscon,
                     double *__restrict__ fvn,
                     const int *__restrict__ kpos = nullptr,
                     double *__restrict__ kvals = nullptr,
                     const int *__restrict__ gptr = nullptr,
                     const int *__restrict__ gidx = nullptr,
                     const double *__restrict__ cellvals = nullptr,
                     int z0 = 0, int r0 = 0, TrapRhs rhs = TrapRhs{}) {
    // (z0, r0: first non-zero / first row of a range that ends at nnz /
    // nrows -- the rows of one rank)
    // kpos / kvals: the new value of F goes straight into its place in the
    // assembled K = [[F, JT], [J, 0]] as well (a scatter kernel of its own
    // before)
    if ((int)blockIdx.x < gm) {
        for (int z = z0 + blockIdx.x * kBlock + threadIdx.x; z < nnz;
             z += gm * kBlock) {
            double s = 0.0;
            // two contributions per pass (a non-zero has 1.5 on average,
            // six to eight on the diagonal): their indices first, then the
            // gathers -- the sum runs in the order of the list as before
            const int k1 = mptr[z + 1];
            for (int k = mptr[z]; k < k1; k += 2) {
                const int i0 = midx[k], i1 = midx[min(k + 1, k1 - 1)];
                const double l0 = L[i0], l1 = L[i1];
                s += l0;
                s += (k + 1 < k1) ? l1 : 0.0;
            }
            nvals[z] = s;
            if (fvals) {
                const double f = mvals[z] + tdt * (avals[z] + s);
                fvals[z] = f;
                if (kvals) kvals[kpos[z]] = f;
            }
        }
        return;
    }
    const double *__restrict__ dbcvals = tab_row(dbctab);
    const int rb = blockIdx.x - gm, nrb = gridDim.x - gm;
    if (rhs.b) {
        const bool hg = gptr != nullptr, wr = rhs.x0 != nullptr,
                   cl = rhs.cells_c != nullptr;
#define DNS_TRAP_ROWS(G, R, C)                                               \
    trap_rows<G, R, C>(rb, nrb, r0, nrows, bptr, bidx, bbc, dbcvals, L,      \
                       mvals, avals, tdt, rhsbc, fv, rhscon, fvn, gptr, gidx, \
                       cellvals, rhs)
        if (hg) {
            if (wr) { if (cl) DNS_TRAP_ROWS(true, true, true); else DNS_TRAP_ROWS(true, true, false); }
            else { if (cl) DNS_TRAP_ROWS(true, false, true); else DNS_TRAP_ROWS(true, false, false); }
        } else {
            if (wr) { if (cl) DNS_TRAP_ROWS(false, true, true); else DNS_TRAP_ROWS(false, true, false); }
            else { if (cl) DNS_TRAP_ROWS(false, false, true); else DNS_TRAP_ROWS(false, false, false); }
        }
#undef DNS_TRAP_ROWS
        return;
    }
    for (int r = r0 + rb * kBlock + threadIdx.x; r < nrows; r += nrb * kBlock) {
        double s = 0.0;
        for (int k = bptr[r]; k < bptr[r + 1]; ++k)
            s = fma(L[bidx[k]], dbcvals[bbc[k]], s);
        rhsbc[r] = -s;
        // gptr: rhscon = N(u)u gathered here from the cell values of the
        // convection vector (a gather kernel of its own before)
        double rc = 0.0;
        if (gptr) {
            for (int k = gptr[r]; k < gptr[r + 1]; ++k) rc += cellvals[gidx[k]];
        } else if (rhscon) {
            rc = rhscon[r];
        }
        if (fvn) fvn[r] = fv[r] - s + rc;
    }
}

}  // namespace dns

// the matrix side of dns_conv: tables bound to ONE CSR pattern
struct dns_conv_mat {
    int nnz = 0;
    dns::DevBuf<int> mptr, midx, bptr, bidx, bbc;
    dns::DevBuf<double> L;            // [ncells][144] local matrices
};

inline dns_conv::~dns_conv() { delete mat; }

inline int dns_conv::enqueue_mat_cells(const double *v_dev, int newton,
                                       hipStream_t s, const int *sel,
                                       int nsel) {
    const int live = sel ? nsel : ncells;
    if (live <= 0) return DNS_OK;
    const int g = (8 * live + dns::kBlock - 1) / dns::kBlock;
    hipLaunchKernelGGL(dns::k_conv_mat_cells, g, dns::kBlock, 0, s, ncells,
                       cellmap.p, glam.p, area.p, v_dev, dbc_ref(), newton,
                       mat->L.p, sel, nsel);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

inline int dns_conv::enqueue_step_cells(const double *v_lin, int newton,
                                        const double *x_c, int row_shift_c,
                                        hipStream_t s, const int *sel,
                                        int nsel, const double *x0,
                                        const dns::UpdJob *upd) {
    const int live = sel ? nsel : ncells;
    if (live <= 0)
        return upd ? dns::fail(DNS_ERR_BAD_ARGUMENT,
                               "update-norm job without cells to ride on")
                   : DNS_OK;
    if (cellvals_c.n < (size_t)12 * ncells)
        DNS_TRY(cellvals_c.alloc((size_t)12 * ncells));
    if (x0 && cellvals_x0.n < (size_t)12 * ncells)
        DNS_TRY(cellvals_x0.alloc((size_t)12 * ncells));
    const int g = (8 * live + dns::kBlock - 1) / dns::kBlock;
    // (Newton's N(v_lin) v_lin: by the matrix family itself, no third one)
    const int gc = 0;
    hipLaunchKernelGGL(dns::k_conv_step_cells,
                       2 * g + gc + (upd ? upd->nblk : 0),
                       dns::kBlock, 0, s, g, g, ncells, cellmap.p, glam.p,
                       area.p, v_lin, dbc_ref(), newton, mat->L.p, x_c,
                       dbc_ref(row_shift_c), cellvals_c.p, cellvals.p, sel,
                       nsel, x0, x0 ? cellvals_x0.p : (double *)nullptr, gc,
                       upd ? *upd : dns::UpdJob{});
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

inline int dns_conv::enqueue_mat_gather(double *nvals, const double *mvals,
                                        const double *avals, double tdt,
                                        double *fvals, hipStream_t s) {
    const int g = std::max(1, std::min((mat->nnz + dns::kBlock - 1) /
                                           dns::kBlock, 4096));
    hipLaunchKernelGGL(dns::k_conv_mat_gather, g, dns::kBlock, 0, s, mat->nnz,
                       mat->mptr.p, mat->midx.p, mat->L.p, nvals, mvals, avals,
                       tdt, fvals);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

inline int dns_conv::enqueue_mat_bc_gather(double *nvals, const double *mvals,
                                           const double *avals, double tdt,
                                           double *fvals, double *rhsbc,
                                           const double *fv,
                                           const double *rhscon, double *fvn,
                                           hipStream_t s, const int *kpos,
                                           double *kvals,
                                           bool rhscon_from_cells,
                                           const RowRange *rr,
                                           const dns::TrapRhs *rhs) {
    // rr: the non-zeros [z0, z1) and rows [r0, r1) of one rank only
    const int z0 = rr ? rr->z0 : 0, z1 = rr ? rr->z1 : mat->nnz;
    const int r0 = rr ? rr->r0 : 0, r1 = rr ? rr->r1 : nv_inner;
    const int gm = std::max(1, std::min((z1 - z0 + dns::kBlock - 1) /
                                            dns::kBlock, 4096));
    // (with the right-hand side: sixteen lanes per row)
    const int gb = std::max(1, std::min(((rhs ? 16 : 1) * (r1 - r0) +
                                         dns::kBlock - 1) / dns::kBlock, 2048)) +
                   ((rhs && rhs->x0) ? rhs->pblocks : 0);
    hipLaunchKernelGGL(dns::k_conv_mat_bc_gather, gm + gb, dns::kBlock, 0, s,
                       gm, z1, mat->mptr.p, mat->midx.p, mat->L.p, nvals,
                       mvals, avals, tdt, fvals, r1, mat->bptr.p,
                       mat->bidx.p, mat->bbc.p, dbc_ref(), rhsbc, fv, rhscon,
                       fvn, kpos, kvals,
                       rhscon_from_cells ? gptr.p : (const int *)nullptr,
                       rhscon_from_cells ? gidx.p : (const int *)nullptr,
                       rhscon_from_cells ? cellvals.p : (const double *)nullptr,
                       z0, r0, rhs ? *rhs : dns::TrapRhs{});
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}

inline int dns_conv::enqueue_bc_gather(double *rhsbc, hipStream_t s) {
    const int g = std::max(1, std::min((nv_inner + dns::kBlock - 1) /
                                           dns::kBlock, 2048));
    hipLaunchKernelGGL(dns::k_conv_bc_gather, g, dns::kBlock, 0, s, nv_inner,
                       mat->bptr.p, mat->bidx.p, mat->bbc.p, mat->L.p,
                       dbc_ref(), rhsbc);
    DNS_HIP(hipGetLastError());
    return DNS_OK;
}
