// Device-resident trapezoidal-rule stepper with the convection linearised
// about a stored trajectory: the Newton/Picard time sweeps of the reference
// (`solve_nse` with `treat_nonl_explicit=False`, snu:1402-1566, and
// `_get_mats_rhs_ts`, snu:1016-1047).  Per step, all on the device:
//   N_n = N1(v_lin) [+ N2(v_lin)]            element kernels + gather
//   F   = M + dt/2 (A + N_n)                 fused into the gather; K, D^-1
//   rhs = M v_c + dt/2 (f_n + f_c - (A + N_c) v_c)
//   K [v_n; p~] = [rhs; fp]                  preconditioned GMRES, warm start
//   update norm dt ||v_n - v_lin||_M^2;  p = -p~/dt   (N_c v_c, f_c of the
//   right-hand side: cell values of the current velocity, taken in the same
//   launch as the element matrices of the next step)
// The linearisation points of a sweep and the velocities it produces live in
// two trajectory buffers (nslots x NV each) in HBM -- they replace the
// per-time-step .npy files of the reference (snu:1012-1014, 1424-1431).
#pragma once
#include <mutex>
#include "convection.hpp"
#include "halo.hpp"
#include "solver.hpp"

namespace dns {

// out = fv + rhsbc (+ rhscon)
__global__ void __launch_bounds__(kBlock)
k_trap_fvn(int nv, const double *__restrict__ fv,
           const double *__restrict__ rhsbc,
           const double *__restrict__ rhscon, double *__restrict__ out) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nv;
         i += gridDim.x * kBlock)
        out[i] = fv[i] + rhsbc[i] + (rhscon ? rhscon[i] : 0.0);
}

// acc[0] += scale * sum(partials)   (one workgroup; the update norm of a
// pipelined sweep stays on the device)
__global__ void __launch_bounds__(kBlock)
k_trap_acc(const double *__restrict__ partials, int nparts, double scale,
           double *__restrict__ acc) {
    __shared__ double red[4];
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += kBlock) s += partials[p];
    s = block_sum(s, red);
    if (threadIdx.x == 0) acc[0] += scale * s;
}

// Partial sums of d^T M d, d = x - y -- the M-norm of the Newton update,
// dt ||v_n - v_lin||_M^2 (snu:1557-1560) -- one partial per workgroup: d is
// formed in the gather, the row is walked once (four kernels before: d, M d,
// dot partials, accumulate).  copy_out: x is stored there on the way (the new
// velocity into its trajectory slot, snu:1012-1014).
template <int LPR>
__global__ void __launch_bounds__(kBlock)
k_trap_updnorm(int nv, const int *__restrict__ rowptr,
               const int *__restrict__ colidx, const double *__restrict__ mvals,
               const double *__restrict__ x, const double *__restrict__ y,
               double *__restrict__ part, double *__restrict__ copy_out,
               int r0 = 0, int r1 = -1, const int *__restrict__ halo = nullptr,
               int nhalo = 0, double scale = 1.0) {
    // scale: the partials are stored times this factor (dt: a pipelined sweep
    // keeps the partials of a whole batch and sums them once)
    // rows [r0, r1) only (a rank's share of the norm; r1 < 0: all); halo:
    // entries of x that are stored as well (what the rank's cells and rows
    // will read of this velocity as a linearisation point)
    if (r1 < 0) r1 = nv;
    __shared__ double red[4];
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int sublane = threadIdx.x % LPR;
    const int nsub = gridDim.x * (kBlock / LPR);
    double acc = 0.0;
    if (copy_out)
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < nhalo;
             i += gridDim.x * kBlock)
            copy_out[halo[i]] = x[halo[i]];
    for (int row = r0 + sub; row < r1; row += nsub) {
        double s = 0.0;
        const int k1 = rowptr[row + 1];
        for (int k = rowptr[row] + sublane; k < k1; k += LPR) {
            const int c = colidx[k];
            s = fma(mvals[k], x[c] - y[c], s);
        }
        s = subwave_sum<LPR>(s);
        if (sublane == 0) {
            const double xr = x[row];
            acc = fma(xr - y[row], s, acc);
            if (copy_out) copy_out[row] = xr;
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = scale * acc;
}

// kpos[z] = where the z-th non-zero of F sits in the assembled K (velocity
// rows of K start with the row of F)
__global__ void __launch_bounds__(kBlock)
k_trap_kpos(int nv, const int *__restrict__ f_rowptr,
            const int *__restrict__ k_rowptr, int *__restrict__ kpos) {
    const int sub = (blockIdx.x * kBlock + threadIdx.x) / 8;
    const int sl = threadIdx.x % 8;
    const int nsub = gridDim.x * (kBlock / 8);
    for (int row = sub; row < nv; row += nsub) {
        const int f0 = f_rowptr[row], f1 = f_rowptr[row + 1];
        const int k0 = k_rowptr[row];
        for (int k = f0 + sl; k < f1; k += 8) kpos[k] = k0 + (k - f0);
    }
}

// d = x - y
__global__ void __launch_bounds__(kBlock)
k_trap_diff(int n, const double *__restrict__ x, const double *__restrict__ y,
            double *__restrict__ d) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
         i += gridDim.x * kBlock)
        d[i] = x[i] - y[i];
}

}  // namespace dns

struct dns_trap {
    dns_saddle *sys = nullptr;
    dns_conv *conv = nullptr;
    int nslots = 0;
    dns::DevBuf<double> mvals, avals, nn_vals;
    dns::DevBuf<double> xs[6];                 // ring of [v; p~] solutions
    int cur = 0, prev = 1, pprev = 2, p3 = 3, p4 = 4, work = 5;
    int nsol = 0;
    dns::DevBuf<double> fv, fp, fvn_n, rhsbc, rhscon, b, dtmp, mtmp;
    dns::DevBuf<double> traj[2];
    // asynchronous export of trajectory slots to the host (the "async writer"
    // of SURVEY 8f4): a copy stream of its own; the solver's stream only
    // waits for a pending export before it overwrites the exported buffer
    struct Export {
        hipStream_t cstream = nullptr;
        hipEvent_t ready = nullptr, done[2] = {nullptr, nullptr};
        bool pending[2] = {false, false};
        // a queued export: the copy engine writes into a page-locked buffer
        // of the library's own; the caller's array is filled by the CPU when
        // the export is waited for (no DMA into caller memory, common.hpp)
        struct Job {
            double *host;
            void *stage;
            size_t bytes;
        };
        std::vector<Job> jobs;
        std::mutex mu;
    } exp_;
    int export_fence(int which);   // solver stream waits for a pending export
    double updnorm = 0.0;                      // sum dt ||v_n - v_lin||_M^2
    dns::DevBuf<double> updnorm_dev;           // ... its part still on the device
    // pipelined sweep: the partials of every step of a batch (times dt) side
    // by side, summed ONCE when the host polls (a one-workgroup launch per
    // step before)
    dns::DevBuf<double> upd_ring;
    int upd_slot = 0, upd_stride = 0;
    static constexpr int kUpdSlots = 128;
    int flush_updnorm();
    // the update norm of the last pipelined step, not launched yet: the
    // element launch of the next step carries it (dns::UpdJob); whoever
    // reads its results first (the partials, the trajectory slot) launches
    // it on its own -- flush_pending_upd()
    struct {
        bool on = false;
        dns::UpdJob job = {};
        int out_which = -1;                    // trajectory it writes a slot of
    } pend;
    bool env_upd_ride = true;                  // DNS_TRAP_UPD_RIDE=0: own launch
    int flush_pending_upd();
    int pipeline_c = 0;                        // > 0: steps do not synchronise
    // oversolve (DnsCtl::stop_frac) of the pipelined batches: > 0 = the
    // solves run their cycle's columns down to this fraction of the tolerance
    double over_frac = 0.0;
    bool env_fuse_r = true;                    // DNS_TRAP_FUSE_R=0: residual kernel
    // pipelined sweep on one GPU: the Krylov cycle as plain launches, not as a
    // graph replay between the plain launches of the assembly and the update
    // norm -- every switch between the two costs the queue 6-9 us (measured,
    // profiles/r05_sweeps: 11.07k -> 11.78k steps/s); DNS_TRAP_GRAPH=1: replay
    bool env_graph = false;
    double last_dt = 0.0;
    // time-dependent data of the sweeps, one row per trajectory slot (= time
    // instance): f_v(t) (forcing + controlled-boundary stiffness terms,
    // snu:1466), f_p(t), M[:, cnt] bcvals(t) (snu:1438-1441,1044-1045); the
    // convection operator's boundary values come from its own table
    dns::DevBuf<double> fv_tab, fp_tab, mbc_tab;
    bool have_fv_tab = false, have_fp_tab = false, have_mbc_tab = false;
    int cur_slot = 0;                          // time instance of xs[cur]
    // the work buffer holds this step's warm start already: the tail kernel of
    // the step before wrote it (dns::TailExtrap) for `pre_sig` = 8 nsol + order
    // and the step size `pre_dt`
    bool pre_ok = false;
    int pre_sig = -1;
    double pre_dt = 0.0;
    // position of every non-zero of F in the assembled K (un-partitioned
    // handle): the assembly kernel writes both
    dns::DevBuf<int> kpos;
    const double *fv_at(int slot) const {
        return have_fv_tab ? fv_tab.p + (size_t)slot * sys->nv : fv.p;
    }
    const double *fp_at(int slot) const {
        return have_fp_tab ? fp_tab.p + (size_t)slot * std::max(1, sys->np)
                           : fp.p;
    }
    // low-rank feedback of one step (snu:1036-1042): system F - dt/2 U V_n,
    // rhs += dt/2 U (V_c v_c); all host pointers
    struct Feedback {
        int r;
        const double *umat;      // NV x r, column major
        const double *vmat_c;    // r x NV, row major (may be null: no rhs term)
        const double *vmat_n;    // r x NV, row major
    };
    dns::DevBuf<double> fb_u, fb_vc, fb_vn, fb_y;
    // Row-partitioned system: the ASSEMBLY is partitioned like the solve --
    // a rank evaluates the cells that touch its rows, gathers the non-zeros
    // and right-hand sides of its rows, takes its share of the update norm
    // (all-reduced when the host asks) -- and no solution is gathered: the
    // preconditioned vectors of the solve are exchanged over `planX` (rows of
    // K + the dofs of those cells), which leaves the new iterate valid
    // wherever this rank reads it.  Trajectory slots hold the own rows and
    // that halo; reading one to the host gathers it first.
    struct Part {
        bool on = false;
        uint64_t gen = ~0ull;
        int v0 = 0, v1 = 0, z0 = 0, z1 = 0, nsel = 0, nhalo = 0;
        dns::DevBuf<int> sel, halo;
        dns_halo_plan planX;
    } part;
    int ensure_partition();
    int gather_slot(int which, int slot);  // collective: whole slot everywhere
    int step_impl(double dt, int lin_which, int lin_slot, int out_slot,
                  int newton, int extrapolate_x0, const dns_solve_opts *opts,
                  dns_solve_stats *stats, const Feedback *fb);
    // checkpoint of the ring (a pipelined batch that did not converge within
    // its cycle length is repeated from here): the three solutions behind
    // `cur`, the host's bookkeeping; N_c / f_c are re-assembled on restore
    dns::DevBuf<double> ck[5];
    struct {
        bool valid = false;
        int nsol = 0, cur_slot = 0;
        double last_dt = 0.0, updnorm = 0.0;
    } ckh;
};
