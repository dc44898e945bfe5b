// Device-resident IMEX stepper state (CNAB / SBDF2 inner loops).
#pragma once
#include "convection.hpp"
#include "solver.hpp"
#include "trap.hpp"
#include "step_kernels.hpp"

struct dns_imex {
    dns_saddle *sys = nullptr;
    dns::CsrDev R1;                // all rows, or this rank's (partitioned)
    dns::HostCsr R1h;              // host copy (row blocks are cut from it)
    bool r1_rows = false;          // R1h holds this rank's rows only
                                   // (dns_imex_create_rows)
    // pair format of R1 (2x2 node blocks, pair.hpp) for the streamed
    // right-hand-side product of the bandwidth regime: R1 = M - theta dt A has
    // the block structure of F (8.5 instead of 10 bytes per non-zero, a third
    // of the gather addresses); absent when R1 is not streamed or NV is odd
    dns::PairDev R1p;
    int build_r1_pair(const dns::HostCsr &rows, int v0);
    // row-partitioned system (dist_solve.inc): the right-hand side is formed
    // for this rank's rows only -- R1 by rows, the convection from the cells
    // that touch them -- and the solution's halo entries are exchanged by
    // index lists (footprints of K, R1 and those cells) instead of gathering
    // the whole vector
    struct Partition {
        bool on = false;
        uint64_t gen = 0;          // sys->dist_generation it was built for
        const dns_conv *conv_for = nullptr;
        dns_halo_plan planX;
        dns::DevBuf<int> conv_sel;
        int nsel = 0;
        bool state_full = true;    // xs[cur] holds every rank's rows
    } part;
    int ensure_partition();
    int gather_state();
    // two solution-space vectors [v; p~] (current, previous) + work
    // current and the four solutions before it, work
    dns::DevBuf<double> xs[6];
    dns::DevBuf<double> ck[6];     // checkpoint of the history (and of the
                                   // work buffer: it holds the warm start)
                                   // for a batch
    dns::DevBuf<double> ckn[2];    // ... and of the convection history
    int cur = 0, prev = 1, pprev = 2, p3 = 3, p4 = 4, work = 5;
    int nsol = 0;                  // how many valid solution vectors (0..5)
    // the work buffer already holds this step's warm start (written by the
    // previous step's tail kernel, dns::TailExtrap) for the coefficient set
    // `pre_sig` = extrap_sig(nsol, order)
    bool pre_ok = false;
    int pre_sig = -1;
    // residual carry-over (dns_imex_coeffs.carry_residual, k_step_front MODE 2):
    // K xs[i] per ring slot and the velocity residual of the previous solve;
    // `b_valid`: b holds the right-hand side whose solution is xs[cur];
    // `carry_ok`: kxs[prev..p4] are K times the ring as it stands (primed by
    // prime_carry or kept current by carry steps)
    dns::DevBuf<double> kxs[6], rcarry, ckb, ckr;
    bool b_valid = false, carry_ok = false;
    int prime_carry(bool zero_r);
    // six-node step (step_kernels.hpp): the warm start lives in x0buf[work & 1]
    // (the tail reads one and writes the other), the residuals of the last two
    // solves in rc6[.] (same parity), the convection cell values are produced
    // by the tail of the step before.  `six_ok`: x0buf / cell values / rc6 are
    // those of the ring as it stands (primed by prime_six or kept by six-node
    // steps); DNS_STEP6=0 keeps the seven-node step
    dns::DevBuf<double> x0buf[2], rc6[2], ckx0, ckrc[2], ckcell, kx6;
    bool six_ok = false, env_six = true;
    bool env_dfront = true;   // DNS_DIST_FRONT: one-launch front of a partitioned step
    // row-partitioned step whose one-step cycle ends in k_arn_tail_lazy1: the
    // tail evaluates the convection cells of the new velocity (DNS_DIST_TAIL,
    // the step is 7 kernels instead of 8).  `dcells_ok`: the cell values on
    // the device are those of xs[cur] for the boundary values of generation
    // `dcells_gen` (left by such a tail or by prime_dcells)
    bool env_dtail = true, dcells_ok = false;
    uint64_t dcells_gen = 0;
    dns::DevBuf<double> x0c;       // the warm start, copied by the front
    // steps BUILT (launched or captured; a replayed graph is not counted), of
    // them: with the cells in the tail / with the cell kernel left out because
    // the tail before had run it (dns_imex_step_counters: tests)
    int64_t n_steps_built = 0, n_steps_tail_cells = 0, n_steps_cells_reused = 0;
    bool dtail_wanted(const dns_solve_opts *o) const;
    int prime_dcells(const dns_solve_opts *o);
    uint64_t six_conv_gen = 0;     // conv->dbc_gen the cell values belong to
    int prime_six(const dns_imex_coeffs *cf, bool keep_r);
    // coefficients of the polynomial warm start from `nsol_` solutions
    static constexpr int kExtrapFit35 = 13;
    static int extrap_coeffs(int nsol_, int order, double e[5]) {
        e[0] = 1.0;
        e[1] = e[2] = e[3] = e[4] = 0.0;
        if (nsol_ >= 5 && order == kExtrapFit35) {
            // value at the new time of the CUBIC least-squares fit through the
            // last FIVE solutions.  A warm start multiplies the final
            // residuals of the solves it is built from by its coefficients:
            // sqrt(sum c^2) = 4.9 here against 15.8 for the interpolating
            // quartic (8.3 cubic), for 1.8 x the cubic's truncation error --
            // once the start residual consists of those residuals rather
            // than of the truncation error (dt <= 1e-3: scripts/
            // recycle_probe.py) that is the better trade
            e[0] = 3.2; e[1] = -2.8; e[2] = -0.8; e[3] = 2.2; e[4] = -0.8;
            return 3;
        }
        if (order == kExtrapFit35) order = 3;      // (history still filling)
        if (nsol_ >= 5 && order >= 4) {
            e[0] = 5.0; e[1] = -10.0; e[2] = 10.0; e[3] = -5.0; e[4] = 1.0;
            return 4;
        }
        if (nsol_ >= 4 && order >= 3) {
            e[0] = 4.0; e[1] = -6.0; e[2] = 4.0; e[3] = -1.0;
            return 3;
        }
        if (nsol_ >= 3 && order >= 2) {
            e[0] = 3.0; e[1] = -3.0; e[2] = 1.0;
            return 2;
        }
        if (nsol_ >= 2 && order >= 1) {
            e[0] = 2.0; e[1] = -1.0;
            return 1;
        }
        return 0;
    }
    long steps_enqueued = 0;       // counts step_device calls (graph replay
                                   // must advance the host state itself)
    void rotate_host() {           // p4 <- p3 <- pprev <- prev <- cur <- new
        const int old = p4;
        p4 = p3;
        p3 = pprev;
        pprev = prev;
        prev = cur;
        cur = work;
        work = old;
        if (nsol < 5) nsol++;
    }
    dns::DevBuf<double> nfc[2];
    int nc = 0, no = 1;
    dns::DevBuf<double> g, gp, b;
    // per-step right-hand sides known in advance (time-dependent forcing,
    // moving Dirichlet data: what the reference's `f_tdp`, `g_tdp`, `applybcs`
    // callbacks return, tiu:114-127): row s of the tables replaces g / gp in
    // step s after the upload; `stepctr` lives on the device so that replayed
    // graphs walk through the tables without the host
    dns::DevBuf<double> gtab, gptab;
    dns::DevBuf<int> stepctr;
    int tab_rows = 0;              // 0: no table, g / gp are used
    bool tab_v = false, tab_p = false;
    int tab_pos = 0;               // host copy of the counter
    bool preparing = false;        // prepare_graphs is capturing (no launch)
    // the last prepare_graphs call went past its "already prepared" exit, i.e.
    // it captured (or tried to): the same on every rank of a partitioned run
    bool prepare_attempted = false;
    // a step counter is needed as soon as anything is tabulated
    bool tables() const {
        return tab_rows > 0 || (conv && conv->dbc_rows > 0);
    }
    int rows_left() const {
        int lim = 1 << 30;
        if (tab_rows > 0) lim = std::min(lim, tab_rows);
        if (conv && conv->dbc_rows > 0) lim = std::min(lim, conv->dbc_rows);
        return lim - tab_pos;
    }
    dns::TabRef g_ref() const {
        if (tab_rows > 0 && tab_v)
            return {gtab.p, stepctr.p, sys->nv, tab_rows};
        return {g.p, nullptr, 0, 1};
    }
    dns::TabRef gp_ref() const {
        if (tab_rows > 0 && tab_p)
            return {gptab.p, stepctr.p, std::max(1, sys->np), tab_rows};
        return {gp.p, nullptr, 0, 1};
    }
    int sync_counter();            // device counter <- tab_pos
    double last_pscale = 1.0;
    dns_conv *conv = nullptr;      // device convection: nfc_c = scale*N(v_c)v_c
    double conv_scale = -1.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // pipelined batches: what the last batch learnt is kept across
    // dns_imex_run calls (a run is then 100 % graph replays from its first
    // step on -- the driver's 20-step window sees what a 400-step window sees)
    int cpred = -1;                // predicted cycle length of a batch
    bool noslack = false;
    int noslack_hold = 1;
    // oversolve policy (multigrid Schur block, DnsCtl::stop_frac): `cpred` is
    // then the cycle length itself; it comes down only when every solve of a
    // batch stood a decade below the tolerance in FRONT of its last column
    // and goes up, without a replay, when a batch ended close to it
    int lower_hold = 0, lower_backoff = 2;
    bool lowered_last = false, spiked = false;
    int batch_len = 8;             // steps per batch: 8 -> 16 -> 32 while the
                                   // predictions hold
    uint64_t prepared_sig = 0;     // configuration the graphs were captured for
    int chi_hi = 0;                // longest cycle length they cover (with hysteresis)
    // record of the last dns_imex_run (dns_imex_run_info)
    int run_unconverged = 0, run_first_bad = -1, run_replayed = 0;
    int run_captures = 0;          // graphs captured inside the last run
    // knobs read ONCE, when the stepper is created
    bool env_step_history = false, env_debug = false, env_slack_adapt = true;
    int env_group = 8;
    double env_noslack_maxrel = 0.85;   // DNS_NOSLACK_MAXREL: no slack step
                                        // while the batch maximum of
                                        // residual / tolerance stays below
    struct HostState {
        int cur, prev, pprev, p3, p4, work, nsol, nc, no, tab_pos;
        long steps_enqueued;
        bool pre_ok;
        int pre_sig;
        bool b_valid, carry_ok, six_ok, dcells_ok;
    };
    HostState host_state() const {
        return {cur, prev, pprev, p3, p4, work, nsol, nc, no, tab_pos,
                steps_enqueued, pre_ok, pre_sig, b_valid, carry_ok, six_ok,
                dcells_ok};
    }
    void set_host_state(const HostState &s) {
        cur = s.cur; prev = s.prev; pprev = s.pprev; p3 = s.p3; p4 = s.p4;
        work = s.work; nsol = s.nsol; nc = s.nc; no = s.no;
        tab_pos = s.tab_pos;
        steps_enqueued = s.steps_enqueued;
        pre_ok = s.pre_ok;
        pre_sig = s.pre_sig;
        b_valid = s.b_valid;
        carry_ok = s.carry_ok;
        six_ok = s.six_ok;
        dcells_ok = s.dcells_ok;
    }
    std::vector<uint64_t> group_key(const dns_imex_coeffs *cf,
                                    const dns_solve_opts *o, int group) const;
    int enqueue_group(const dns_imex_coeffs *cf, const dns_solve_opts *o,
                      int group, bool launch);
    int prepare_graphs(const dns_imex_coeffs *cf, const dns_solve_opts *o);
    ~dns_imex() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    uint64_t step_key(const dns_imex_coeffs *cf) const;
    int step_device(const dns_imex_coeffs *cf, const dns_solve_opts *o,
                    dns_solve_stats *st, bool with_true_residual);
};
