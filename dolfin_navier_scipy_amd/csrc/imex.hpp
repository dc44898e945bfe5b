// Device-resident IMEX stepper state (CNAB / SBDF2 inner loops).
#pragma once
#include "convection.hpp"
#include "solver.hpp"
#include "trap.hpp"

struct dns_imex {
    dns_saddle *sys = nullptr;
    dns::CsrDev R1;
    // two solution-space vectors [v; p~] (current, previous) + work
    // current and the four solutions before it, work
    dns::DevBuf<double> xs[6];
    dns::DevBuf<double> ck[5];     // checkpoint of the history for a batch
    dns::DevBuf<double> ckn[2];    // ... and of the convection history
    int cur = 0, prev = 1, pprev = 2, p3 = 3, p4 = 4, work = 5;
    int nsol = 0;                  // how many valid solution vectors (0..5)
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
    double last_pscale = 1.0;
    dns_conv *conv = nullptr;      // device convection: nfc_c = scale*N(v_c)v_c
    double conv_scale = -1.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~dns_imex() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    uint64_t step_key(const dns_imex_coeffs *cf) const;
    int step_device(const dns_imex_coeffs *cf, const dns_solve_opts *o,
                    dns_solve_stats *st, bool with_true_residual);
};
