// Host-side CSR algebra used only at preconditioner SET-UP time (building the
// explicit polynomial approximate inverse); nothing here runs per solve.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

#include "../../include/dns_amd.h"

// fn(0) .. fn(nt-1) on host threads.  A thread that cannot be created (EAGAIN
// under a task limit shared with other processes of the user) must not take
// the process down -- std::thread's constructor throws, and unwinding past
// joinable threads is std::terminate: its share runs on the calling thread
// instead; every started thread is joined; an exception out of `fn` (a
// std::bad_alloc of a worker's scratch vectors) is rethrown by the caller
// after the join.
// host threads of the set-up algebra: up to 16, the cores there are, or
// DNS_HOST_THREADS (several ranks sharing the cores of one box)
inline int dns_host_threads() {
    static const int nt = [] {
        int n = (int)std::min<unsigned>(
            16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *e = getenv("DNS_HOST_THREADS"))
            n = std::max(1, std::min(64, atoi(e)));
        return n;
    }();
    return nt;
}

template <typename Fn>
inline void dns_run_threads(int nt, Fn fn) {
    if (nt <= 1) {
        fn(0);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)nt);
    auto guarded = [&](int t) {
        try {
            fn(t);
        } catch (...) {
            err[(size_t)t] = std::current_exception();
        }
    };
    th.reserve((size_t)nt);
    std::vector<int> inline_work;
    for (int t = 1; t < nt; ++t) {
        try {
            th.emplace_back(guarded, t);
        } catch (...) {
            inline_work.push_back(t);
        }
    }
    guarded(0);
    for (int t : inline_work) guarded(t);
    for (auto &x : th) x.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

namespace dns {

struct HostCsr {
    int nrows = 0, ncols = 0;
    std::vector<int> rowptr, colidx;
    std::vector<double> vals;
    int64_t nnz() const { return (int64_t)vals.size(); }
    dns_csr view() const {
        dns_csr v;
        v.nrows = nrows;
        v.ncols = ncols;
        v.nnz = nnz();
        v.rowptr = rowptr.data();
        v.colidx = colidx.data();
        v.vals = vals.data();
        return v;
    }
};

inline HostCsr host_copy(const dns_csr *a) {
    HostCsr h;
    h.nrows = a->nrows;
    h.ncols = a->ncols;
    h.rowptr.assign(a->rowptr, a->rowptr + a->nrows + 1);
    h.colidx.assign(a->colidx, a->colidx + a->nnz);
    h.vals.assign(a->vals, a->vals + a->nnz);
    return h;
}

inline HostCsr host_diag(const std::vector<double> &d) {
    HostCsr h;
    h.nrows = h.ncols = (int)d.size();
    h.rowptr.resize(d.size() + 1);
    h.colidx.resize(d.size());
    h.vals = d;
    for (size_t i = 0; i < d.size(); ++i) {
        h.rowptr[i] = (int)i;
        h.colidx[i] = (int)i;
    }
    h.rowptr[d.size()] = (int)d.size();
    return h;
}

// Rank-local construction: a matrix keeps its GLOBAL shape and row numbering,
// but only some rows are populated (the rank's own rows, later the ghost rows
// fetched from their owners); every other row is empty.  The row-wise set-up
// algebra (polynomial rows, products, slices) reads populated rows only.
// rows [r0, r0 + a->nrows) of an `nrows`-row matrix from a block of rows
inline HostCsr host_embed_rows(const dns_csr *a, int r0, int nrows) {
    HostCsr h;
    h.nrows = nrows;
    h.ncols = a->ncols;
    h.rowptr.assign((size_t)nrows + 1, 0);
    for (int i = 0; i < a->nrows; ++i)
        h.rowptr[(size_t)r0 + i + 1] = a->rowptr[i + 1];
    for (int i = r0 + a->nrows; i < nrows; ++i)
        h.rowptr[(size_t)i + 1] = (int)a->nnz;
    h.colidx.assign(a->colidx, a->colidx + a->nnz);
    h.vals.assign(a->vals, a->vals + a->nnz);
    return h;
}

// the populated rows of A and of B in one matrix (same shape; a row populated
// in both is taken from A)
inline HostCsr host_merge_rows(const HostCsr &A, const HostCsr &B) {
    HostCsr C;
    C.nrows = A.nrows;
    C.ncols = A.ncols;
    C.rowptr.assign((size_t)A.nrows + 1, 0);
    C.colidx.reserve(A.colidx.size() + B.colidx.size());
    C.vals.reserve(A.vals.size() + B.vals.size());
    for (int i = 0; i < A.nrows; ++i) {
        const HostCsr &S = (A.rowptr[i + 1] > A.rowptr[i]) ? A : B;
        C.colidx.insert(C.colidx.end(), S.colidx.begin() + S.rowptr[i],
                        S.colidx.begin() + S.rowptr[i + 1]);
        C.vals.insert(C.vals.end(), S.vals.begin() + S.rowptr[i],
                      S.vals.begin() + S.rowptr[i + 1]);
        C.rowptr[(size_t)i + 1] = (int)C.colidx.size();
    }
    return C;
}

// the rows `R` (sorted) of A as a matrix of their own, in that order; with
// `loc_of` (global index -> position in R, -1: not there) the columns are
// renumbered too and entries outside R dropped (a square matrix on the index
// set R), without it they stay global
inline HostCsr host_compact_rows(const HostCsr &A, const std::vector<int> &R,
                                 const std::vector<int> *loc_of) {
    HostCsr C;
    C.nrows = (int)R.size();
    C.ncols = loc_of ? (int)R.size() : A.ncols;
    C.rowptr.assign(R.size() + 1, 0);
    size_t cap = 0;
    for (int i : R) cap += (size_t)(A.rowptr[i + 1] - A.rowptr[i]);
    C.colidx.reserve(cap);
    C.vals.reserve(cap);
    for (size_t q = 0; q < R.size(); ++q) {
        const int i = R[q];
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
            int c = A.colidx[k];
            if (loc_of) {
                c = (*loc_of)[c];
                if (c < 0) continue;
            }
            C.colidx.push_back(c);
            C.vals.push_back(A.vals[k]);
        }
        C.rowptr[q + 1] = (int)C.colidx.size();
    }
    return C;
}

// rows [r0, r1) of C = A * B  (Gustavson, dense accumulator per row, sorted
// columns); rowlen[i - r0] = entries of row i
inline void host_spgemm_rows(const HostCsr &A, const HostCsr &B, int r0, int r1,
                             std::vector<int> &rowlen, std::vector<int> &ci,
                             std::vector<double> &va) {
    std::vector<double> acc((size_t)B.ncols, 0.0);
    std::vector<int> mark((size_t)B.ncols, -1), cols;
    rowlen.assign((size_t)(r1 - r0), 0);
    for (int i = r0; i < r1; ++i) {
        cols.clear();
        for (int ka = A.rowptr[i]; ka < A.rowptr[i + 1]; ++ka) {
            const int k = A.colidx[ka];
            const double av = A.vals[ka];
            for (int kb = B.rowptr[k]; kb < B.rowptr[k + 1]; ++kb) {
                const int j = B.colidx[kb];
                if (mark[j] != i) {
                    mark[j] = i;
                    acc[j] = 0.0;
                    cols.push_back(j);
                }
                acc[j] += av * B.vals[kb];
            }
        }
        std::sort(cols.begin(), cols.end());
        for (int j : cols) {
            ci.push_back(j);
            va.push_back(acc[j]);
        }
        rowlen[i - r0] = (int)cols.size();
    }
}

// C = A * B, row ranges on up to 16 host threads (the set-up of the explicit
// polynomial is five such products; results do not depend on the thread count)
inline HostCsr host_spgemm(const HostCsr &A, const HostCsr &B) {
    HostCsr C;
    C.nrows = A.nrows;
    C.ncols = B.ncols;
    C.rowptr.assign((size_t)A.nrows + 1, 0);
    int nt = dns_host_threads();
    if (A.nnz() < 20000) nt = 1;
    nt = std::max(1, std::min(nt, A.nrows));
    std::vector<std::vector<int>> rl((size_t)nt), cis((size_t)nt);
    std::vector<std::vector<double>> vas((size_t)nt);
    std::vector<int> start((size_t)nt + 1, 0);
    // split by non-zeros of A (work is roughly proportional)
    for (int t = 1; t < nt; ++t) {
        const int64_t target = A.nnz() * t / nt;
        start[t] = (int)(std::upper_bound(A.rowptr.begin(), A.rowptr.end(),
                                          (int)target) - A.rowptr.begin()) - 1;
        start[t] = std::max(start[t], start[t - 1]);
    }
    start[nt] = A.nrows;
    dns_run_threads(nt, [&](int t) {
        host_spgemm_rows(A, B, start[t], start[t + 1], rl[t], cis[t], vas[t]);
    });
    size_t total = 0;
    for (int t = 0; t < nt; ++t) total += cis[t].size();
    C.colidx.reserve(total);
    C.vals.reserve(total);
    for (int t = 0; t < nt; ++t) {
        for (int i = start[t]; i < start[t + 1]; ++i)
            C.rowptr[i + 1] = C.rowptr[i] + rl[t][i - start[t]];
        C.colidx.insert(C.colidx.end(), cis[t].begin(), cis[t].end());
        C.vals.insert(C.vals.end(), vas[t].begin(), vas[t].end());
    }
    return C;
}

// C = alpha*A + beta*B  (sorted columns in both)
inline HostCsr host_add(double alpha, const HostCsr &A, double beta,
                        const HostCsr &B) {
    HostCsr C;
    C.nrows = A.nrows;
    C.ncols = A.ncols;
    C.rowptr.assign((size_t)A.nrows + 1, 0);
    for (int i = 0; i < A.nrows; ++i) {
        int ka = A.rowptr[i], kb = B.rowptr[i];
        const int ea = A.rowptr[i + 1], eb = B.rowptr[i + 1];
        while (ka < ea || kb < eb) {
            const int ca = ka < ea ? A.colidx[ka] : INT32_MAX;
            const int cb = kb < eb ? B.colidx[kb] : INT32_MAX;
            if (ca == cb) {
                C.colidx.push_back(ca);
                C.vals.push_back(alpha * A.vals[ka++] + beta * B.vals[kb++]);
            } else if (ca < cb) {
                C.colidx.push_back(ca);
                C.vals.push_back(alpha * A.vals[ka++]);
            } else {
                C.colidx.push_back(cb);
                C.vals.push_back(beta * B.vals[kb++]);
            }
        }
        C.rowptr[i + 1] = (int)C.colidx.size();
    }
    return C;
}

inline void host_scale_rows(const std::vector<double> &d, HostCsr &A) {
    for (int i = 0; i < A.nrows; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) A.vals[k] *= d[i];
}

// drop entries below tol * (largest magnitude of their row); diagonal kept
inline void host_drop_small(HostCsr &A, double tol) {
    std::vector<int> rp((size_t)A.nrows + 1, 0), ci;
    std::vector<double> va;
    ci.reserve(A.colidx.size());
    va.reserve(A.vals.size());
    for (int i = 0; i < A.nrows; ++i) {
        double mx = 0.0;
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            mx = std::max(mx, std::fabs(A.vals[k]));
        const double thr = tol * mx;
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if (A.colidx[k] == i || std::fabs(A.vals[k]) >= thr) {
                ci.push_back(A.colidx[k]);
                va.push_back(A.vals[k]);
            }
        rp[i + 1] = (int)ci.size();
    }
    A.rowptr.swap(rp);
    A.colidx.swap(ci);
    A.vals.swap(va);
}

// G = p(DF) D with p the Chebyshev polynomial of the device recurrence
// (k_cheb_init / k_cheb_step):  X += Dm;  R -= DF Dm;  Dm = c1 Dm + c2 R,
// started from R = D, Dm = D / theta, G = X + Dm at the end.  All iterates are
// polynomials in DF times D, and polynomials in DF commute with DF, so ROW i of
// every iterate obeys the same recurrence with the row vector times DF:
//   x += d;  r -= d DF;  d = c1 d + c2 r        (r = e_i, d = e_i / theta)
// and row i of G is (x + d) D.  Rows are independent: dense accumulators with a
// touched list per thread, no matrix temporaries (the matrix form spent 0.4 s
// of the 0.45 s set-up at N=2 adding matrices with the pattern of F^5).
// Entries below tol * (row maximum) are dropped at the end; diagonal kept.
// `rows` (sorted): only these rows are formed, the others stay empty -- a rank
// of a row-partitioned set-up forms the rows its blocks are made of.
inline HostCsr host_cheb_poly(const HostCsr &DF, const std::vector<double> &dv,
                              double theta, const std::vector<double> &c1,
                              const std::vector<double> &c2, double tol,
                              const std::vector<int> *rows = nullptr) {
    const int n = DF.nrows;
    const int nr = rows ? (int)rows->size() : n;
    HostCsr G;
    G.nrows = G.ncols = n;
    G.rowptr.assign((size_t)n + 1, 0);
    int nt = dns_host_threads();
    nt = std::max(1, std::min(nt, nr / 64 + 1));
    std::vector<std::vector<int>> rl((size_t)nt), cis((size_t)nt);
    std::vector<std::vector<double>> vas((size_t)nt);
    auto work = [&](int t) {
        const int r0 = (int)((int64_t)nr * t / nt), r1 = (int)((int64_t)nr * (t + 1) / nt);
        std::vector<double> x((size_t)n, 0.0), r((size_t)n, 0.0),
            d((size_t)n, 0.0), w((size_t)n, 0.0);
        std::vector<char> in((size_t)n, 0);
        std::vector<int> touched;
        rl[t].assign((size_t)(r1 - r0), 0);
        for (int ii = r0; ii < r1; ++ii) {
            const int i = rows ? (*rows)[ii] : ii;
            touched.clear();
            touched.push_back(i);
            in[i] = 1;
            r[i] = 1.0;
            d[i] = 1.0 / theta;
            for (size_t s = 0; s < c1.size(); ++s) {
                const size_t nold = touched.size();
                // w = d DF (only the old entries of d are non-zero)
                for (size_t q = 0; q < nold; ++q) {
                    const int k = touched[q];
                    const double dk = d[k];
                    if (dk == 0.0) continue;
                    for (int kb = DF.rowptr[k]; kb < DF.rowptr[k + 1]; ++kb) {
                        const int j = DF.colidx[kb];
                        if (!in[j]) {
                            in[j] = 1;
                            touched.push_back(j);
                        }
                        w[j] += dk * DF.vals[kb];
                    }
                }
                for (int j : touched) {
                    x[j] += d[j];
                    r[j] -= w[j];
                    w[j] = 0.0;
                    d[j] = c1[s] * d[j] + c2[s] * r[j];
                }
            }
            std::sort(touched.begin(), touched.end());
            double mx = 0.0;
            for (int j : touched) {
                x[j] = (x[j] + d[j]) * dv[j];
                mx = std::max(mx, std::fabs(x[j]));
            }
            const double thr = tol * mx;
            int cnt = 0;
            for (int j : touched) {
                if (j == i || (tol > 0.0 ? std::fabs(x[j]) >= thr : true)) {
                    cis[t].push_back(j);
                    vas[t].push_back(x[j]);
                    ++cnt;
                }
                x[j] = r[j] = d[j] = 0.0;
                in[j] = 0;
            }
            rl[t][ii - r0] = cnt;
        }
    };
    dns_run_threads(nt, work);
    size_t total = 0;
    for (int t = 0; t < nt; ++t) total += cis[t].size();
    G.colidx.reserve(total);
    G.vals.reserve(total);
    // row lengths first (rows outside the list: zero), then the prefix sum
    for (int t = 0; t < nt; ++t) {
        const int r0 = (int)((int64_t)nr * t / nt);
        for (size_t q = 0; q < rl[t].size(); ++q) {
            const int i = rows ? (*rows)[(size_t)r0 + q] : r0 + (int)q;
            G.rowptr[(size_t)i + 1] = rl[t][q];
        }
        G.colidx.insert(G.colidx.end(), cis[t].begin(), cis[t].end());
        G.vals.insert(G.vals.end(), vas[t].begin(), vas[t].end());
    }
    for (int i = 0; i < n; ++i) G.rowptr[(size_t)i + 1] += G.rowptr[i];
    return G;
}

// A^T (sorted columns)
inline HostCsr host_transpose(const HostCsr &A) {
    HostCsr T;
    T.nrows = A.ncols;
    T.ncols = A.nrows;
    T.rowptr.assign((size_t)A.ncols + 1, 0);
    for (int c : A.colidx) T.rowptr[(size_t)c + 1]++;
    for (int i = 0; i < A.ncols; ++i) T.rowptr[i + 1] += T.rowptr[i];
    T.colidx.resize(A.colidx.size());
    T.vals.resize(A.vals.size());
    std::vector<int> pos(T.rowptr.begin(), T.rowptr.end() - 1);
    for (int i = 0; i < A.nrows; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
            const int q = pos[A.colidx[k]]++;
            T.colidx[q] = i;
            T.vals[q] = A.vals[k];
        }
    return T;
}

// largest eigenvalue (in modulus) of D^-1 A by power iterations, D = diag(A)
inline double host_jacobi_lmax(const HostCsr &A, int iters = 20) {
    const int n = A.nrows;
    std::vector<double> d((size_t)n, 1.0), x((size_t)n), y((size_t)n);
    for (int i = 0; i < n; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if (A.colidx[k] == i && A.vals[k] != 0.0) d[i] = 1.0 / A.vals[k];
    for (int i = 0; i < n; ++i) x[i] = 1.0 + 0.5 * std::sin(0.37 * i + 1.0);
    double lam = 1.0;
    for (int it = 0; it < iters; ++it) {
        double nx = 0.0, ny = 0.0;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
                s += A.vals[k] * x[A.colidx[k]];
            y[i] = d[i] * s;
            nx += x[i] * x[i];
            ny += y[i] * y[i];
        }
        lam = std::sqrt(ny / nx);
        const double sc = 1.0 / std::sqrt(ny);
        for (int i = 0; i < n; ++i) x[i] = y[i] * sc;
    }
    return lam;
}

// y = A x (host, set-up time only)
inline void host_spmv(const HostCsr &A, const std::vector<double> &x,
                      std::vector<double> &y) {
    y.assign((size_t)A.nrows, 0.0);
    for (int i = 0; i < A.nrows; ++i) {
        double s = 0.0;
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            s += A.vals[k] * x[A.colidx[k]];
        y[i] = s;
    }
}

// Bounds (lmin, lmax) of the spectrum of D^-1 A, D = diag(A), for a matrix with
// real eigenvalues (A symmetric): power iteration for lmax, shifted power
// iteration for lmin -- the host twin of dns_saddle::estimate_bounds
inline void host_jacobi_bounds(const HostCsr &A, double *lmin, double *lmax) {
    const int n = A.nrows;
    std::vector<double> d((size_t)n, 1.0), x((size_t)n), y;
    for (int i = 0; i < n; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if (A.colidx[k] == i && A.vals[k] != 0.0) d[i] = 1.0 / A.vals[k];
    auto norm = [&](const std::vector<double> &v) {
        double s = 0.0;
        for (double e : v) s += e * e;
        return std::sqrt(s);
    };
    for (int i = 0; i < n; ++i) x[i] = 1.0 + 0.5 * std::sin(0.37 * i + 7.0);
    double lam = 1.0;
    for (int it = 0; it < 30; ++it) {
        host_spmv(A, x, y);
        for (int i = 0; i < n; ++i) y[i] *= d[i];
        const double nx = norm(x), ny = norm(y);
        lam = ny / nx;
        for (int i = 0; i < n; ++i) x[i] = y[i] / ny;
    }
    *lmax = lam;
    const double shift = 1.05 * lam;
    for (int i = 0; i < n; ++i) x[i] = 1.0 + 0.5 * std::cos(0.61 * i + 7.0);
    double mu = 0.0;
    for (int it = 0; it < 80; ++it) {
        host_spmv(A, x, y);
        for (int i = 0; i < n; ++i) y[i] = shift * x[i] - d[i] * y[i];
        const double nx = norm(x), ny = norm(y);
        if (!(ny > 0.0) || !(nx > 0.0)) break;
        mu = ny / nx;
        for (int i = 0; i < n; ++i) x[i] = y[i] / ny;
    }
    *lmin = shift - mu;
}

// eta = spectral radius of the skew part of A in the metric of its diagonal:
// D^-1/2 (A - A^T)/2 D^-1/2 (real skew-symmetric: eigenvalues +- i eta) --
// Bendixson: the eigenvalues of D^-1 A have |Im| <= eta.  0 for a symmetric A.
inline double host_skew_radius(const HostCsr &A, const HostCsr &AT,
                               int iters = 30) {
    HostCsr K = host_add(0.5, A, -0.5, AT);
    double mx = 0.0, amax = 0.0;
    for (double v : K.vals) mx = std::max(mx, std::fabs(v));
    for (double v : A.vals) amax = std::max(amax, std::fabs(v));
    if (!(mx > 1e-12 * amax)) return 0.0;     // symmetric up to rounding
    const int n = A.nrows;
    std::vector<double> ds((size_t)n, 1.0), x((size_t)n), y, z;
    for (int i = 0; i < n; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if (A.colidx[k] == i && A.vals[k] > 0.0)
                ds[i] = 1.0 / std::sqrt(A.vals[k]);
    for (int i = 0; i < n; ++i)
        for (int k = K.rowptr[i]; k < K.rowptr[i + 1]; ++k)
            K.vals[k] *= ds[i] * ds[K.colidx[k]];
    for (int i = 0; i < n; ++i) x[i] = 1.0 + 0.5 * std::sin(0.53 * i + 3.0);
    double eta2 = 0.0;
    for (int it = 0; it < iters; ++it) {
        host_spmv(K, x, y);            // K^T K = -K^2: x <- -K (K x)
        host_spmv(K, y, z);
        double nx = 0.0, nz = 0.0;
        for (int i = 0; i < n; ++i) {
            nx += x[i] * x[i];
            nz += z[i] * z[i];
        }
        if (!(nz > 0.0)) return 0.0;
        eta2 = std::sqrt(nz / nx);
        const double sc = 1.0 / std::sqrt(nz);
        for (int i = 0; i < n; ++i) x[i] = -z[i] * sc;
    }
    return std::sqrt(eta2);
}

// [A, B] side by side (same number of rows)
inline HostCsr host_hstack(const HostCsr &A, const HostCsr &B) {
    HostCsr C;
    C.nrows = A.nrows;
    C.ncols = A.ncols + B.ncols;
    C.rowptr.assign((size_t)A.nrows + 1, 0);
    C.colidx.reserve(A.colidx.size() + B.colidx.size());
    C.vals.reserve(A.vals.size() + B.vals.size());
    for (int i = 0; i < A.nrows; ++i) {
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
            C.colidx.push_back(A.colidx[k]);
            C.vals.push_back(A.vals[k]);
        }
        for (int k = B.rowptr[i]; k < B.rowptr[i + 1]; ++k) {
            C.colidx.push_back(A.ncols + B.colidx[k]);
            C.vals.push_back(B.vals[k]);
        }
        C.rowptr[i + 1] = (int)C.colidx.size();
    }
    return C;
}

// this rank's rows of K = [[F, JT], [J, 0]]: velocity rows, then pressure rows
inline HostCsr host_k_slice(const HostCsr &F, const HostCsr &JT,
                            const HostCsr &J, int nv, int v0, int v1,
                                 int p0, int p1) {
    HostCsr S;
    S.nrows = (v1 - v0) + (p1 - p0);
    S.ncols = nv + J.nrows;
    S.rowptr.assign(1, 0);
    for (int r = v0; r < v1; ++r) {
        for (int k = F.rowptr[r]; k < F.rowptr[r + 1]; ++k) {
            S.colidx.push_back(F.colidx[k]);
            S.vals.push_back(F.vals[k]);
        }
        for (int k = JT.rowptr[r]; k < JT.rowptr[r + 1]; ++k) {
            S.colidx.push_back(nv + JT.colidx[k]);
            S.vals.push_back(JT.vals[k]);
        }
        S.rowptr.push_back((int)S.colidx.size());
    }
    for (int r = p0; r < p1; ++r) {
        for (int k = J.rowptr[r]; k < J.rowptr[r + 1]; ++k) {
            S.colidx.push_back(J.colidx[k]);
            S.vals.push_back(J.vals[k]);
        }
        S.rowptr.push_back((int)S.colidx.size());
    }
    return S;
}

}  // namespace dns
