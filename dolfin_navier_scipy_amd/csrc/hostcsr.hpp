// Host-side CSR algebra used only at preconditioner SET-UP time (building the
// explicit polynomial approximate inverse); nothing here runs per solve.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/dns_amd.h"

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

// C = A * B  (Gustavson, dense accumulator per row, sorted columns)
inline HostCsr host_spgemm(const HostCsr &A, const HostCsr &B) {
    HostCsr C;
    C.nrows = A.nrows;
    C.ncols = B.ncols;
    C.rowptr.assign((size_t)A.nrows + 1, 0);
    std::vector<double> acc((size_t)B.ncols, 0.0);
    std::vector<int> mark((size_t)B.ncols, -1), cols;
    for (int i = 0; i < A.nrows; ++i) {
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
            C.colidx.push_back(j);
            C.vals.push_back(acc[j]);
        }
        C.rowptr[i + 1] = (int)C.colidx.size();
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

}  // namespace dns
