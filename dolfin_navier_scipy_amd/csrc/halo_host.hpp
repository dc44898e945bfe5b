// Host-side arithmetic of the row partition: block starts, halo index lists,
// row slices.  Plain C++, no HIP (see pair_host.hpp: the sanitizer build of the
// host logic compiles these headers as they are).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "hostcsr.hpp"

namespace dns {

// contiguous block partition of `n` rows over `nranks`: starts[r] .. starts[r+1]
inline std::vector<int> partition_starts(int n, int nranks) {
    std::vector<int> st((size_t)nranks + 1);
    // (even chunks: the two velocity dofs of a node stay on one rank -- the
    // pair format of a rank's row block needs that)
    const int chunk = ((n + nranks - 1) / nranks + 1) & ~1;
    for (int r = 0; r <= nranks; ++r)
        st[r] = (int)std::min<int64_t>((int64_t)n, (int64_t)r * chunk);
    return st;
}

// sorted distinct columns < ncols_part that the rows [row0, row1) of `a`
// reference in every rank's column range (need[q]); own range left empty
inline void halo_need(const dns_csr *a, int row0, int row1, int nranks,
                      int rank, const int *col_starts, int ncols_part,
                      std::vector<std::vector<int>> &need) {
    need.assign((size_t)nranks, std::vector<int>());
    std::vector<unsigned char> mark((size_t)std::max(1, ncols_part), 0);
    for (int r = row0; r < row1; ++r)
        for (int k = a->rowptr[r]; k < a->rowptr[r + 1]; ++k) {
            const int c = a->colidx[k];
            if (c < ncols_part) mark[c] = 1;
        }
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        for (int c = col_starts[q]; c < col_starts[q + 1]; ++c)
            if (mark[c]) need[q].push_back(c);
    }
}

}  // namespace dns

// rows [r0, r1) of a host CSR matrix as a CSR of its own (rebased row pointers)
inline dns::HostCsr host_row_slice(const dns::HostCsr &A, int r0, int r1) {
    dns::HostCsr S;
    S.nrows = r1 - r0;
    S.ncols = A.ncols;
    S.rowptr.resize((size_t)S.nrows + 1);
    const int k0 = A.rowptr[r0];
    for (int i = 0; i <= S.nrows; ++i) S.rowptr[i] = A.rowptr[r0 + i] - k0;
    S.colidx.assign(A.colidx.begin() + k0, A.colidx.begin() + A.rowptr[r1]);
    S.vals.assign(A.vals.begin() + k0, A.vals.begin() + A.rowptr[r1]);
    return S;
}
