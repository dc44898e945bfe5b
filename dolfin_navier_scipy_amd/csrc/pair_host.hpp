// Host-side builder of the pair format (layout: pair.hpp).  Plain C++, no HIP:
// this header, hostcsr.hpp and halo_host.hpp are what the sanitizer build of
// the host logic compiles (tests/host_sanitize.cpp, -fsanitize=address,
// undefined, CPU only).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "hostcsr.hpp"

namespace dns {

constexpr int kPairTileA = 1024;   // A entries per row block (2 x 8 KB of LDS)
constexpr int kPairTileB = 256;    // B entries per row block (2 x 2 KB)
constexpr int kPairRowsV = 128;    // pair rows per velocity row block
constexpr int kPairRowsP = 256;    // rows per pressure row block

struct HostPair {
    int nv = 0, np = 0, nvp = 0;              // nvp = nv / 2 pair rows
    std::vector<int> rpA, rpB, rpC;           // row pointers of the streams
    std::vector<int> colA, colB, colC;
    std::vector<double> vA, vB, vC;           // 4 / 2 / 2 values per entry
    std::vector<unsigned short> cA, cB, cC;   // 16-bit offsets
    std::vector<int> rowblocks;               // rows 0..nvp-1 pair rows, then
                                              // nvp + p pressure rows
    std::vector<int> base;                    // two bases per row block
    int aoff = 0, poff = 0;                   // row block: first global pair
                                              // row / pressure row
    int64_t nnz_equiv = 0;
};

// K: (nv + np) x (nv + np), rows / columns < nv = velocity; nv even.
// Row block of a partitioned system (`nvl >= 0`): K holds `nvl` velocity rows
// that start at the EVEN global row `v0`, then its pressure rows from global
// pressure row `p0` on; the columns stay global (nv = global velocity count).
inline bool host_pair_from_k(const HostCsr &K, int nv, HostPair &P,
                             const char **why = nullptr, int nvl = -1,
                             int v0 = 0, int p0 = 0) {
    const char *dummy = nullptr;
    if (!why) why = &dummy;
    const int n = K.nrows;
    const bool block = nvl >= 0;
    if (!block) nvl = nv;
    if (n <= 0 || nv <= 0 || (nv & 1) || (nvl & 1) || (v0 & 1) || nvl > n ||
        K.ncols < nv || (!block && K.ncols != n)) {
        *why = "odd (or no) number of velocity dofs";
        return false;
    }
    P = HostPair();
    P.nv = nv;
    P.np = n - nvl;
    P.nvp = nvl / 2;
    P.aoff = v0 / 2;
    P.poff = p0;
    P.nnz_equiv = K.nnz();
    const int nvp = P.nvp, np = P.np;
    P.rpA.assign(1, 0);
    P.rpB.assign(1, 0);
    P.rpC.assign(1, 0);
    for (int a = 0; a < nvp; ++a) {
        int k0 = K.rowptr[2 * a], e0 = K.rowptr[2 * a + 1];
        int k1 = K.rowptr[2 * a + 1], e1 = K.rowptr[2 * a + 2];
        // velocity columns: 2x2 blocks
        while ((k0 < e0 && K.colidx[k0] < nv) ||
               (k1 < e1 && K.colidx[k1] < nv)) {
            const int b0 = (k0 < e0 && K.colidx[k0] < nv) ? K.colidx[k0] / 2
                                                          : INT32_MAX;
            const int b1 = (k1 < e1 && K.colidx[k1] < nv) ? K.colidx[k1] / 2
                                                          : INT32_MAX;
            const int b = std::min(b0, b1);
            double blk[4] = {0.0, 0.0, 0.0, 0.0};
            while (k0 < e0 && K.colidx[k0] / 2 == b) {
                blk[K.colidx[k0] & 1] = K.vals[k0];
                ++k0;
            }
            while (k1 < e1 && K.colidx[k1] / 2 == b) {
                blk[2 + (K.colidx[k1] & 1)] = K.vals[k1];
                ++k1;
            }
            P.colA.push_back(b);
            P.vA.insert(P.vA.end(), blk, blk + 4);
        }
        // pressure columns: union of the two patterns
        while (k0 < e0 || k1 < e1) {
            const int c0 = k0 < e0 ? K.colidx[k0] : INT32_MAX;
            const int c1 = k1 < e1 ? K.colidx[k1] : INT32_MAX;
            const int c = std::min(c0, c1);
            P.colB.push_back(c - nv);
            P.vB.push_back(c0 == c ? K.vals[k0++] : 0.0);
            P.vB.push_back(c1 == c ? K.vals[k1++] : 0.0);
        }
        if ((int64_t)P.colA.size() - P.rpA.back() > kPairTileA ||
            (int64_t)P.colB.size() - P.rpB.back() > kPairTileB) {
            *why = "a velocity row pair too long for one tile";
            return false;
        }
        P.rpA.push_back((int)P.colA.size());
        P.rpB.push_back((int)P.colB.size());
    }
    for (int r = nvl; r < n; ++r) {
        int k = K.rowptr[r];
        const int e = K.rowptr[r + 1];
        while (k < e) {
            const int c = K.colidx[k];
            if (c >= nv) {
                *why = "an entry in the pressure-pressure block";
                return false;
            }
            double vx = 0.0, vy = 0.0;
            if (c & 1) {
                vy = K.vals[k++];
            } else {
                vx = K.vals[k++];
                if (k < e && K.colidx[k] == c + 1) vy = K.vals[k++];
            }
            P.colC.push_back(c / 2);
            P.vC.push_back(vx);
            P.vC.push_back(vy);
        }
        if ((int64_t)P.colC.size() - P.rpC.back() > kPairTileA) {
            *why = "a pressure row too long for one tile";
            return false;
        }
        P.rpC.push_back((int)P.colC.size());
    }
    // row blocks (velocity pair rows, then pressure rows; never mixed) and
    // their bases: V blocks (base A: velocity pairs, base B: pressure), P
    // blocks (base C: velocity pairs, unused)
    P.rowblocks.assign(1, 0);
    int start = 0;
    while (start < nvp) {
        int end = start;
        int64_t accA = 0, accB = 0;
        while (end < nvp && end - start < kPairRowsV) {
            const int64_t la = P.rpA[end + 1] - P.rpA[end];
            const int64_t lb = P.rpB[end + 1] - P.rpB[end];
            if (accA + la > kPairTileA || accB + lb > kPairTileB) break;
            accA += la;
            accB += lb;
            ++end;
        }
        P.rowblocks.push_back(end);
        start = end;
    }
    start = 0;
    while (start < np) {
        int end = start;
        int64_t acc = 0;
        while (end < np && end - start < kPairRowsP) {
            const int64_t lc = P.rpC[end + 1] - P.rpC[end];
            if (acc + lc > kPairTileA) break;
            acc += lc;
            ++end;
        }
        P.rowblocks.push_back(nvp + end);
        start = end;
    }
    const int nb = (int)P.rowblocks.size() - 1;
    P.cA.resize(P.colA.size());
    P.cB.resize(P.colB.size());
    P.cC.resize(P.colC.size());
    P.base.assign((size_t)2 * nb, 0);
    auto encode = [&](const std::vector<int> &col, int k0, int k1,
                      std::vector<unsigned short> &out, int *base) -> bool {
        int lo = INT32_MAX;
        for (int k = k0; k < k1; ++k) lo = std::min(lo, col[k]);
        if (lo == INT32_MAX) lo = 0;
        *base = lo;
        for (int k = k0; k < k1; ++k) {
            const int off = col[k] - lo;
            if (off > 0xffff) return false;
            out[k] = (unsigned short)off;
        }
        return true;
    };
    for (int b = 0; b < nb; ++b) {
        const int r0 = P.rowblocks[b], r1 = P.rowblocks[b + 1];
        bool ok;
        if (r0 < nvp) {
            ok = encode(P.colA, P.rpA[r0], P.rpA[r1], P.cA, &P.base[2 * b]) &&
                 encode(P.colB, P.rpB[r0], P.rpB[r1], P.cB, &P.base[2 * b + 1]);
        } else {
            ok = encode(P.colC, P.rpC[r0 - nvp], P.rpC[r1 - nvp], P.cC,
                        &P.base[2 * b]);
        }
        if (!ok) {
            *why = "the columns of a row block span more than 65536 entries";
            return false;
        }
    }
    return true;
}

}  // namespace dns
