// CPU-only sanitizer target for the HOST logic of the library (no HIP, no GPU):
// hostcsr.hpp (SpGEMM, polynomial rows, transposes, slices), pair_host.hpp
// (pair-format builder) and halo_host.hpp (partition, halo index lists) are
// compiled as they are with
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover
// and driven over a small saddle-point system with the structure of the
// package's matrices (2x2 velocity node blocks, pressure rows on node
// patches), whole and in row blocks of 1..4 ranks, with the edge cases the
// set-up meets (empty blocks, odd sizes, rows without entries).  Every result
// is checked against a dense / scalar restatement, so a wrong index that stays
// inside its allocation is caught as well.  tests/test_host_sanitized.py
// builds and runs it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../dolfin_navier_scipy_amd/csrc/halo_host.hpp"
#include "../dolfin_navier_scipy_amd/csrc/hostcsr.hpp"
#include "../dolfin_navier_scipy_amd/csrc/pair_host.hpp"

using dns::HostCsr;

static int g_fail = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__,   \
                    __LINE__);                                               \
            ++g_fail;                                                        \
        }                                                                    \
    } while (0)

static HostCsr from_map(int nr, int nc,
                        const std::vector<std::map<int, double>> &rows) {
    HostCsr A;
    A.nrows = nr;
    A.ncols = nc;
    A.rowptr.assign(1, 0);
    for (int i = 0; i < nr; ++i) {
        for (const auto &kv : rows[i]) {
            A.colidx.push_back(kv.first);
            A.vals.push_back(kv.second);
        }
        A.rowptr.push_back((int)A.colidx.size());
    }
    return A;
}

static std::vector<double> dense(const HostCsr &A) {
    std::vector<double> d((size_t)A.nrows * A.ncols, 0.0);
    for (int i = 0; i < A.nrows; ++i)
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            d[(size_t)i * A.ncols + A.colidx[k]] += A.vals[k];
    return d;
}

static double lcg(unsigned &s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 65536.0 - 0.5;
}

// nodes on an nx x ny grid, two interleaved velocity dofs per node; F couples
// a node with its 4 neighbours through dense 2x2 blocks (diagonally dominant),
// J has one pressure row per grid cell over the 8 dofs of its corners
static void build_system(int nx, int ny, HostCsr &F, HostCsr &J) {
    const int nn = nx * ny, nv = 2 * nn, np = (nx - 1) * (ny - 1);
    unsigned seed = 12345u;
    std::vector<std::map<int, double>> fr((size_t)nv), jr((size_t)np);
    auto node = [&](int i, int j) { return j * nx + i; };
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            const int a = node(i, j);
            const int nb[5][2] = {{i, j}, {i - 1, j}, {i + 1, j}, {i, j - 1},
                                  {i, j + 1}};
            for (int q = 0; q < 5; ++q) {
                const int bi = nb[q][0], bj = nb[q][1];
                if (bi < 0 || bi >= nx || bj < 0 || bj >= ny) continue;
                const int b = node(bi, bj);
                for (int r = 0; r < 2; ++r)
                    for (int c = 0; c < 2; ++c) {
                        double v = 0.2 * lcg(seed);
                        if (q == 0 && r == c) v += 4.0;
                        fr[(size_t)2 * a + r][2 * b + c] = v;
                    }
            }
        }
    for (int j = 0; j + 1 < ny; ++j)
        for (int i = 0; i + 1 < nx; ++i) {
            const int p = j * (nx - 1) + i;
            const int cs[4] = {node(i, j), node(i + 1, j), node(i, j + 1),
                               node(i + 1, j + 1)};
            for (int q = 0; q < 4; ++q) {
                jr[(size_t)p][2 * cs[q]] = lcg(seed);
                // (some rows touch only the x dof of a corner: half-filled
                // (vx, vy) entries of the pair format)
                if ((p + q) % 3) jr[(size_t)p][2 * cs[q] + 1] = lcg(seed);
            }
        }
    F = from_map(nv, nv, fr);
    J = from_map(np, nv, jr);
}

// y = K x from the pair format, scalar restatement of k_spmv_pair16x's
// arithmetic incl. the 16-bit column decoding
static std::vector<double> pair_apply(const dns::HostPair &P,
                                      const std::vector<double> &x) {
    const int nblocks = (int)P.rowblocks.size() - 1;
    std::vector<double> y((size_t)2 * P.nvp + P.np, 0.0);
    auto col = [&](unsigned short c, int b, int which) {
        return P.base[(size_t)2 * b + which] + (int)c;
    };
    for (int b = 0; b < nblocks; ++b)
        for (int r = P.rowblocks[b]; r < P.rowblocks[b + 1]; ++r) {
            if (r < P.nvp) {
                double y0 = 0.0, y1 = 0.0;
                for (int k = P.rpA[r]; k < P.rpA[r + 1]; ++k) {
                    const int c = col(P.cA[k], b, 0);
                    CHECK(c == P.colA[k]);
                    const double x0 = x[(size_t)2 * c], x1 = x[(size_t)2 * c + 1];
                    y0 += P.vA[(size_t)4 * k] * x0 + P.vA[(size_t)4 * k + 1] * x1;
                    y1 += P.vA[(size_t)4 * k + 2] * x0 + P.vA[(size_t)4 * k + 3] * x1;
                }
                for (int k = P.rpB[r]; k < P.rpB[r + 1]; ++k) {
                    const int c = col(P.cB[k], b, 1);
                    CHECK(c == P.colB[k]);
                    y0 += P.vB[(size_t)2 * k] * x[(size_t)P.nv + c];
                    y1 += P.vB[(size_t)2 * k + 1] * x[(size_t)P.nv + c];
                }
                y[(size_t)2 * r] = y0;
                y[(size_t)2 * r + 1] = y1;
            } else {
                const int p = r - P.nvp;
                double yp = 0.0;
                for (int k = P.rpC[p]; k < P.rpC[p + 1]; ++k) {
                    const int c = col(P.cC[k], b, 0);
                    CHECK(c == P.colC[k]);
                    yp += P.vC[(size_t)2 * k] * x[(size_t)2 * c] +
                          P.vC[(size_t)2 * k + 1] * x[(size_t)2 * c + 1];
                }
                y[(size_t)2 * P.nvp + p] = yp;
            }
        }
    return y;
}

static void test_products_and_slices(const HostCsr &F, const HostCsr &J) {
    const int nv = F.nrows, np = J.nrows;
    const HostCsr JT = dns::host_transpose(J);
    CHECK(JT.nrows == nv && JT.ncols == np && JT.nnz() == J.nnz());
    const HostCsr JTT = dns::host_transpose(JT);
    CHECK(JTT.rowptr == J.rowptr && JTT.colidx == J.colidx && JTT.vals == J.vals);
    // S = J JT against the dense product
    const HostCsr S = dns::host_spgemm(J, JT);
    const std::vector<double> jd = dense(J), sd = dense(S);
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) {
            double ref = 0.0;
            for (int k = 0; k < nv; ++k)
                ref += jd[(size_t)i * nv + k] * jd[(size_t)j * nv + k];
            CHECK(std::fabs(ref - sd[(size_t)i * np + j]) <= 1e-13);
        }
    // [F, JT] and K slices of every block partition
    const HostCsr FJ = dns::host_hstack(F, JT);
    CHECK(FJ.ncols == nv + np && FJ.nnz() == F.nnz() + JT.nnz());
    const HostCsr K = dns::host_k_slice(F, JT, J, nv, 0, nv, 0, np);
    CHECK(K.nrows == nv + np && K.nnz() == F.nnz() + 2 * J.nnz());
    for (int P = 1; P <= 4; ++P) {
        const std::vector<int> sv = dns::partition_starts(nv, P),
                               sp = dns::partition_starts(np, P);
        CHECK(sv[0] == 0 && sv[P] == nv && sp[0] == 0 && sp[P] == np);
        int64_t nnz = 0;
        for (int r = 0; r < P; ++r) {
            CHECK((sv[r] & 1) == 0 && sv[r] <= sv[r + 1]);
            const HostCsr Kl = dns::host_k_slice(F, JT, J, nv, sv[r], sv[r + 1],
                                                 sp[r], sp[r + 1]);
            nnz += Kl.nnz();
            const HostCsr Fl = host_row_slice(F, sv[r], sv[r + 1]);
            CHECK(Fl.nrows == sv[r + 1] - sv[r]);
            for (int i = 0; i < Fl.nrows; ++i)
                CHECK(Fl.rowptr[i + 1] - Fl.rowptr[i] ==
                      F.rowptr[sv[r] + i + 1] - F.rowptr[sv[r] + i]);
        }
        CHECK(nnz == K.nnz());
    }
}

static void test_polynomial_rows(const HostCsr &F) {
    // D^-1 F, a degree-3 polynomial; the rows of a subset equal the rows of
    // the full build bit for bit (the partitioned set-up rests on that)
    const int nv = F.nrows;
    std::vector<double> dv((size_t)nv);
    HostCsr DF = F;
    for (int i = 0; i < nv; ++i) {
        double d = 1.0;
        for (int k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k)
            if (F.colidx[k] == i) d = F.vals[k];
        dv[i] = 1.0 / d;
        for (int k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) DF.vals[k] *= dv[i];
    }
    const std::vector<double> c1 = {0.1, 0.05, 0.02}, c2 = {0.9, 0.8, 0.7};
    for (double tol : {0.0, 1e-3}) {
        const HostCsr G = dns::host_cheb_poly(DF, dv, 1.0, c1, c2, tol);
        CHECK(G.nrows == nv && G.rowptr[nv] == (int)G.colidx.size());
        std::vector<int> rows;
        for (int i = 0; i < nv; i += 3) rows.push_back(i);
        rows.push_back(nv - 1);
        std::sort(rows.begin(), rows.end());
        rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
        const HostCsr Gs = dns::host_cheb_poly(DF, dv, 1.0, c1, c2, tol, &rows);
        size_t q = 0;
        for (int i = 0; i < nv; ++i) {
            const bool in = q < rows.size() && rows[q] == i;
            if (in) ++q;
            const int len = Gs.rowptr[i + 1] - Gs.rowptr[i];
            CHECK(len == (in ? G.rowptr[i + 1] - G.rowptr[i] : 0));
            for (int k = 0; in && k < len; ++k) {
                CHECK(Gs.colidx[Gs.rowptr[i] + k] == G.colidx[G.rowptr[i] + k]);
                CHECK(Gs.vals[Gs.rowptr[i] + k] == G.vals[G.rowptr[i] + k]);
            }
        }
        // empty row list: an empty matrix, not a crash
        const std::vector<int> none;
        const HostCsr G0 = dns::host_cheb_poly(DF, dv, 1.0, c1, c2, tol, &none);
        CHECK(G0.nnz() == 0 && (int)G0.rowptr.size() == nv + 1);
    }
}

// Rank-local construction: a block of rows embedded in the global shape, ghost
// rows merged in ring by ring, the polynomial rows of the block formed on the
// renumbered index set they reach -- equal, bit for bit, to the rows of the
// polynomial of the whole matrix (dist_solve.inc, build_explicit_part;
// rank_local.inc, extend_rows_for_setup)
static void test_rows_on_their_reach(const HostCsr &F) {
    const int nv = F.nrows;
    std::vector<double> dv((size_t)nv);
    for (int i = 0; i < nv; ++i) {
        double d = 1.0;
        for (int k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k)
            if (F.colidx[k] == i) d = F.vals[k];
        dv[i] = 1.0 / d;
    }
    const std::vector<double> c1 = {0.1, 0.05, 0.02}, c2 = {0.9, 0.8, 0.7};
    const int degree = (int)c1.size() + 1;
    HostCsr DFw = F;
    dns::host_scale_rows(dv, DFw);
    const HostCsr Gw = dns::host_cheb_poly(DFw, dv, 1.0, c1, c2, 1e-3);
    for (int nranks : {1, 2, 3}) {
        const std::vector<int> st = dns::partition_starts(nv, nranks);
        for (int me = 0; me < nranks; ++me) {
            const int v0 = st[me], v1 = st[me + 1];
            // the own rows in the global shape
            const HostCsr blk = host_row_slice(F, v0, v1);
            const dns_csr bv = blk.view();
            HostCsr Fx = dns::host_embed_rows(&bv, v0, nv);
            CHECK(Fx.nrows == nv && Fx.nnz() == blk.nnz());
            for (int i = 0; i < nv; ++i)
                CHECK(Fx.rowptr[i + 1] - Fx.rowptr[i] ==
                      ((i >= v0 && i < v1) ? F.rowptr[i + 1] - F.rowptr[i] : 0));
            // ghost rows ring by ring (here: cut out of the whole matrix)
            std::vector<char> in((size_t)nv, 0);
            std::vector<int> fresh, rows;
            for (int i = v0; i < v1; ++i) {
                in[i] = 1;
                fresh.push_back(i);
                rows.push_back(i);
            }
            for (int s = 0; s < degree; ++s) {
                std::vector<int> want;
                for (int i : fresh)
                    if (i < v0 || i >= v1) want.push_back(i);
                HostCsr got;
                got.nrows = nv;
                got.ncols = nv;
                got.rowptr.assign((size_t)nv + 1, 0);
                for (int i : want) got.rowptr[(size_t)i + 1] = F.rowptr[i + 1] - F.rowptr[i];
                for (int i = 0; i < nv; ++i) got.rowptr[(size_t)i + 1] += got.rowptr[i];
                for (int i : want)
                    for (int k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                        got.colidx.push_back(F.colidx[k]);
                        got.vals.push_back(F.vals[k]);
                    }
                if (!want.empty()) Fx = dns::host_merge_rows(Fx, got);
                if (s + 1 == degree) break;
                std::vector<int> next;
                for (int i : fresh)
                    for (int k = Fx.rowptr[i]; k < Fx.rowptr[i + 1]; ++k)
                        if (!in[Fx.colidx[k]]) {
                            in[Fx.colidx[k]] = 1;
                            next.push_back(Fx.colidx[k]);
                        }
                std::sort(next.begin(), next.end());
                fresh.swap(next);
            }
            // the index set the rows reach, renumbered
            std::vector<int> R, loc_of((size_t)nv, -1);
            for (int i = 0; i < nv; ++i)
                if (in[i]) R.push_back(i);
            for (size_t q = 0; q < R.size(); ++q) loc_of[R[q]] = (int)q;
            HostCsr DFx = Fx;
            dns::host_scale_rows(dv, DFx);
            const HostCsr DFc = dns::host_compact_rows(DFx, R, &loc_of);
            CHECK(DFc.nrows == (int)R.size() && DFc.ncols == (int)R.size());
            std::vector<double> dvc(R.size());
            for (size_t q = 0; q < R.size(); ++q) dvc[q] = dv[R[q]];
            std::vector<int> rows_c;
            for (int i : rows) rows_c.push_back(loc_of[i]);
            const HostCsr Gc = dns::host_cheb_poly(DFc, dvc, 1.0, c1, c2, 1e-3,
                                                   &rows_c);
            for (int i = v0; i < v1; ++i) {
                const int q = loc_of[i];
                const int len = Gc.rowptr[q + 1] - Gc.rowptr[q];
                CHECK(len == Gw.rowptr[i + 1] - Gw.rowptr[i]);
                for (int k = 0; k < len && k < Gw.rowptr[i + 1] - Gw.rowptr[i]; ++k) {
                    CHECK(R[Gc.colidx[Gc.rowptr[q] + k]] ==
                          Gw.colidx[Gw.rowptr[i] + k]);
                    CHECK(Gc.vals[Gc.rowptr[q] + k] == Gw.vals[Gw.rowptr[i] + k]);
                }
            }
            // rows kept global when no map is given
            const HostCsr Fr = dns::host_compact_rows(Fx, R, nullptr);
            CHECK(Fr.ncols == nv && Fr.nnz() == Fx.nnz());
        }
    }
}

static void test_pair_format(const HostCsr &F, const HostCsr &J) {
    const int nv = F.nrows, np = J.nrows, n = nv + np;
    const HostCsr JT = dns::host_transpose(J);
    const HostCsr K = dns::host_k_slice(F, JT, J, nv, 0, nv, 0, np);
    std::vector<double> x((size_t)n), yref((size_t)n);
    unsigned seed = 99u;
    for (double &v : x) v = lcg(seed);
    dns::host_spmv(K, x, yref);
    dns::HostPair P;
    const char *why = nullptr;
    CHECK(dns::host_pair_from_k(K, nv, P, &why));
    const std::vector<double> y = pair_apply(P, x);
    for (int i = 0; i < n; ++i) CHECK(std::fabs(y[i] - yref[i]) <= 1e-13);
    // row blocks of 1..4 ranks: local rows, global columns
    for (int R = 2; R <= 4; ++R) {
        const std::vector<int> sv = dns::partition_starts(nv, R),
                               sp = dns::partition_starts(np, R);
        for (int r = 0; r < R; ++r) {
            const HostCsr Kl = dns::host_k_slice(F, JT, J, nv, sv[r], sv[r + 1],
                                                 sp[r], sp[r + 1]);
            dns::HostPair Pl;
            if (Kl.nrows == 0) continue;
            CHECK(dns::host_pair_from_k(Kl, nv, Pl, &why, sv[r + 1] - sv[r],
                                        sv[r], sp[r]));
            const std::vector<double> yl = pair_apply(Pl, x);
            for (int i = 0; i < sv[r + 1] - sv[r]; ++i)
                CHECK(std::fabs(yl[i] - yref[sv[r] + i]) <= 1e-13);
            for (int i = 0; i < sp[r + 1] - sp[r]; ++i)
                CHECK(std::fabs(yl[(size_t)(sv[r + 1] - sv[r]) + i] -
                                yref[(size_t)nv + sp[r] + i]) <= 1e-13);
        }
    }
    // refusals instead of out-of-range writes: odd velocity count, an entry in
    // the pressure-pressure block
    dns::HostPair Q;
    HostCsr Kodd = dns::host_k_slice(F, JT, J, nv, 0, nv, 0, np);
    CHECK(!dns::host_pair_from_k(Kodd, nv - 1, Q, &why));
    HostCsr Kpp = K;
    Kpp.colidx.back() = n - 1;          // last pressure row, last column
    CHECK(!dns::host_pair_from_k(Kpp, nv, Q, &why));
}

static void test_halo_lists(const HostCsr &F, const HostCsr &J) {
    const int nv = F.nrows, np = J.nrows;
    const dns_csr fv = F.view(), jv = J.view();
    for (int P = 1; P <= 4; ++P) {
        const std::vector<int> sv = dns::partition_starts(nv, P),
                               sp = dns::partition_starts(np, P);
        std::vector<std::vector<std::vector<int>>> need((size_t)P);
        for (int r = 0; r < P; ++r) {
            dns::halo_need(&fv, sv[r], sv[r + 1], P, r, sv.data(), nv, need[r]);
            CHECK(need[r][r].empty());
            std::vector<std::vector<int>> nj;
            dns::halo_need(&jv, sp[r], sp[r + 1], P, r, sv.data(), nv, nj);
            CHECK(nj[r].empty());
            // every off-rank column of the rank's rows is in exactly one list
            for (int i = sv[r]; i < sv[r + 1]; ++i)
                for (int k = F.rowptr[i]; k < F.rowptr[i + 1]; ++k) {
                    const int c = F.colidx[k];
                    if (c >= sv[r] && c < sv[r + 1]) continue;
                    int hits = 0;
                    for (int q = 0; q < P; ++q)
                        hits += (int)std::count(need[r][q].begin(),
                                                need[r][q].end(), c);
                    CHECK(hits == 1);
                }
            for (int q = 0; q < P; ++q)
                for (size_t k = 0; k < need[r][q].size(); ++k) {
                    CHECK(need[r][q][k] >= sv[q] && need[r][q][k] < sv[q + 1]);
                    CHECK(k == 0 || need[r][q][k - 1] < need[r][q][k]);
                }
        }
    }
    // more ranks than rows, no rows at all
    const std::vector<int> tiny = dns::partition_starts(3, 8);
    CHECK(tiny[0] == 0 && tiny[8] == 3);
    for (int r = 0; r < 8; ++r) CHECK(tiny[r] <= tiny[r + 1]);
    const std::vector<int> zero = dns::partition_starts(0, 4);
    for (int r = 0; r <= 4; ++r) CHECK(zero[r] == 0);
}

static void test_misc(const HostCsr &F) {
    HostCsr A = F;
    dns::host_drop_small(A, 0.03);
    CHECK(A.nnz() <= F.nnz() && A.rowptr.back() == (int)A.colidx.size());
    const HostCsr T = dns::host_transpose(F);
    const HostCsr sum = dns::host_add(1.0, F, -1.0, T);
    CHECK(sum.nrows == F.nrows);
    double lmin = 0.0, lmax = 0.0;
    dns::host_jacobi_bounds(F, &lmin, &lmax);
    CHECK(lmax >= lmin && lmax > 0.0);
    const double eta = dns::host_skew_radius(F, T);
    CHECK(eta >= 0.0 && std::isfinite(eta));
    std::vector<double> d((size_t)F.nrows, 2.0);
    const HostCsr D = dns::host_diag(d);
    const HostCsr DF = dns::host_spgemm(D, F);
    CHECK(DF.nnz() == F.nnz());
    for (int64_t k = 0; k < F.nnz(); ++k)
        CHECK(std::fabs(DF.vals[k] - 2.0 * F.vals[k]) <= 1e-15);
    // a matrix with empty rows and an empty matrix go through every routine
    HostCsr E;
    E.nrows = 5;
    E.ncols = 5;
    E.rowptr.assign(6, 0);
    const HostCsr ET = dns::host_transpose(E);
    CHECK(ET.nnz() == 0);
    const HostCsr EE = dns::host_spgemm(E, E);
    CHECK(EE.nnz() == 0 && (int)EE.rowptr.size() == 6);
    const HostCsr Es = host_row_slice(E, 2, 2);
    CHECK(Es.nrows == 0 && Es.nnz() == 0);
}

int main() {
    for (const auto &dims : {std::pair<int, int>{7, 5}, {12, 9}, {3, 2}}) {
        HostCsr F, J;
        build_system(dims.first, dims.second, F, J);
        test_products_and_slices(F, J);
        test_polynomial_rows(F);
        test_rows_on_their_reach(F);
        test_pair_format(F, J);
        test_halo_lists(F, J);
        test_misc(F);
    }
    if (g_fail) {
        fprintf(stderr, "%d checks failed\n", g_fail);
        return 1;
    }
    printf("host logic: all checks passed under the sanitizers\n");
    return 0;
}
