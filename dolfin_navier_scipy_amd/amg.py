"""Algebraic coarsening for the multigrid Schur block.

`dns_saddle_set_schur_mg` takes a list of prolongation matrices; the
geometric ones (`fem.pressure_prolongations`) exist only for meshes this
repository refined itself.  Any other mesh with a pressure space beyond the
dense Schur inverse -- the reference's `karman2D-rotcyl_lvl3/lvl4`
(`tests/mesh/`, `problem_setups.py:773-987`) -- used to fall back to
`diag(J D^-1 J^T)^-1`.  This module builds the same list from the matrices
alone, by smoothed aggregation (Vanek, Mandel, Brezina 1996) on the graph of

    L = J D^-1 J^T ,     D = diag(F)

-- the pressure operator whose sparse form `S_0 = J G J^T` (G the polynomial
approximation of F^-1) the device cycle runs on; both are discrete Laplacians
of the pressure space scaled by dt-dependent weights, and L is available
before the set-up:

  1. strength of connection: `|l_ij| >= theta sqrt(l_ii l_jj)`;
  2. greedy aggregation (a root and its three most strongly coupled free
     neighbours -- the coarsening ratio of a red-refined mesh; leftovers join
     the neighbouring aggregate they are connected to most strongly);
  3. tentative prolongation = the aggregates' indicator functions (the
     near-null space of a Laplacian: constants);
  4. one damped-Jacobi smoothing step `P = (I - omega D_L^-1 L_f) P_tent`
     with the filtered matrix (weak entries lumped onto the diagonal) and
     `omega = 4 / (3 rho(D_L^-1 L_f))`;
  5. Galerkin coarse operator `P^T L P`, repeat until the level is small
     enough for the dense inverse.

Host code (NumPy / SciPy), a set-up cost like the geometric hierarchy's; the
cycle itself, the Galerkin products on the true `S_0` and the fused level
operators are the library's, unchanged.  No reference counterpart (the
reference factorises the saddle-point matrix).
"""
import numpy as np
import scipy.sparse as sps

__all__ = ['pressure_operator', 'aggregate', 'smoothed_prolongation',
           'contraction_estimate', 'algebraic_prolongations']


def pressure_operator(F, J, JT=None):
    """`L = J diag(F)^-1 J^T` (CSR, symmetrised)"""
    F = sps.csr_matrix(F)
    J = sps.csr_matrix(J)
    JT = J.T.tocsr() if JT is None else sps.csr_matrix(JT)
    d = F.diagonal()
    d = np.where(d != 0.0, d, 1.0)
    L = (J @ sps.diags(1.0/d) @ JT).tocsr()
    L = (0.5*(L + L.T)).tocsr()
    L.sum_duplicates()
    L.sort_indices()
    return L


def _strength(L, theta):
    """the strong off-diagonal connections of `L` as a symmetric CSR pattern
    with the coupling `|l_ij| / sqrt(l_ii l_jj)` as values"""
    L = sps.csr_matrix(L)
    d = np.abs(L.diagonal())
    d = np.where(d > 0.0, d, 1.0)
    C = sps.coo_matrix(L)
    off = C.row != C.col
    w = np.abs(C.data[off])/np.sqrt(d[C.row[off]]*d[C.col[off]])
    keep = w >= theta
    S = sps.csr_matrix((w[keep], (C.row[off][keep], C.col[off][keep])),
                       shape=L.shape)
    S = S.maximum(S.T).tocsr()
    S.sort_indices()
    return S


def aggregate(L, theta=0.08, cap=4):
    """aggregate number per node (`-1` never remains) and the number of
    aggregates; greedy, in the node order of `L` (which for an RCM-ordered
    pressure space sweeps the mesh front by front).  `cap`: largest aggregate
    the first pass forms -- a root and its `cap - 1` most strongly coupled
    free neighbours.  Four (the coarsening ratio of a red-refined mesh) gives
    V(2,2) contraction 0.25-0.29 on the pressure operators of the reference's
    meshes against 0.22 for the geometric hierarchy; whole neighbourhoods
    (ratio 7) give 0.36-0.42."""
    S = _strength(L, theta)
    n = S.shape[0]
    ip, ix, wv = S.indptr, S.indices, S.data
    agg = np.full(n, -1, dtype=np.int64)
    nagg = 0
    # pass 1: a node with enough free strong neighbours roots an aggregate
    for i in range(n):
        if agg[i] >= 0:
            continue
        nb = ix[ip[i]:ip[i+1]]
        if nb.size == 0:
            continue
        free = agg[nb] < 0
        if free.sum() < min(cap - 1, nb.size):
            continue
        order = np.argsort(-wv[ip[i]:ip[i+1]][free], kind='stable')[:cap - 1]
        agg[i] = nagg
        agg[nb[free][order]] = nagg
        nagg += 1
    # pass 2: the rest joins the aggregate (as it stood after pass 1) it is
    # coupled to most strongly
    frozen = agg.copy()
    for i in np.flatnonzero(agg < 0):
        nb = ix[ip[i]:ip[i+1]]
        if nb.size == 0:
            continue
        a = frozen[nb]
        ok = a >= 0
        if ok.any():
            cand, inv = np.unique(a[ok], return_inverse=True)
            tot = np.bincount(inv, weights=wv[ip[i]:ip[i+1]][ok])
            agg[i] = cand[np.argmax(tot)]
    # pass 3: what is left (isolated nodes, islands of weak couplings) forms
    # aggregates with its free strong neighbours, or on its own
    for i in np.flatnonzero(agg < 0):
        if agg[i] >= 0:
            continue
        agg[i] = nagg
        nb = ix[ip[i]:ip[i+1]]
        free = nb[agg[nb] < 0]
        agg[free] = nagg
        nagg += 1
    return agg, nagg


def smoothed_prolongation(L, agg, nagg, theta=0.08, omega_scale=4.0/3.0):
    """`(I - omega D^-1 L_f) P_tent`; `L_f`: `L` with its weak off-diagonal
    entries lumped onto the diagonal (row sums, hence constants, are kept)"""
    L = sps.csr_matrix(L)
    n = L.shape[0]
    Pt = sps.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nagg))
    S = _strength(L, theta)
    mask = (S != 0).astype(np.float64)
    off = L - sps.diags(L.diagonal())
    strong = off.multiply(mask).tocsr()
    weak = (off - strong).tocsr()
    dl = L.diagonal() + np.asarray(weak.sum(axis=1)).reshape(-1)
    dl = np.where(np.abs(dl) > 0.0, dl, 1.0)
    Lf = (strong + sps.diags(dl)).tocsr()
    DinvL = sps.diags(1.0/dl) @ Lf
    # spectral radius of D^-1 L_f: a few power iterations (an over-estimate
    # by Gershgorin bounds it from above)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n)
    lam = 1.0
    for _ in range(20):
        y = DinvL @ x
        lam = np.linalg.norm(y)/max(np.linalg.norm(x), 1e-300)
        x = y/max(np.linalg.norm(y), 1e-300)
    gersh = np.abs(DinvL).sum(axis=1).max()
    lam = min(max(lam, 1e-12)*1.05, float(gersh))
    P = (Pt - (omega_scale/lam)*(DinvL @ Pt)).tocsr()
    P.sum_duplicates()
    P.eliminate_zeros()
    P.sort_indices()
    return P


def contraction_estimate(L, prols, nu=2, its=12):
    """error reduction per V(nu, nu) cycle (damped Jacobi, Galerkin coarse
    operators, exact coarsest solve) of the hierarchy `prols` on `L`, by a
    few cycles on a random error -- the host-side figure a hierarchy is
    judged by before it is handed to the device"""
    ops = [sps.csr_matrix(L)]
    for P in prols:
        ops.append((P.T @ ops[-1] @ P).tocsr())
    dinv, om = [], []
    rng = np.random.default_rng(0)
    for o in ops[:-1]:
        d = o.diagonal()
        d = np.where(d != 0.0, 1.0/d, 1.0)
        x = rng.standard_normal(o.shape[0])
        lam = 1.0
        for _ in range(20):
            y = d*(o @ x)
            lam = np.linalg.norm(y)/max(np.linalg.norm(x), 1e-300)
            x = y/max(np.linalg.norm(y), 1e-300)
        dinv.append(d)
        om.append(4.0/(3.0*lam))
    coarse = np.linalg.pinv(ops[-1].toarray()) if prols else None

    def cycle(l, b):
        if l == len(ops) - 1:
            return coarse @ b
        x = np.zeros_like(b)
        for _ in range(nu):
            x = x + om[l]*dinv[l]*(b - ops[l] @ x)
        x = x + prols[l] @ cycle(l + 1, prols[l].T @ (b - ops[l] @ x))
        for _ in range(nu):
            x = x + om[l]*dinv[l]*(b - ops[l] @ x)
        return x
    if not prols:
        return 0.0
    e = rng.standard_normal(L.shape[0])
    fac = 1.0
    for _ in range(its):
        e2 = e - cycle(0, ops[0] @ e)
        fac = np.linalg.norm(e2)/max(np.linalg.norm(e), 1e-300)
        e = e2/max(np.linalg.norm(e2), 1e-300)
    return float(fac)


def _hierarchy(L, coarsest, theta, cap, max_levels, min_ratio):
    prols = []
    while L.shape[0] > coarsest and len(prols) < max_levels:
        agg, nagg = aggregate(L, theta, cap)
        if nagg*min_ratio > L.shape[0]:
            break
        P = smoothed_prolongation(L, agg, nagg, theta)
        prols.append(P)
        L = (P.T @ L @ P).tocsr()
        L.sum_duplicates()
        L.sort_indices()
    return prols


def algebraic_prolongations(F, J, JT=None, coarsest=1500, theta=0.08, cap=4,
                            max_levels=8, min_ratio=1.5, accept=0.5,
                            info=None):
    """the list `dns_saddle_set_schur_mg` takes (finest first), from `F` and
    `J` alone.  Coarsening stops at the first level with at most `coarsest`
    unknowns (it gets the dense inverse) or when a level no longer shrinks
    by `min_ratio`.  The hierarchy is judged by `contraction_estimate` on
    `L`; above `accept` two more parameter sets are tried and the best one is
    kept (an unlucky strength threshold can leave chains instead of
    aggregates: 0.99 instead of 0.3).  `info` (dict): sizes, parameters and
    the estimate of what was returned."""
    L = pressure_operator(F, J, JT)
    best = None
    for th, cp in ((theta, cap), (0.5*theta, cap), (theta, cap + 2)):
        prols = _hierarchy(L, coarsest, th, cp, max_levels, min_ratio)
        est = contraction_estimate(L, prols)
        if best is None or est < best[0]:
            best = (est, prols, th, cp)
        if est <= accept:
            break
    est, prols, th, cp = best
    if info is not None:
        info.update(levels=[L.shape[0]] + [P.shape[1] for P in prols],
                    theta=th, cap=cp, contraction_estimate=est,
                    prolongation_nnz_per_row=[P.nnz/float(P.shape[0])
                                              for P in prols])
    return prols
