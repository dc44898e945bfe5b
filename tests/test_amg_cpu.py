"""Algebraic coarsening of the multigrid Schur block (`amg.py`), host side:
what can be judged without a GPU -- the hierarchy itself and its V-cycle
contraction on the pressure operator, next to the geometric hierarchy of a
mesh that has one."""
import numpy as np
import scipy.sparse as sps

from dolfin_navier_scipy_amd import amg
from dolfin_navier_scipy_amd.fem import (get_sysmats, TaylorHood,
                                         cylinder_mesh_hierarchy,
                                         pressure_prolongations)


def _wake(refine):
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, refine=refine,
                                 Re=100.)
    dt = 1./(512*2**refine)
    return (sm['M'] + .5*dt*sm['A']).tocsr(), sm['J']


def test_hierarchy_from_the_matrices_alone_is_close_to_the_geometric_one():
    F, J = _wake(1)                                   # NP = 4991
    info = {}
    prols = amg.algebraic_prolongations(F, J, coarsest=1500, info=info)
    assert len(prols) >= 1
    NP = J.shape[0]
    assert prols[0].shape[0] == NP
    for a, b in zip(prols[:-1], prols[1:]):
        assert a.shape[1] == b.shape[0]
    assert prols[-1].shape[1] <= 1500
    ratios = np.array(info['levels'][:-1], float)/np.array(info['levels'][1:])
    assert (ratios > 2.5).all() and (ratios < 8.).all(), info
    # every fine dof interpolates from a few aggregates
    assert 1.5 <= prols[0].nnz/float(NP) <= 8.
    L = amg.pressure_operator(F, J)
    assert abs(L - L.T).max() <= 1e-12*abs(L).max()
    est = amg.contraction_estimate(L, prols)
    assert abs(est - info['contraction_estimate']) <= 1e-12
    hier = cylinder_mesh_hierarchy(N=2, refine=1)
    spaces = [TaylorHood(m) for m, _ in hier][::-1]
    geo = pressure_prolongations(spaces, [p for _, p in hier][::-1])
    est_geo = amg.contraction_estimate(L, geo)
    print('V(2,2) contraction on J D^-1 J^T: algebraic', est, 'levels',
          info['levels'], '; geometric', est_geo)
    assert est < 0.4
    assert est < 2.*est_geo
    # deterministic: the same matrices give the same hierarchy
    again = amg.algebraic_prolongations(F, J, coarsest=1500)
    assert all(abs(a - b).max() == 0. for a, b in zip(prols, again))


def test_aggregates_cover_every_dof_and_respect_the_cap():
    F, J = _wake(0)
    L = amg.pressure_operator(F, J)
    agg, nagg = amg.aggregate(L, theta=0.08, cap=4)
    assert agg.min() == 0 and agg.max() == nagg - 1
    sizes = np.bincount(agg)
    assert sizes.min() >= 1
    # the first pass forms aggregates of at most four; joiners may add a few
    assert np.median(sizes) <= 5 and sizes.max() <= 12
    P = amg.smoothed_prolongation(L, agg, nagg)
    assert P.shape == (L.shape[0], nagg)
    # smoothing with the filtered matrix (weak entries lumped: row sums kept)
    # leaves the constants alone wherever the rows of L sum to zero
    ones = P @ np.ones(nagg)
    rowsum = np.abs(np.asarray(L.sum(axis=1)).reshape(-1))
    interior = rowsum <= 1e-10*np.abs(L.diagonal())
    assert interior.sum() > 0.5*L.shape[0]
    assert np.abs(ones[interior] - 1.).max() <= 1e-9


def test_a_system_below_the_dense_limit_gets_no_hierarchy():
    class _Sys(object):
        def set_schur_mg(self, prols):
            self.prols = prols
    from dolfin_navier_scipy_amd.saddle import choose_schur
    F, J = _wake(0)                                   # NP = 1289
    s = _Sys()
    assert choose_schur(s, F, J, schur='auto') == 'dense'
    assert s.schur_hierarchy is None
    assert choose_schur(s, F, J, schur='auto', dense_max=500) == 'mg'
    assert s.schur_hierarchy['kind'] == 'algebraic'
    assert s.prols[0].shape[0] == J.shape[0]
    # the caller's nested spaces win over the algebraic ones
    fake = [sps.identity(J.shape[0], format='csr')[:, :600]]
    assert choose_schur(s, F, J, schur='auto', prolongations=fake,
                        dense_max=500) == 'mg'
    assert s.schur_hierarchy['kind'] == 'geometric'
    assert choose_schur(s, F, J, schur='jacobi') == 'jacobi'
