"""Problem data for the hot path without FEniCS (input scaffolding).

`get_sysmats` returns what `dolfin_navier_scipy.problem_setups.get_sysmats`
(reference dnsps:34-220) hands to `solve_nse`: condensed CSR matrices
`M, A, J, JT, MP`, the right-hand sides with the Dirichlet contributions
merged in, and the index sets -- with the FE space object replaced by a
`TaylorHood` instance that provides the convection forms.
"""
import os

import numpy as np
import scipy.sparse as sps

from .mesh2d import (load_npz_mesh, refine_uniform, channel_cylinder_mesh,
                     rectangle_mesh)
from .taylor_hood import TaylorHood

__all__ = ['get_sysmats', 'condense_sysmatsbybcs', 'cylinder_mesh',
           'cylinder_mesh_hierarchy', 'pressure_prolongations', 'GOLDEN_DIR']

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))), 'tests', 'golden')


def condense_sysmatsbybcs(stms, dbcinds=None, dbcvals=None):
    """resolve Dirichlet BCs, restrict to the inner dofs (reference dts:475-573)
    """
    nv = stms['A'].shape[0]
    invinds = np.setdiff1d(np.arange(nv), dbcinds).astype(np.int32)
    auxu = np.zeros((nv, 1))
    auxu[dbcinds, 0] = dbcvals
    fvbc = -(stms['A'] @ auxu)[invinds, :]
    fpbc = -(stms['J'] @ auxu)
    stokesmatsc = dict(M=stms['M'][invinds, :][:, invinds].tocsr(),
                       A=stms['A'][invinds, :][:, invinds].tocsr(),
                       J=stms['J'][:, invinds].tocsr(),
                       JT=stms['JT'][invinds, :].tocsr(),
                       MP=stms['MP'])
    for mat in stokesmatsc.values():
        mat.sort_indices()
    return stokesmatsc, dict(fv=fvbc, fp=fpbc), invinds


def _snap_to_cylinder(xc=0.2, yc=0.2, radius=0.05, tol=0.2):
    def snap(mid, onbnd):
        d = mid - np.array([[xc, yc]])
        r = np.sqrt((d*d).sum(axis=1))
        sel = onbnd & (r < radius*(1+tol))
        mid = mid.copy()
        mid[sel] = np.array([[xc, yc]]) + d[sel]*(radius/r[sel])[:, None]
        return mid
    return snap


def cylinder_mesh(N=2, refine=0):
    """Schaefer-Turek meshes `cylinder_N` (fixtures converted from the
    reference's `tests/mesh/cylinder_N.xml`) plus `refine` red refinements;
    `N='toy'` is a generated coarse stand-in for CPU-sized tests"""
    if N == 'toy':
        mesh = channel_cylinder_mesh()
        snap = None
    else:
        mesh = load_npz_mesh(os.path.join(GOLDEN_DIR,
                                          'mesh_cylinder_{0}.npz'.format(N)))
        snap = _snap_to_cylinder()
    for _ in range(refine):
        mesh, _ = refine_uniform(mesh, snap=snap)
    return mesh


def cylinder_mesh_hierarchy(N=2, refine=0):
    """`[(mesh_0, None), (mesh_1, parents_1), ...]`: the level-`N` mesh and its
    `refine` red refinements with the parent vertices of every fine vertex
    (`refine_uniform`) -- what a pressure multigrid needs"""
    mesh = load_npz_mesh(os.path.join(GOLDEN_DIR,
                                      'mesh_cylinder_{0}.npz'.format(N)))
    snap = _snap_to_cylinder()
    out = [(mesh, None)]
    for _ in range(refine):
        mesh, parents = refine_uniform(mesh, snap=snap)
        out.append((mesh, parents))
    return out


def pressure_prolongations(spaces, parents):
    """P1 prolongations between the pressure spaces of nested meshes, FINEST
    FIRST: `P[l]` maps level `l+1` (coarser) to level `l`, in the pressure-dof
    numbering of the `TaylorHood` spaces (`spaces[0]` the finest; `parents[l]`
    the parent table of `spaces[l]`'s mesh)"""
    prols = []
    for l in range(len(spaces) - 1):
        fine, coarse = spaces[l], spaces[l + 1]
        par = parents[l]
        nvf = fine.mesh.nverts
        rows = np.repeat(fine.vert_pdof[np.arange(nvf)], 2)
        cols = coarse.vert_pdof[par].reshape(-1)
        P = sps.coo_matrix((np.full(2*nvf, 0.5), (rows, cols)),
                           shape=(fine.pdim, coarse.pdim)).tocsr()
        P.sum_duplicates()          # coarse vertices: 0.5 + 0.5 = 1
        P.sort_indices()
        prols.append(P)
    return prols


def get_sysmats(problem='cylinderwake', N=2, refine=0, Re=None, nu=None,
                charvel=1., mesh=None):
    """system matrices of the (Navier-)Stokes problem, BCs resolved

    Returns
    -------
    femp : dict with `V` (a `TaylorHood`), `invinds`, `dbcinds`, `dbcvals`,
        `nu`, `Re`, `charlen`, `ppin`
    stokesmatsc : dict `M, A, J, JT, MP` (condensed, CSR)
    rhsd : dict `fv (NV,1)`, `fp (NP,1)` -- forcing (zero) + BC contributions
        (as `mergerhs=True`, reference dnsps:215-218)
    """
    if problem == 'cylinderwake':
        mesh = cylinder_mesh(N=N, refine=refine) if mesh is None else mesh
        th = TaylorHood(mesh)
        charlen = 0.1                                   # dnsps:612
        toy = (N == 'toy')
        dbcinds, dbcvals, invinds = th.cylinderwake_bcs(
            obstacle_halfwidth=0.05 if toy else None)
        ppin = None                                     # dnsps:171-174
    elif problem == 'drivencavity':
        mesh = rectangle_mesh(N, N) if mesh is None else mesh
        th = TaylorHood(mesh)
        charlen = 1.0
        dbcinds, dbcvals, invinds = th.drivencavity_bcs()
        ppin = -1                                       # dnsps:178-182
    else:
        raise NotImplementedError(problem)
    if Re is not None:
        nu = charvel*charlen/Re                         # dnsps:138-141
    else:
        Re = charvel*charlen/nu
    stms = th.stokes_mats(nu=nu)
    if ppin == -1:
        stms['J'] = stms['J'][:-1, :].tocsr()
        stms['JT'] = stms['JT'][:, :-1].tocsr()
    stokesmatsc, rhsd, invinds = condense_sysmatsbybcs(stms, dbcinds, dbcvals)
    femp = dict(V=th, invinds=invinds, dbcinds=dbcinds, dbcvals=dbcvals,
                nu=nu, Re=Re, charlen=charlen, ppin=ppin, mesh=mesh)
    return femp, stokesmatsc, rhsd
