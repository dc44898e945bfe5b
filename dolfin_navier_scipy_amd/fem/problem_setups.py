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
           'cylinder_mesh_hierarchy', 'pressure_prolongations', 'DATA_DIR',
           'gen_bccont_fems', 'classify_boundary']

# mesh fixtures (converted from the reference's `tests/mesh/*`, see
# scripts/convert_meshes.py) and the geometry descriptions that go with them
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


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
        mesh = load_npz_mesh(os.path.join(DATA_DIR,
                                          'mesh_cylinder_{0}.npz'.format(N)))
        snap = _snap_to_cylinder()
    for _ in range(refine):
        mesh, _ = refine_uniform(mesh, snap=snap)
    return mesh


def cylinder_mesh_hierarchy(N=2, refine=0):
    """`[(mesh_0, None), (mesh_1, parents_1), ...]`: the level-`N` mesh and its
    `refine` red refinements with the parent vertices of every fine vertex
    (`refine_uniform`) -- what a pressure multigrid needs"""
    mesh = load_npz_mesh(os.path.join(DATA_DIR,
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


def classify_boundary(th, geo, tol=1e-8):
    """boundary edges by physical part, from the GEOMETRY in the reference's
    `*_geo_cntrlbc*.json` (the `*_facet_region.xml.gz` mesh functions carry
    dolfin's own edge numbering and cannot be used without dolfin): the inflow
    segment `xone -> xtwo`, circles (control boundaries / moving walls) by
    centre and radius, the outflow as the side opposite to the inflow, walls =
    the rest of the bounding box.  Returns `dict(part -> edge ids)`; control
    circles are `('control', k)`."""
    eids, mid = th.boundary_edges()
    left = np.ones(eids.size, dtype=bool)
    parts = {}

    def take(name, mask):
        mask = mask & left
        parts[name] = eids[mask]
        left[mask] = False
    xone = np.array(geo['inflow']['xone'], dtype=float)
    xtwo = np.array(geo['inflow']['xtwo'], dtype=float)
    seg = xtwo - xone
    lseg = np.sqrt((seg**2).sum())
    rel = mid - xone[None, :]
    along = rel @ seg/lseg**2
    dist = np.abs(rel[:, 0]*seg[1] - rel[:, 1]*seg[0])/lseg
    take('inflow', (dist < tol*lseg) & (along > -tol) & (along < 1 + tol))
    circles = []
    for key in ('controlbcs', 'moving walls'):
        for k, cb in enumerate(geo.get(key, [])):
            if cb.get('type') not in ('rotating circle', 'circle'):
                continue
            g = cb.get('geometry', cb)
            cen, rad = np.array(g['center'], dtype=float), float(g['radius'])
            r = np.sqrt(((mid - cen[None, :])**2).sum(axis=1))
            name = ('control', k) if key == 'controlbcs' else ('moving', k)
            take(name, np.abs(r - rad) < 0.05*rad)   # chord midpoints sit
            circles.append(name)                     # slightly inside
    # outflow: the straight side of the bounding box opposite to the inflow
    nin = np.array(geo['inflow']['inward normal'], dtype=float)
    proj = mid @ nin
    take('outflow', proj > (th.mesh.verts @ nin).max() - tol*lseg)
    take('walls', left.copy())
    return parts


def gen_bccont_fems(meshname='2D-double-rotcyl_lvl1',
                    geodata='2D-double-rotcyl_geo_cntrlbc_rotcntrl',
                    bccontrol=True, inflowvel=1., mesh=None, geo=None):
    """general 2D inflow/outflow set-up with boundary control (reference
    `problem_setups.gen_bccont_fems`, dnsps:773-987): parabolic inflow
    (`InflowParabola`, dnsps:1014-1041: `6 s (1 - s)` along the inflow
    segment), no-slip walls, do-nothing outflow, control boundaries either
    Robin-penalised (`bccontrol=True`; shape functions `RotatingCircle`,
    dnsps:1083-1104: tangential unit rotation) or no-slip walls"""
    import json
    if mesh is None:
        mesh = load_npz_mesh(os.path.join(DATA_DIR,
                                          'mesh_{0}.npz'.format(meshname)))
    if geo is None:
        with open(os.path.join(DATA_DIR, geodata + '.json')) as fh:
            geo = json.load(fh)
    th = TaylorHood(mesh)
    parts = classify_boundary(th, geo)
    xone = np.array(geo['inflow']['xone'], dtype=float)
    xtwo = np.array(geo['inflow']['xtwo'], dtype=float)
    nin = np.array(geo['inflow']['inward normal'], dtype=float)
    lenb = np.linalg.norm(xone - xtwo)
    vals = {}

    def setnodes(eids, fun):
        nodes = np.unique(th._edge_nodes(eids))
        uv = fun(th.nodecoords[nodes])
        for nd, (ux, uy) in zip(nodes, uv):
            vals[2*nd], vals[2*nd + 1] = ux, uy

    def parabola(xy):
        s = np.sqrt(((xy - xone[None, :])**2).sum(axis=1))/lenb
        return (inflowvel*6*s*(1 - s))[:, None]*nin[None, :]

    def zero(xy):
        return np.zeros_like(xy)
    # order as the reference's `diribcu` list (later entries win at shared
    # nodes: the corner nodes of the inflow belong to the walls)
    setnodes(parts['inflow'], parabola)
    setnodes(parts['walls'], zero)
    cntrl = sorted(k for k in parts if isinstance(k, tuple)
                   and k[0] == 'control')
    moving = sorted(k for k in parts if isinstance(k, tuple)
                    and k[0] == 'moving')
    for k in moving:                       # `movingwallcntrl=False`: walls
        setnodes(parts[k], zero)
    if not bccontrol:
        for k in cntrl:
            setnodes(parts[k], zero)
    dbcinds = np.array(sorted(vals.keys()), dtype=np.int32)
    dbcvals = np.array([vals[k] for k in dbcinds])
    shapefuns = []
    if bccontrol:
        for k in cntrl:
            cb = geo['controlbcs'][k[1]]
            cen, rad = np.array(cb['center'], dtype=float), float(cb['radius'])

            def rot(xy, cen=cen, rad=rad):
                # RotatingCircle(omega=1): anglevel = radius
                curn = (xy - cen[None, :])/rad
                return rad*np.stack([-curn[:, 1], curn[:, 0]], axis=1)
            shapefuns.append(rot)
    return dict(V=th, mesh=mesh, parts=parts, dbcinds=dbcinds,
                dbcvals=dbcvals, charlen=float(geo['characteristic length']),
                contrbcsedges=[parts[k] for k in cntrl] if bccontrol else [],
                contrbcsshapefuns=shapefuns, outflowedges=parts['outflow'])


def get_sysmats(problem='cylinderwake', N=2, refine=0, Re=None, nu=None,
                charvel=1., mesh=None, bccontrol=False, meshparams=None):
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
    elif problem == 'gen_bccont':
        gfem = gen_bccont_fems(bccontrol=bccontrol, inflowvel=charvel,
                               mesh=mesh, **(meshparams or {}))
        th, mesh, charlen = gfem['V'], gfem['mesh'], gfem['charlen']
        dbcinds, dbcvals = gfem['dbcinds'], gfem['dbcvals']
        ppin = None
    else:
        raise NotImplementedError(problem)
    if Re is not None:
        nu = charvel*charlen/Re                         # dnsps:138-141
    else:
        Re = charvel*charlen/nu
    stms = th.stokes_mats(nu=nu)
    if problem == 'gen_bccont':
        # symmetric-gradient form corrected at the outflow (dts:246-248)
        stms['A'] = (stms['A'] + th.outflow_stress_correction(
            gfem['outflowedges'], nu)).tocsr()
    if ppin == -1:
        stms['J'] = stms['J'][:-1, :].tocsr()
        stms['JT'] = stms['JT'][:, :-1].tocsr()
    stokesmatsc, rhsd, invinds = condense_sysmatsbybcs(stms, dbcinds, dbcvals)
    femp = dict(V=th, invinds=invinds, dbcinds=dbcinds, dbcvals=dbcvals,
                nu=nu, Re=Re, charlen=charlen, ppin=ppin, mesh=mesh)
    if problem == 'gen_bccont' and bccontrol:
        # Robin-penalised control boundaries (dts:277-320, dnsps:201-210):
        # `Arob` = sum of the boundary mass matrices, `Brob[:, k]` = boundary
        # load of the k-th shape function; the caller adds `Arob/alpha` to `A`
        # and `Brob/alpha * u(t)` to the rhs
        # (tests/time_dep_nse_double_rotcyl_bcrob.py:37-47)
        amatrob = None
        bcols = []
        for eids, gfun in zip(gfem['contrbcsedges'],
                              gfem['contrbcsshapefuns']):
            mb = th.boundary_mass(eids)
            amatrob = mb if amatrob is None else amatrob + mb
            bcols.append(th.boundary_load(eids, gfun))
        auxu = np.zeros((th.vdim, 1))
        auxu[dbcinds, 0] = dbcvals
        if np.linalg.norm((amatrob @ auxu)[invinds]) > 1e-15:
            raise UserWarning('diri and control bc must not intersect')
        stokesmatsc.update(
            Arob=amatrob[invinds, :][:, invinds].tocsr(),
            Brob=np.hstack(bcols)[invinds, :])
        femp.update(parts=gfem['parts'])
    return femp, stokesmatsc, rhsd
