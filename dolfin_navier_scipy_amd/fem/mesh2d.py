"""Triangular 2-D meshes: dolfin-XML reader, uniform refinement, edge tables.

Input scaffolding only (SURVEY.md §7 stage 1 / §8f row 3): neither FEniCS nor
the reference package is importable on the build or GPU boxes, so the
saddle-point matrices the hot path consumes are produced by this host-side
NumPy code.  It plays the role of `dolfin.Mesh("mesh/cylinder_%d.xml")`
(reference `dolfin_navier_scipy/problem_setups.py:476-478`).
"""
import gzip
import re

import numpy as np

__all__ = ['Mesh2D', 'read_dolfin_xml', 'load_npz_mesh', 'save_npz_mesh',
           'refine_uniform', 'rectangle_mesh', 'channel_cylinder_mesh']


class Mesh2D(object):
    """vertices `(nv, 2)` float64, cells `(nc, 3)` int32 (counter-clockwise)"""

    def __init__(self, verts, cells):
        self.verts = np.ascontiguousarray(verts, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int64)
        # orient all cells counter-clockwise
        p0, p1, p2 = (self.verts[cells[:, k]] for k in range(3))
        det = (p1[:, 0]-p0[:, 0])*(p2[:, 1]-p0[:, 1]) \
            - (p1[:, 1]-p0[:, 1])*(p2[:, 0]-p0[:, 0])
        flip = det < 0
        cells[flip, 1], cells[flip, 2] = cells[flip, 2].copy(), \
            cells[flip, 1].copy()
        self.cells = cells
        self._edges = None

    @property
    def nverts(self):
        return self.verts.shape[0]

    @property
    def ncells(self):
        return self.cells.shape[0]

    def edges(self):
        """unique edges

        Returns
        -------
        edges : (ne, 2) int64, sorted vertex pairs
        celledges : (nc, 3) int64, edge `k` of a cell is opposite to vertex `k`
        nadj : (ne,) number of adjacent cells (1 = boundary edge)
        """
        if self._edges is None:
            c = self.cells
            loc = np.stack([c[:, [1, 2]], c[:, [0, 2]], c[:, [0, 1]]], axis=1)
            loc = np.sort(loc.reshape(-1, 2), axis=1)
            key = loc[:, 0]*self.nverts + loc[:, 1]
            ukey, inv, cnt = np.unique(key, return_inverse=True,
                                       return_counts=True)
            edges = np.stack([ukey // self.nverts, ukey % self.nverts], axis=1)
            self._edges = (edges, inv.reshape(-1, 3), cnt)
        return self._edges

    def cell_areas(self):
        p0, p1, p2 = (self.verts[self.cells[:, k]] for k in range(3))
        return 0.5*((p1[:, 0]-p0[:, 0])*(p2[:, 1]-p0[:, 1])
                    - (p1[:, 1]-p0[:, 1])*(p2[:, 0]-p0[:, 0]))


def read_dolfin_xml(path):
    """read a dolfin-XML triangle mesh (`tests/mesh/*.xml(.gz)` format)"""
    opener = gzip.open if str(path).endswith('.gz') else open
    with opener(path, 'rt') as fh:
        txt = fh.read()
    vre = re.compile(r'<vertex index="(\d+)" x="([^"]+)" y="([^"]+)"')
    cre = re.compile(r'<triangle index="(\d+)" v0="(\d+)" v1="(\d+)" '
                     r'v2="(\d+)"')
    vl = vre.findall(txt)
    cl = cre.findall(txt)
    verts = np.zeros((len(vl), 2))
    for idx, x, y in vl:
        verts[int(idx)] = (float(x), float(y))
    cells = np.zeros((len(cl), 3), dtype=np.int64)
    for idx, a, b, c in cl:
        cells[int(idx)] = (int(a), int(b), int(c))
    return Mesh2D(verts, cells)


def save_npz_mesh(path, mesh):
    np.savez_compressed(path, verts=mesh.verts,
                        cells=mesh.cells.astype(np.int32))


def load_npz_mesh(path):
    dat = np.load(path)
    return Mesh2D(dat['verts'], dat['cells'])


def refine_uniform(mesh, snap=None):
    """red refinement: every triangle is split into four

    Parameters
    ----------
    snap : callable, optional
        `snap(xy_mid, on_boundary_mask) -> xy_mid` to project new boundary
        midpoints onto a curved boundary

    Returns
    -------
    fine : Mesh2D
    parents : (nv_fine, 2) int64
        the two coarse vertices a fine vertex interpolates (equal for the
        coarse vertices themselves) -- defines the P1 prolongation
    """
    edges, celledges, nadj = mesh.edges()
    nv = mesh.nverts
    mid = 0.5*(mesh.verts[edges[:, 0]] + mesh.verts[edges[:, 1]])
    if snap is not None:
        mid = snap(mid, nadj == 1)
    verts = np.vstack([mesh.verts, mid])
    c = mesh.cells
    m0, m1, m2 = (nv + celledges[:, k] for k in range(3))  # opp. to vertex k
    fine = np.vstack([
        np.stack([c[:, 0], m2, m1], axis=1),
        np.stack([m2, c[:, 1], m0], axis=1),
        np.stack([m1, m0, c[:, 2]], axis=1),
        np.stack([m0, m1, m2], axis=1)])
    parents = np.vstack([np.stack([np.arange(nv), np.arange(nv)], axis=1),
                         edges])
    return Mesh2D(verts, fine), parents


def rectangle_mesh(nx, ny, lx=1.0, ly=1.0):
    """structured triangulation of `[0,lx]x[0,ly]` (for small tests)"""
    xs = np.linspace(0, lx, nx+1)
    ys = np.linspace(0, ly, ny+1)
    X, Y = np.meshgrid(xs, ys, indexing='ij')
    verts = np.stack([X.ravel(), Y.ravel()], axis=1)

    def vid(i, j):
        return i*(ny+1) + j
    cells = []
    for i in range(nx):
        for j in range(ny):
            a, b, c, d = vid(i, j), vid(i+1, j), vid(i+1, j+1), vid(i, j+1)
            if (i + j) % 2 == 0:
                cells += [(a, b, c), (a, c, d)]
            else:
                cells += [(a, b, d), (b, c, d)]
    return Mesh2D(verts, np.array(cells))


def channel_cylinder_mesh(nx=22, ny=8, lx=2.2, ly=0.41,
                          xc=0.2, yc=0.2, radius=0.05):
    """a crude channel-with-obstacle mesh for CPU-sized tests

    cells whose centroid lies in the bounding square of the cylinder are
    removed from a structured triangulation -- a polygonal stand-in with
    the topology (one hole) and boundary classes of the Schaefer-Turek
    domain (reference `problem_setups.py:371-379`)
    """
    base = rectangle_mesh(nx, ny, lx, ly)
    cen = base.verts[base.cells].mean(axis=1)
    keep = ~((np.abs(cen[:, 0]-xc) < radius) & (np.abs(cen[:, 1]-yc) < radius))
    cells = base.cells[keep]
    used = np.unique(cells)
    remap = -np.ones(base.nverts, dtype=np.int64)
    remap[used] = np.arange(used.size)
    return Mesh2D(base.verts[used], remap[cells])
