"""P2/P1 Taylor-Hood assembly on triangles (NumPy, host side).

Input scaffolding (SURVEY.md §7 stage 1): produces the SciPy CSR matrices and
vectors that `dolfin_to_sparrays` hands to the hot path, from the same weak
forms:

 * `M`  = inner(u, v) dx                         (reference dts:243)
 * `A`  = nu*inner(2*eps(u), grad(v)) dx, eps = symmetric gradient (dts:236-245)
 * `J`  = q*div(u) dx, `JT` = div(v)*p dx        (dts:254-255)
 * `MP` = inner(p, q) dx                         (dts:244)
 * `N(u)u` = inner(grad(u)*u, v) dx              (dts:463)
 * `N1(u0)` = inner(grad(u)*u0, v) dx, `N2(u0)` = inner(grad(u0)*u, v) dx
                                                 (dts:358-359)

Degrees of freedom: P2 nodes (mesh vertices and edge midpoints) are renumbered
by reverse Cuthill-McKee, as dolfin's default `reorder_dofs_serial` does for
the reference's matrices; velocity dof `2*node + component`; pressure dofs
follow the order of their vertices.  The numbering still differs from
dolfin's -- parity is with the reference *algorithm* on identical matrices,
not with dolfin's assembly.
"""
import numpy as np
import scipy.sparse as sps

__all__ = ['TaylorHood']

# 7-point, degree-5 Gauss rule on the triangle (barycentric coordinates)
_s15 = np.sqrt(15.0)
_t1, _t2 = (6.0 - _s15)/21.0, (6.0 + _s15)/21.0
_w1, _w2 = (155.0 - _s15)/1200.0, (155.0 + _s15)/1200.0
_QP = np.array([[1/3., 1/3., 1/3.],
                [1-2*_t1, _t1, _t1], [_t1, 1-2*_t1, _t1], [_t1, _t1, 1-2*_t1],
                [1-2*_t2, _t2, _t2], [_t2, 1-2*_t2, _t2], [_t2, _t2, 1-2*_t2]])
_QW = np.array([9/40., _w1, _w1, _w1, _w2, _w2, _w2])  # sums to 1

# local P2 node k = 3,4,5 sits on the edge opposite to vertex k-3
_EDGE_OF = [(1, 2), (0, 2), (0, 1)]


def _p2_basis(lam):
    """P2 shape functions at barycentric points `lam (nq, 3)` -> `(nq, 6)`"""
    phi = np.empty((lam.shape[0], 6))
    for k in range(3):
        phi[:, k] = lam[:, k]*(2*lam[:, k] - 1)
    for k, (i, j) in enumerate(_EDGE_OF):
        phi[:, 3+k] = 4*lam[:, i]*lam[:, j]
    return phi


def _p2_dbasis_dlam(lam):
    """d(phi_a)/d(lam_i) at `lam (nq,3)` -> `(nq, 6, 3)`"""
    d = np.zeros((lam.shape[0], 6, 3))
    for k in range(3):
        d[:, k, k] = 4*lam[:, k] - 1
    for k, (i, j) in enumerate(_EDGE_OF):
        d[:, 3+k, i] = 4*lam[:, j]
        d[:, 3+k, j] = 4*lam[:, i]
    return d


class TaylorHood(object):
    """Taylor-Hood spaces on a `Mesh2D` and the forms of the NSE"""

    def __init__(self, mesh, reorder=True):
        self.mesh = mesh
        edges, celledges, nadj = mesh.edges()
        self.edges, self.celledges, self.edge_nadj = edges, celledges, nadj
        nv = mesh.nverts
        self.nnodes = nv + edges.shape[0]
        self.vdim = 2*self.nnodes
        self.pdim = nv
        cellnodes = np.hstack([mesh.cells, nv + celledges])       # (nc, 6)
        nodecoords = np.vstack(
            [mesh.verts, 0.5*(mesh.verts[edges[:, 0]]+mesh.verts[edges[:, 1]])])
        # locality-preserving dof numbering (what dolfin's default
        # `reorder_dofs_serial` gives the reference's matrices): reverse
        # Cuthill-McKee on the P2 node graph; pressure dofs follow the order
        # of their vertices in it
        if reorder:
            from scipy.sparse.csgraph import reverse_cuthill_mckee
            nn = self.nnodes
            r = np.repeat(cellnodes, 6, axis=1).ravel()
            c = np.tile(cellnodes, (1, 6)).ravel()
            graph = sps.coo_matrix((np.ones(r.size, dtype=np.int8), (r, c)),
                                   shape=(nn, nn)).tocsr()
            order = reverse_cuthill_mckee(graph, symmetric_mode=True)
            newid = np.empty(nn, dtype=np.int64)
            newid[order] = np.arange(nn)
        else:
            newid = np.arange(self.nnodes, dtype=np.int64)
        self.node_newid = newid                  # (vertex|edge id) -> P2 node
        self.cellnodes = newid[cellnodes]
        self.nodecoords = np.empty_like(nodecoords)
        self.nodecoords[newid] = nodecoords
        vorder = np.argsort(newid[:nv])
        self.vert_pdof = np.empty(nv, dtype=np.int64)
        self.vert_pdof[vorder] = np.arange(nv)   # vertex -> pressure dof
        self.cellpdofs = self.vert_pdof[mesh.cells]                 # (nc, 3)
        # geometry: gradients of the barycentric coordinates, (nc, 3, 2)
        p = mesh.verts[mesh.cells]
        self.area = mesh.cell_areas()
        glam = np.empty((mesh.ncells, 3, 2))
        for k in range(3):
            i, j = (k+1) % 3, (k+2) % 3
            glam[:, k, 0] = (p[:, i, 1] - p[:, j, 1])/(2*self.area)
            glam[:, k, 1] = (p[:, j, 0] - p[:, i, 0])/(2*self.area)
        self.glam = glam
        self._phi = _p2_basis(_QP)                  # (nq, 6)
        self._dphi_dlam = _p2_dbasis_dlam(_QP)      # (nq, 6, 3)
        # physical P2 gradients at quadrature points, (nc, nq, 6, 2)
        self._gphi = np.einsum('qai,cid->cqad', self._dphi_dlam, glam)
        self._psi = _QP                              # P1 basis = lambdas

    # -- helpers ---------------------------------------------------------
    def _vdofs(self):
        """cell velocity dofs `(nc, 6, 2)`"""
        return 2*self.cellnodes[:, :, None] + np.arange(2)[None, None, :]

    def _assemble(self, rows, cols, vals, shape, keep_pattern=False):
        mat = sps.coo_matrix((vals.ravel(), (rows.ravel(), cols.ravel())),
                             shape=shape).tocsr()
        mat.sum_duplicates()
        if not keep_pattern:
            mat.eliminate_zeros()   # as `mat_dolfin2sparse`, reference dts:80
        mat.sort_indices()
        return mat

    # -- linear forms and matrices ---------------------------------------
    def stokes_mats(self, nu=1.0):
        """`dict(M, A, J, JT, MP)` on the full spaces (no BCs resolved)"""
        nc = self.mesh.ncells
        w = _QW[None, :]*self.area[:, None]                       # (nc, nq)
        phi, gphi, psi = self._phi, self._gphi, self._psi
        vd = self._vdofs()                                        # (nc,6,2)
        # scalar P2 mass and stiffness blocks
        mloc = np.einsum('cq,qa,qb->cab', w, phi, phi)
        kloc = np.einsum('cq,cqad,cqbd->cab', w, gphi, gphi)
        # d_c phi_a * d_d phi_b
        gloc = np.einsum('cq,cqae,cqbd->cabed', w, gphi, gphi)
        rows, cols, mv, av = [], [], [], []
        for d in range(2):           # test component
            for c in range(2):       # trial component
                r = np.broadcast_to(vd[:, :, None, d], (nc, 6, 6))
                cc = np.broadcast_to(vd[:, None, :, c], (nc, 6, 6))
                # A[(a,d),(b,c)] = nu*(delta_cd grad a.grad b + d_c a d_d b)
                aval = nu*gloc[:, :, :, c, d]
                if c == d:
                    aval = aval + nu*kloc
                    mv.append(mloc)
                else:
                    mv.append(np.zeros_like(mloc))
                rows.append(r), cols.append(cc), av.append(aval)
        rows, cols = np.array(rows), np.array(cols)
        M = self._assemble(rows, cols, np.array(mv), (self.vdim, self.vdim))
        A = self._assemble(rows, cols, np.array(av), (self.vdim, self.vdim))
        # divergence: J[q_a, (b,c)] = int psi_a d_c phi_b
        jloc = np.einsum('cq,qa,cqbd->cabd', w, psi, gphi)      # (nc,3,6,2)
        jr = np.broadcast_to(self.cellpdofs[:, :, None, None], (nc, 3, 6, 2))
        jc = np.broadcast_to(vd[:, None, :, :], (nc, 3, 6, 2))
        J = self._assemble(jr, jc, jloc, (self.pdim, self.vdim))
        mploc = np.einsum('cq,qa,qb->cab', w, psi, psi)
        pr = np.broadcast_to(self.cellpdofs[:, :, None], (nc, 3, 3))
        pc = np.broadcast_to(self.cellpdofs[:, None, :], (nc, 3, 3))
        MP = self._assemble(pr, pc, mploc, (self.pdim, self.pdim))
        return dict(M=M, A=A, J=J, JT=sps.csr_matrix(J.T), MP=MP)

    def _vel_at_qp(self, uvec):
        """velocity `(nc,nq,2)` and its gradient `(nc,nq,2,2)` [i, j]=d_j u_i"""
        uloc = np.asarray(uvec).reshape(-1)[self._vdofs()]         # (nc,6,2)
        # (batched matrix products instead of einsum: 5x faster, the host
        # convection callback of the oracle legs runs thousands of times)
        uq = np.matmul(self._phi[None, :, :], uloc)                # cqi
        guq = np.matmul(uloc.transpose(0, 2, 1)[:, None, :, :],
                        self._gphi)                                # cqij
        return uq, guq

    def convection_vec(self, uvec, utwo=None):
        """`N(u1)u2 = inner(grad(u1)*u2, v) dx` as `(vdim, 1)` (dts:427-472)"""
        uq, guq = self._vel_at_qp(uvec)
        if utwo is not None:
            uq, _ = self._vel_at_qp(utwo)
        w = _QW[None, :]*self.area[:, None]
        conv = np.matmul(guq, uq[:, :, :, None])[:, :, :, 0]     # (nc,nq,2)
        floc = np.matmul(self._phi.T[None, :, :],
                         w[:, :, None]*conv)                      # (nc,6,2)
        fvec = np.bincount(self._vdofs().ravel(), weights=floc.ravel(),
                           minlength=self.vdim)
        return fvec.reshape((-1, 1))

    def convection_mats(self, u0vec, keep_pattern=False):
        """`N1(u0), N2(u0), N(u0)u0` on the full space (dts:325-376)

        `keep_pattern`: keep the structural pattern (explicit zeros) so that
        `A + N(v)` has ONE sparsity pattern for every `v` -- what lets the
        device keep the system resident and only re-value it per step"""
        nc = self.mesh.ncells
        uq, guq = self._vel_at_qp(u0vec)
        w = _QW[None, :]*self.area[:, None]
        vd = self._vdofs()
        # N1 = delta_cd int (u0 . grad phi_b) phi_a
        ugphi = np.einsum('cqj,cqbj->cqb', uq, self._gphi)
        n1loc = np.einsum('cq,qa,cqb->cab', w, self._phi, ugphi)
        # N2[(a,d),(b,c)] = int d_c u0_d phi_b phi_a
        n2loc = np.einsum('cq,qa,qb,cqdk->cabdk', w, self._phi, self._phi, guq)
        rows, cols, v1, v2 = [], [], [], []
        for d in range(2):
            for c in range(2):
                rows.append(np.broadcast_to(vd[:, :, None, d], (nc, 6, 6)))
                cols.append(np.broadcast_to(vd[:, None, :, c], (nc, 6, 6)))
                v1.append(n1loc if c == d else np.zeros_like(n1loc))
                v2.append(n2loc[:, :, :, d, c])
        rows, cols = np.array(rows), np.array(cols)
        N1 = self._assemble(rows, cols, np.array(v1), (self.vdim, self.vdim),
                            keep_pattern=keep_pattern)
        N2 = self._assemble(rows, cols, np.array(v2), (self.vdim, self.vdim),
                            keep_pattern=keep_pattern)
        return N1, N2, self.convection_vec(u0vec)

    # -- boundary forms (Robin-penalised control boundaries, outflow) ------
    def boundary_edges(self):
        """ids (into `self.edges`) of the boundary edges and their midpoints"""
        eids = np.where(self.edge_nadj == 1)[0]
        mid = 0.5*(self.mesh.verts[self.edges[eids, 0]]
                   + self.mesh.verts[self.edges[eids, 1]])
        return eids, mid

    def _edge_nodes(self, eids):
        """P2 nodes `(k, 3)` of edges: two vertices, then the midpoint"""
        nv = self.mesh.nverts
        return np.stack([self.node_newid[self.edges[eids, 0]],
                         self.node_newid[self.edges[eids, 1]],
                         self.node_newid[nv + eids]], axis=1)

    def boundary_mass(self, eids):
        """`inner(u, v) ds` over the edges `eids` on the full velocity space
        (the Robin form `arob`, reference dts:304)"""
        eids = np.asarray(eids)
        nodes = self._edge_nodes(eids)
        x = self.mesh.verts[self.edges[eids]]
        length = np.sqrt(((x[:, 1] - x[:, 0])**2).sum(axis=1))
        ref = np.array([[4., -1., 2.], [-1., 4., 2.], [2., 2., 16.]])/30.
        loc = length[:, None, None]*ref[None, :, :]               # (k,3,3)
        rows, cols, vals = [], [], []
        for comp in range(2):
            rows.append(np.broadcast_to(2*nodes[:, :, None] + comp,
                                        loc.shape))
            cols.append(np.broadcast_to(2*nodes[:, None, :] + comp,
                                        loc.shape))
            vals.append(loc)
        return self._assemble(np.array(rows), np.array(cols), np.array(vals),
                              (self.vdim, self.vdim))

    def boundary_load(self, eids, gfun):
        """`inner(v, g) ds` with `g` interpolated at the P2 nodes of the edges
        (how dolfin treats a degree-2 `UserExpression`): `(vdim, 1)`
        (the Robin input form `brob`, reference dts:305)"""
        eids = np.asarray(eids)
        nodes = np.unique(self._edge_nodes(eids))
        gvals = np.zeros((self.nnodes, 2))
        gvals[nodes] = gfun(self.nodecoords[nodes])
        return self.boundary_mass(eids) @ gvals.reshape((-1, 1))

    def outflow_stress_correction(self, eids, nu):
        """`- nu * inner(grad(u).T * n, v) ds` over the (outflow) edges `eids`:
        the term that turns the symmetric-gradient form into the do-nothing
        condition there (reference dts:246-248)"""
        eids = np.asarray(eids)
        emap = -np.ones(self.edges.shape[0], dtype=np.int64)
        emap[eids] = np.arange(eids.size)
        hit = emap[self.celledges] >= 0                       # (nc, 3)
        cells, kloc = np.nonzero(hit)
        gp = np.array([0.5 - 0.5*np.sqrt(0.6), 0.5, 0.5 + 0.5*np.sqrt(0.6)])
        gw = np.array([5., 8., 5.])/18.
        rows, cols, vals = [], [], []
        vd = self._vdofs()
        for c, k in zip(cells, kloc):
            i, j = _EDGE_OF[k]
            vi, vj, vk = (self.mesh.verts[self.mesh.cells[c, q]]
                          for q in (i, j, k))
            tang = vj - vi
            length = np.sqrt((tang**2).sum())
            nrm = np.array([tang[1], -tang[0]])/length
            if np.dot(nrm, 0.5*(vi + vj) - vk) < 0:
                nrm = -nrm                                    # outward
            lam = np.zeros((3, 3))
            lam[:, i], lam[:, j] = 1. - gp, gp
            phi = _p2_basis(lam)                              # (3, 6)
            gphi = np.einsum('qai,id->qad', _p2_dbasis_dlam(lam),
                             self.glam[c])                    # (3, 6, 2)
            # loc[(a,i),(b,j)] = -nu * sum_q w phi_a d_i phi_b n_j
            loc = -nu*length*np.einsum('q,qa,qbi,j->aibj', gw, phi, gphi, nrm)
            r = np.broadcast_to(vd[c][:, :, None, None], (6, 2, 6, 2))
            cc = np.broadcast_to(vd[c][None, None, :, :], (6, 2, 6, 2))
            rows.append(r.ravel()), cols.append(cc.ravel())
            vals.append(loc.ravel())
        if not rows:
            return sps.csr_matrix((self.vdim, self.vdim))
        return self._assemble(np.concatenate(rows), np.concatenate(cols),
                              np.concatenate(vals), (self.vdim, self.vdim))

    # -- boundary conditions ---------------------------------------------
    def boundary_nodes(self):
        """P2 nodes on the boundary and their coordinates"""
        bedges = np.where(self.edge_nadj == 1)[0]
        nodes = np.unique(self.node_newid[np.concatenate(
            [self.edges[bedges].ravel(), self.mesh.nverts + bedges])])
        return nodes, self.nodecoords[nodes]

    def cylinderwake_bcs(self, xmin=0.0, xmax=2.2, ymin=0.0, ymax=0.41,
                         xcenter=0.2, ycenter=0.2, radius=0.05,
                         bmarg=1.e-3, obstacle_halfwidth=None):
        """Dirichlet data of the 2D cylinder wake (problem_setups.py:371-599)

        inflow parabola `4*y*(ymax-y)/ymax**2` at `x=xmin`, no-slip at the
        channel walls and at the obstacle, do-nothing at `x=xmax`

        Returns
        -------
        dbcinds : (K,) int32 velocity dofs with Dirichlet data
        dbcvals : (K,) float64
        invinds : (NV,) int32 inner velocity dofs (as reference dts:533)
        """
        nodes, xy = self.boundary_nodes()
        x, y = xy[:, 0], xy[:, 1]
        inflow = x < xmin + bmarg
        walls = (y < ymin + bmarg) | (y > ymax - bmarg)
        outflow = (x > xmax - bmarg) & ~walls
        obstacle = ~inflow & ~walls & ~outflow
        if obstacle_halfwidth is None:
            r = np.sqrt((x-xcenter)**2 + (y-ycenter)**2)
            if not np.all(r[obstacle] < radius + bmarg):
                raise ValueError('unclassified boundary nodes')
        vals = {}
        for nd, yy in zip(nodes[inflow], y[inflow]):
            vals[2*nd] = 4*yy*(ymax-yy)/(ymax*ymax)
            vals[2*nd+1] = 0.0
        for nd in np.concatenate([nodes[walls], nodes[obstacle]]):
            vals[2*nd] = 0.0
            vals[2*nd+1] = 0.0
        dbcinds = np.array(sorted(vals.keys()), dtype=np.int32)
        dbcvals = np.array([vals[k] for k in dbcinds])
        invinds = np.setdiff1d(np.arange(self.vdim), dbcinds).astype(np.int32)
        return dbcinds, dbcvals, invinds

    def drivencavity_bcs(self, lidvel=1.0, bmarg=1e-10):
        """enclosed flow: lid `u=(lidvel,0)` at `y=ymax`, no-slip elsewhere"""
        nodes, xy = self.boundary_nodes()
        ytop = self.mesh.verts[:, 1].max()
        vals = {}
        for nd, yy in zip(nodes, xy[:, 1]):
            vals[2*nd] = lidvel if yy > ytop - bmarg else 0.0
            vals[2*nd+1] = 0.0
        dbcinds = np.array(sorted(vals.keys()), dtype=np.int32)
        dbcvals = np.array([vals[k] for k in dbcinds])
        invinds = np.setdiff1d(np.arange(self.vdim), dbcinds).astype(np.int32)
        return dbcinds, dbcvals, invinds
