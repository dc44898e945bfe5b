"""Boundary-condition plumbing of the caller side of the hot path, on the
device (`dns_bcmap_*`, `dns_op_*` of include/dns_amd.h):

 * `append_bcs_vec`  (reference dts:49-64)  scatter inner + Dirichlet values
   into the full velocity vector, `NaN` where neither set reaches
 * `ResidentOperator`  a CSR matrix kept in HBM for repeated products
 * `make_applybcs`   the `applybcs` closure of `solve_nse` (snu:1103-1115):
   `(-A[:, cnt] vals, -J[:, cnt] vals, M[:, cnt] vals)` for the current values
   of the controlled Dirichlet dofs
 * `condense_velmatsbybcs(get_rhs_only=True)` (dts:576-630)
"""
import ctypes as ct

import numpy as np
import scipy.sparse as sps

from . import _capi as C

__all__ = ['BcMap', 'append_bcs_vec', 'ResidentOperator', 'make_applybcs',
           'condense_velmatsbybcs_rhs', 'clear_cache']

_maps = {}


def _i32(arr):
    return np.ascontiguousarray(np.asarray(arr).reshape(-1), dtype=np.int32)


class BcMap(object):
    """index sets of `append_bcs_vec` resident on the device"""

    def __init__(self, vdim, invinds, bcinds, device=0):
        self.lib = C.load_library()
        self.vdim = int(vdim)
        self._inv, self._bc = _i32(invinds), _i32(bcinds)
        self._h = ct.c_void_p()
        C.check(self.lib.dns_bcmap_create(
            device, self.vdim, self._inv.size,
            self._inv.ctypes.data_as(C.c_int32_p), self._bc.size,
            self._bc.ctypes.data_as(C.c_int32_p), ct.byref(self._h)))

    def scatter(self, vvec, bcvals):
        v = C.as_f64(vvec, size=self._inv.size)
        b = C.as_f64(bcvals, size=self._bc.size)
        out = np.empty(self.vdim)
        C.check(self.lib.dns_bc_scatter(self._h, C.dptr(v), C.dptr(b),
                                        C.dptr(out)))
        return out.reshape((-1, 1))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.dns_bcmap_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def clear_cache():
    for m in _maps.values():
        m.close()
    _maps.clear()


def _unroll(bcinds, bcvals):
    if bcinds is None or len(bcinds) == 0:
        return [], []
    if not isinstance(bcinds[0], (list, tuple, np.ndarray)):
        return list(bcinds), list(bcvals)
    ui, uv = [], []
    for k, cbci in enumerate(bcinds):
        ui.extend(list(cbci))
        uv.extend(list(bcvals[k]))
    return ui, uv


def append_bcs_vec(vvec, V=None, vdim=None, bcinds=None, bcvals=None,
                   invinds=None, device=0, **kwargs):
    """append given boundary conditions to a vector of inner-node values
    (reference dts:49-64): `(vdim, 1)` array, `NaN` where no index reaches"""
    if vdim is None:
        vdim = V.vdim
    cbcinds, cbcvals = _unroll(bcinds, bcvals)
    inv, bci = _i32(invinds), _i32(cbcinds)
    key = (int(vdim), inv.size, bci.size, hash(inv.tobytes()),
           hash(bci.tobytes()), device)
    m = _maps.get(key)
    if m is None:
        m = BcMap(vdim, inv, bci, device=device)
        _maps[key] = m
        while len(_maps) > 8:
            _maps.pop(next(iter(_maps))).close()
    return m.scatter(vvec, cbcvals)


class ResidentOperator(object):
    """`y = alpha A x + beta y` with `A` resident in HBM (host vectors)"""

    def __init__(self, A, device=0):
        self.lib = C.load_library()
        self._view = C.CsrView(sps.csr_matrix(A))
        self.shape = self._view.shape
        self._h = ct.c_void_p()
        C.check(self.lib.dns_op_create(device, self._view.byref(),
                                       ct.byref(self._h)))

    def apply(self, x, y=None, alpha=1., beta=0.):
        x = C.as_f64(x, size=self.shape[1])
        out = np.zeros(self.shape[0]) if y is None else \
            C.as_f64(y, size=self.shape[0]).copy()
        C.check(self.lib.dns_op_apply(self._h, C.dptr(x), C.dptr(out),
                                      float(alpha), float(beta)))
        return out.reshape((-1, 1))

    def __matmul__(self, x):
        return self.apply(x)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.dns_op_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_applybcs(A, J, M, loccntbcinds, locinvinds, device=0,
                  reference_literal=True):
    """the `applybcs(bcs_n)` closure of `solve_nse` (snu:1103-1115)

    `reference_literal=True` (default): what the reference EXECUTES.  snu:1112
    has the assignment `cauxvec[loccntbcinds, 0] = bcs_n` commented out, so
    its closure multiplies `A`, `J`, `M` with a vector that stays zero and
    returns zero columns of the shapes `(len(locinvinds), 1)`, `(NP, 1)`,
    `(len(locinvinds), 1)` whatever the control does (the controlled values
    reach the flow through `appndbcs` / the convection only).
    `reference_literal=False`: the values ARE written -- what the commented
    line evidently meant: `(-A[:, cnt] vals, -J[:, cnt] vals, M[:, cnt] vals)`
    from one resident operator, one launch per call."""
    if loccntbcinds is None or len(loccntbcinds) == 0:
        def applybcs(bcs_n):
            return 0., 0., 0.              # snu:1104-1105
        return applybcs
    cnt = np.asarray(loccntbcinds)
    inv = np.asarray(locinvinds)
    NV, NP = A.shape[0], J.shape[0]
    if reference_literal:
        zv, zp = np.zeros((inv.size, 1)), np.zeros((NP, 1))

        def applybcs(bcs_n):               # snu:1111-1115 as it runs
            return -zv, -zp, zv.copy()
        applybcs.operator = None
        return applybcs
    # only the controlled columns matter: [A; J; M][:, cnt] as ONE resident
    # operator -- a single launch per call
    stack = sps.vstack([sps.csr_matrix(A)[:, cnt], sps.csr_matrix(J)[:, cnt],
                        sps.csr_matrix(M)[:, cnt]], format='csr')
    op = ResidentOperator(stack, device=device)

    def applybcs(bcs_n):
        out = op.apply(np.asarray(bcs_n, dtype=np.float64))
        return (-out[:NV][inv, :], -out[NV:NV+NP], out[NV+NP:][inv, :])
    applybcs.operator = op
    return applybcs


def condense_velmatsbybcs_rhs(A, invinds=None, dbcinds=None, dbcvals=None,
                              op=None):
    """`condense_velmatsbybcs(A, ..., get_rhs_only=True)` (dts:576-630):
    `-(A bcsv)[invinds]` with `bcsv` the Dirichlet values scattered into a zero
    vector; `op`: a `ResidentOperator` of `A` to reuse"""
    bci, bcv = _unroll(dbcinds, dbcvals)
    nv = A.shape[1]
    bcsv = np.zeros(nv)
    bcsv[bci] = bcv
    own = op is None
    op = ResidentOperator(A) if own else op
    try:
        out = -op.apply(bcsv)
    finally:
        if own:
            op.close()
    return out[np.asarray(invinds), :]
