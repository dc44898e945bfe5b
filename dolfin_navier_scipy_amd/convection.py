"""Device convection `N(u)u` for P2 velocities on triangles (`dns_conv_*`).

Replaces the per-step host callback `f_vdp` of the reference
(`stokes_navier_utils.py:1136-1140` -> `dolfin_to_sparrays.get_convvec`,
dts:427-472): with an operator attached (`ImexStepper.set_convection`) the
CNAB/SBDF2 loop needs no host round trip at all.
"""
import ctypes as ct

import numpy as np

from . import _capi as C

__all__ = ['ConvectionP2']


def _i32(arr):
    return np.ascontiguousarray(arr, dtype=np.int32)


class ConvectionP2(object):
    def __init__(self, cell_vdofs, glam, area, vdim, invinds, dbcinds,
                 dbcvals, device=0):
        """`cell_vdofs (nc, 6, 2)`, `glam (nc, 3, 2)`, `area (nc,)` -- the
        element data any P2 FE library has (see include/dns_amd.h)"""
        self.lib = C.load_library()
        cv = _i32(np.asarray(cell_vdofs).reshape(-1))
        gl = np.ascontiguousarray(np.asarray(glam, dtype=np.float64).reshape(-1))
        ar = C.as_f64(area)
        self.ncells = ar.size
        self._cell_vdofs = cv.reshape((-1, 12))
        self._vdim = int(vdim)
        inv, dbi = _i32(invinds), _i32(dbcinds)
        self._invinds = inv
        dbv = C.as_f64(dbcvals, size=dbi.size)
        self.ndbc = int(dbi.size)
        self.nv = inv.size
        self._h = ct.c_void_p()
        C.check(self.lib.dns_conv_create_p2(
            device, self.ncells, cv.ctypes.data_as(C.c_int32_p), C.dptr(gl),
            C.dptr(ar), int(vdim), inv.size, inv.ctypes.data_as(C.c_int32_p),
            dbi.size, dbi.ctypes.data_as(C.c_int32_p), C.dptr(dbv),
            ct.byref(self._h)))

    @classmethod
    def from_taylor_hood(cls, th, invinds, dbcinds, dbcvals, device=0):
        """element data of the scaffolding assembler (`fem.TaylorHood`)"""
        return cls(th._vdofs(), th.glam, th.area, th.vdim, invinds, dbcinds,
                   dbcvals, device=device)

    def set_dbcvals(self, dbcvals):
        """one constant set of Dirichlet values (drops a value table)"""
        dbv = C.as_f64(dbcvals, size=self.ndbc)
        C.check(self.lib.dns_conv_set_dbcvals(self._h, C.dptr(dbv)))

    def set_dbc_table(self, table):
        """`(nrows, ndbc)` Dirichlet values per step (IMEX stepper: row s for
        the s-th step after the stepper's tables were set) or per trajectory
        slot (trapezoidal sweeps)"""
        tab = np.ascontiguousarray(table, dtype=np.float64).reshape(
            (-1, max(self.ndbc, 1)) if self.ndbc else (len(table), 0))
        nrows = tab.shape[0]
        flat = C.as_f64(tab) if tab.size else np.zeros(1)
        C.check(self.lib.dns_conv_set_dbc_table(self._h, int(nrows),
                                                C.dptr(flat)))

    def set_dbc_row(self, row):
        C.check(self.lib.dns_conv_set_dbc_row(self._h, int(row)))

    def apply(self, v_inner, scale=1.0):
        v = C.as_f64(v_inner, size=self.nv)
        out = np.empty(self.nv)
        C.check(self.lib.dns_conv_apply(self._h, C.dptr(v), float(scale),
                                        C.dptr(out)))
        return out.reshape((-1, 1))

    # -- linearised convection matrices (Newton/Picard sweeps) ---------------
    def connectivity(self):
        """CSR of ones: all pairs of inner velocity dofs sharing a cell -- the
        pattern of the condensed `N1(u) + N2(u)` for any `u`"""
        import scipy.sparse as sps
        code = np.full(self._vdim, -1, dtype=np.int64)
        code[self._invinds] = np.arange(self.nv)
        loc = code[self._cell_vdofs]                      # (nc, 12)
        rows = np.repeat(loc[:, :, None], 12, axis=2).reshape(-1)
        cols = np.repeat(loc[:, None, :], 12, axis=1).reshape(-1)
        keep = (rows >= 0) & (cols >= 0)
        pat = sps.coo_matrix((np.ones(int(keep.sum())),
                              (rows[keep], cols[keep])),
                             shape=(self.nv, self.nv)).tocsr()
        pat.sum_duplicates()
        pat.sort_indices()
        pat.data[:] = 1.
        return pat

    def bind_pattern(self, pattern):
        """the CSR pattern `assemble` (and the trapezoidal stepper) fill; it
        must contain `connectivity()`"""
        view = C.CsrView(pattern)
        C.check(self.lib.dns_conv_bind_pattern(self._h, view.byref()))
        self._pattern = view

    def assemble(self, u_inner, newton=False, dbcvals_lin=None,
                 dbcvals_rhs=None):
        """`(N, rhsbc, rhscon)`: `N1(u)` (Picard) or `N1(u) + N2(u)` (Newton)
        condensed, in the bound pattern; `-N[:, bc] bcvals`; `N(u)u`
        (reference `get_v_conv_conts`, snu:109-133).  `dbcvals_lin` /
        `dbcvals_rhs`: Dirichlet values of the linearisation field / the ones
        that go to `rhsbc` (default: the operator's own set)"""
        import scipy.sparse as sps
        u = C.as_f64(u_inner, size=self.nv)
        pv = self._pattern
        nvals = np.empty(pv.data.size)
        rhsbc, rhscon = np.empty(self.nv), np.empty(self.nv)
        if dbcvals_lin is None:
            C.check(self.lib.dns_conv_assemble(
                self._h, C.dptr(u), int(bool(newton)), C.dptr(nvals),
                C.dptr(rhsbc), C.dptr(rhscon)))
        else:
            dl = C.as_f64(dbcvals_lin, size=self.ndbc) if self.ndbc else \
                np.zeros(1)
            dr = None if dbcvals_rhs is None else (
                C.as_f64(dbcvals_rhs, size=self.ndbc) if self.ndbc
                else np.zeros(1))
            C.check(self.lib.dns_conv_assemble2(
                self._h, C.dptr(u), C.dptr(dl), C.dptr(dr), int(bool(newton)),
                C.dptr(nvals), C.dptr(rhsbc), C.dptr(rhscon)))
        N = sps.csr_matrix((nvals, pv.indices, pv.indptr), shape=pv.shape)
        return N, rhsbc.reshape((-1, 1)), rhscon.reshape((-1, 1))

    def host_callback(self, invinds, scale=-1.0):
        """an `f_vdp(vfull)` callable (reference tiu:113) backed by this
        operator: the inner part of `vfull` in, `scale*N(v)v` out"""
        inv = np.asarray(invinds)

        def f_vdp(vfull):
            return self.apply(np.asarray(vfull).reshape(-1)[inv], scale=scale)
        return f_vdp

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.dns_conv_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
