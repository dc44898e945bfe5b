"""Python handles on the device objects of `include/dns_amd.h`."""
import ctypes as ct

import numpy as np
import scipy.sparse as sps

from . import _capi as C

__all__ = ['streaming_precond_defaults', 'choose_schur', 'SaddleSystem', 'ImexStepper', 'spmv', 'dot', 'axpy', 'gemv',
           'dense_inverse', 'spmv_bench', 'spmv_pair', 'solve_opts', 'precond_opts']

_METHODS = {'gmres': C.DNS_METHOD_GMRES, 'bicgstab': C.DNS_METHOD_BICGSTAB}
_SCHUR = {'dense': C.DNS_SCHUR_DENSE, 'jacobi': C.DNS_SCHUR_JACOBI, 'mg': 2}
_VARIANTS = {'vector': C.DNS_SPMV_VECTOR, 'stream': C.DNS_SPMV_STREAM,
             'stream16': 2}
_FHAT = {'cheb': C.DNS_FHAT_CHEB, 'explicit': C.DNS_FHAT_EXPLICIT,
         'auto': C.DNS_FHAT_AUTO}


def solve_opts(method='gmres', restart=60, maxiter=400, reorth=True,
               rtol=1e-10, atol=0., check_every=4, use_graph=False):
    o = C.dns_solve_opts()
    C.load_library().dns_default_solve_opts(ct.byref(o))
    o.method = _METHODS[method.lower()] if isinstance(method, str) else method
    o.restart, o.maxiter = int(restart), int(maxiter)
    # True/1: CGS2, False/0: CGS once, 2: CGS once fused into the head kernel
    o.reorth = 2 if (reorth == 2 and reorth is not True) else (1 if reorth else 0)
    o.rtol, o.atol = float(rtol), float(atol)
    o.check_every = int(check_every)
    o.use_graph = 1 if use_graph else 0
    return o


def precond_opts(cheb_degree=4, schur='dense', fhat='auto', eig_lo=0.,
                 eig_hi=0., eig_lo_safety=0.9, eig_hi_safety=1.05,
                 fp32_store=None, drop_tol=None, factorization='triangular'):
    o = C.dns_precond_opts()
    C.load_library().dns_default_precond_opts(ct.byref(o))
    if fp32_store is not None:
        o.fp32_store = 1 if fp32_store else 0
    if drop_tol is not None:
        o.drop_tol = float(drop_tol)
    o.factorization = {'triangular': 0, 'full': 1}[factorization] \
        if isinstance(factorization, str) else int(factorization)
    o.cheb_degree = int(cheb_degree)
    o.schur = _SCHUR[schur] if isinstance(schur, str) else schur
    o.fhat = _FHAT[fhat] if isinstance(fhat, str) else fhat
    o.eig_lo, o.eig_hi = float(eig_lo), float(eig_hi)
    o.eig_lo_safety, o.eig_hi_safety = float(eig_lo_safety), \
        float(eig_hi_safety)
    return o


class SaddleSystem(object):
    """`[[F, JT], [J, 0]]` resident in HBM with its block preconditioner"""

    def __init__(self, F, J, JT=None, device=0):
        self.lib = C.load_library()
        self._f = C.CsrView(F)
        self._j = C.CsrView(J)
        self._jt = C.CsrView(JT) if JT is not None else None
        self.NP, self.NV = self._j.shape
        self.n = self.NV + self.NP
        self.device = device
        self._h = ct.c_void_p()
        C.check(self.lib.dns_saddle_create(
            device, self._f.byref(), self._j.byref(),
            self._jt.byref() if self._jt is not None else None,
            ct.byref(self._h)))
        self.last_stats = None
        self.precond_ready = False

    @classmethod
    def from_rows(cls, F_rows, JT_rows, J_rows, NV, NP, comm, device=0):
        """rank-local construction (`dns_saddle_create_rows`): this rank's rows
        of `F` and `J^T` -- rows `comm.partition_range(NV)` -- and of `J` --
        rows `partition_range(NP)` -- with global column indices; no rank
        holds a whole matrix.  Collective over `comm`."""
        self = cls.__new__(cls)
        self.lib = C.load_library()
        self._f = C.CsrView(F_rows)
        self._jt = C.CsrView(JT_rows)
        self._j = C.CsrView(J_rows)
        self.NP, self.NV = int(NP), int(NV)
        self.n = self.NV + self.NP
        self.device = device
        self._comm = comm
        self._h = ct.c_void_p()
        C.check(self.lib.dns_saddle_create_rows(
            device, comm._h, self.NV, self.NP, self._f.byref(),
            self._jt.byref(), self._j.byref(), ct.byref(self._h)))
        self.last_stats = None
        self.precond_ready = False
        self.from_rows_handle = True
        return self

    @classmethod
    def from_rows_of(cls, F, J, comm, device=0):
        """`from_rows` with the rows cut out of whole SciPy matrices (tests,
        the bench: a distributed assembler hands its rows over directly)"""
        import scipy.sparse as sps
        from .comm import partition_range
        NP, NV = J.shape
        v0, v1 = partition_range(NV, comm.nranks, comm.rank)
        p0, p1 = partition_range(NP, comm.nranks, comm.rank)
        F, J = sps.csr_matrix(F), sps.csr_matrix(J)
        JT = sps.csr_matrix(J[:, v0:v1].T)
        return cls.from_rows(F[v0:v1, :], JT, J[p0:p1, :], NV, NP, comm,
                             device=device)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.dns_saddle_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_comm(self, comm):
        """attach a `comm.Comm` (row-partitioned solve); call before
        `setup_precond`; `None` detaches"""
        C.check(self.lib.dns_saddle_set_comm(
            self._h, comm._h if comm is not None else None))
        # (only once the library has taken it: a handle created from rows
        # refuses another communicator and keeps the one it has)
        self._comm = comm
        self.precond_ready = False

    def device_matrix_bytes(self):
        """bytes of HBM this rank's matrices occupy (`dns_saddle_device_bytes`)
        -- shrinks with the number of ranks of a row-partitioned handle"""
        out = ct.c_int64(0)
        C.check(self.lib.dns_saddle_device_bytes(self._h, ct.byref(out)))
        return out.value

    def host_matrix_bytes(self):
        """`(kept, setup)` bytes of host matrices (`dns_saddle_host_bytes`):
        what the handle keeps, and what its last partitioned set-up held"""
        a, b = ct.c_int64(0), ct.c_int64(0)
        C.check(self.lib.dns_saddle_host_bytes(self._h, ct.byref(a),
                                               ct.byref(b)))
        return a.value, b.value

    def set_schur_mg(self, prolongations, smooth_steps=2):
        """nested pressure spaces for `schur='mg'`: `prolongations[l]` maps
        level `l+1` (coarser) to level `l`, finest first; the coarsest level
        gets a dense inverse"""
        views = [C.CsrView(p) for p in prolongations]
        arr = (C.dns_csr*max(len(views), 1))(*[v.struct for v in views])
        C.check(self.lib.dns_saddle_set_schur_mg(self._h, len(views), arr,
                                                 int(smooth_steps)))
        self.precond_ready = False

    def update_values(self, fdata):
        fdata = C.as_f64(fdata, size=self._f.data.size)
        C.check(self.lib.dns_saddle_update_values(self._h, C.dptr(fdata)))

    def setup_precond(self, **kw):
        o = precond_opts(**kw)
        C.check(self.lib.dns_saddle_setup_precond(self._h, ct.byref(o)))
        self.precond_ready = True

    def cheb_bounds(self):
        lo, hi = ct.c_double(), ct.c_double()
        C.check(self.lib.dns_saddle_cheb_bounds(self._h, ct.byref(lo),
                                                ct.byref(hi)))
        return lo.value, hi.value

    def set_option(self, name, value):
        """a tuning knob of this handle (`dns_saddle_set_option`: `stream_nnz`,
        `pair`, `mg_dense_max`, `mg_dense_half_max`, `mg_part_min`,
        `mg_fused`, `mg_cheb`, `mg_cheb_alpha`, `mg_cycles`, `mg_rho`, `dist_graph`, `oversolve`,
        `oversolve_cmin`); before
        `setup_precond`"""
        C.check(self.lib.dns_saddle_set_option(self._h, name.encode(),
                                               float(value)))
        return self

    def precond_info(self):
        """sizes of the resident preconditioner (bench.py's byte counts)"""
        buf = (ct.c_int64*64)()
        cnt = ct.c_int32(0)
        C.check(self.lib.dns_saddle_precond_info(self._h, 64, buf,
                                                 ct.byref(cnt)))
        v = list(buf[:cnt.value])
        out = dict(nnz_K=v[0], nnz_Gc=v[1], nnz_JG=v[2], NP=v[3],
                   fp32_store=bool(v[4]), cheb_degree=v[5],
                   schur={0: 'dense', 1: 'jacobi', 2: 'mg'}[v[6]],
                   NV=self.NV, nnz_F=int(self._f.data.size),
                   nnz_J=int(self._j.data.size), mg_nu=v[8], mg_levels=[])
        for l in range(v[7]):
            n, nnzs, nnzp = v[9 + 3*l:12 + 3*l]
            out['mg_levels'].append(dict(n=n, nnz_S=nnzs, nnz_P=nnzp))
        tail = 9 + 3*v[7]
        out['pair_format_bytes'] = v[tail] if len(v) > tail else 0
        out['mg_coarse_val_bytes'] = v[tail + 1] if len(v) > tail + 1 else 8
        out['mg_cycles'] = max(1, v[tail + 2]) if len(v) > tail + 2 else 1
        out['mg_two_cycle_maxc'] = v[tail + 3] if len(v) > tail + 3 else 0
        return out

    def solve(self, rhsv, rhsp=None, x0=None, raise_on_fail=True, **kw):
        """returns `[v; p~]` as a 1-D array of length NV+NP"""
        o = kw.pop('opts', None)
        o = solve_opts(**kw) if o is None else o
        rv = C.as_f64(rhsv, size=self.NV)
        rp = None if rhsp is None else C.as_f64(rhsp, size=self.NP)
        xi = None if x0 is None else C.as_f64(x0, size=self.n)
        out = np.empty(self.n)
        st = C.dns_solve_stats()
        C.check(self.lib.dns_saddle_solve(self._h, C.dptr(rv), C.dptr(rp),
                                          C.dptr(xi), C.dptr(out),
                                          ct.byref(o), ct.byref(st)))
        self.last_stats = st.asdict()
        if raise_on_fail and st.status != C.DNS_OK:
            cls = C.NotConverged if st.status == C.DNS_NOT_CONVERGED \
                else C.Breakdown
            raise cls(st.status, 'Krylov solve stopped after {0} iterations '
                      'at relative residual {1:.3e}'.format(st.iters,
                                                            st.est_relres))
        return out

    PROBES = ('head', 'fhat', 'kapply_dots', 'orth', 'resid_norm', 'tail',
              'combine', 'precond', 'arnoldi_step_seq')

    def probe(self, which, chain=64, reps=20):
        """microseconds per launch of one cycle kernel in a replayed graph"""
        idx = self.PROBES.index(which) if isinstance(which, str) else which
        out = ct.c_double(0.)
        C.check(self.lib.dns_saddle_probe(self._h, idx, chain, reps,
                                          ct.byref(out)))
        return out.value

    def solve_multi(self, rhsv, rhsp=None, x0=None, raise_on_fail=True, **kw):
        """`k` right-hand sides in one call (`dns_saddle_solve_multi`): `rhsv`
        `(NV, k)`, `rhsp` `(NP, k)` or None, `x0` `(NV+NP, 1 or k)` or None;
        returns the `(NV+NP, k)` solutions; `last_stats_cols` holds one
        record per column, `last_stats` the last column's"""
        o = kw.pop('opts', None)
        o = solve_opts(**kw) if o is None else o
        rv = np.asarray(rhsv, dtype=np.float64).reshape((self.NV, -1))
        k = rv.shape[1]
        rvf = np.ascontiguousarray(rv.T).reshape(-1)          # column by column
        rpf = None
        if rhsp is not None:
            rpf = np.ascontiguousarray(np.asarray(
                rhsp, dtype=np.float64).reshape((self.NP, k)).T).reshape(-1)
        x0f, x0c = None, 0
        if x0 is not None:
            xa = np.asarray(x0, dtype=np.float64).reshape((self.n, -1))
            x0c = xa.shape[1]
            if x0c not in (1, k):
                raise ValueError('`x0` must have 1 or {0} columns'.format(k))
            x0f = np.ascontiguousarray(xa.T).reshape(-1)
        out = np.empty(k*self.n)
        sts = (C.dns_solve_stats*k)()
        C.check(self.lib.dns_saddle_solve_multi(
            self._h, k, C.dptr(rvf), C.dptr(rpf), C.dptr(x0f), x0c,
            C.dptr(out), ct.byref(o), sts))
        self.last_stats_cols = [st.asdict() for st in sts]
        self.last_stats = self.last_stats_cols[-1]
        if raise_on_fail:
            for c, st in enumerate(sts):
                if st.status != C.DNS_OK:
                    cls = C.NotConverged if st.status == C.DNS_NOT_CONVERGED \
                        else C.Breakdown
                    raise cls(st.status, 'column {0}: Krylov solve stopped '
                              'after {1} iterations at relative residual '
                              '{2:.3e}'.format(c, st.iters, st.est_relres))
        return out.reshape((k, self.n)).T.copy()

    def residual_history_col(self, col):
        cnt = ct.c_int32(0)
        C.check(self.lib.dns_saddle_residual_history_col(
            self._h, int(col), None, 0, ct.byref(cnt)))
        out = np.zeros(max(cnt.value, 1))
        C.check(self.lib.dns_saddle_residual_history_col(
            self._h, int(col), C.dptr(out), cnt.value, ct.byref(cnt)))
        return out[:cnt.value]

    def residual_history(self):
        cnt = ct.c_int32(0)
        C.check(self.lib.dns_saddle_residual_history(self._h, None, 0,
                                                     ct.byref(cnt)))
        out = np.zeros(max(cnt.value, 1))
        C.check(self.lib.dns_saddle_residual_history(
            self._h, C.dptr(out), cnt.value, ct.byref(cnt)))
        return out[:cnt.value]

    def apply(self, x):
        x = C.as_f64(x, size=self.n)
        y = np.empty(self.n)
        C.check(self.lib.dns_saddle_apply(self._h, C.dptr(x), C.dptr(y)))
        return y

    def apply_precond(self, r):
        r = C.as_f64(r, size=self.n)
        z = np.empty(self.n)
        C.check(self.lib.dns_saddle_apply_precond(self._h, C.dptr(r),
                                                  C.dptr(z)))
        return z


class ImexStepper(object):
    """device-resident CNAB/SBDF2 state (`dns_imex_*`)"""

    def __init__(self, system, R1, rows=None):
        """`rows=True` (default on a system created from rows): `R1` holds
        this rank's rows only, or is cut down to them here when it is whole
        (`dns_imex_create_rows`)"""
        self.sys = system
        self.lib = system.lib
        if rows is None:
            rows = getattr(system, 'from_rows_handle', False)
        self._h = ct.c_void_p()
        if rows:
            from .comm import partition_range
            cm = system._comm
            v0, v1 = partition_range(system.NV, cm.nranks, cm.rank)
            if R1.shape[0] == system.NV and (v1 - v0) != system.NV:
                import scipy.sparse as sps
                R1 = sps.csr_matrix(R1)[v0:v1, :]
            self._r1 = C.CsrView(R1)
            C.check(self.lib.dns_imex_create_rows(
                system._h, self._r1.byref(), ct.byref(self._h)))
        else:
            self._r1 = C.CsrView(R1)
            C.check(self.lib.dns_imex_create(system._h, self._r1.byref(),
                                             ct.byref(self._h)))
        self.last_stats = None
        # time steps and Krylov steps since the stepper exists
        self.total_steps, self.total_iters = 0, 0

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.dns_imex_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, v_c, v_p=None, ptilde_c=None, nfc_c=None, nfc_o=None):
        NV, NP = self.sys.NV, self.sys.NP
        args = [C.as_f64(v_c, NV),
                None if v_p is None else C.as_f64(v_p, NV),
                None if ptilde_c is None else C.as_f64(ptilde_c, NP),
                None if nfc_c is None else C.as_f64(nfc_c, NV),
                None if nfc_o is None else C.as_f64(nfc_o, NV)]
        C.check(self.lib.dns_imex_set_state(self._h, *[C.dptr(a)
                                                       for a in args]))

    def set_convection(self, conv, scale=-1.0):
        """attach a `convection.ConvectionP2`: every step then evaluates
        `nfc_c = scale*N(v_c)v_c` on the device (`None` detaches)"""
        self._conv = conv
        C.check(self.lib.dns_imex_set_convection(
            self._h, conv._h if conv is not None else None, float(scale)))

    def set_rhs(self, gvec=None, rhsp=None):
        g = None if gvec is None else C.as_f64(gvec, self.sys.NV)
        gp = None if rhsp is None else C.as_f64(rhsp, self.sys.NP)
        C.check(self.lib.dns_imex_set_rhs(self._h, C.dptr(g), C.dptr(gp)))

    def set_rhs_table(self, gv=None, gp=None):
        """per-step right-hand sides known in advance: row s of `gv`
        `(nsteps, NV)` / `gp` `(nsteps, NP)` is used by the s-th step from now
        on (`dns_imex_set_rhs_table`); `set_rhs` returns to constant vectors"""
        NV, NP = self.sys.NV, self.sys.NP
        tv = None if gv is None else \
            np.ascontiguousarray(gv, dtype=np.float64).reshape((-1, NV))
        tp = None if gp is None else \
            np.ascontiguousarray(gp, dtype=np.float64).reshape((-1, NP))
        nsteps = (tv if tv is not None else tp).shape[0]
        if tv is not None and tp is not None and tp.shape[0] != nsteps:
            raise ValueError('tables of different length')
        C.check(self.lib.dns_imex_set_rhs_table(
            self._h, int(nsteps), C.dptr(None if tv is None else tv.reshape(-1)),
            C.dptr(None if tp is None else tp.reshape(-1))))

    def table_position(self):
        pos, left = ct.c_int32(0), ct.c_int32(0)
        C.check(self.lib.dns_imex_table_position(self._h, ct.byref(pos),
                                                 ct.byref(left)))
        return pos.value, left.value

    @staticmethod
    def coeffs(a_c=1., a_p=0., cn_c=0., cn_o=0., pscale=1., extrapolate=True,
               carry_residual=True):
        """`carry_residual`: the velocity residual of a step's inexact solve
        goes into the next step's right-hand side (include/dns_amd.h): the
        distance to the direct-solve trajectory stays at the size of ONE
        solve's error instead of adding up step after step"""
        return C.dns_imex_coeffs(a_c=a_c, a_p=a_p, cn_c=cn_c, cn_o=cn_o,
                                 pscale=pscale,
                                 extrapolate_x0=int(extrapolate),
                                 carry_residual=int(bool(carry_residual)))

    def step(self, cf, nfc_new=None, opts=None, raise_on_fail=True):
        o = solve_opts() if opts is None else opts
        nf = None if nfc_new is None else C.as_f64(nfc_new, self.sys.NV)
        st = C.dns_solve_stats()
        C.check(self.lib.dns_imex_step(self._h, C.dptr(nf), ct.byref(cf),
                                       ct.byref(o), ct.byref(st)))
        self.last_stats = st.asdict()
        self.total_steps += 1
        self.total_iters += int(st.iters)
        if raise_on_fail and st.status != C.DNS_OK:
            raise C.NotConverged(st.status, 'time step solve failed: '
                                 '{0}'.format(self.last_stats))
        return self.last_stats

    def run(self, nsteps, cf, opts=None, raise_on_fail=True):
        """`nsteps` steps without host callbacks (device convection if
        attached, else the convection history stays frozen); returns
        `(device_seconds, total_iters, last_stats)`.  `last_run` then holds the
        record of the call: time steps whose solve ended at `maxiter`
        (`unconverged`, `first_bad`), steps repeated after a mispredicted
        batch (`replayed`) and graphs captured inside the call (`captures`).
        Raises `NotConverged` if any step did not converge, like `step()`."""
        o = solve_opts() if opts is None else opts
        st = C.dns_solve_stats()
        secs = ct.c_double(0.)
        its = ct.c_int64(0)
        status = self.lib.dns_imex_run(self._h, int(nsteps), ct.byref(cf),
                                       ct.byref(o), ct.byref(st),
                                       ct.byref(secs), ct.byref(its))
        if status != C.DNS_NOT_CONVERGED:
            C.check(status)
        self.last_stats = st.asdict()
        self.total_steps += int(nsteps)
        self.total_iters += int(its.value)
        vals = [ct.c_int32(0) for _ in range(4)]
        C.check(self.lib.dns_imex_run_info(self._h,
                                           *[ct.byref(v) for v in vals]))
        self.last_run = dict(zip(('unconverged', 'first_bad', 'replayed',
                                  'captures'), [v.value for v in vals]))
        if status == C.DNS_NOT_CONVERGED and raise_on_fail:
            C.check(status)
        return secs.value, its.value, self.last_stats

    def step_counters(self):
        """`(built, tail_cells, cells_reused)` of the row-partitioned step
        (`dns_imex_step_counters`)"""
        buf = (ct.c_int64*3)()
        C.check(self.lib.dns_imex_step_counters(self._h, buf))
        return tuple(int(b) for b in buf)

    def get_state(self):
        v = np.empty(self.sys.NV)
        p = np.empty(self.sys.NP)
        C.check(self.lib.dns_imex_get_state(self._h, C.dptr(v), C.dptr(p)))
        return v.reshape((-1, 1)), p.reshape((-1, 1))

    def vnorm(self):
        out = ct.c_double(0.)
        C.check(self.lib.dns_imex_vnorm(self._h, ct.byref(out)))
        return out.value


def choose_schur(system, F, J, schur='auto', prolongations=None,
                 dense_max=6000):
    """the Schur block of a drop-in's system: `'auto'` = the dense inverse up
    to `dense_max` pressure dofs, beyond it the multigrid block -- on the
    caller's nested pressure spaces (`prolongations`) or, for a mesh that
    comes with none, on a hierarchy built from `F` and `J` alone
    (`amg.algebraic_prolongations`; `'amg'` asks for it by name).  Returns
    the `schur=` keyword for `setup_precond` (the prolongations are attached
    here); what was built is left in `system.schur_hierarchy`."""
    NP = J.shape[0]
    system.schur_hierarchy = None
    if prolongations is not None and schur in ('auto', 'mg'):
        system.set_schur_mg(prolongations)
        system.schur_hierarchy = dict(kind='geometric',
                                      levels=[NP] + [P.shape[1]
                                                     for P in prolongations])
        return 'mg'
    if schur == 'auto' and NP <= dense_max:
        return 'dense'
    if schur in ('auto', 'amg'):
        from . import amg
        info = dict(kind='algebraic')
        prols = amg.algebraic_prolongations(F, J, info=info,
                                            coarsest=min(1500, dense_max))
        if not prols:
            return 'dense' if NP <= 4*dense_max else 'jacobi'
        system.set_schur_mg(prols)
        system.schur_hierarchy = info
        return 'mg'
    return schur


def streaming_precond_defaults(n):
    """degree and drop tolerance of the explicit polynomial `Fh^-1` for a
    system of `n` unknowns.  Where the step is bandwidth bound the polynomial
    matrix `Gc` is the largest stream of a Krylov step (3.6 x the non-zeros of
    `K` at degree 8, drop 1e-3): a HIGHER degree with a LARGER drop tolerance
    keeps the Krylov-step count and cuts the stream -- measured on the refined
    wake (profiles/r04_gc_pareto/table.txt): n = 173k 3651 -> 4039 steps/s,
    n = 693k 1301 -> 1595, n = 2.78M 308 -> 431.  At the reference sizes (cache
    resident, ONE Krylov step per time step) degree 6 / 1e-3 stays.

    `extrapolate`: order of the warm start.  With the multigrid Schur block a
    solve runs the two columns of its cycle whatever its residual (oversolve,
    `solver.hpp`); what the next start residual then consists of is the final
    residuals of the last solves times the warm start's coefficients, and the
    cubic's (sum |c| = 15) keep a two-column cycle contracting where the
    quartic's (31) do not: n = 173k 4175 -> 4490 steps/s, n = 693k 1434 ->
    1870, n = 2.78M 465 -> 604 (profiles/r05_oversolve/)."""
    if n >= 100000:
        return dict(cheb_degree=8, drop_tol=7e-3, extrapolate=3)
    return dict(cheb_degree=6, drop_tol=1e-3, extrapolate=4)


# ---- standalone kernels ---------------------------------------------------
def spmv(A, x, y=None, alpha=1., beta=0., variant='vector', device=0):
    lib = C.load_library()
    view = C.CsrView(A)
    x = C.as_f64(x, size=view.shape[1])
    out = np.zeros(view.shape[0]) if y is None else C.as_f64(y).copy()
    C.check(lib.dns_spmv(device, view.byref(), C.dptr(x), C.dptr(out),
                         float(alpha), float(beta),
                         _VARIANTS.get(variant, variant)))
    return out


def spmv_pair(K, nv, x, reps=0, warmup=1, device=0):
    """`y = K x` through the pair format (`dns_spmv_pair`); with `reps > 0`
    returns `(y, seconds_per_launch, format_bytes)`"""
    lib = C.load_library()
    view = C.CsrView(K)
    x = C.as_f64(x, size=view.shape[1])
    out = np.zeros(view.shape[0])
    secs, fb = ct.c_double(0.), ct.c_int64(0)
    C.check(lib.dns_spmv_pair(device, view.byref(), int(nv), C.dptr(x),
                              C.dptr(out), int(reps), int(warmup),
                              ct.byref(secs), ct.byref(fb)))
    if reps > 0:
        return out, secs.value, fb.value
    return out


def dot(x, y, device=0):
    x, y = C.as_f64(x), C.as_f64(y)
    out = ct.c_double(0.)
    C.check(C.load_library().dns_dot(device, x.size, C.dptr(x), C.dptr(y),
                                     ct.byref(out)))
    return out.value


def axpy(a, x, y, device=0):
    x = C.as_f64(x)
    out = C.as_f64(y).copy()
    C.check(C.load_library().dns_axpy(device, x.size, float(a), C.dptr(x),
                                      C.dptr(out)))
    return out


def gemv(A, x, alpha=1., device=0):
    A = np.ascontiguousarray(A, dtype=np.float64)
    x = C.as_f64(x, size=A.shape[0])
    y = np.empty(A.shape[0])
    C.check(C.load_library().dns_gemv(device, A.shape[0], C.dptr(A),
                                      C.dptr(x), C.dptr(y), float(alpha)))
    return y


def dense_inverse(A, device=0):
    out = np.array(A, dtype=np.float64, order='C', copy=True)
    C.check(C.load_library().dns_dense_inverse(device, out.shape[0],
                                               C.dptr(out)))
    return out


def spmv_bench(A, variant='stream', reps=50, warmup=5, device=0):
    """average seconds per `y = A x` on resident data, and `||y||^2`"""
    view = C.CsrView(sps.csr_matrix(A))
    secs, chk = ct.c_double(0.), ct.c_double(0.)
    vid = _VARIANTS[variant] if isinstance(variant, str) else int(variant)
    C.check(C.load_library().dns_spmv_bench(
        device, view.byref(), vid, int(reps), int(warmup),
        ct.byref(secs), ct.byref(chk)))
    return secs.value, chk.value


def hbm_probe(nbytes=2 << 30, kind='read', reps=20, device=0):
    """attainable HBM bandwidth [GB/s] of plain streaming kernels over
    `nbytes` (`kind`: 'read' | 'copy' | 'triad')"""
    out = ct.c_double(0.)
    C.check(C.load_library().dns_hbm_probe(
        int(device), int(nbytes),
        {'read': 0, 'copy': 1, 'triad': 2, 'read8a': 3, 'read8b': 4,
         'read8c': 5, 'read_b64': 6, 'read_tiles8': 7,
         'read_tiles1': 8}[kind],
        int(reps), ct.byref(out)))
    return out.value
