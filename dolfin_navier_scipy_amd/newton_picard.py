"""Newton/Picard trapezoidal time sweeps, device resident (`dns_trap_*`).

What the reference does in `solve_nse(..., treat_nonl_explicit=False)`
(stokes_navier_utils.py:1304-1334, 1402-1587) with FEniCS assemblies, sparse
matrix sums and one saddle-point solve per time step:

 * `get_v_conv_conts` (snu:109-133): `N1(v_lin)` (Picard) or `N1 + N2` (Newton)
   condensed to the inner dofs, the Dirichlet-column rhs and `N(v)v`
   -> `ConvectionP2.bind_pattern` + the element kernels of `csrc/convection.hpp`
 * `_get_mats_rhs_ts` (snu:1016-1047): `M + dt/2 C_n`, `M v + dt/2 (f_n + f_c -
   C_c v)` -> formed on the device in the ONE resident system (same pattern,
   new values every step)
 * the per-step `.npy` linearisation points (snu:1012-1014, 1424-1431)
   -> two trajectory buffers in HBM

Host code only drives the loop (`TrapezoidalStepper.sweep`, `newton_picard`).
"""
import ctypes as ct
import os
import sys

import numpy as np
import scipy.sparse as sps

from . import _capi as C
from .saddle import SaddleSystem, solve_opts, choose_schur

__all__ = ['TrapezoidalStepper', 'newton_picard', 'union_pattern',
           'values_in_pattern']


def union_pattern(*mats):
    """CSR matrix of ones on the union of the patterns (explicit zeros of the
    inputs count), canonical format"""
    acc = None
    for m in mats:
        m = sps.csr_matrix(m)
        one = sps.csr_matrix((np.ones(m.indices.size), m.indices, m.indptr),
                             shape=m.shape)
        acc = one if acc is None else acc + one
    acc = sps.csr_matrix(acc)
    acc.sum_duplicates()
    acc.sort_indices()
    acc.data[:] = 1.
    return acc


def values_in_pattern(mat, pattern):
    """the values of `mat` laid out in `pattern` (a superset of its pattern);
    entries of the pattern `mat` lacks are explicit zeros"""
    mat = sps.csr_matrix(mat)
    mat.sum_duplicates()
    ncols = pattern.shape[1]
    prow = np.repeat(np.arange(pattern.shape[0]), np.diff(pattern.indptr))
    pkeys = prow.astype(np.int64)*ncols + pattern.indices
    mrow = np.repeat(np.arange(mat.shape[0]), np.diff(mat.indptr))
    mkeys = mrow.astype(np.int64)*ncols + mat.indices
    pos = np.searchsorted(pkeys, mkeys)
    if pos.size and (pos.max() >= pkeys.size or
                     not np.array_equal(pkeys[pos], mkeys)):
        raise ValueError('matrix has entries outside the pattern')
    out = np.zeros(pkeys.size)
    out[pos] = mat.data
    return out


class TrapezoidalStepper(object):
    """`M, A, J` condensed (inner dofs), `conv` a `ConvectionP2` of the same
    space; `dt` is the step size the preconditioner is set up for (other
    step sizes still converge, with more iterations)"""

    def __init__(self, M, A, J, conv, nslots, dt, device=0, precond=None,
                 JT=None, comm=None, precond_linpoint=None, refresh_iters=3.0,
                 batch=64, oversolve=1e-3):
        """`comm` (a `comm.Comm`): the saddle solves of the sweeps run
        row-partitioned over its ranks, and so does the
        assembly: a rank evaluates the cells that touch its rows and forms
        its rows of `N(v_lin)`, `F` and the right-hand side; no solution is
        gathered per step.  `read_traj`, the trajectory exports, `state()`
        and `update_norm()` are collective then (all ranks call them).
        `precond_linpoint` (inner velocity, NV): the preconditioner -- set up
        ONCE, the system matrix is re-valued every step -- is built for
        `M + dt/2 (A + N1(v))` at this velocity instead of `M + dt/2 A`: the
        Oseen term of a representative state is then inside the polynomial
        and the Schur block (fewer Krylov steps per time step).
        `refresh_iters`: bound on the Krylov steps per time step of a batch of
        `batch` steps beyond which `sweep` rebuilds the preconditioner about
        the current operator (`None` / 0: never, the set-up of the start is
        kept).  `oversolve`: the solves of a pipelined batch run the columns
        of their cycle down to this fraction of the tolerance instead of
        stopping at it (`set_oversolve`; 0: stop at the tolerance, with a
        slack column per cycle as until round 4)"""
        self.lib = C.load_library()
        self.refresh_iters = refresh_iters
        self.batch = int(batch)
        self.refreshes = 0
        self._level, self._tried = None, False    # (state of the policy)
        self.conv = conv
        self.M, self.A, self.J = (sps.csr_matrix(M), sps.csr_matrix(A),
                                  sps.csr_matrix(J))
        self.NP, self.NV = self.J.shape
        self.pattern = union_pattern(self.M, self.A, conv.connectivity())
        self.mvals = values_in_pattern(self.M, self.pattern)
        self.avals = values_in_pattern(self.A, self.pattern)
        conv.bind_pattern(self.pattern)
        f0 = self.mvals + .5*dt*self.avals
        if precond_linpoint is not None:
            Nref, _, _ = conv.assemble(precond_linpoint, newton=False)
            nd = Nref.data if Nref.data.size == f0.size else \
                values_in_pattern(Nref, self.pattern)
            f0 = f0 + .5*dt*nd
        F0 = sps.csr_matrix((f0, self.pattern.indices, self.pattern.indptr),
                            shape=self.pattern.shape)
        self.system = SaddleSystem(F0, self.J, JT=JT, device=device)
        if comm is not None:
            self.system.set_comm(comm)
        pkw = dict(cheb_degree=6, schur='auto', fhat='auto', fp32_store=True,
                   drop_tol=3e-3)
        if comm is not None:
            pkw['fhat'] = 'explicit'       # (the partitioned solve needs it)
        pkw.update(precond or {})
        pkw['schur'] = choose_schur(self.system, F0, self.J,
                                    schur=pkw['schur'],
                                    prolongations=pkw.pop('prolongations',
                                                          None))
        self._pkw = dict(pkw)
        self.system.setup_precond(**pkw)
        conv.bind_pattern(self.pattern)
        self.nslots = int(nslots)
        self._h = ct.c_void_p()
        C.check(self.lib.dns_trap_create(
            self.system._h, conv._h, C.dptr(self.mvals), C.dptr(self.avals),
            self.nslots, ct.byref(self._h)))
        self.last_stats = None
        self._over = False
        self.set_oversolve(oversolve or 0.0)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.dns_trap_destroy(self._h)
            self._h = ct.c_void_p()
            self.system.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_rhs(self, fv=None, fp=None):
        fv = None if fv is None else C.as_f64(fv, self.NV)
        fp = None if fp is None else C.as_f64(fp, self.NP)
        C.check(self.lib.dns_trap_set_rhs(self._h, C.dptr(fv), C.dptr(fp)))

    def set_tables(self, fv_tab=None, fp_tab=None, mbc_tab=None,
                   dbc_tab=None):
        """time-dependent data of the sweeps, one row per trajectory slot
        (`nslots` rows): `fv_tab (nslots, NV)` replaces `fv` (forcing `fvtd(t)`
        plus the stiffness terms of controlled boundary values, snu:1466),
        `fp_tab (nslots, NP)`, `mbc_tab (nslots, NV)` = `M[:, cnt] bcvals(t)`
        (snu:1438-1441: `rhs += mbcs_n - mbcs_c`), `dbc_tab (nslots, ndbc)` the
        convection operator's Dirichlet values per time instance"""
        def _tab(arr, width):
            if arr is None:
                return None
            out = np.ascontiguousarray(arr, dtype=np.float64)
            if out.shape != (self.nslots, width):
                raise ValueError('table must be {0} x {1}'.format(
                    self.nslots, width))
            return out.reshape(-1)
        tabs = [_tab(fv_tab, self.NV), _tab(fp_tab, self.NP),
                _tab(mbc_tab, self.NV)]
        C.check(self.lib.dns_trap_set_tables(self._h,
                                             *[C.dptr(t) for t in tabs]))
        if dbc_tab is not None:
            self.conv.set_dbc_table(dbc_tab)

    def write_linpoint(self, which, slot, v):
        v = C.as_f64(v, self.NV)
        C.check(self.lib.dns_trap_traj_write(self._h, which, slot, C.dptr(v)))

    def read_traj(self, which, slot):
        out = np.empty(self.NV)
        C.check(self.lib.dns_trap_traj_read(self._h, which, slot, C.dptr(out)))
        return out.reshape((-1, 1))

    def export_async(self, which, slot0=0, count=None):
        """queue the copy of trajectory slots to the host on a copy stream of
        its own and return the (not yet filled) `count x NV` array; the next
        sweep may be started at once.  `export_wait()` before reading it."""
        count = self.nslots - slot0 if count is None else count
        out = np.empty((count, self.NV))
        C.check(self.lib.dns_trap_traj_export_async(self._h, which, slot0,
                                                    count, C.dptr(out)))
        self._exports = getattr(self, '_exports', []) + [out]   # keep alive
        return out

    def export_wait(self):
        C.check(self.lib.dns_trap_traj_export_wait(self._h))
        self._exports = []

    def save_trajectory_async(self, which, times, namer, slot0=0):
        """the asynchronous writer of the trajectory store: what the
        reference does with a blocking `dou.save_npa` per time step
        (snu:1012-1014, 1424-1431).  Queues the export of the slots of `times`
        and hands the waiting + `np.save(namer(t), v_t)` to a background
        thread; returns the thread (`join()` it before the files are used)."""
        import threading
        times = list(times)
        arr = self.export_async(which, slot0, len(times))

        def work():
            C.check(self.lib.dns_trap_traj_export_wait(self._h))
            for k, t in enumerate(times):
                np.save(namer(t), arr[k].reshape((-1, 1)))
        th = threading.Thread(target=work, name='dns-traj-writer')
        th.start()
        return th

    def start(self, iniv, newton):
        v = C.as_f64(iniv, self.NV)
        C.check(self.lib.dns_trap_start(self._h, C.dptr(v), int(bool(newton))))

    def step(self, dt, lin_which, lin_slot, out_slot, newton, opts=None,
             extrapolate=4, raise_on_fail=True, feedback=None):
        """`feedback=(umat (NV, r), vmat_c (r, NV) or None, vmat_n (r, NV))`:
        the low-rank terms of `_get_mats_rhs_ts` (snu:1036-1042) -- system
        `F - dt/2 U V_n` by Sherman-Morrison-Woodbury, rhs `+ dt/2 U V_c v_c`"""
        o = solve_opts() if opts is None else opts
        st = C.dns_solve_stats()
        if feedback is None:
            C.check(self.lib.dns_trap_step(
                self._h, float(dt), int(lin_which), int(lin_slot),
                int(out_slot), int(bool(newton)), int(extrapolate),
                ct.byref(o), ct.byref(st)))
        else:
            umat, vmat_c, vmat_n = feedback
            umat = np.asarray(umat, dtype=np.float64).reshape((self.NV, -1))
            r = umat.shape[1]
            uf = np.ascontiguousarray(umat.T).reshape(-1)     # column major
            vn = np.ascontiguousarray(vmat_n, dtype=np.float64).reshape(
                (r, self.NV)).reshape(-1)
            vc = None if vmat_c is None else np.ascontiguousarray(
                vmat_c, dtype=np.float64).reshape((r, self.NV)).reshape(-1)
            C.check(self.lib.dns_trap_step_fb(
                self._h, float(dt), int(lin_which), int(lin_slot),
                int(out_slot), int(bool(newton)), int(extrapolate),
                ct.byref(o), ct.byref(st), int(r), C.dptr(uf), C.dptr(vc),
                C.dptr(vn)))
        self.last_stats = st.asdict()
        if raise_on_fail and st.status != C.DNS_OK:
            raise C.NotConverged(st.status, 'time step solve failed: '
                                 '{0}'.format(self.last_stats))
        return self.last_stats

    def run(self, dt, lin_which, slot0, count, newton, opts=None,
            extrapolate=4):
        """`count` pipelined steps in a row (`dns_trap_run`): step `k` is
        linearised about slot `slot0 + k` and stores its velocity there --
        the time loop of a sweep without a trip through Python per step"""
        o = solve_opts() if opts is None else opts
        C.check(self.lib.dns_trap_run(
            self._h, float(dt), int(lin_which), int(slot0), int(count),
            int(bool(newton)), int(extrapolate), ct.byref(o)))

    def state(self):
        v, p = np.empty(self.NV), np.empty(self.NP)
        C.check(self.lib.dns_trap_get_state(self._h, C.dptr(v), C.dptr(p)))
        return v.reshape((-1, 1)), p.reshape((-1, 1))

    def set_pipeline(self, cycle_len):
        """`cycle_len > 0`: `step` only enqueues (no host synchronisation, no
        stats); `poll()` collects the batch's counters"""
        C.check(self.lib.dns_trap_set_pipeline(self._h, int(cycle_len)))

    def poll(self):
        """counters of the pipelined batch since `set_pipeline` / the last
        poll (`dns_trap_poll_ext`): solves, failures, Krylov steps, the
        longest solve, the most columns a solve NEEDED to meet the tolerance,
        the batch maxima of final residual / tolerance and of the residual in
        front of the last column / tolerance"""
        iv, dv = (ct.c_int32*6)(), (ct.c_double*2)()
        C.check(self.lib.dns_trap_poll_ext(self._h, iv, dv))
        return dict(solves=iv[0], fails=iv[1], iters=iv[2], maxit=iv[3],
                    maxneed=iv[4], sumneed=iv[5], maxrel=dv[0],
                    maxprev=dv[1])

    def set_oversolve(self, stop_frac):
        """`stop_frac > 0`: the solves of a pipelined batch run the columns of
        their cycle (down to `stop_frac` x tolerance) instead of stopping at
        the tolerance -- `dns_trap_set_oversolve`; 0 switches it off"""
        C.check(self.lib.dns_trap_set_oversolve(self._h, float(stop_frac)))
        self._over = float(stop_frac) > 0.0

    def update_norm(self):
        out = ct.c_double(0.)
        C.check(self.lib.dns_trap_update_norm(self._h, ct.byref(out)))
        return out.value

    def checkpoint(self):
        C.check(self.lib.dns_trap_checkpoint(self._h))

    def restore(self, newton):
        C.check(self.lib.dns_trap_restore(self._h, int(bool(newton))))

    def refresh_precond(self):
        """set the preconditioner up again for the system matrix as it stands
        on the device -- `M + dt/2 (A + N(v_lin))` of the step that ran last.
        The reference factorises the current operator in every step
        (snu:1484-1512) and therefore never solves with a stale one; here the
        polynomial and the Schur block are rebuilt when the policy of `sweep`
        asks for it (0.07 s at N=2, the price of a few hundred time steps)"""
        self.system.setup_precond(**self._pkw)
        self.refreshes += 1

    def sweep(self, trange, iniv, lin_which, picard, opts=None, extrapolate=4,
              record=True, pipeline=True, batch=None, feedback=None):
        """one sweep over `trange` linearised about trajectory `lin_which`
        (slot k <-> trange[k]); the new velocities go to the other trajectory.
        Returns `(vdict, pdict, norm_nwtnupd, stats)` (dicts empty unless
        `record`).

        Uniform time grids run PIPELINED in batches of `batch` steps
        (`self.batch`, 64): nobody waits for a step, the device counts Krylov
        steps and failures, the host looks once per batch.  A batch in which a
        step did not converge within the agreed cycle length is repeated from
        its checkpoint with synchronous steps.  REFRESH POLICY
        (`refresh_iters`, constructor): when the Krylov steps per time step of
        a batch exceed the bound AND have risen by a fifth over what this
        preconditioner gave in the first batch behind its set-up, it is rebuilt
        about the current operator before the next batch (a first batch above
        the bound gets one rebuild straight away: a set-up made for another
        state) -- the system matrix follows the flow every step, the
        preconditioner follows it when it pays.

        `feedback`: callable `t -> (umat (NV, r), vmat (r, NV))`, the low-rank
        terms of the closed loop per time instance (snu:1367-1384, 1461-1483;
        `umat` the same at every instance -- the reference's `b_mat`); such
        sweeps run step by step (every Woodbury column is a solve of its
        own)."""
        trange = np.asarray(trange, dtype=np.float64)
        if trange.size > self.nslots:
            raise ValueError('trajectory buffers hold {0} slots'.format(
                self.nslots))
        newton = not picard
        steps = np.diff(trange)
        uniform = steps.size > 7 and \
            np.abs(steps - steps[0]).max() <= 1e-12*abs(steps[0])
        batch = self.batch if batch is None else int(batch)
        nt = trange.size
        self.start(iniv, newton)
        self.write_linpoint(1 - lin_which, 0, iniv)
        vdict, pdict = {}, {}
        if record:
            vdict[trange[0]] = np.asarray(iniv, dtype=float).reshape((-1, 1))
        tot = dict(iters=0, device_seconds=0., refreshes=0, replayed_batches=0,
                   batches=[])

        def lowrank(k):
            if feedback is None:
                return None
            umat_c, vmat_c = feedback(trange[k-1])
            umat_n, vmat_n = feedback(trange[k])
            if umat_c is not umat_n and not np.array_equal(umat_c, umat_n):
                raise NotImplementedError('`umat` must not depend on the time '
                                          '(one input matrix, snu:1476-1478)')
            return umat_n, vmat_c, vmat_n

        def one(k):
            st = self.step(trange[k] - trange[k-1], lin_which, k, k, newton,
                           opts=opts, extrapolate=extrapolate,
                           feedback=lowrank(k))
            if record:
                vdict[trange[k]], pdict[trange[k]] = self.state()
            return st

        def sync_steps(k0, k1):
            its, worst = 0, 0
            for k in range(k0, k1):
                st = one(k)
                its += st['iters']
                worst = max(worst, st['iters'])
                tot['device_seconds'] += st['device_seconds']
            return its, worst

        def policy(per_step, k):
            bound = self.refresh_iters
            tot['batches'].append(per_step)
            if not bound or k >= nt:
                return
            if self._level is None:
                # the first batch behind a set-up: what THIS preconditioner
                # gives at its best.  Above the bound, one rebuild about the
                # current operator is tried (a set-up made for another state)
                self._level = per_step
                if per_step > bound and not self._tried:
                    self._tried = True
                    self._level = None
                    self.refresh_precond()
                    tot['refreshes'] += 1
                return
            # later: rebuild when the count is above the bound AND has risen by
            # a fifth over that level -- a bound that no rebuild reaches (tight
            # tolerance) must not rebuild batch after batch
            if per_step > bound and per_step > 1.2*self._level:
                self._level = None
                self._tried = True
                self.refresh_precond()
                tot['refreshes'] += 1

        pipelined = pipeline and uniform and feedback is None
        # (synchronous until the warm start has its full order -- five
        # solutions for the quartic one: the steps before need more Krylov
        # steps than the run will, and a first batch sized by them fails)
        k = min(nt, 7) if pipelined else 1
        over = pipelined and getattr(self, '_over', False)
        hints = self.__dict__.setdefault('_cycle_hint', {})
        learning = False
        if pipelined:
            its, worst = sync_steps(1, k)
            tot['iters'] += its
            cycle = max(1, worst) if over else max(2, worst + 1)
            if over and newton in hints:
                # what the last sweep of this kind settled at (sweeps repeat
                # themselves: Picard, then Newton after Newton over the same
                # range); the first steps above only bound it from above
                # (+ 1: a sweep STARTS in a transient -- from the initial state,
                # with an empty warm-start history -- that the end of the last
                # sweep did not see; a first batch that fails is repeated
                # step by step)
                cycle = min(cycle, hints[newton] + 1)
            learning = over and newton not in hints
        hold, backoff, lowered = 0, 2, False
        while k < nt:
            kend = min(nt, k + batch)
            if lowered or learning:
                # a trial batch -- and the first batches of a stepper that
                # does not know its cycle length yet -- are short
                kend = min(nt, k + min(batch, 16))
            if not pipelined:
                its, _ = sync_steps(k, kend)
            else:
                self.checkpoint()
                self.set_pipeline(cycle)
                if record:
                    for kk in range(k, kend):
                        one(kk)
                else:
                    # (uniform grid, no low-rank terms: the loop is the
                    # library's)
                    self.run(trange[k] - trange[k-1], lin_which, k, kend - k,
                             newton, opts=opts, extrapolate=extrapolate)
                acc = self.poll()
                self.set_pipeline(0)
                if os.environ.get('DNS_DEBUG'):
                    print('[sweep] steps {0}..{1} cycle {2}: {3}'.format(
                        k, kend, cycle, acc), file=sys.stderr)
                if acc['fails'] == 0 and not over:
                    its, cycle = acc['iters'], max(2, acc['maxit'] + 1)
                elif acc['fails'] == 0:
                    # oversolve (as `dns_imex_run`, imex_capi.inc): every solve
                    # ran the `cycle` columns of the replayed graph.  The
                    # cycle grows when the batch ended close to the tolerance,
                    # shrinks to what was run when every solve reached the
                    # floor earlier, and is TRIED one column shorter (on a
                    # short batch) when every solve stood a decade below the
                    # tolerance in front of its last column
                    its = acc['iters']
                    was = cycle
                    if acc['maxrel'] > 0.9:
                        cycle = was + 1
                        if lowered:
                            backoff = min(4096, 4*backoff)
                        hold = backoff
                    elif acc['maxit'] < was:
                        cycle = max(1, acc['maxit'])
                    elif was > 1 and hold == 0 and (
                            0. < acc['maxprev'] < 0.25
                            or (learning and acc['maxneed'] < was
                                and acc['maxprev'] < 0.5)):
                        cycle = was - 1
                    lowered = cycle < was and acc['maxit'] >= was
                    hold = max(0, hold - 1)
                    learning = learning and cycle != was
                else:
                    # a step of this batch was not through after `cycle` Krylov
                    # steps: the batch again, every step run to convergence
                    self.restore(newton)
                    its, worst = sync_steps(k, kend)
                    if over:
                        cycle = max(cycle + 1, worst)
                        if lowered:
                            backoff = min(4096, 4*backoff)
                        hold, lowered = max(backoff, 4), False
                    else:
                        cycle = max(2, worst + 1)
                    tot['replayed_batches'] += 1
            tot['iters'] += its
            # (oversolve: what the refresh policy goes by is the columns a
            # solve NEEDED, not the columns the cycle ran)
            per = its/float(kend - k)
            if over and pipelined and acc['fails'] == 0:
                per = min(per, acc['sumneed']/float(max(1, acc['solves'])))
            policy(per, kend)
            k = kend
        if over and nt > 2*batch:
            hints[newton] = cycle      # (a real sweep, not a warm-up pass)
        tot['cycle'] = cycle if pipelined else None
        return vdict, pdict, self.update_norm(), tot


def time_sections(trange, nsects=1, addfullsweep=False):
    """local time ranges of the sweeps: `nsects` sections of
    `floor(len(trange)/nsects)` steps sharing their end points, the last one
    takes the rest; `addfullsweep` appends the whole range (snu:1076-1086)"""
    trange = np.asarray(trange, dtype=np.float64)
    if nsects == 1:
        return [trange]
    lensect = int(np.floor(trange.size/nsects))
    out = [trange[k*lensect:(k+1)*lensect+1] for k in range(nsects-1)]
    out.append(trange[(nsects-1)*lensect:])
    if addfullsweep:
        out.append(trange)
    return out


def newton_picard(stepper, trange, iniv, linpoints0, vel_pcrd_stps=1,
                  vel_nwtn_stps=2, vel_nwtn_tol=1e-14, opts=None,
                  extrapolate=4, rhs_table=None, nsects=1,
                  loc_nwtn_tol=5e-15, loc_pcrd_stps=True, addfullsweep=False,
                  tables=None, feedback=None):
    """Picard sweeps first, then Newton sweeps, each linearised about the
    previous sweep's trajectory (snu:1304-1334, 1562-1587).  `linpoints0`:
    `{t: v_inner}` for the first sweep (key `None`: the value for every time
    without one of its own, snu:1427-1431).  `rhs_table (NV, len(trange))`:
    the momentum rhs per time instance (`cfv + fvtd(t)`).  `tables`: dict of
    per-time-instance data, one COLUMN per entry of `trange` -- `fv` (NV),
    `fp` (NP), `mbc` (NV; `condense_velmatsbybcs(M, ..., get_rhs_only=True)`
    of the controlled boundary values, snu:1438-1441) and `dbc` (the
    convection operator's Dirichlet values): controlled Dirichlet values that
    are functions of the time (snu:1433-1466).  `feedback`: see
    `TrapezoidalStepper.sweep`.

    `nsects`, `loc_nwtn_tol`, `loc_pcrd_stps`, `addfullsweep` as in the
    reference (snu:1076-1091, 1576-1587): the time range is cut into sections
    that are iterated one after the other -- each starts from the end of the
    one before, has its own Picard count and tolerance -- and an optional last
    sweep over the whole range restarts from the initial value.  Returns
    `(vdict, pdict, hist)` over all times computed."""
    trange = np.asarray(trange, dtype=np.float64)
    sections = time_sections(trange, nsects, addfullsweep)
    if nsects == 1:
        loc_nwtn_tol, addfullsweep = vel_nwtn_tol, False
    tabs_all = {k: np.asarray(v).T for k, v in (tables or {}).items()
                if v is not None}
    if rhs_table is not None:
        tabs_all['fv'] = np.asarray(rhs_table).T
    index = {t: k for k, t in enumerate(trange)}
    plain = len(sections) == 1
    vel_loc_pcrd_steps = vel_pcrd_stps
    realiniv = np.array(iniv, dtype=np.float64).reshape((-1, 1))
    cur = dict(linpoints0)
    newtk, norm_nwtnupd = 0, 1.
    hist = []
    vall, pall = {}, {}
    for si, loctrng in enumerate(sections):
        if tabs_all:
            rows = [index[t] for t in loctrng]
            sect = {}
            for name, full in tabs_all.items():
                tab = np.zeros((stepper.nslots, full.shape[1]))
                tab[:len(rows)] = full[rows]
                tab[len(rows):] = full[rows[-1]]
                sect[name + '_tab'] = tab
            stepper.set_tables(**sect)
        which = 0
        first = True
        while newtk < vel_nwtn_stps and norm_nwtnupd > loc_nwtn_tol:
            if vel_pcrd_stps > 0:
                vel_pcrd_stps -= 1
                picard = True
            else:
                picard = False
                newtk += 1
            if first or not plain:
                # slot k <-> loctrng[k]; later sweeps of the ONE section of a
                # plain run read the trajectory the sweep before has left on
                # the device, sectioned runs hand the points over per sweep
                which = 0
                for k, t in enumerate(loctrng):
                    stepper.write_linpoint(
                        which, k, cur[t] if t in cur else cur[None])
                first = False
            vdict, pdict, norm_nwtnupd, _ = stepper.sweep(
                loctrng, iniv, which, picard, opts=opts,
                extrapolate=extrapolate, feedback=feedback)
            hist.append(('picard' if picard else 'newton', norm_nwtnupd))
            which = 1 - which       # the new trajectory = next lin. points
            cur.update(vdict)
            vall.update(vdict)
            pall.update(pdict)
        iniv = vall[loctrng[-1]]
        if addfullsweep and si == len(sections) - 2:
            iniv = realiniv
            loc_nwtn_tol = vel_nwtn_tol
        elif loc_pcrd_stps:
            vel_pcrd_stps = vel_loc_pcrd_steps
        norm_nwtnupd, newtk = 1., 0
    return vall, pall, hist
