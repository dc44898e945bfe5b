"""ORACLE (test infrastructure, never shipped, never the thing measured).

CPU restatement of the linear-algebra boundary the reference calls as
`sadptprj_riclyap_adi.lin_alg_utils` (`lau`).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

Provenance / pinning
--------------------
`lau` is a third-party dependency of the reference, listed UNPINNED in
`/root/reference/requirements.txt:6`; its source is not in the reference tree.
What is restated here is its published contract as consumed at the reference's
call sites:

 * `solve_sadpnt_smw(amat, jmat, jmatT, rhsv, rhsp, umat, vmat, return_alu)`
   solves `[[A - U V, J^T], [J, 0]] [v; p] = [rhsv; rhsp]` by a sparse direct
   factorisation (SuperLU) -- call sites `time_int_utils.py:402,466,605`,
   `stokes_navier_utils.py:401,458,497,904,1505,1629`; the low-rank term by
   Sherman-Morrison-Woodbury (hence `_smw`).
 * `app_prj_via_sadpnt` applies `Pi^T`, `Pi = I - M^-1 J^T (J M^-1 J^T)^-1 J`
   (`residual_checks.py:16-38`).

The reference ships no golden vectors for `lau` (its tests need dolfin), so the
direct solve is pinned by its mathematical definition: the tests check
`||K x - b|| / ||b||` of every oracle solve.  For the Krylov variant
(`krylov='Gmres'`) there is no reference test at all: PARITY UNPINNED for the
iteration history; the converged solution is compared with the direct solve.
"""
import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spsla

__all__ = ['saddle_matrix', 'solve_sadpnt_smw', 'app_prj_via_sadpnt',
           'apply_massinv', 'SaddleLU', 'RefinedSolve']


def saddle_matrix(amat, jmat, jmatT=None):
    """`K = [[A, J^T], [J, 0]]` in CSC"""
    jT = jmat.T if jmatT is None else jmatT
    return sps.bmat([[amat, jT], [jmat, None]], format='csc')


class SaddleLU(object):
    """factor once, solve many -- what `return_alu=True` hands back
    (`time_int_utils.py:605-615`) and what `spsla.factorized` is in the
    CNAB/SBDF2 loops (`time_int_utils.py:89-91,134`)"""

    def __init__(self, amat, jmat, jmatT=None):
        self.NP, self.NV = jmat.shape
        self.lu = spsla.splu(saddle_matrix(amat, jmat, jmatT))

    def __call__(self, rhs):
        rhs = np.asarray(rhs, dtype=np.float64)
        if rhs.ndim == 1:
            return self.lu.solve(rhs)
        return self.lu.solve(rhs).reshape(rhs.shape)


class RefinedSolve(object):
    """`solve_sadpnt_smw` for a long SEQUENCE of nearby systems (the re-valued
    matrices of a Newton/Picard sweep over thousands of steps, where a fresh
    sparse LU per step -- what the direct path amounts to -- takes minutes):
    the LU of one system of the sequence serves the following ones through
    iterative refinement, `x += K_ref^-1 (b - K x)`, carried on until the
    residual is at the rounding floor (`rtol`) or stops shrinking; a system
    that needs more than `max_inner` corrections gets a factorisation of its
    own (so the fallback IS the direct solve).  Same answer as the direct
    solve up to rounding -- `tests/test_newton_picard.py` holds the two side
    by side."""

    def __init__(self, rtol=2e-16, max_inner=10, floor=1e-13, refactor_at=6):
        self.rtol, self.max_inner, self.floor = rtol, max_inner, floor
        # a solve that needed this many corrections makes the NEXT system
        # factor its own matrix (the sequence has drifted away)
        self.refactor_at = refactor_at
        self.lu = None
        self.factorisations = 0
        self.corrections = 0
        self.calls = 0

    def _factor(self, K):
        self.lu = spsla.splu(K.tocsc())
        self.factorisations += 1

    def __call__(self, amat=None, jmat=None, jmatT=None, rhsv=None, rhsp=None,
                 **kw):
        NP, NV = jmat.shape
        rhsv = np.asarray(rhsv, dtype=np.float64).reshape((NV, 1))
        rhsp = np.zeros((NP, 1)) if rhsp is None else \
            np.asarray(rhsp, dtype=np.float64).reshape((NP, 1))
        K = saddle_matrix(amat, jmat, jmatT).tocsr()
        b = np.vstack([rhsv, rhsp])[:, 0]
        bn = np.linalg.norm(b)
        self.calls += 1
        for attempt in (0, 1):
            if self.lu is None:
                self._factor(K)
            x = self.lu.solve(b)
            r = b - K @ x
            rn, it = np.linalg.norm(r), 0
            while rn > self.rtol*bn and it < self.max_inner:
                xn = x + self.lu.solve(r)
                r2 = b - K @ xn
                r2n = np.linalg.norm(r2)
                it += 1
                if r2n >= rn:              # no further: the rounding floor
                    break
                stalled = r2n > 0.5*rn and r2n <= self.floor*bn
                x, r, rn = xn, r2, r2n
                if stalled:
                    break
            self.corrections += it
            if rn <= self.floor*bn:
                if it >= self.refactor_at:
                    self.lu = None
                return x.reshape((-1, 1))
            self.lu = None                 # this system gets its own LU
        raise RuntimeError('refined solve failed (relative residual {0:.2e})'
                           .format(rn/bn))


def solve_sadpnt_smw(amat=None, jmat=None, jmatT=None, rhsv=None, rhsp=None,
                     umat=None, vmat=None, return_alu=False,
                     krylov=None, krpslvprms={}, krplsprms={},
                     decouplevp=False, solve_A=None, symmetric=False,
                     cgtol=1e-8, **kw):
    """direct solve of the saddle-point system, returns `(NV+NP, k)`"""
    NP, NV = jmat.shape
    rhsv = np.asarray(rhsv, dtype=np.float64).reshape((NV, -1))
    ncols = rhsv.shape[1]
    rhsp = np.zeros((NP, ncols)) if rhsp is None else \
        np.asarray(rhsp, dtype=np.float64).reshape((NP, -1))
    if amat is None:
        # the decoupled variant of snu:1622-1628: `amat` omitted, `A^-1` given
        # as the callable `solve_A` -- Schur complement CG
        #   S p = J A^-1 rhsv - rhsp,  S = J A^-1 J^T;  v = A^-1 (rhsv - J^T p)
        if solve_A is None:
            raise ValueError('decoupled solve needs `solve_A`')
        jT = sps.csr_matrix(jmat.T) if jmatT is None else jmatT
        sol = np.zeros((NV + NP, ncols))
        for k in range(ncols):
            ainvf = np.asarray(solve_A(rhsv[:, k])).reshape(-1)

            def _schur(pvec):
                return jmat @ np.asarray(solve_A(jT @ pvec)).reshape(-1)
            sop = spsla.LinearOperator((NP, NP), matvec=_schur,
                                       dtype=np.float64)
            pk, info = spsla.cg(sop, jmat @ ainvf - rhsp[:, k], rtol=cgtol,
                                atol=0., maxiter=10*NP)
            if info != 0:
                raise RuntimeError('Schur complement CG did not converge')
            sol[:NV, k] = np.asarray(solve_A(rhsv[:, k] - jT @ pk)).reshape(-1)
            sol[NV:, k] = pk
        return sol
    alu = SaddleLU(amat, jmat, jmatT)
    rhs = np.vstack([rhsv, rhsp])
    sol = alu(rhs)
    if umat is not None:
        # (K - Ue Ve)^-1 = K^-1 + K^-1 Ue (I - Ve K^-1 Ue)^-1 Ve K^-1
        umat = np.asarray(sps.csr_matrix(umat).todense()) \
            if sps.issparse(umat) else np.asarray(umat)
        vmat = np.asarray(sps.csr_matrix(vmat).todense()) \
            if sps.issparse(vmat) else np.asarray(vmat)
        r = umat.shape[1]
        ue = np.vstack([umat, np.zeros((NP, r))])
        ve = np.hstack([vmat, np.zeros((r, NP))])
        kiu = alu(ue)
        small = np.eye(r) - ve.dot(kiu)
        sol = sol + kiu.dot(np.linalg.solve(small, ve.dot(sol)))
    if return_alu:
        return sol, alu
    return sol


def app_prj_via_sadpnt(amat=None, jmat=None, rhsv=None, jmatT=None,
                       umat=None, vmat=None, transposedprj=False):
    """apply the discrete Leray projector (or its transpose) via a saddle solve

    `Pi = I - A^-1 J^T S^-1 J`,  `Pi^T = I - J^T S^-1 J A^-1`,
    `S = J A^-1 J^T` (reference `residual_checks.py:21-24`)
    """
    NP, NV = jmat.shape
    rhsv = np.asarray(rhsv).reshape((NV, -1))
    jT = jmat.T if jmatT is None else jmatT
    if transposedprj:
        # [A J^T; J 0][w; q] = [f; 0]  ->  Pi^T f = A w = f - J^T q
        wq = solve_sadpnt_smw(amat=amat, jmat=jmat, jmatT=jT, rhsv=rhsv,
                              umat=umat, vmat=vmat)
        return rhsv - jT @ wq[NV:, :]
    # Pi f = f - A^-1 J^T S^-1 J f : solve with rhs [A f; 0]
    wq = solve_sadpnt_smw(amat=amat, jmat=jmat, jmatT=jT, rhsv=amat @ rhsv,
                          umat=umat, vmat=vmat)
    return wq[:NV, :]


def apply_massinv(M, rhsa, output=None):
    """`M^-1 rhsa` by a sparse direct solve (`time_dep_nse_bigchannel.py:33`)"""
    mlu = spsla.splu(sps.csc_matrix(M))
    if sps.issparse(rhsa):
        sol = mlu.solve(np.asarray(rhsa.todense()))
    else:
        sol = mlu.solve(np.asarray(rhsa))
    return sps.csr_matrix(sol) if output == 'sparse' else sol
