"""Schaefer-Turek 2D-2 (periodic vortex shedding, Re = 100) with the scaffolding
assembler and the device time stepper: Strouhal number and the maxima of the
drag and lift coefficients.

    python scripts/schaefer_turek_unsteady.py [mesh level N] [nts] [tend]

In the units of `fem.get_sysmats` (inflow peak 1, mean Ubar = 2/3, D = 0.1) the
benchmark's Re = Ubar*D/nu = 100 is `Re = 150`; times scale by 1.5.
Reference intervals (Schaefer & Turek 1996): St 0.2950-0.3050,
c_D,max 3.22-3.24, c_L,max 0.99-1.01.
Forces: consistent nodal forces, F = - sum over the cylinder's dofs of
`M dv/dt + A v + N(v) v - J^T p`, evaluated on the host from the state the
device hands back every step (only the cells at the cylinder contribute).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from dolfin_navier_scipy_amd import saddle, convection  # noqa: E402
from dolfin_navier_scipy_amd import lin_alg_utils as lau  # noqa: E402
from dolfin_navier_scipy_amd.fem import get_sysmats  # noqa: E402
from dolfin_navier_scipy_amd.fem import taylor_hood as thm  # noqa: E402

REF = dict(St=(0.2950, 0.3050), cDmax=(3.22, 3.24), cLmax=(0.99, 1.01))


class CylinderForces(object):
    """row sums of the un-condensed operators over the cylinder's dofs"""

    def __init__(self, femp, th):
        inv = femp['invinds']
        stms = th.stokes_mats(nu=femp['nu'])
        nodes, xy = th.boundary_nodes()
        r = np.sqrt((xy[:, 0] - 0.2)**2 + (xy[:, 1] - 0.2)**2)
        cyl = nodes[r < 0.05 + 1e-3]
        self.th, self.inv = th, inv
        self.bcs = np.zeros(th.vdim)
        self.bcs[femp['dbcinds']] = femp['dbcvals']
        self.rows = [2*cyl, 2*cyl + 1]
        self.a, self.m, self.j, self.a0 = [], [], [], []
        for rows in self.rows:
            sel = np.zeros(th.vdim)
            sel[rows] = 1.
            arow = stms['A'].T @ sel
            self.a.append(arow[inv])
            self.a0.append(float(arow @ self.bcs))
            self.m.append((stms['M'].T @ sel)[inv])
            self.j.append(stms['JT'].T @ sel)
        # cells with a node on the cylinder: the only ones whose convection
        # term reaches the cylinder's test functions
        oncyl = np.zeros(th.vdim // 2, dtype=bool)
        oncyl[cyl] = True
        self.cells = np.where(oncyl[th.cellnodes].any(axis=1))[0]
        self.vd = th._vdofs()[self.cells]                    # (k, 6, 2)
        self.mask = oncyl[th.cellnodes[self.cells]]          # (k, 6)
        self.w = thm._QW[None, :]*th.area[self.cells][:, None]
        self.gphi = th._gphi[self.cells]

    def __call__(self, v, vdot, p):
        th = self.th
        full = self.bcs.copy()
        full[self.inv] = v
        uloc = full[self.vd]                                  # (k, 6, 2)
        uq = np.einsum('qa,cai->cqi', th._phi, uloc)
        guq = np.einsum('cqaj,cai->cqij', self.gphi, uloc)
        conv = np.einsum('cqij,cqj->cqi', guq, uq)
        floc = np.einsum('cq,qa,cqi->cai', self.w, th._phi, conv)
        out = []
        for d in range(2):
            nd = float((floc[:, :, d]*self.mask).sum())
            out.append(-(self.a[d] @ v + self.a0[d] + self.m[d] @ vdot
                         - self.j[d] @ p + nd))
        return out


def run(N=3, nts=1024, tend=18.0, Re=150., verbose=True):
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=N, Re=Re)
    th, inv = femp['V'], femp['invinds']
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    dt = 1./nts
    vp0 = lau.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'],
                               rhsp=rhsd['fp'])
    lau.clear_cache()
    v0 = vp0[:NV]
    system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
    system.setup_precond(cheb_degree=6, schur='dense', drop_tol=1e-3,
                         factorization='full')
    cv = convection.ConvectionP2.from_taylor_hood(th, inv, femp['dbcinds'],
                                                  femp['dbcvals'])
    stp = saddle.ImexStepper(system, (M - .5*dt*A).tocsr())
    nfc = cv.apply(v0, scale=-1.0)
    stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
    stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
    stp.set_convection(cv, scale=-1.0)
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt, extrapolate=4)
    opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
    forces = CylinderForces(femp, th)
    ubar, diam = 2./3, 0.1
    nsteps = int(round(tend*nts))
    # free run first (pipelined), forces only over the last third
    nfree = (2*nsteps)//3
    stp.run(nfree, cf, opts)
    vprev = stp.get_state()[0][:, 0]
    ts, cds, cls = [], [], []
    for k in range(nfree, nsteps):
        stp.run(1, cf, opts)
        v, p = stp.get_state()
        v, p = v[:, 0], p[:, 0]
        if not np.isfinite(v).all():
            raise RuntimeError('diverged at step {0}'.format(k))
        # p belongs to the step's end; dv/dt by the backward difference
        fx, fy = forces(v, (v - vprev)/dt, p)
        vprev = v
        ts.append((k + 1)*dt)
        cds.append(2*fx/(ubar**2*diam))
        cls.append(2*fy/(ubar**2*diam))
    stp.close()
    system.close()
    cv.close()
    ts, cds, cls = np.array(ts), np.array(cds), np.array(cls)
    # shedding frequency from the upward zero crossings of the lift
    cl0 = cls - cls.mean()
    up = np.where((cl0[:-1] < 0) & (cl0[1:] >= 0))[0]
    tc = ts[up] - cl0[up]*(ts[up+1] - ts[up])/(cl0[up+1] - cl0[up])
    if tc.size >= 3:
        period = float(np.mean(np.diff(tc)))
        st = diam/(period*ubar)
        last = ts >= tc[-3]                 # the last two periods
    else:
        period, st, last = float('nan'), float('nan'), slice(None)
    out = dict(level=N, NV=int(NV), NP=int(NP), dt=dt, t_end=tend,
               periods_seen=int(max(tc.size - 1, 0)), period=period, St=st,
               cDmax=float(cds[last].max()), cDmin=float(cds[last].min()),
               cLmax=float(cls[last].max()), cLmin=float(cls[last].min()),
               reference=REF)
    if verbose:
        sys.stderr.write(json.dumps(out) + '\n')
    return out


if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    nts = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    tend = float(sys.argv[3]) if len(sys.argv) > 3 else 18.0
    print(json.dumps(run(N, nts, tend)))
