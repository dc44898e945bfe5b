"""Schaefer-Turek 2D-1 (steady flow around the cylinder, Re = 20) with the
scaffolding assembler and the device time stepper: drag, lift and pressure
difference against the benchmark's reference values.

    python scripts/schaefer_turek.py [mesh level N] [refine] [nts] [tend]

In the units of `fem.get_sysmats` (inflow peak 1, mean 2/3, D = 0.1,
`Re = 1*0.1/nu`) the benchmark's Re = Ubar*D/nu = 20 is `Re = 30`.  Marches
CNAB steps on the GPU until the velocity stops changing, then evaluates
  c_D, c_L = 2 F / (Ubar^2 D),   F = - sum over the cylinder's dofs of the
             momentum residual  A v + N(v) v - J^T p   (consistent nodal forces)
  dp       = p(0.15, 0.2) - p(0.25, 0.2)              (scaled to Ubar = 0.2)
Reference (Schaefer & Turek 1996, John & Matthies 2001):
  c_D = 5.5795, c_L = 0.010619, dp = 0.11752
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from dolfin_navier_scipy_amd import saddle, convection  # noqa: E402
from dolfin_navier_scipy_amd import lin_alg_utils as lau  # noqa: E402
from dolfin_navier_scipy_amd.fem import get_sysmats  # noqa: E402

REF = dict(cD=5.5795, cL=0.010619, dp=0.11752)


def p1_eval(mesh, pvert, pt):
    """P1 function with vertex values `pvert` at the point `pt`"""
    v, c = mesh.verts, mesh.cells
    a, b, d = v[c[:, 0]], v[c[:, 1]], v[c[:, 2]]
    det = (b[:, 0]-a[:, 0])*(d[:, 1]-a[:, 1]) - (d[:, 0]-a[:, 0])*(b[:, 1]-a[:, 1])
    l1 = ((pt[0]-a[:, 0])*(d[:, 1]-a[:, 1]) - (d[:, 0]-a[:, 0])*(pt[1]-a[:, 1]))/det
    l2 = ((b[:, 0]-a[:, 0])*(pt[1]-a[:, 1]) - (pt[0]-a[:, 0])*(b[:, 1]-a[:, 1]))/det
    l0 = 1 - l1 - l2
    worst = np.minimum(np.minimum(l0, l1), l2)
    k = int(np.argmax(worst))            # the cell that contains it best
    lam = np.array([l0[k], l1[k], l2[k]])
    return float(lam @ pvert[c[k]]), float(worst[k])


def run(N=2, refine=0, nts=512, tend=8.0, Re=30., verbose=True):
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=N, refine=refine,
                                 Re=Re)
    th, inv, mesh = femp['V'], femp['invinds'], femp['mesh']
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
    nsteps, chunk = int(round(tend*nts)), nts
    vprev = v0
    done = 0
    secs = 0.
    while done < nsteps:
        n = min(chunk, nsteps - done)
        ds, its, last = stp.run(n, cf, opts)
        secs += ds
        done += n
        v, p = stp.get_state()
        chg = np.linalg.norm(v - vprev)/np.linalg.norm(v)
        vprev = v
        if verbose:
            sys.stderr.write('t = {0:.2f}: change over the last {1} steps '
                             '{2:.2e}\n'.format(done*dt, n, chg))
        if not np.isfinite(chg):
            raise RuntimeError('diverged')
        if chg < 1e-8:
            break
    stp.close()
    system.close()
    # consistent nodal forces from the un-condensed operators
    stms = th.stokes_mats(nu=femp['nu'])
    vfull = np.zeros((th.vdim, 1))
    vfull[inv] = v
    vfull[femp['dbcinds'], 0] = femp['dbcvals']
    res = stms['A'] @ vfull + th.convection_vec(vfull) - stms['JT'] @ p
    nodes, xy = th.boundary_nodes()
    r = np.sqrt((xy[:, 0] - 0.2)**2 + (xy[:, 1] - 0.2)**2)
    cyl = nodes[r < 0.05 + 1e-3]
    ubar, diam = 2./3, 0.1
    fx = -res[2*cyl, 0].sum()
    fy = -res[2*cyl + 1, 0].sum()
    cD, cL = 2*fx/(ubar**2*diam), 2*fy/(ubar**2*diam)
    pvert = np.zeros(mesh.nverts)
    pvert[:] = p[th.vert_pdof, 0] if hasattr(th, 'vert_pdof') else p[:, 0]
    pa, wa = p1_eval(mesh, pvert, (0.15, 0.2))
    pe, we = p1_eval(mesh, pvert, (0.25, 0.2))
    dp = (pa - pe)*(0.2/ubar)**2          # in the benchmark's units
    cv.close()
    return dict(level=N, refine=refine, NV=int(NV), NP=int(NP), dt=dt,
                t_end=done*dt, steps=done, device_seconds=secs,
                last_change=float(chg), cD=float(cD), cL=float(cL),
                dp=float(dp), reference=REF,
                rel_err=dict(cD=abs(cD/REF['cD'] - 1),
                             dp=abs(dp/REF['dp'] - 1)),
                point_in_cell=[wa, we])


if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    refine = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nts = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    tend = float(sys.argv[4]) if len(sys.argv) > 4 else 8.0
    print(json.dumps(run(N, refine, nts, tend)))
