"""Distance between the device-resident CNAB run (one Krylov step per time step,
bench settings) and the oracle's direct-solve trajectory, chunk by chunk, with
and without the residual carry-over; beside it how far the ORACLE moves when
its initial value is perturbed by 1e-13 (the conditioning of the trajectory).

    python scripts/horizon_probe.py [nsteps] [Re] [rtol] [chunk]
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spsla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402
from dolfin_navier_scipy_amd import saddle, convection  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Re = float(sys.argv[2]) if len(sys.argv) > 2 else 100.
rtol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-10
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dt = 1./512
femp, sm, rhsd = bench.build_problem(N=2, Re=Re)
M, A, J = sm['M'].tocsr(), sm['A'].tocsr(), sm['J'].tocsr()
NP, NV = J.shape
th, inv = femp['V'], femp['invinds']
dflt = bench.DEFAULTS
v0, pt0, st0 = bench.initial_state(sm, rhsd,
                                   lambda F, Jm: saddle.SaddleSystem(F, Jm))
cv = convection.ConvectionP2.from_taylor_hood(th, inv, femp['dbcinds'],
                                              femp['dbcvals'])
nfc0 = cv.apply(v0, scale=-1.0)
R1 = (M - .5*dt*A).tocsr()
K = sps.bmat([[M + .5*dt*A, J.T], [J, None]], format='csc')
lu = spsla.splu(K)
fv, fp = rhsd['fv'], rhsd['fp']


def oracle(vstart):
    """factor-once CNAB (tiu:104-143) from `vstart`, convection by the device
    operator applied to the oracle's own iterate (same N(v)v on both sides)"""
    v = vstart.copy()
    nfo = cv.apply(v, scale=-1.0)
    out = {}
    for k in range(1, nsteps + 1):
        nfc = cv.apply(v, scale=-1.0)
        b = np.vstack([R1 @ v + dt*fv + 1.5*dt*nfc - .5*dt*nfo, fp])
        x = lu.solve(b[:, 0]).reshape((-1, 1))
        v, nfo = x[:NV], nfc
        if k % chunk == 0:
            out[k] = (v.copy(), -x[NV:]/dt)
    return out


def mn(x):
    return float(np.sqrt((x.T @ (M @ x)).item()))


t0 = time.time()
ref = oracle(v0)
print('# oracle', round(time.time() - t0, 1), 's', file=sys.stderr)
rng = np.random.default_rng(0)
pert = oracle(v0*(1. + 1e-13*rng.standard_normal(v0.shape)))
res = {}
for carry in (False, True):
    system = saddle.SaddleSystem((M + .5*dt*A).tocsr(), J)
    system.setup_precond(cheb_degree=dflt['cheb'], schur='dense',
                         fp32_store=bool(dflt['fp32']), drop_tol=dflt['drop'],
                         factorization=dflt['fact'])
    stp = saddle.ImexStepper(system, R1)
    stp.set_state(v0, nfc_c=nfc0, nfc_o=nfc0)
    stp.set_rhs(dt*fv, fp)
    stp.set_convection(cv, scale=-1.0)
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt, extrapolate=dflt['extrap'],
                                   carry_residual=carry)
    opts = saddle.solve_opts(method='gmres', rtol=rtol, maxiter=400,
                             restart=60, check_every=2, use_graph=True,
                             reorth=dflt['reorth'])
    tot_s, tot_it = 0., 0
    for k in range(chunk, nsteps + 1, chunk):
        ds, its, last = stp.run(chunk, cf, opts)
        tot_s += ds
        tot_it += its
        vg, pg = stp.get_state()
        vo, po = ref[k]
        vq, pq = pert[k]
        row = dict(step=k, carry=carry, rtol=rtol, Re=Re,
                   v=mn(vg - vo)/mn(vo),
                   p=float(np.linalg.norm(pg - po)/np.linalg.norm(po)),
                   oracle_sens_v=mn(vq - vo)/mn(vo),
                   oracle_sens_p=float(np.linalg.norm(pq - po)
                                       / np.linalg.norm(po)),
                   its_per_step=its/float(chunk),
                   relres=last['true_relres'], replayed=stp.last_run['replayed'],
                   steps_per_s=chunk/ds)
        print(json.dumps(row))
        sys.stdout.flush()
    stp.close()
    system.close()
cv.close()
