"""Several simulations on ONE GPU, one XCD each (DNS_CU_MASK): K independent
CNAB runs of the benchmark workload advance concurrently from K host threads.

    python scripts/xcd_ensemble.py [K] [nsteps] [same]

`same` = 1 puts all K on XCD 0 (contention check).
"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench  # noqa: E402
from dolfin_navier_scipy_amd import saddle, convection, _capi  # noqa: E402


def run(K=8, nsteps=400, same=False, warmup=40, device=0):
    dt = 1./512
    femp, sm, rhsd = bench.build_problem(N=2, Re=100.)
    M, A, J = sm['M'], sm['A'], sm['J']
    th, inv = femp['V'], femp['invinds']
    v0, pt0, st0 = bench.initial_state(
        sm, rhsd, lambda F, Jm: saddle.SaddleSystem(F, Jm, device=device))
    F = (M + .5*dt*A).tocsr()
    R1 = (M - .5*dt*A).tocsr()
    cf = saddle.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt, extrapolate=4)
    opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=True, reorth=2)
    sims = []
    for k in range(K):
        os.environ['DNS_CU_MASK'] = '8={0}'.format(0 if same else k % 8)
        system = saddle.SaddleSystem(F, J, device=device)
        system.setup_precond(cheb_degree=6, schur='dense', fp32_store=True,
                             drop_tol=1e-3, factorization='full')
        cv = convection.ConvectionP2.from_taylor_hood(
            th, inv, femp['dbcinds'], femp['dbcvals'], device=device)
        stp = saddle.ImexStepper(system, R1)
        nfc = cv.apply(v0, scale=-1.0)
        stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
        stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
        stp.set_convection(cv, scale=-1.0)
        sims.append((system, cv, stp))
    os.environ.pop('DNS_CU_MASK', None)

    def work(stp, n):
        stp.run(n, cf, opts)

    for n in (warmup, nsteps):
        _capi.device_synchronize(device)
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(s[2], n)) for s in sims]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        _capi.device_synchronize(device)
        wall = time.perf_counter() - t0
    finals = [s[2].get_state()[0] for s in sims]
    spread = max(np.abs(f - finals[0]).max() for f in finals)
    for system, cv, stp in sims:
        stp.close()
        cv.close()
        system.close()
    return dict(simulations=K, steps_each=nsteps, same_xcd=bool(same),
                steps_per_s_total=K*nsteps/wall,
                steps_per_s_each=nsteps/wall, max_abs_spread=float(spread))


if __name__ == '__main__':
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    same = len(sys.argv) > 3 and sys.argv[3] == '1'
    print(json.dumps(run(K, n, same)))
