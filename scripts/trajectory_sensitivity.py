"""How far apart do two runs of the ORACLE itself drift when the initial
velocity differs by a relative 1e-13?  (The conditioning of the trajectory: no
solver that differs from the reference's SuperLU path by rounding can agree with
it better than this over the same horizon.)

  python scripts/trajectory_sensitivity.py config5 6000
  python scripts/trajectory_sensitivity.py config2 512 [Re]
writes a JSON line per checkpoint; CPU only."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import imex_oracle, saddle_oracle          # noqa: E402
from dolfin_navier_scipy_amd.fem import get_sysmats    # noqa: E402


def setup(which, nsteps, Re=None):
    if which == 'config5':
        femp, sm, rhsd = get_sysmats(problem='gen_bccont', Re=60,
                                     bccontrol=True)
        palpha, tE, Nts = 1e-5, 15., 6000
        A = (sm['A'] + 1./palpha*sm['Arob']).tocsr()
        Brob = 1./palpha*sm['Brob']
        bsum = Brob[:, :1] + Brob[:, 1:]

        def f_tdp(t):
            return rhsd['fv'] + np.sin(t/tE*2*np.pi)*bsum
    else:
        femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2,
                                     Re=Re or 80)
        tE, Nts = (1., 512)
        A = sm['A']

        def f_tdp(t):
            return rhsd['fv']
    th, inv = femp['V'], femp['invinds']
    M, J = sm['M'], sm['J']
    NP, NV = J.shape
    dbcinds, dbcvals = femp['dbcinds'], femp['dbcvals']
    vp0 = saddle_oracle.solve_sadpnt_smw(amat=A, jmat=J, rhsv=f_tdp(0.),
                                         rhsp=rhsd['fp'])
    trange = np.linspace(0., tE, Nts + 1)[:nsteps + 1]
    if nsteps > Nts:
        trange = np.arange(nsteps + 1)*(tE/Nts)

    def appnd(vvec, bcs):
        full = np.full((th.vdim, 1), np.nan)
        full[inv] = vvec
        full[dbcinds, 0] = dbcvals
        return full

    def f_vdp(vf):
        return -th.convection_vec(vf)[inv, :]

    def kw(inivel, rec):
        return dict(trange=trange, inivel=inivel, inip=-vp0[NV:], bcs_ini=[],
                    M=M, A=A, J=J, f_vdp=f_vdp, f_tdp=f_tdp,
                    g_tdp=lambda t: rhsd['fp'], scalep=-1.,
                    getbcs=lambda t, v, p, mode=None: [],
                    applybcs=lambda b: (0., 0., 0.), appndbcs=appnd,
                    savevp=rec, check_ff_maxv=1e8)
    return M, inv, vp0[:NV], kw, trange


class Snap(object):
    def __init__(self, inv, every):
        self.inv, self.every, self.k, self.v, self.p = inv, every, 0, {}, {}

    def __call__(self, v, p, time=None):
        if self.k % self.every == 0:
            self.v[self.k] = v[self.inv].copy()
            self.p[self.k] = np.array(p).copy()
        self.k += 1


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'config5'
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    Re = float(sys.argv[3]) if len(sys.argv) > 3 else None
    eps = float(os.environ.get('DNS_SENS_EPS', '1e-13'))
    M, inv, v0, kw, trange = setup(which, nsteps, Re)
    every = max(1, nsteps//20)
    rng = np.random.default_rng(0)
    t0 = time.time()
    ra, rb = Snap(inv, every), Snap(inv, every)
    imex_oracle.cnab(**kw(v0, ra))
    print('# run A', time.time() - t0, 's', file=sys.stderr)
    imex_oracle.cnab(**kw(v0*(1. + eps*rng.standard_normal(v0.shape)), rb))

    def mn(x):
        return float(np.sqrt((x.T @ (M @ x)).item()))
    for k in sorted(ra.v):
        ev = mn(ra.v[k] - rb.v[k])/mn(ra.v[k])
        ep = float(np.linalg.norm(ra.p[k] - rb.p[k])/np.linalg.norm(ra.p[k]))
        print(json.dumps(dict(workload=which, Re=Re, step=k, eps=eps,
                              v_rel_M=ev, p_rel=ep)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
