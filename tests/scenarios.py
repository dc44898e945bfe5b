"""Seeded test scenarios for the semi-explicit integrators.

One scenario = the full keyword set of `time_int_utils.cnab/sbdftwo`
(reference tiu:23-34, 260-268) built on Taylor-Hood matrices of a small
channel-with-obstacle mesh.  The same builder feeds (i) the reference module
when `tests/golden/make_golden.py` records the golden vectors, (ii) the CPU
oracle and (iii) the HIP path, so all three see identical inputs.

Variants
 * 'plain'   -- constant rhs, static Dirichlet data (what `solve_nse` sets up
               for the cylinder wake, reference snu:1103-1126)
 * 'forced'  -- time-dependent `f_tdp`, nonzero `g_tdp`, a `dynamic_rhs` with
               memory and a state dependent `f_tvdp`
 * 'movingbc'-- time-dependent controlled Dirichlet values: `getbcs` /
               `applybcs` / `appndbcs` as in reference snu:1003-1006,1111-1115,
               1152-1157 (with the control values actually written)
"""
import numpy as np

from dolfin_navier_scipy_amd.fem import (TaylorHood, channel_cylinder_mesh,
                                         condense_sysmatsbybcs)

VARIANTS = ('plain', 'forced', 'movingbc')


class Recorder(object):
    """`savevp` callback collecting `(time, v_with_bcs, p)`"""

    def __init__(self):
        self.times, self.vels, self.prss = [], [], []

    def __call__(self, vvec, pvec, time=None):
        self.times.append(time)
        self.vels.append(np.array(vvec, dtype=np.float64).reshape(-1))
        self.prss.append(np.array(pvec, dtype=np.float64).reshape(-1))

    def arrays(self):
        return (np.array(self.times, dtype=np.float64), np.array(self.vels),
                np.array(self.prss))


def toy_problem(nx=22, ny=8, Re=40.):
    mesh = channel_cylinder_mesh(nx=nx, ny=ny)
    th = TaylorHood(mesh)
    nu = 0.1/Re
    stms = th.stokes_mats(nu=nu)
    dbcinds, dbcvals, invinds = th.cylinderwake_bcs(obstacle_halfwidth=0.05)
    smc, rhsd, invinds = condense_sysmatsbybcs(stms, dbcinds, dbcvals)
    return dict(th=th, stms=stms, smc=smc, rhsd=rhsd, invinds=invinds,
                dbcinds=dbcinds, dbcvals=dbcvals, nu=nu)


def build(variant='plain', seed=0, Nts=12, tE=0.06, nx=22, ny=8, Re=40.,
          prob=None):
    """returns `(kwargs_for_integrator, recorder, aux)`"""
    prob = toy_problem(nx=nx, ny=ny, Re=Re) if prob is None else prob
    th, smc, rhsd = prob['th'], prob['smc'], prob['rhsd']
    invinds, dbcinds, dbcvals = (prob['invinds'], prob['dbcinds'],
                                 prob['dbcvals'])
    M, A, J = smc['M'], smc['A'], smc['J']
    NP, NV = J.shape
    rng = np.random.default_rng(seed)
    vdim = th.vdim
    trange = np.linspace(0., tE, Nts+1)

    # a discretely divergence-free-ish smooth initial state: a scaled, masked
    # parabola plus a seeded perturbation (projected by the first solves)
    xy = th.nodecoords
    ufull = np.zeros((vdim, 1))
    ufull[0::2, 0] = 4*xy[:, 1]*(0.41-xy[:, 1])/0.41**2
    ufull[dbcinds, 0] = dbcvals
    inivel = ufull[invinds] + 1e-2*rng.standard_normal((NV, 1))
    inip = np.zeros((NP, 1))
    cfv, cfp = rhsd['fv'], rhsd['fp']

    # control boundary: the dofs of the inflow rows get a time modulation
    if variant == 'movingbc':
        cntinds = dbcinds[np.abs(dbcvals) > 0]        # inflow dofs
        cntbase = dbcvals[np.abs(dbcvals) > 0]
        statinds = dbcinds[np.abs(dbcvals) == 0]
        statvals = dbcvals[np.abs(dbcvals) == 0]
        stms = prob['stms']
        Afull, Mfull, Jfull = stms['A'], stms['M'], stms['J']
        # static part of the BC rhs only
        aux0 = np.zeros((vdim, 1))
        aux0[statinds, 0] = statvals
        cfv = -(Afull @ aux0)[invinds, :]
        cfp = -(Jfull @ aux0)

        def getbcs(time, vvec, pvec, mode=None):
            return (cntbase*(1. + 0.3*np.sin(20*time))).tolist()

        def applybcs(bcs_n):
            caux = np.zeros((vdim, 1))
            caux[cntinds, 0] = bcs_n
            return (-(Afull @ caux)[invinds, :], -(Jfull @ caux),
                    (Mfull @ caux)[invinds, :])

        def appndbcs(vvec, ccntrlldbcvals):
            full = np.full((vdim, 1), np.nan)      # dts:58
            full[invinds] = vvec
            full[statinds, 0] = statvals
            full[cntinds, 0] = ccntrlldbcvals
            return full
        bcs_ini = getbcs(trange[0], None, None)
    else:
        def getbcs(time, vvec, pvec, mode=None):
            return []

        def applybcs(bcs_n):
            return 0., 0., 0.                      # snu:1104-1105

        def appndbcs(vvec, ccntrlldbcvals):
            full = np.full((vdim, 1), np.nan)
            full[invinds] = vvec
            full[dbcinds, 0] = dbcvals
            return full
        bcs_ini = []

    def f_vdp(vfull):
        # minus sign: goes to the rhs (snu:1128-1140, snu:103-107)
        return -th.convection_vec(vfull)[invinds, :]

    kw = dict(trange=trange, inivel=inivel, inip=inip, bcs_ini=bcs_ini,
              M=M, A=A, J=J, f_vdp=f_vdp, scalep=-1.,
              getbcs=getbcs, applybcs=applybcs, appndbcs=appndbcs,
              check_ff_maxv=1e8, verbose=False)

    if variant == 'forced':
        fdir = rng.standard_normal((NV, 1))
        fdir = 1e-3*(M @ fdir)
        gdir = 1e-4*rng.standard_normal((NP, 1))
        bdir = 1e-3*(M @ rng.standard_normal((NV, 1)))
        cdir = rng.standard_normal((1, NV))/np.sqrt(NV)

        def f_tdp(t):
            return cfv + np.sin(30*t)*fdir

        def g_tdp(t):
            return cfp + np.cos(10*t)*gdir

        def dynamic_rhs(t, vc=None, memory={}, mode=None):
            # a scalar observer state driven by an output y = c v
            if mode == 'init':
                memory = dict(x=0.0, lastt=t)
                return 0.*bdir, memory
            y = (cdir @ vc).item()
            x = memory['x']
            if mode in ('heuncorr', 'abtwo'):
                x = x + (t - memory['lastt'])*(y - 2*x)
                memory = dict(x=x, lastt=t)
            return x*bdir, memory

        def f_tvdp(t, vc):
            return 1e-2*np.cos(5*t)*(M @ vc)
        kw.update(f_tdp=f_tdp, g_tdp=g_tdp, dynamic_rhs=dynamic_rhs,
                  dynamic_rhs_memory={}, f_tvdp=f_tvdp)
    else:
        kw.update(f_tdp=lambda t: cfv, g_tdp=lambda t: cfp)

    rec = Recorder()
    kw.update(savevp=rec)
    return kw, rec, dict(prob=prob, cfv=cfv, cfp=cfp)
