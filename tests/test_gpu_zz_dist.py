"""Row-partitioned solve on the GPU (per-rank row blocks of K, Fh^-1,
J Fh^-1; halo exchange by index lists; one all-reduce per Arnoldi step).

 * one rank, RCCL communicator (world size 1): every RCCL call of the
   distributed code path runs and the answer equals the plain solve;
 * two ranks sharing ONE GPU through the gloo-staged communicator: the real
   HIP kernels on rank > 0 row blocks with halo exchanges in between, against
   the CPU oracle and against the single-rank run -- on the toy system and on
   BASELINE config 4's mesh (`cylinder_3`: NV=19 468, NP=2 592,
   tests/time_dep_nse_krylov.py); per-rank matrix storage < 0.6 x serial.
(RCCL itself refuses two ranks on one device, and a gpurun box has one GPU.)
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    import scenarios
    prob = scenarios.toy_problem()
    M, A, J = (prob['smc'][k] for k in 'MAJ')
    dt = 5e-3
    F = (M + .5*dt*A).tocsr()
    R1 = (M - .5*dt*A).tocsr()
    rng = np.random.default_rng(11)
    NP, NV = J.shape
    rhsv = M @ rng.standard_normal(NV)
    rhsp = 1e-3*rng.standard_normal(NP)
    return dict(M=M, A=A, J=J, F=F, R1=R1, rhsv=rhsv, rhsp=rhsp, dt=dt,
                v0=rng.standard_normal(NV), nfc=M @ rng.standard_normal(NV))


def _mg_problem():
    """the toy mesh refined once, with the prolongation of its pressure space"""
    from dolfin_navier_scipy_amd.fem import (
        get_sysmats, channel_cylinder_mesh, refine_uniform, TaylorHood,
        pressure_prolongations)
    coarse = channel_cylinder_mesh()
    fine, parents = refine_uniform(coarse)
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N='toy', Re=40.,
                                 mesh=fine)
    prols = pressure_prolongations([femp['V'], TaylorHood(coarse)],
                                   [parents, None])
    M, A, J = sm['M'], sm['A'], sm['J']
    dt = 5e-3
    rng = np.random.default_rng(11)
    NP, NV = J.shape
    return dict(M=M, A=A, J=J, F=(M + .5*dt*A).tocsr(),
                R1=(M - .5*dt*A).tocsr(), rhsv=M @ rng.standard_normal(NV),
                rhsp=1e-3*(J @ rng.standard_normal(NV)), dt=dt,
                v0=rng.standard_normal(NV), nfc=M @ rng.standard_normal(NV),
                prols=prols)


def _mg3_problem():
    """the toy mesh refined twice: three multigrid levels once the dense
    coarsest level is capped at 300 dofs (DNS_MG_DENSE_MAX)"""
    from dolfin_navier_scipy_amd.fem import (
        get_sysmats, channel_cylinder_mesh, refine_uniform, TaylorHood,
        pressure_prolongations)
    coarse = channel_cylinder_mesh()
    mid, par1 = refine_uniform(coarse)
    fine, par2 = refine_uniform(mid)
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N='toy', Re=40.,
                                 mesh=fine)
    prols = pressure_prolongations(
        [femp['V'], TaylorHood(mid), TaylorHood(coarse)], [par2, par1, None])
    M, A, J = sm['M'], sm['A'], sm['J']
    dt = 2.5e-3
    rng = np.random.default_rng(11)
    NP, NV = J.shape
    return dict(M=M, A=A, J=J, F=(M + .5*dt*A).tocsr(),
                R1=(M - .5*dt*A).tocsr(), rhsv=M @ rng.standard_normal(NV),
                rhsp=1e-3*(J @ rng.standard_normal(NV)), dt=dt,
                v0=rng.standard_normal(NV), nfc=M @ rng.standard_normal(NV),
                prols=prols)


def _cyl3_problem():
    """config 4: cylinder wake on `cylinder_3`, Re=40, dt = 0.5/256
    (`tests/time_dep_nse_krylov.py:52`)"""
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=3, Re=40)
    M, A, J = sm['M'], sm['A'], sm['J']
    dt = 0.5/256
    rng = np.random.default_rng(11)
    NP, NV = J.shape
    return dict(M=M, A=A, J=J, F=(M + .5*dt*A).tocsr(),
                R1=(M - .5*dt*A).tocsr(), rhsv=M @ rng.standard_normal(NV),
                rhsp=1e-3*(J @ rng.standard_normal(NV)), dt=dt,
                v0=1e-2*rng.standard_normal(NV),
                nfc=M @ rng.standard_normal(NV), femp=femp)


def _solve_and_step(sad, comm, fhat, reorth, fact='triangular', info=None):
    # 'mgpart' / 'mg3part': the multigrid levels row-partitioned too (every
    # level but the dense coarsest one; the knobs are read when the handle is
    # created)
    knobs = {}
    if fhat.endswith('_lazy') or fhat.endswith('_nolazy'):
        # every solve starts with a ONE-step cycle: with 'dist_lazy1' (default)
        # its first vector stays un-normalised and the norms of r and b are
        # all-reduced together with the step's dots; later cycles of the solve
        # are the general ones
        knobs['DNS_CYCLE_FIRST'] = '1'
        if fhat.endswith('_nolazy'):
            knobs['DNS_DIST_LAZY1'] = '0'
        fhat = fhat[:fhat.rindex('_')]
    from_rows = fhat.endswith('_loc')
    if from_rows:
        # rank-local construction: the handle is created from this rank's rows
        # of F, JT and J only (`dns_saddle_create_rows`)
        fhat = fhat[:-4]
    if fhat.endswith('_rep'):
        # the set-up replicated (every rank forms every row) instead of
        # partitioned (the default with a communicator)
        knobs['DNS_PART_SETUP'] = '0'
        fhat = fhat[:-4]
    if fhat in ('mg', 'mgpart', 'mg3', 'mg3part', 'mg3parts'):
        # ONE V-cycle per application: what a row-partitioned solve runs
        # (`mg_cycles_eff`); the single-rank runs it is compared with run the
        # same block, not their default of two cycles
        knobs['DNS_MG_CYCLES'] = '1'
    if fhat in ('mgpart', 'mg3part', 'mg3parts'):
        knobs['DNS_MG_PART_MIN'] = '0'
    if fhat in ('mg3', 'mg3part', 'mg3parts'):
        knobs['DNS_MG_DENSE_MAX'] = '300'
    if fhat in ('mg', 'mgpart'):
        knobs['DNS_MG_DENSE_MAX'] = '100'      # (two levels at least)
    if fhat in ('cyl3s', 'mg3parts'):
        # the row blocks through the streaming kernels (bandwidth regime):
        # K in the pair format, J Fh^-1 and the level operators as fp32 streams
        knobs['DNS_STREAM_NNZ'] = '1'
    three = fhat in ('mg3', 'mg3part', 'mg3parts')
    mgs = fhat in ('mg', 'mgpart', 'mg3', 'mg3part', 'mg3parts')
    cyl3 = fhat in ('cyl3', 'cyl3s')
    pr = (_mg3_problem() if three else _mg_problem()) if mgs else (
        _cyl3_problem() if cyl3 else _problem())
    if from_rows:
        system = sad.SaddleSystem.from_rows_of(pr['F'], pr['J'], comm)
    else:
        system = sad.SaddleSystem(pr['F'], pr['J'])
    for k, v in knobs.items():        # per-handle options, no process state
        system.set_option(k[len('DNS_'):].lower(), float(v))
    if comm is not None and not from_rows:
        system.set_comm(comm)
    if cyl3:
        fhat, fact = 'explicit', 'full'
        degree = 6
    if fhat == 'full':          # full block factorisation (explicit Fh^-1)
        fhat, fact = 'explicit', 'full'
    if mgs:                     # multigrid Schur block + full factorisation
        system.set_schur_mg(pr['prols'])
        fhat, fact = 'explicit', 'full'
    system.setup_precond(cheb_degree=6 if cyl3 else 4,
                         schur='mg' if mgs else 'dense', fhat=fhat,
                         factorization=fact, drop_tol=1e-3 if cyl3 else None)
    if info is not None:
        info['matrix_bytes'] = system.device_matrix_bytes()
        info['precond'] = system.precond_info()
    x = system.solve(pr['rhsv'], pr['rhsp'], rtol=1e-12, reorth=reorth)
    stats = dict(system.last_stats)
    if info is not None and info.get('revalue'):
        # new values of F (the same ones) through `dns_saddle_update_values` --
        # the values of the own rows on a handle created from rows -- and the
        # set-up once more: the same answer
        system.update_values(system._f.data.copy())
        system.setup_precond(cheb_degree=6 if cyl3 else 4,
                             schur='mg' if mgs else 'dense', fhat=fhat,
                             factorization=fact,
                             drop_tol=1e-3 if cyl3 else None)
        x2 = system.solve(pr['rhsv'], pr['rhsp'], rtol=1e-12, reorth=reorth)
        assert np.array_equal(x, x2)
    # a few device-resident CNAB steps through the same communicator
    stepper = sad.ImexStepper(system, pr['R1'])
    dt = pr['dt']
    cvop = None
    if 'femp' in pr:
        # the convection on the device too: in the partitioned run every rank
        # evaluates only the cells that touch its rows
        from dolfin_navier_scipy_amd import convection
        femp = pr['femp']
        cvop = convection.ConvectionP2.from_taylor_hood(
            femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
        nfc = cvop.apply(pr['v0'].reshape((-1, 1)), scale=-1.0)
        stepper.set_state(pr['v0'], nfc_c=nfc, nfc_o=nfc)
        stepper.set_convection(cvop, scale=-1.0)
    else:
        stepper.set_state(pr['v0'], nfc_c=pr['nfc'], nfc_o=pr['nfc'])
    stepper.set_rhs(dt*pr['rhsv'], pr['rhsp'])
    cf = sad.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                pscale=-1./dt)
    opts = sad.solve_opts(rtol=1e-12, reorth=reorth, use_graph=True)
    stepper.run(5, cf, opts)
    v, p = stepper.get_state()
    stepper.close()
    if cvop is not None:
        cvop.close()
    system.close()
    return x, stats, v, p


def system_levels(info):
    return info['precond'].get('mg_levels', [])


CFG4_NTS = 256


def _config4_setup():
    """BASELINE config 4 (`tests/time_dep_nse_krylov.py:4-7,52`): cylinder
    wake on `cylinder_3`, Re=40, tE=0.5, Nts=256, Stokes start"""
    from dolfin_navier_scipy_amd.fem import get_sysmats
    from oracle import saddle_oracle
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=3, Re=40)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    th, inv = femp['V'], femp['invinds']
    dbcinds, dbcvals = femp['dbcinds'], femp['dbcvals']
    vp0 = saddle_oracle.solve_sadpnt_smw(amat=A, jmat=J, rhsv=rhsd['fv'],
                                         rhsp=rhsd['fp'])
    inivel, inip = vp0[:NV], -vp0[NV:]

    def appnd(vvec, bcs):
        full = np.full((th.vdim, 1), np.nan)
        full[inv] = vvec
        full[dbcinds, 0] = dbcvals
        return full

    def f_vdp(vf):
        return -th.convection_vec(vf)[inv, :]

    def make_kw(rec, nts=CFG4_NTS):
        return dict(trange=np.linspace(0, .5, CFG4_NTS + 1)[:nts + 1],
                    inivel=inivel, inip=inip, bcs_ini=[], M=M, A=A, J=J,
                    f_vdp=f_vdp, f_tdp=lambda t: rhsd['fv'],
                    g_tdp=lambda t: rhsd['fp'], scalep=-1.,
                    getbcs=lambda t, v, p, mode=None: [],
                    applybcs=lambda b: (0., 0., 0.), appndbcs=appnd,
                    savevp=rec, check_ff_maxv=1e8)
    return femp, sm, rhsd, make_kw


def _config4_partitioned(sad, comm):
    """Heun start on the host (oracle arithmetic, identical on every rank),
    then the remaining 255 CNAB steps device resident on the row-partitioned
    system with the device convection"""
    import scenarios
    from oracle import imex_oracle
    from dolfin_navier_scipy_amd import convection
    femp, sm, rhsd, make_kw = _config4_setup()
    M, A, J = sm['M'], sm['A'], sm['J']
    dt = .5/CFG4_NTS
    kw = make_kw(scenarios.Recorder(), nts=2)
    (v1, p1, _, _, _, _, _, nfc0, nfc1, _, _) = imex_oracle.heun_start(
        vc=kw['inivel'], pc=kw['inip'], tc=0., tn=dt, M=M, A=A, J=J,
        scalep=-1., dfv_c=0., dynamic_rhs=lambda t, vc=None, memory={},
        mode=None: (np.zeros_like(kw['inivel']), memory), drm={}, bcs_c=[],
        applybcs=kw['applybcs'], appndbcs=kw['appndbcs'], getbcs=kw['getbcs'],
        f_tdp=kw['f_tdp'], f_vdp=kw['f_vdp'], g_tdp=kw['g_tdp'])
    system = sad.SaddleSystem((M + .5*dt*A).tocsr(), J)
    if comm is not None:
        system.set_comm(comm)
    system.setup_precond(cheb_degree=6, schur='dense', fhat='explicit',
                         factorization='full', drop_tol=1e-3)
    stp = sad.ImexStepper(system, (M - .5*dt*A).tocsr())
    stp.set_state(v1, ptilde_c=-dt*p1, nfc_c=nfc0)
    stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
    cvop = convection.ConvectionP2.from_taylor_hood(
        femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
    stp.set_convection(cvop, scale=-1.0)
    cf = sad.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                pscale=-1./dt, extrapolate=4)
    opts = sad.solve_opts(rtol=1e-13, maxiter=400, reorth=2, use_graph=True)
    _, its, _ = stp.run(CFG4_NTS - 1, cf, opts)
    v, p = stp.get_state()
    stp.close()
    cvop.close()
    system.close()
    return v, p, its


def test_rccl_world_size_one_equals_plain_solve():
    from dolfin_navier_scipy_amd import saddle, comm as dcomm, _capi
    assert _capi.device_count() > 0
    x0, st0, v0, p0 = _solve_and_step(saddle, None, 'explicit', True)
    cm = dcomm.Comm.rccl(0, 1, 0, dcomm.rccl_unique_id())
    for fhat in ('explicit', 'full', 'full_loc'):
        x1, st1, v1, p1 = _solve_and_step(saddle, cm, fhat, True)
        assert st1['status'] == 0 and st1['true_relres'] <= 5e-12
        assert np.linalg.norm(x1 - x0) <= 1e-9*np.linalg.norm(x0)
        assert np.linalg.norm(v1 - v0) <= 1e-9*np.linalg.norm(v0)
    calls = cm.stats()
    assert calls['allreduce'] > 0 and calls['allgatherv'] > 0
    assert calls['halo_exchange'] > 0 and calls['halo_bytes'] == 0  # 1 rank
    # the recurrence form of Fh^-1 has no row-partitioned path
    with pytest.raises(_capi.DnsError):
        _solve_and_step(saddle, cm, 'cheb', True)
    # construction from rows: rows that do not fit the partition are refused
    # (by all ranks together), and so are the set-ups that need whole matrices
    pr = _problem()
    NP, NV = pr['J'].shape
    JT = pr['J'].T.tocsr()
    with pytest.raises(_capi.DnsError):
        saddle.SaddleSystem.from_rows(pr['F'][:-2, :], JT[:-2, :], pr['J'],
                                      NV, NP, cm)
    rows = saddle.SaddleSystem.from_rows_of(pr['F'], pr['J'], cm)
    with pytest.raises(_capi.DnsError):
        rows.setup_precond(cheb_degree=4, schur='jacobi', fhat='explicit')
    with pytest.raises(_capi.DnsError):
        rows.set_comm(None)
    kept, _ = rows.host_matrix_bytes()
    assert kept > 0
    rows.close()
    cm.close()


def test_rccl_selftest_and_the_all_gather_forms_on_one_rank():
    """`dns_comm_selftest` (the first-contact leg of `bench.py --gpus N`):
    every primitive of the partitioned path, plain and captured in a hipGraph,
    checked entry by entry -- and the all-gather as ONE ncclAllGather: in
    place for equal blocks, through the staging buffer + unpack kernel
    otherwise (forced here: one rank has no unequal blocks), against the
    group of broadcasts it replaces"""
    from dolfin_navier_scipy_amd import saddle, comm as dcomm, _capi
    cm = dcomm.Comm.rccl(0, 1, 0, dcomm.rccl_unique_id())
    for form in ('auto', 'staged', 'bcast'):
        cm.set_gather_form(form)
        for name in cm.SELFTEST_PRIMITIVES:
            for graph in (False, True):
                one = cm.selftest(name, graph=graph, count=5, reps=10)
                assert one['ok'], (form, name, graph)
                assert 0.0 < one['us_per_call'] < 1e5, one
    forms = cm.gather_forms()
    assert min(forms.values()) > 0, forms
    # the same solve whichever way the blocks travel
    ref = None
    for form in ('auto', 'staged', 'bcast'):
        cm.set_gather_form(form)
        x, st, v, p = _solve_and_step(saddle, cm, 'full', True)
        assert st['status'] == 0
        if ref is None:
            ref = (x, v)
        assert np.array_equal(x, ref[0]) and np.array_equal(v, ref[1])
    with pytest.raises(_capi.DnsError):
        cm.selftest(7)
    cm.close()


def test_one_step_cycles_carry_the_norms_with_the_dots():
    """pipelined CNAB steps of the bench workload on one RCCL rank: once the
    warm start is good every solve is ONE Krylov step, and with 'dist_lazy1'
    the norms of r and b are all-reduced together with that step's dots --
    two all-reduces of scalars per time step become one"""
    from dolfin_navier_scipy_amd import saddle as sad, comm as dcomm
    from dolfin_navier_scipy_amd import convection
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=100.)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    dt = 1./512
    out = {}
    for lazy in (1, 0):
        cm = dcomm.Comm.rccl(0, 1, 0, dcomm.rccl_unique_id())
        system = sad.SaddleSystem((M + .5*dt*A).tocsr(), J)
        system.set_option('dist_lazy1', lazy)
        system.set_comm(cm)
        system.setup_precond(cheb_degree=6, schur='dense', fhat='explicit',
                             factorization='full', drop_tol=1e-3)
        cvop = convection.ConvectionP2.from_taylor_hood(
            femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
        v0 = np.zeros((NV, 1))
        nfc = cvop.apply(v0, scale=-1.0)
        stp = sad.ImexStepper(system, (M - .5*dt*A).tocsr())
        stp.set_state(v0, nfc_c=nfc, nfc_o=nfc)
        stp.set_rhs(dt*rhsd['fv'], rhsd['fp'])
        stp.set_convection(cvop, scale=-1.0)
        cf = sad.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                    pscale=-1./dt, extrapolate=4)
        opts = sad.solve_opts(rtol=1e-10, maxiter=400, reorth=2,
                              use_graph=True)
        stp.run(256, cf, opts)               # past the start-up transient
        c0 = cm.stats()
        _, its, lst = stp.run(128, cf, opts)
        c1 = cm.stats()
        v, p = stp.get_state()
        out[lazy] = (v, p, its, c1['allreduce'] - c0['allreduce'],
                     lst['true_relres'], dict(stp.last_run))
        stp.close()
        cvop.close()
        system.set_comm(None)
        cm.close()
        system.close()
    (vl, pl, il, arl, trl, recl), (vn, pn, inn, arn, trn, recn) = out[1], out[0]
    print('128 steps: Krylov steps', il, '/', inn, 'all-reduces', arl, '/', arn,
          recl)
    assert trl <= 1e-9 and trn <= 1e-9
    assert np.linalg.norm(vl - vn) <= 1e-8*np.linalg.norm(vn)
    assert np.linalg.norm(pl - pn) <= 1e-6*np.linalg.norm(pn)
    assert abs(il - inn) <= 8
    # one rank, dense Schur block by rows: per one-step solve 2 all-reduces
    # (norms, dots) become 1
    assert arl <= arn - 64, (arl, arn)


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME='lo')   # loopback, deterministically
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(100, exit=True)   # never hang a GPU box
    dist.init_process_group('gloo', rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60))
    from dolfin_navier_scipy_amd import saddle, comm as dcomm
    cm = dcomm.Comm.gloo(0)
    out = {}
    for fhat, reorth in (('explicit', False), ('full', False), ('mg', False),
                         ('mgpart', False), ('mg3', False), ('mg3part', False),
                         ('mg3parts', False), ('cyl3', False),
                         ('cyl3s', False), ('full_lazy', 2),
                         ('full_nolazy', 2), ('mg3part_lazy', 2),
                         ('full_rep', False),
                         ('mg3part_rep', False), ('cyl3_rep', False),
                         ('full_loc', False), ('mg3part_loc', False),
                         ('cyl3_loc', False)):
        info = {}
        before = cm.stats()
        x, st, v, p = _solve_and_step(saddle, cm, fhat, reorth, info=info)
        after = cm.stats()
        out[fhat] = (x, v, p, st['iters'], st['true_relres'],
                     info['matrix_bytes'],
                     after['halo_bytes'] - before['halo_bytes'],
                     after['halo_exchange'] - before['halo_exchange'],
                     after['allreduce'] - before['allreduce'])
    np.savez(os.path.join(outdir, 'rank{0}.npz'.format(rank)),
             **{'{0}_{1}'.format(k, i): np.asarray(val)
                for k, tup in out.items() for i, val in enumerate(tup)})
    cm.close()
    dist.destroy_process_group()


def _worker_rows(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME='lo')
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(100, exit=True)
    dist.init_process_group('gloo', rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60))
    from dolfin_navier_scipy_amd import saddle, comm as dcomm
    cm = dcomm.Comm.gloo(0)
    out = {}
    for fhat in ('full', 'full_loc', 'mg3part', 'mg3part_loc', 'mg3part_rep'):
        info = {'revalue': True}
        x, st, v, p = _solve_and_step(saddle, cm, fhat, False, info=info)
        out[fhat] = (x, v, p, st['iters'], st['true_relres'])
    np.savez(os.path.join(outdir, 'rank{0}.npz'.format(rank)),
             **{'{0}_{1}'.format(k, i): np.asarray(val)
                for k, tup in out.items() for i, val in enumerate(tup)})
    cm.close()
    dist.destroy_process_group()


def test_construction_from_rows_over_three_ranks(tmp_path):
    """`dns_saddle_create_rows` / `dns_imex_create_rows` over THREE ranks
    sharing the GPU (the middle rank has neighbours on both sides, ghost rows
    come from two owners): solutions and time steps equal, bit for bit, the
    ones of handles created from whole matrices on the same communicator"""
    from spawn_util import spawn_ranks
    spawn_ranks(_worker_rows, 3, str(tmp_path))
    rs = [np.load(tmp_path / 'rank{0}.npz'.format(r)) for r in range(3)]
    for key in ('full', 'mg3part'):
        for rr in rs:
            assert float(rr[key + '_loc_4']) <= 5e-12
            for i in (0, 1, 2):
                assert np.array_equal(rr[key + '_%d' % i],
                                      rr[key + '_loc_%d' % i]), (key, i)
                assert np.array_equal(rr[key + '_%d' % i],
                                      rs[0][key + '_%d' % i]), (key, i)
            assert int(rr[key + '_3']) == int(rr[key + '_loc_3'])
    # the multigrid levels formed by rows (mg_rows.inc) == cut out of the whole
    # hierarchy every rank builds in the replicated set-up
    for rr in rs:
        for i in (0, 1, 2):
            assert np.array_equal(rr['mg3part_%d' % i],
                                  rr['mg3part_rep_%d' % i]), i


def test_two_ranks_one_gpu_gloo_staged(tmp_path):
    from dolfin_navier_scipy_amd import saddle
    from oracle import saddle_oracle
    from spawn_util import spawn_ranks
    spawn_ranks(_worker, 2, str(tmp_path))
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / 'rank1.npz')
    pr = _problem()
    ref = saddle_oracle.solve_sadpnt_smw(amat=pr['F'], jmat=pr['J'],
                                         rhsv=pr['rhsv'],
                                         rhsp=pr['rhsp']).reshape(-1)
    NV = pr['F'].shape[0]
    xs, sts, vs, ps = _solve_and_step(saddle, None, 'explicit', False)
    xf, stf, _, _ = _solve_and_step(saddle, None, 'full', False)
    for fhat in ('explicit', 'full'):
        x_a, x_b = r0[fhat + '_0'], r1[fhat + '_0']
        # both ranks end with the same full iterate ...
        assert np.array_equal(x_a, x_b)
        # ... which is the oracle's solution
        assert float(r0[fhat + '_4']) <= 5e-12
        assert np.linalg.norm(x_a[:NV] - ref[:NV]) <= 1e-9*np.linalg.norm(ref[:NV])
        assert np.linalg.norm(x_a[NV:] - ref[NV:]) <= 1e-7*np.linalg.norm(ref[NV:])
        # time stepping through the communicator == single-rank stepping
        assert np.array_equal(r0[fhat + '_1'], r1[fhat + '_1'])
        assert np.linalg.norm(r0[fhat + '_1'] - vs) <= 1e-9*np.linalg.norm(vs)
        assert np.linalg.norm(r0[fhat + '_2'] - ps) <= 1e-7*np.linalg.norm(ps)
    assert abs(int(r0['explicit_3']) - sts['iters']) <= 3
    # multigrid Schur block (replicated per rank) on the refined toy mesh
    xm, stm, vm, pm = _solve_and_step(saddle, None, 'mg', False)
    assert np.array_equal(r0['mg_0'], r1['mg_0'])
    assert float(r0['mg_4']) <= 5e-12
    assert np.linalg.norm(r0['mg_0'] - xm) <= 1e-9*np.linalg.norm(xm)
    assert np.linalg.norm(r0['mg_1'] - vm) <= 1e-9*np.linalg.norm(vm)
    assert abs(int(r0['mg_3']) - stm['iters']) <= 1
    # the multigrid levels row-partitioned as well: the same iterates, and the
    # level operators are stored by row blocks
    info3 = {}
    x3, st3, v3, p3 = _solve_and_step(saddle, None, 'mg3', False, info=info3)
    for key, (xr, vr, strf) in (('mgpart', (xm, vm, stm)),
                                ('mg3part', (x3, v3, st3)),
                                ('mg3parts', (x3, v3, st3))):
        assert np.array_equal(r0[key + '_0'], r1[key + '_0'])
        assert float(r0[key + '_4']) <= 5e-12
        assert np.linalg.norm(r0[key + '_0'] - xr) <= 1e-9*np.linalg.norm(xr)
        assert np.linalg.norm(r0[key + '_1'] - vr) <= 1e-9*np.linalg.norm(vr)
        assert abs(int(r0[key + '_3']) - strf['iters']) <= 1
    assert len(system_levels(info3)) >= 3
    # (the partitioned cycle exchanges halos per level: more exchanges than the
    # replicated one, fewer bytes of matrices)
    for a, b in (('mg', 'mgpart'), ('mg3', 'mg3part')):
        assert int(r0[b + '_7']) > int(r0[a + '_7']), (a, b)
        assert int(r0[b + '_5']) < int(r0[a + '_5']), (a, b)
    print('multigrid over 2 ranks: matrix bytes per rank, replicated cycle',
          int(r0['mg3_5']), 'partitioned', int(r0['mg3part_5']), 'serial',
          info3['matrix_bytes'], '; halo exchanges', int(r0['mg3_7']), '->',
          int(r0['mg3part_7']), '; levels',
          [lv['n'] for lv in system_levels(info3)])
    for rr in (r0, r1):
        assert int(rr['mg3part_5']) < 0.62*info3['matrix_bytes'], \
            (int(rr['mg3part_5']), info3['matrix_bytes'])
    # the full block factorisation is partitioned too (one more all-gather
    # per apply) and needs the same few steps as on one GPU
    assert abs(int(r0['full_3']) - stf['iters']) <= 3
    assert int(r0['full_3']) < int(r0['explicit_3'])
    # config 4's mesh, row-partitioned over the two ranks
    info = {}
    xc, stc, vc, pc = _solve_and_step(saddle, None, 'cyl3', False, info=info)
    assert np.array_equal(r0['cyl3_0'], r1['cyl3_0'])
    assert float(r0['cyl3_4']) <= 5e-12
    assert np.linalg.norm(r0['cyl3_0'] - xc) <= 1e-9*np.linalg.norm(xc)
    assert np.linalg.norm(r0['cyl3_1'] - vc) <= 1e-9*np.linalg.norm(vc)
    assert np.linalg.norm(r0['cyl3_2'] - pc) <= 1e-7*np.linalg.norm(pc)
    assert abs(int(r0['cyl3_3']) - stc['iters']) <= 3
    # the same with the row blocks going through the streaming kernels
    assert np.array_equal(r0['cyl3s_0'], r1['cyl3s_0'])
    assert float(r0['cyl3s_4']) <= 5e-12
    assert np.linalg.norm(r0['cyl3s_0'] - xc) <= 1e-9*np.linalg.norm(xc)
    assert np.linalg.norm(r0['cyl3s_1'] - vc) <= 1e-9*np.linalg.norm(vc)
    assert np.linalg.norm(r0['cyl3s_2'] - pc) <= 1e-7*np.linalg.norm(pc)
    assert abs(int(r0['cyl3s_3']) - stc['iters']) <= 3
    # one-step first cycles with the norms travelling behind the step: the
    # same answers with fewer all-reduces
    for key, ref in (('full', 'full'), ('mg3part', 'mg3part')):
        lz = key + '_lazy'
        assert np.array_equal(r0[lz + '_0'], r1[lz + '_0'])
        assert float(r0[lz + '_4']) <= 5e-12
        for i, tol in ((0, 1e-9), (1, 1e-9), (2, 1e-7)):
            a, b = r0[lz + '_%d' % i], r0[ref + '_%d' % i]
            assert np.linalg.norm(a - b) <= tol*np.linalg.norm(b), (lz, i)
    # (the counts of collectives: test_one_step_cycles_* below)
    # partitioned set-up (every rank forms its rows of Fh^-1, J Fh^-1 and of
    # the Schur complement only) == replicated set-up, bit for bit
    for key in ('full', 'mg3part', 'cyl3'):
        for i in (0, 1, 2):
            assert np.array_equal(r0[key + '_%d' % i],
                                  r0[key + '_rep_%d' % i]), (key, i)
        assert int(r0[key + '_3']) == int(r0[key + '_rep_3'])
    # rank-local construction (every rank hands over its rows of F, JT, J
    # only; ghost rows fetched ring by ring, spectral bounds by rows) == the
    # handle created from whole matrices, bit for bit
    for key in ('full', 'mg3part', 'cyl3'):
        for rr in (r0, r1):
            for i in (0, 1, 2):
                assert np.array_equal(rr[key + '_%d' % i],
                                      rr[key + '_loc_%d' % i]), (key, i)
            assert int(rr[key + '_3']) == int(rr[key + '_loc_3'])
            assert int(rr[key + '_loc_5']) <= int(rr[key + '_5'])
    # per-rank storage: the row blocks, not the matrices
    for rr in (r0, r1):
        assert int(rr['cyl3_5']) < 0.6*info['matrix_bytes'], \
            (int(rr['cyl3_5']), info['matrix_bytes'])
    # the halo is a halo: bytes sent per exchange << a whole velocity vector
    NVc = vc.size
    per_exchange = int(r0['cyl3_6'])/max(1, int(r0['cyl3_7']))
    assert 0 < per_exchange < 0.35*8*NVc, per_exchange
    print('cylinder_3 over 2 ranks: matrix bytes per rank',
          int(r0['cyl3_5']), int(r1['cyl3_5']), 'serial',
          info['matrix_bytes'], '; halo bytes per exchange', per_exchange,
          'of', 8*NVc, '; all-reduces', int(r0['cyl3_8']), 'for',
          int(r0['cyl3_3']), 'Krylov steps + 5 time steps')


def test_config4_256_steps_partitioned_over_two_ranks(tmp_path):
    """config 4's own horizon (Nts = 256, `tests/time_dep_nse_krylov.py:52`)
    on `cylinder_3`, row-partitioned over two ranks (gloo-staged on one GPU):
    velocity AND pressure within 1e-8 of the oracle's factor-once
    CNAB loop; both ranks bitwise equal"""
    import scenarios
    from oracle import imex_oracle
    from spawn_util import spawn_ranks
    spawn_ranks(_worker_cfg4, 2, str(tmp_path))
    r0 = np.load(tmp_path / 'cfg4_rank0.npz')
    r1 = np.load(tmp_path / 'cfg4_rank1.npz')
    assert np.array_equal(r0['v'], r1['v']) and np.array_equal(r0['p'], r1['p'])
    femp, sm, rhsd, make_kw = _config4_setup()
    vo, po, ff = imex_oracle.cnab(**make_kw(scenarios.Recorder()))
    assert ff == 0
    M = sm['M']
    mn = lambda x: float(np.sqrt((x.T @ (M @ x)).item()))
    ev = mn(r0['v'].reshape((-1, 1)) - vo)/mn(vo)
    ep = np.linalg.norm(r0['p'].reshape((-1, 1)) - po)/np.linalg.norm(po)
    print('config 4, 256 steps over 2 ranks: v', ev, 'p', ep, 'Krylov steps',
          int(r0['its']), 'halo exchanges', int(r0['halo']), 'all-reduces',
          int(r0['allreduce']))
    assert ev <= 1e-8, ev
    assert ep <= 1e-8, ep
    assert int(r0['halo']) > 0 and int(r0['allreduce']) > 0


def _worker_cfg4(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME='lo')
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(200, exit=True)   # never hang a GPU box
    dist.init_process_group('gloo', rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    from dolfin_navier_scipy_amd import saddle, comm as dcomm
    cm = dcomm.Comm.gloo(0)
    before = cm.stats()
    v4, p4, its4 = _config4_partitioned(saddle, cm)
    after = cm.stats()
    np.savez(os.path.join(outdir, 'cfg4_rank{0}.npz'.format(rank)), v=v4, p=p4,
             its=its4, halo=after['halo_exchange'] - before['halo_exchange'],
             allreduce=after['allreduce'] - before['allreduce'])
    cm.close()
    dist.destroy_process_group()


def _trap_sweeps(comm):
    """one Picard and two Newton sweeps of the toy problem (snu:1304-1587)
    through the device stepper, optionally on a row-partitioned handle"""
    import scenarios
    import test_gpu_newton_picard as tnp
    from dolfin_navier_scipy_amd import convection, saddle
    from dolfin_navier_scipy_amd import newton_picard as dnp
    s = tnp._setup(scenarios.toy_problem())
    tr = s['trange']
    cv = convection.ConvectionP2.from_taylor_hood(s['th'], s['inv'],
                                                  s['dbcinds'], s['dbcvals'])
    stp = dnp.TrapezoidalStepper(s['M'], s['A'], s['J'], cv, nslots=tr.size,
                                 dt=tr[1] - tr[0], comm=comm,
                                 precond=dict(cheb_degree=4,
                                              factorization='full'))
    stp.set_rhs(s['fv'], s['fp'])
    opts = saddle.solve_opts(rtol=1e-12, maxiter=400, use_graph=True)
    c0 = comm.stats() if comm is not None else None
    vd, pd, hist = dnp.newton_picard(stp, tr, s['iniv'], s['lin0'],
                                     vel_pcrd_stps=1, vel_nwtn_stps=2,
                                     opts=opts)
    if comm is not None:
        c1 = comm.stats()
        # all-gathers of the three sweeps and how many time steps they made
        s['sweep_gathers'] = c1['allgatherv'] - c0['allgatherv']
        s['sweep_steps'] = 3*(tr.size - 1)
        s['nslots'] = tr.size
    # ... and three steps with the low-rank feedback terms of
    # `_get_mats_rhs_ts` (snu:1036-1042: `F - dt/2 U V_n` by Sherman-Morrison-
    # Woodbury; every Woodbury column is one more partitioned solve)
    rng = np.random.default_rng(21)
    NV = s['NV']
    umat = 1e-1*(s['M'] @ rng.standard_normal((NV, 2)))
    vmats = [rng.standard_normal((2, NV))/np.sqrt(NV) for _ in tr]
    for k, t in enumerate(tr):
        stp.write_linpoint(0, k, s['lin0'][t])
    stp.start(s['iniv'], newton=True)
    osync = saddle.solve_opts(rtol=1e-12, maxiter=400, use_graph=False,
                              reorth=1)
    dt = tr[1] - tr[0]
    for k in range(1, 4):
        stp.step(dt, 0, k, k, True, opts=osync,
                 feedback=(umat, vmats[k-1], vmats[k]))
    s['fb_state'] = stp.state()
    # a preconditioner REFRESH about the operator of the step that ran last.
    # A partitioned assembly leaves only the rank's own rows of F current on
    # its device; the set-up reads rings of rows beyond them (polynomial,
    # bounds, skew radius): they are gathered from their owners first, so
    # that every rank sets up from the same, whole matrix
    stp.refresh_precond()
    s['refresh_bounds'] = np.array(stp.system.cheb_bounds(), dtype=float)
    its = []
    for k in range(4, min(7, tr.size)):
        st = stp.step(dt, 0, k, k, True, opts=osync)
        its.append(st['iters'])
    s['refresh_iters'] = np.array(its)
    s['refresh_state'] = stp.state()
    stp.close()
    cv.close()
    tl = tr[-1]
    return vd[tl], pd[tl], np.array([h[1] for h in hist]), s


def _worker_trap(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME='lo')
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(200, exit=True)   # never hang a GPU box
    dist.init_process_group('gloo', rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    from dolfin_navier_scipy_amd import comm as dcomm
    cm = dcomm.Comm.gloo(0)
    before = cm.stats()
    v, p, hist, sw = _trap_sweeps(cm)
    after = cm.stats()
    np.savez(os.path.join(outdir, 'trap_rank{0}.npz'.format(rank)), v=v, p=p,
             fbv=sw['fb_state'][0], fbp=sw['fb_state'][1],
             rbounds=sw['refresh_bounds'], rits=sw['refresh_iters'],
             rv=sw['refresh_state'][0], hist=hist, halo=after['halo_exchange'] - before['halo_exchange'],
             gathers=after['allgatherv'] - before['allgatherv'],
             sweep_gathers=sw['sweep_gathers'], sweep_steps=sw['sweep_steps'],
             nslots=sw['nslots'])
    cm.close()
    dist.destroy_process_group()


def test_newton_picard_sweeps_on_a_partitioned_handle(tmp_path):
    """the trapezoidal Newton/Picard stepper row-partitioned over two ranks,
    ASSEMBLY included (a rank evaluates the cells that touch its rows, gathers
    the non-zeros and right-hand sides of its rows, takes its share of the
    update norm; no solution is gathered per step): same sweeps, update norms
    and iterates as on one GPU and as the oracle's restatement"""
    from oracle import newton_picard_oracle as npo
    from spawn_util import spawn_ranks
    spawn_ranks(_worker_trap, 2, str(tmp_path))
    r0 = np.load(tmp_path / 'trap_rank0.npz')
    r1 = np.load(tmp_path / 'trap_rank1.npz')
    assert np.array_equal(r0['v'], r1['v']) and np.array_equal(r0['p'], r1['p'])
    v1, p1, hist1, s = _trap_sweeps(None)
    assert np.linalg.norm(r0['v'] - v1) <= 1e-9*np.linalg.norm(v1)
    assert np.linalg.norm(r0['p'] - p1) <= 1e-8*np.linalg.norm(p1)
    assert np.allclose(r0['hist'], hist1, rtol=1e-5, atol=1e-16)
    # low-rank feedback on the partitioned handle == on one GPU
    assert np.array_equal(r0['fbv'], r1['fbv'])
    fv1, fp1 = s['fb_state']
    assert np.linalg.norm(r0['fbv'] - fv1) <= 1e-9*np.linalg.norm(fv1)
    assert np.linalg.norm(r0['fbp'] - fp1) <= 1e-8*np.linalg.norm(fp1)
    # the refreshed preconditioner: the same on both ranks and on one GPU
    # (bounds of the WHOLE current F), hence the same Krylov step counts
    assert np.array_equal(r0['rbounds'], r1['rbounds'])
    assert np.allclose(r0['rbounds'], s['refresh_bounds'], rtol=1e-12)
    assert np.array_equal(r0['rits'], r1['rits'])
    assert np.array_equal(r0['rits'], s['refresh_iters']), \
        (r0['rits'], s['refresh_iters'])
    assert np.array_equal(r0['rv'], r1['rv'])
    rv1 = s['refresh_state'][0]
    assert np.linalg.norm(r0['rv'] - rv1) <= 1e-9*np.linalg.norm(rv1)
    tr = s['trange']
    lin_full = {t: s['appnd'](v) for t, v in s['lin0'].items()}
    vo, po, _ = npo.newton_picard(
        tr, s['iniv'], lin_full, vel_pcrd_stps=1, vel_nwtn_stps=2,
        invinds=s['inv'], M=s['M'], A=s['A'], J=s['J'], fv=s['fv'],
        fp=s['fp'], conv=s['conv'], appndbcs=s['appnd'])
    tl = tr[-1]
    assert np.linalg.norm(r0['v'] - vo[tl]) <= 1e-8*np.linalg.norm(vo[tl])
    assert np.linalg.norm(r0['p'] - po[tl]) <= 1e-8*np.linalg.norm(po[tl])
    assert int(r0['halo']) > 0 and int(r0['gathers']) > 0
    # no all-gather inside the time steps of a sweep: what is left are the
    # whole trajectories (and final states) handed to the host between the
    # sweeps -- at most a few per trajectory slot, not two per time step
    print('all-gathers of three sweeps:', int(r0['sweep_gathers']), 'for',
          int(r0['sweep_steps']), 'time steps,', int(r0['nslots']), 'slots')
    assert int(r0['sweep_gathers']) <= 3*(int(r0['nslots']) + 2)
    assert int(r0['sweep_gathers']) < 2*int(r0['sweep_steps'])


def test_newton_picard_sweeps_on_one_rccl_rank():
    """the same sweeps with an RCCL communicator of size one: every RCCL call
    of the partitioned stepper runs (captured into the cycle graphs) and the
    iterates equal the plain run's"""
    from dolfin_navier_scipy_amd import comm as dcomm
    v1, p1, hist1, s1 = _trap_sweeps(None)
    cm = dcomm.Comm.rccl(0, 1, 0, dcomm.rccl_unique_id())
    v, p, hist, sc = _trap_sweeps(cm)
    calls = cm.stats()
    cm.close()
    for a, b, tol in zip(sc['fb_state'], s1['fb_state'], (1e-9, 1e-8)):
        assert np.linalg.norm(a - b) <= tol*np.linalg.norm(b)
    assert np.linalg.norm(v - v1) <= 1e-9*np.linalg.norm(v1)
    assert np.linalg.norm(p - p1) <= 1e-8*np.linalg.norm(p1)
    assert np.allclose(hist, hist1, rtol=1e-5, atol=1e-16)
    assert calls['allgatherv'] > 0 and calls['allreduce'] > 0


def _bench_two_ranks_one_gpu(extra_env, timeout=900, self_launch=False):
    """`bench.py --gpus 2` as the driver launches it, both ranks on this
    box's one GPU through the gloo-staged communicator"""
    import json
    import subprocess
    from spawn_util import free_port
    launcher = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                '--master-port', str(free_port())]
    if self_launch:       # no launcher: bench.py starts its two ranks itself
        launcher = [sys.executable]
    cmd = launcher + [os.path.join(ROOT, 'bench.py'),
                      '--gpus', '2', '--steps', '20', '--warmup', '5',
                      '--spinup', '24', '--no-bandwidth', '--no-strong',
                      '--no-ensemble', '--partitioned-timeout', '400']
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(DNS_BENCH_REHEARSE_ONE_GPU='1', GLOO_SOCKET_IFNAME='lo',
               **extra_env)
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines()
             if ln.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0]), out.stderr.decode()


def test_bench_n2_line_carries_its_parity_and_a_wrong_halo_turns_it_red():
    """the N > 1 bench line proves itself: the gathered final state of the
    row-partitioned run against an un-partitioned run of the same steps from
    the same (Stokes) state on rank 0, asserted at 1e-8 -- and a halo list
    with ONE wrong entry (`DNS_TEST_CORRUPT_HALO`) fails the leg instead of
    producing a rate (a solver that merely converges would not notice: the
    partitioned code computes its own `true_relres`)"""
    rec, _ = _bench_two_ranks_one_gpu({}, self_launch=True)
    assert rec['n_gpus'] == 2
    weak = rec['config']['weak_scaling']
    assert 'error' not in weak, weak.get('error')
    # first contact of the two (gloo-staged) ranks: the plain legs ran
    st = rec['config']['rccl_selftest']
    assert st['complete'] and st['ok'], st
    assert st['primitives']['sendrecv_ring_eager']['ok']
    assert 'skipped' in st['primitives']['allreduce_graph']
    par = rec['parity']
    assert par == weak['parity'] and par['ok'] is True
    assert par['v_rel_Mnorm'] <= 1e-8 and par['p_rel_l2'] <= 1e-8, par
    assert par['steps'] == 24 + 5 + 20
    assert 'Stokes' in weak['start_state']
    assert 'FALLBACK' not in rec['config']['parallelism']
    tm = weak['collectives_device_time']
    for kind in ('allreduce', 'halo_exchange'):
        assert tm[kind]['calls'] > 0 and tm[kind]['device_ms'] > 0.0, tm
    print('two ranks, one GPU (gloo staged): parity', par['v_rel_Mnorm'],
          par['p_rel_l2'], '; device us per call',
          {k: tm[k]['us_per_call'] for k in ('allreduce', 'halo_exchange')})
    bad, err = _bench_two_ranks_one_gpu({'DNS_TEST_CORRUPT_HALO': '1'})
    assert 'TEST HOOK' in err
    weak = bad['config']['weak_scaling']
    # (either the corrupted run no longer converges, or it converges to
    # something else: in both cases the leg is an error and the line says so)
    assert 'error' in weak, weak
    assert 'FALLBACK' in bad['config']['parallelism']
    first = weak.get('first_attempt_with_graphs', weak)
    assert 'parity' in first['error'] or 'parity' in weak['error'] \
        or 'child' in weak['error'], weak['error']


def _dtail_steps(sad, comm, tail_on, nsteps=40):
    """config 4's mesh with the convection on the device, every solve starting
    with a ONE-step cycle, a tolerance such that most steps ARE that one step:
    the lazy tail then evaluates the convection cells of the new velocity and
    the next step's front leaves its cell kernel out (`DNS_DIST_TAIL`)"""
    from dolfin_navier_scipy_amd import convection
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=3, Re=40)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    dt = 0.5/256
    os.environ['DNS_DIST_TAIL'] = '1' if tail_on else '0'
    system = sad.SaddleSystem((M + .5*dt*A).tocsr(), J)
    system.set_option('cycle_first', 1.)
    if comm is not None:
        system.set_comm(comm)
    system.setup_precond(cheb_degree=6, schur='dense', fhat='explicit',
                         factorization='full', drop_tol=1e-3)
    cvop = convection.ConvectionP2.from_taylor_hood(
        femp['V'], femp['invinds'], femp['dbcinds'], femp['dbcvals'])
    stepper = sad.ImexStepper(system, (M - .5*dt*A).tocsr())
    v0 = np.zeros((NV, 1))                       # impulsive start from rest
    nfc = cvop.apply(v0, scale=-1.0)
    stepper.set_state(v0, nfc_c=nfc, nfc_o=nfc)
    stepper.set_convection(cvop, scale=-1.0)
    stepper.set_rhs(dt*rhsd['fv'], rhsd['fp'])
    cf = sad.ImexStepper.coeffs(a_c=1., cn_c=1.5*dt, cn_o=-.5*dt,
                                pscale=-1./dt, extrapolate=4)
    opts = sad.solve_opts(rtol=1e-7, reorth=2, use_graph=True)
    its = 0
    for _ in range(nsteps):
        its += stepper.step(cf, opts=opts)['iters']
    v, p = stepper.get_state()
    counters = stepper.step_counters()
    stepper.close()
    cvop.close()
    system.close()
    os.environ.pop('DNS_DIST_TAIL', None)
    return v, p, its, counters


def _worker_dtail(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME='lo')
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(200, exit=True)   # never hang a GPU box
    dist.init_process_group('gloo', rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    from dolfin_navier_scipy_amd import saddle, comm as dcomm
    cm = dcomm.Comm.gloo(0)
    on = _dtail_steps(saddle, cm, True)
    off = _dtail_steps(saddle, cm, False)
    np.savez(os.path.join(outdir, 'dtail_rank{0}.npz'.format(rank)),
             v_on=on[0], p_on=on[1], v_off=off[0], p_off=off[1],
             its=[on[2], off[2]], c_on=on[3], c_off=off[3])
    cm.close()
    dist.destroy_process_group()


def test_convection_cells_in_the_lazy_tail_over_two_ranks(tmp_path):
    """the cells of the new velocity evaluated by the tail of a one-step cycle
    (from the copied warm start and Z_0, own rows AND halo) are the cells the
    cell kernel computes from the new solution: two ranks, tail on / off --
    bitwise the same trajectories, both ranks equal, and the single-GPU run's"""
    from dolfin_navier_scipy_amd import saddle
    from spawn_util import spawn_ranks
    spawn_ranks(_worker_dtail, 2, str(tmp_path))
    r0 = np.load(tmp_path / 'dtail_rank0.npz')
    r1 = np.load(tmp_path / 'dtail_rank1.npz')
    built, tail_cells, reused = [int(c) for c in r0['c_on']]
    print('two ranks, 40 steps: tail evaluated the cells in', tail_cells,
          'steps, the front left its cell kernel out in', reused, 'of', built,
          '; Krylov steps', list(r0['its']))
    assert tail_cells >= 10 and reused >= 10, (built, tail_cells, reused)
    assert [int(c) for c in r0['c_off']][1:] == [0, 0]
    for key in ('v_on', 'p_on', 'v_off', 'p_off'):
        assert np.array_equal(r0[key], r1[key]), key
    assert np.array_equal(r0['v_on'], r0['v_off'])
    assert np.array_equal(r0['p_on'], r0['p_off'])
    v1, p1, _, _ = _dtail_steps(saddle, None, True)
    assert np.linalg.norm(r0['v_on'] - v1) <= 1e-6*np.linalg.norm(v1)
    assert np.linalg.norm(r0['p_on'] - p1) <= 1e-5*np.linalg.norm(p1)
