#!/usr/bin/env python
"""Benchmark of the hot path: CNAB time steps of the 2D cylinder wake.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path of `time_int_utils.cnab` (reference
tiu:104-143) with everything resident in HBM: convection vector N(v)v (device
element kernel -- the reference's host FEniCS callback, tiu:113), fused
right-hand-side SpMV, block-preconditioned GMRES solve of
`[[M + dt/2 A, J^T],[J, 0]]`, pressure rescale.  `--scheme sbdf2` times the
resident SBDF2 loop (tiu:320-353) instead.

Workload at N=1: Schaefer-Turek cylinder wake, mesh level N=2 (NV=9356,
NP=1289), Re=100, dt=1/512, Taylor-Hood, CNAB -- the configuration
BASELINE.json quotes the metric on.

For N>1 the headline is ONE row-partitioned simulation:
every rank holds its row blocks of K, Fh^-1 and J Fh^-1 only, exchanges halo
entries by index lists (ncclSend/Recv) and takes part in ONE all-reduce per
Arnoldi step.  Weak scaling: the mesh grows with the ranks so that the rows per
rank stay about constant -- `WEAK_LADDER` below: level 2 (n=10 645) on 1 rank,
level 3 (n=22 060) on 2, level 2 refined once (n=43 009) on 4, level 3 refined
once (n=88 789) on 8, dt halved per refinement (and once more on the level-3
meshes: the explicit convection needs it).  `value` is the time steps/s of
that one simulation (no normalisation); `config.weak_scaling` carries n, the
rows per rank and dof-steps/s, `config.collectives` the RCCL call counts of
the timed window from `dns_comm_stats2`.  Secondary figures on the same ranks:
`config.strong_scaling` (the N=1 workload itself partitioned over the N ranks,
reported as it is -- at n ~ 1e4 a step is ~45 us of work on one GPU and every
collective costs 10-20 us, so this is SLOWER than one GPU) and
`config.ensemble` (N independent simulations, one per GPU, no collective).
`config.weak_scaling_bandwidth` is the same partitioned loop in the BANDWIDTH
regime (`BANDWIDTH_LADDER`: >= 7e5 rows per rank -- refine 3 on 1 rank, level 3
refined 3x on 2, refine 4 on 4, level 3 refined 4x on 8), whose N=1 point is
`config.weak_scaling_bandwidth_base` of the N=1 line.
The partitioned runs execute in child processes (one per rank, their own
rendezvous) under a time limit, so that a collective that never completes
leaves an error in the JSON line instead of a hung benchmark; if the headline
run fails the line says so and falls back to the ensemble figure.

The JSON line also carries
  roofline     : CSR SpMV `y = K x` (the kernel family that dominates the
                 step), HIP-event timed inside this script, at the benchmark
                 size AND on a uniformly refined mesh that leaves the caches
  cpu_baseline : the oracle's prefactored-SuperLU CNAB step (tiu:89-91,
                 125-137 restated in oracle/) timed on this box's host cores;
                 its convection callback runs on the host and is NOT timed
  parity       : final GPU iterate vs the CPU leg's after the same steps of
                 the same nonlinear trajectory
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec

# solver settings of the timed loop (tests/test_gpu_long_horizon.py runs the
# 512 steps of BASELINE config 2 with exactly these and checks the 1e-8 parity)
DEFAULTS = dict(cheb=6, rtol=1e-10, extrap=4, fp32=1, drop=1e-3, fact='full',
                reorth=2)


# (mesh level, red refinements): unknowns relative to the N=1 workload
WEAK_LADDER = [((2, 0), 1.0), ((3, 0), 2.07), ((2, 1), 4.04), ((3, 1), 8.34),
               ((2, 2), 16.2), ((3, 2), 33.5)]


# bandwidth regime: (mesh level, red refinements) with >= 7e5 rows per rank
# (n = 693k / 1.43M / 2.78M / 5.71M: 693k-715k rows per rank; every rank forms
# only its rows of the preconditioner -- on ONE rank the 5.71M mesh needs 90 s
# from start to result, 46 s of them set-up, 27 s of those the polynomial that
# is now divided among the ranks); multigrid Schur block, explicit degree-8
# polynomial from 1e6 unknowns on
BANDWIDTH_LADDER = {1: (2, 3), 2: (3, 3), 4: (2, 4), 8: (3, 4)}


def bandwidth_ladder(world):
    keys = sorted(BANDWIDTH_LADDER)
    best = min(keys, key=lambda k: abs(k - world))
    return BANDWIDTH_LADDER[best]


def weak_ladder(world):
    """the mesh whose size relative to the N=1 workload is closest to the
    number of ranks (rows per rank ~ constant)"""
    import math
    return min(WEAK_LADDER,
               key=lambda e: abs(math.log(e[1]) - math.log(world)))[0]


class stdout_to_stderr(object):
    """RCCL prints a version banner on fd 1 when a communicator is created;
    the bench contract is ONE JSON line on stdout"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


LINE_LIMIT = 7600       # bytes: the driver's record keeps an 8 KB tail


def compact_line(out, limit=LINE_LIMIT, side=None):
    """the ONE JSON line under `limit` bytes.  The published op lists (`ops`:
    what `roofline.step` is the sum of) and, if that is not enough, the
    longest explanatory strings move to a side file -- `profiles/bench_ops.json`
    next to this script (best effort: a read-only tree keeps them out) and
    `gpurun_out/bench_ops.json`; the line keeps a reference.  Returns the
    line's dict and what was moved."""
    moved = {}

    def walk(node, path):
        if isinstance(node, dict):
            for key in list(node.keys()):
                sub = path + '.' + key if path else key
                if key == 'ops' and isinstance(node[key], list):
                    moved[sub] = node[key]
                    node[key] = 'profiles/bench_ops.json#' + sub
                else:
                    walk(node[key], sub)
        elif isinstance(node, list):
            for i, item in enumerate(node):
                walk(item, '{0}[{1}]'.format(path, i))
    def rounded(node):
        # (six significant digits: a rate or a residual needs no more)
        if isinstance(node, float):
            return float('{0:.6g}'.format(node)) if node == node and \
                abs(node) != float('inf') else node
        if isinstance(node, dict):
            return {k: rounded(v) for k, v in node.items()}
        if isinstance(node, list):
            return [rounded(v) for v in node]
        return node
    out = rounded(json.loads(json.dumps(out)))     # (a copy to edit)
    walk(out, '')

    def strings(node, path, acc):
        if isinstance(node, dict):
            for key, val in node.items():
                sub = path + '.' + key if path else key
                if isinstance(val, str) and len(val) > 120 \
                        and key not in ('metric', 'error', 'stage') \
                        and 'rccl_selftest' not in sub:
                    acc.append((len(val), sub, node, key))
                else:
                    strings(val, sub, acc)
        return acc
    while len(json.dumps(out)) > limit:
        cands = sorted(strings(out, '', []), key=lambda c: c[0], reverse=True)
        # (what the bench contract names stays in the line whatever its
        # length: the workload, the CPU baseline's sample, the kernel)
        cands = [c for c in cands
                 if c[1] not in ('config.workload', 'cpu_baseline.sample',
                                 'roofline.kernel', 'config.parallelism')]
        if cands:
            _, sub, node, key = cands[0]
            moved[sub] = node[key]
            node[key] = 'bench_ops.json#' + sub
            continue
        # still too long: the largest SECONDARY record keeps its figures of
        # merit, the rest of it moves to the side file
        keep = ('steps_per_s', 'ms_per_step', 'krylov_iters_per_step',
                'parity', 'ranks', 'unknowns', 'rows_per_rank', 'error',
                'frac', 'achieved', 'true_relres_last', 'level', 'refine',
                'gpu_steps_per_s', 'cpu_steps_per_s', 'speedup', 'picard',
                'newton', 'ratio_to_unpartitioned', 'start_state')
        nests = []

        def nested(node, path, depth):
            if isinstance(node, dict):
                for key, val in node.items():
                    sub = path + '.' + key
                    if isinstance(val, dict) and depth >= 1 and \
                            len(json.dumps(val)) > 400 and \
                            not set(val) <= set(keep) | {'details'}:
                        nests.append((len(json.dumps(val)), sub, node, key))
                    nested(val, sub, depth + 1)
        nested(out.get('config', {}), 'config', 1)
        if not nests:
            break
        _, sub, node, key = max(nests, key=lambda c: c[0])
        moved[sub] = node[key]
        node[key] = dict({k: v for k, v in node[key].items() if k in keep},
                         details='bench_ops.json#' + sub)
    if moved:
        for path in (side or [os.path.join(ROOT, 'profiles', 'bench_ops.json'),
                              os.path.join(ROOT, 'gpurun_out',
                                           'bench_ops.json')]):
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, 'w') as fh:
                    json.dump(moved, fh, indent=1, sort_keys=True)
            except OSError:
                pass
    return out, moved


def spmv_bytes(A):
    """algorithmic bytes of y = A x"""
    r, c = A.shape
    return 12*A.nnz + 4*(r+1) + 8*c + 8*r


def build_problem(N=2, Re=100., refine=0):
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=N, Re=Re,
                                 refine=refine)
    return femp, sm, rhsd


def saddle_csr(F, J):
    import scipy.sparse as sps
    return sps.bmat([[F, J.T], [J, None]], format='csr')


def initial_state(sm, rhsd, system_factory):
    """start from the steady Stokes solution like `start_ssstokes=True`
    (reference snu:903-907) -- computed here by the GPU path itself with a
    generous Chebyshev degree (A alone is stiff)"""
    A, J = sm['A'], sm['J']
    NP, NV = J.shape
    system = system_factory(A.tocsr(), J)
    system.setup_precond(cheb_degree=24, schur='dense')
    vp = system.solve(rhsd['fv'], rhsd['fp'], rtol=1e-9, maxiter=3000,
                      raise_on_fail=False)
    stats = dict(system.last_stats)
    system.close()
    return vp[:NV].reshape((-1, 1)), vp[NV:].reshape((-1, 1)), stats


def scheme_setup(scheme, M, A, dt):
    """system matrix, rhs matrix and step coefficients of the resident loop:
    CNAB (tiu:104-143) or SBDF2 (tiu:320-353)"""
    if scheme == 'sbdf2':
        return ((M + 2./3*dt*A).tocsr(), M.tocsr(),
                dict(a_c=4./3, a_p=-1./3, cn_c=4./3*dt, cn_o=-2./3*dt),
                2./3*dt)
    return ((M + .5*dt*A).tocsr(), (M - .5*dt*A).tocsr(),
            dict(a_c=1., a_p=0., cn_c=1.5*dt, cn_o=-.5*dt), dt)


def cpu_baseline(sm, rhsd, v0, nfc0, dt, conv_host, nsteps, scheme='cnab'):
    """oracle leg: the reference's CNAB step (tiu:125-137) with the SuperLU
    factorisation done once (untimed, as the reference does once per run).
    The convection vector is the host callback in the reference (FEniCS, not
    part of this path): it is evaluated here by the scaffolding assembler and
    EXCLUDED from the timing, so the CPU figure is the linear algebra of the
    step alone; the GPU figure it stands next to includes the convection."""
    from oracle.saddle_oracle import SaddleLU
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    F, R1, cfd, gdt = scheme_setup(scheme, M, A, dt)
    t0 = time.perf_counter()
    klu = SaddleLU(F.tocsc(), J)
    tfac = time.perf_counter() - t0
    v = v0.copy()
    vprev = v0.copy()
    fv, fp = rhsd['fv'], rhsd['fp']
    nfc_c = nfc0
    el = 0.
    for k in range(nsteps):
        nfc_o, nfc_c = nfc_c, conv_host(v)           # untimed (host callback)
        t0 = time.perf_counter()
        rhs = R1 @ (cfd['a_c']*v + cfd['a_p']*vprev) + cfd['cn_c']*nfc_c \
            + cfd['cn_o']*nfc_o + gdt*fv
        vp = klu(np.vstack([rhs, fp]).flatten())
        vprev = v
        v = vp[:NV].reshape((NV, 1))
        p = -1./dt*vp[NV:].reshape((NP, 1))
        el += time.perf_counter() - t0
    return dict(value=nsteps/el, unit='timesteps/s', cores=1, kind='port',
                sample='{0} {2} steps with the oracle: SuperLU factor once '
                '({1:.3f} s, untimed) + per-step rhs, 2 triangular solves, '
                'rescale; convection callback evaluated on the host but not '
                'timed'.format(nsteps, tfac, scheme.upper())), v, p


def cpu_baseline_extras(sm, rhsd, v0, nfc0, dt, nsteps_gmres=1,
                        nsteps_threads=40):
    """the other two CPU lines of SURVEY 8d:
    (iii) un-preconditioned `scipy.sparse.linalg.gmres(rtol=1e-3,
          maxiter=800)` per step -- the stand-in for the reference's krypy path
          (`tests/time_dep_nse_krylov.py:4-7`: tol 1e-3, maxiter 800); a cold
          start needs ~700 inner iterations = ~11 s per time step on one core,
          so the sample is ONE step (with the initial Stokes solve below a
          bounded ~25 s of CPU work);
    "all cores": the prefactored SuperLU step again with every host core the
          process may use handed to the BLAS/OpenMP runtimes (SuperLU's
          triangular solves and SciPy's CSR products are single-threaded, so
          this mostly shows that the reference algorithm does not scale with
          cores)"""
    import scipy.sparse as sps
    import scipy.sparse.linalg as spsla
    from oracle.saddle_oracle import SaddleLU
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    F = (M + .5*dt*A).tocsr()
    K = sps.bmat([[F, J.T], [J, None]], format='csr')
    fv, fp = rhsd['fv'], rhsd['fp']
    out = {}
    # (iii) Krylov stand-in: what `krylov='Gmres'` with `tol 1e-3, maxiter 800`
    # (tests/time_dep_nse_krylov.py:4-7) asks of every boundary solve; COLD
    # start (x0 = 0), as krypy starts at snu:903-907 when no `x0` is handed over
    v = v0.copy()
    el, cnt = 0., [0]

    def cb(_):
        cnt[0] += 1
    relres = []
    # one core, as the line says: the BLAS pool of a 256-core host makes the
    # orthogonalisation of 10 645-entry vectors five times SLOWER
    try:
        from threadpoolctl import threadpool_limits
        one_core = threadpool_limits(limits=1)
    except ImportError:
        one_core = None
    for k in range(nsteps_gmres):
        t0 = time.perf_counter()
        rhs = M @ v - .5*dt*(A @ v) + dt*nfc0 + dt*fv
        b = np.vstack([rhs, fp]).flatten()
        x, info = spsla.gmres(K, b, x0=np.zeros(NV + NP), rtol=1e-3, atol=0.,
                              restart=800, maxiter=1, callback=cb,
                              callback_type='pr_norm')
        el += time.perf_counter() - t0
        relres.append(float(np.linalg.norm(K @ x - b)/np.linalg.norm(b)))
        v = x[:NV].reshape((NV, 1))
    out['krylov_unpreconditioned'] = dict(
        value=nsteps_gmres/el, unit='timesteps/s', cores=1, kind='port',
        inner_iterations_per_step=cnt[0]/float(nsteps_gmres),
        relres_reached_max=max(relres),
        sample='{0} CNAB steps, each solved by scipy.sparse.linalg.gmres('
        'rtol=1e-3, 800 inner iterations at most, no preconditioner, cold '
        'start x0 = 0) -- stand-in for the krypy path of '
        'tests/time_dep_nse_krylov.py (tol 1e-3, maxiter 800); convection '
        'frozen'.format(nsteps_gmres))
    # ... and the place the `krylov=` keywords reach in the reference: the
    # initial Stokes solve (snu:903-907), same call on `[[A, JT], [J, 0]]`
    Ks = sps.bmat([[A, J.T], [J, None]], format='csr')
    bs = np.vstack([fv, fp]).flatten()
    cnt[0] = 0
    t0 = time.perf_counter()
    xs, info = spsla.gmres(Ks, bs, x0=np.zeros(NV + NP), rtol=1e-3, atol=0.,
                           restart=800, maxiter=1, callback=cb,
                           callback_type='pr_norm')
    ts = time.perf_counter() - t0
    out['initial_stokes_krylov'] = dict(
        seconds=ts, inner_iterations=cnt[0], converged=bool(info == 0),
        relres_reached=float(np.linalg.norm(Ks @ xs - bs)/np.linalg.norm(bs)),
        cores=1, kind='port',
        sample='the initial Stokes solve (snu:903-907) by '
        'scipy.sparse.linalg.gmres(rtol=1e-3, one cycle of at most 800 '
        'iterations, no preconditioner, x0 = 0); the device solves the same '
        'system to 1e-9 (config.initial_stokes)')
    if one_core is not None:
        one_core.restore_original_limits()
    # all cores
    ncores = len(os.sched_getaffinity(0))
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    klu = SaddleLU(F.tocsc(), J)
    v = v0.copy()
    el = 0.
    ctx = threadpool_limits(limits=ncores) if threadpool_limits else None
    try:
        for k in range(nsteps_threads):
            t0 = time.perf_counter()
            rhs = M @ v - .5*dt*(A @ v) + dt*nfc0 + dt*fv
            vp = klu(np.vstack([rhs, fp]).flatten())
            v = vp[:NV].reshape((NV, 1))
            el += time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.restore_original_limits()
    out['all_cores'] = dict(
        value=nsteps_threads/el, unit='timesteps/s', cores=ncores,
        kind='port',
        sample='{0} prefactored-SuperLU CNAB steps with the BLAS/OpenMP '
        'thread pools opened to all {1} cores of the process (SuperLU '
        'triangular solves and CSR products stay single-threaded)'.format(
            nsteps_threads, ncores))
    return out


def picard_sweep_figures(femp, sm, rhsd, v0, dt, device, nsteps=256,
                         precond_at_v0=True, use_graph=True):
    """secondary workload (BASELINE config 3): one Picard
    and one Newton trapezoidal sweep over `nsteps` steps, everything on the
    device (convection matrices, F = M + dt/2 (A + N), solve); beside it what
    the reference pays per step on the CPU: a fresh SuperLU factorisation +
    solve of the re-valued saddle matrix (FEniCS assembly not counted)"""
    import scipy.sparse as sps
    import scipy.sparse.linalg as spsla
    from dolfin_navier_scipy_amd import saddle, convection
    from dolfin_navier_scipy_amd import newton_picard as dnp
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    th, inv = femp['V'], femp['invinds']
    cvop = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'], device=device)
    trange = dt*np.arange(nsteps + 1)
    # the preconditioner (set up once, the system is re-valued every step) is
    # built about the initial state: M + dt/2 (A + N1(v0))
    ts = dnp.TrapezoidalStepper(M, A, J, cvop, nslots=nsteps + 1, dt=dt,
                                device=device,
                                precond=dict(cheb_degree=6, drop_tol=1e-3,
                                             factorization='full'),
                                precond_linpoint=(v0 if precond_at_v0
                                                  else None))
    ts.set_rhs(rhsd['fv'], rhsd['fp'])
    for k in range(nsteps + 1):       # first linearisation: the initial state
        ts.write_linpoint(0, k, v0)
    opts = saddle.solve_opts(rtol=1e-10, maxiter=400, use_graph=use_graph,
                             reorth=2)
    out = {}
    which = 0
    # untimed pass: graph capture of the solver cycles (trajectory 0 = the
    # linearisation points is not modified by a sweep)
    ts.sweep(trange[:9], v0, which, True, opts=opts, record=False)
    for name, picard in (('picard', True), ('newton', False)):
        t0 = time.perf_counter()
        _, _, upd, st = ts.sweep(trange, v0, which, picard, opts=opts,
                                 record=False)
        wl = time.perf_counter() - t0
        out[name] = dict(steps_per_s=nsteps/wl,
                         krylov_iters_per_step=st['iters']/float(nsteps),
                         update_norm=upd)
        which = 1 - which
    Nc, _, _ = cvop.assemble(v0, newton=False)
    Fm = sps.csr_matrix((ts.mvals + .5*dt*(ts.avals + Nc.data),
                         ts.pattern.indices, ts.pattern.indptr), shape=(NV, NV))
    K = sps.bmat([[Fm, J.T], [J, None]], format='csc')
    rhs = np.ones(NV + NP)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        spsla.splu(K).solve(rhs)
    out['cpu_splu_factor_solve_ms_per_step'] = \
        1e3*(time.perf_counter() - t0)/reps
    out['steps'] = nsteps
    out['preconditioner'] = ('M + dt/2 (A + N1(v0)), set up once' if
                             precond_at_v0 else 'M + dt/2 A, set up once')
    ts.close()
    cvop.close()
    return out


def kernel_sources_sha256():
    """fingerprint of the sources the roofline kernels are compiled from"""
    import hashlib
    out = {}
    for name in ('pair.hpp', 'kernels.hpp'):
        path = os.path.join(ROOT, 'dolfin_navier_scipy_amd', 'csrc', name)
        out[name] = hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16]
    return out


def pmc_traffic(Kmat, kernel='k_spmv_stream16'):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes
    (profiles/spmv_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950) -- only if they were taken on this very matrix AND
    on these very kernel sources (the record is stamped with the hashes of
    pair.hpp / kernels.hpp by scripts/stamp_traffic.py; a stale record is not
    printed)"""
    path = os.path.join(ROOT, 'profiles', 'spmv_traffic.json')
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    if rec.get('nnz') != int(Kmat.nnz) or rec.get('rows') != Kmat.shape[0]:
        return None
    if rec.get('sources_sha256') != kernel_sources_sha256():
        sys.stderr.write('profiles/spmv_traffic.json was taken on other '
                         'kernel sources: roofline.traffic withheld\n')
        return None
    if kernel.startswith('k_spmv_pair'):
        return (rec.get('pair_format') or {}).get('hbm_bytes_per_launch')
    return rec['hbm_bytes_per_launch']


def traffic_source(traffic):
    """where `roofline.traffic` comes from: never measured by the run that
    prints it -- replayed from the committed PMC passes while the kernel
    sources are the ones the passes were taken on"""
    if traffic is None:
        return ('none: profiles/spmv_traffic.json is absent, was taken on '
                'another matrix, or on other kernel sources (withheld)')
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'spmv_traffic.json')))
    return ('stamped, not measured by this run: replayed from '
            'profiles/spmv_traffic.json (rocprofv3 --pmc passes of '
            'scripts/final_profiles.sh, FETCH_SIZE x 2 + WRITE_SIZE per '
            'launch) taken at commit {0}; printed only while the SHA-256 of '
            'pair.hpp / kernels.hpp equal the stamped ones'.format(
                rec.get('taken_at_commit', '?')))


def roofline_pair(saddle, Kmat, nv, reps, label):
    """`y = K x` through the pair format (2x2 node blocks, csrc/pair.hpp) --
    the kernel that applies K inside the solver at this size; same HIP-event
    timing and the same ALGORITHMIC bytes (those of the CSR product, SURVEY
    8d) as the CSR figures"""
    x = np.sin(0.37*np.arange(Kmat.shape[1]))
    # three timed regions, each on buffers of its own (every call uploads the
    # matrix afresh): where the 0.7 GB of streams land in the HBM moves the
    # average launch time by up to 10 % from one allocation to the next; the
    # MEDIAN region is reported, all three are in `regions_avg_us`
    runs = []
    try:
        # the first launches of a process run at idle clocks (148-152 us
        # measured first thing, 132-134 us behind half a second of work on the
        # same box): 0.3 s of the kernel itself untimed, then the regions
        saddle.spmv_pair(Kmat, nv, x, reps=2000, warmup=5)
        for _ in range(3):
            _, secs, fbytes = saddle.spmv_pair(Kmat, nv, x, reps=reps,
                                               warmup=5)
            runs.append(secs)
    except Exception as exc:            # odd NV, ...: the CSR kernel applies K
        sys.stderr.write('pair format not available: {0}\n'.format(exc))
        return None
    secs = sorted(runs)[1]
    return dict(kernel='k_spmv_pair16x', achieved=spmv_bytes(Kmat)/secs/1e9,
                avg_us=secs*1e6, bytes=spmv_bytes(Kmat), nnz=int(Kmat.nnz),
                rows=int(Kmat.shape[0]), matrix=label,
                format_bytes=int(fbytes),
                regions_avg_us=[1e6*r for r in runs])


def hbm_roofline(saddle, args, dt, device):
    """the roofline kernel on the refined matrix and what plain streaming
    kernels get out of this HBM -- measured FIRST in the process: after the
    other legs have allocated and freed a few GB of device memory the same
    launches run 10-15 % slower (133 us in a fresh process, 147-152 us behind
    the refined and Newton/Picard legs, `profiles/r03_f_final/`: every timed
    region inside one process agrees to 1 %, the processes do not), so the
    order of the legs would otherwise decide the figure"""
    if args.roofline_refine <= 0:
        return None, None, None
    _, smr, _ = build_problem(N=args.level, Re=args.Re,
                              refine=args.roofline_refine)
    Kr = saddle_csr((smr['M'] + .5*dt*smr['A']).tocsr(), smr['J'])
    # only the LDS-streaming kernel (16-bit column offsets) runs on the
    # refined matrix, so its rocprofv3 average is this measurement and
    # nothing else
    label = 'K on the mesh refined {0}x (n={1})'.format(
        args.roofline_refine, Kr.shape[0])
    # the solver applies K through the pair format at this size (if K has an
    # even number of velocity dofs): that is the kernel on record; the CSR
    # streaming kernel stays next to it
    roof_pair = roofline_pair(saddle, Kr, smr['M'].shape[0], 30, label)
    roof_csr = roofline_spmv(saddle, Kr, 30, label, variants=('stream16',))
    roof_hbm = roof_pair if roof_pair is not None else roof_csr
    roof_hbm['csr_kernel'] = None if roof_pair is None else dict(
        roof_csr, traffic=pmc_traffic(Kr, roof_csr['kernel']))
    traffic = pmc_traffic(Kr, roof_hbm['kernel'])
    # what plain streaming kernels get out of this HBM (2 GiB, fp64)
    attain = {k: saddle.hbm_probe(2 << 30, k, reps=20, device=device)
              for k in ('read', 'read8c', 'read_tiles1', 'copy', 'triad')}
    # best read pattern: grid-stride 16-byte loads or workgroup-contiguous
    # 16 KiB tiles (what the CSR stream kernels do)
    attain['read'] = max(attain['read'], attain.pop('read8c'),
                         attain.pop('read_tiles1'))
    return roof_hbm, traffic, attain


def roofline_spmv(saddle, Kmat, reps, label, variants=('vector', 'stream')):
    best = None
    for variant in variants:
        secs, chk = saddle.spmv_bench(Kmat, variant=variant, reps=reps,
                                      warmup=5)
        gbs = spmv_bytes(Kmat)/secs/1e9
        rec = dict(kernel='k_spmv_{0}'.format(variant), achieved=gbs,
                   avg_us=secs*1e6, bytes=spmv_bytes(Kmat), nnz=int(Kmat.nnz),
                   rows=int(Kmat.shape[0]), matrix=label)
        if best is None or gbs > best['achieved']:
            best = rec
    return best


PARITY_TOL = 1e-8           # v (M-norm) and p (l2), relative: north-star
PARITY_PREFIX_STEPS = 10    # bandwidth ladder: the first steps are compared


def partitioned_run(args, world, rank, device, dist, one_gpu, start=None,
                    reference=None):
    """ONE row-partitioned simulation on `world` ranks:
    `--spinup` + `--warmup` untimed steps from the start state, then `--steps`
    timed ones between barriers; the maximum over the ranks counts.

    The line carries its own proof: the final state is gathered
    (`dns_imex_get_state`, collective) and rank 0 repeats the SAME steps from
    the SAME state on an un-partitioned handle -- `parity` = the distance of
    the two final states (v in the M-norm, p in l2, relative); above
    `PARITY_TOL` the leg is an error.  Meshes of more than
    `--parity-full-max` unknowns compare the first `PARITY_PREFIX_STEPS` steps
    instead (the un-partitioned set-up alone takes a minute at n = 5.7M).
    `start`: `(v0,)` handed over by the caller; else `--start` decides (the
    steady Stokes state like the N=1 headline, solved un-partitioned on rank
    0 and broadcast, or rest).  `reference`: `(v, p)` of an un-partitioned run
    of the same steps the caller has already made (N=1 line)."""
    from dolfin_navier_scipy_amd import saddle, _capi, convection, perfmodel
    from dolfin_navier_scipy_amd import comm as dcomm
    dt = 1./args.nts
    femp, sm, rhsd = build_problem(N=args.level, Re=args.Re,
                                   refine=args.refine)
    M, A, J = sm['M'], sm['A'], sm['J']
    NP, NV = J.shape
    th, inv = femp['V'], femp['invinds']
    schur_kind = 'dense' if NP <= args.dense_max else 'mg'
    prols = None
    if schur_kind == 'mg':
        from dolfin_navier_scipy_amd.fem import (
            cylinder_mesh_hierarchy, pressure_prolongations, TaylorHood)
        hier = cylinder_mesh_hierarchy(N=args.level, refine=args.refine)
        spaces = [TaylorHood(m) for m, _ in hier][::-1]
        prols = pressure_prolongations(spaces, [p for _, p in hier][::-1])
    F = (M + .5*dt*A).tocsr()
    R1 = (M - .5*dt*A).tocsr()

    def barrier():
        _capi.device_synchronize(device)
        if dist is not None:
            dist.barrier()
            _capi.device_synchronize(device)

    def over_ranks(vals, op):
        if dist is None:
            return list(vals)
        import torch
        t = torch.tensor(list(vals), dtype=torch.float64,
                         device='cpu' if one_gpu else 'cuda')
        dist.all_reduce(t, op=op)
        return [float(x) for x in t.cpu()]

    # ---- start state: the same on every rank, bit for bit
    stokes = None
    if start is not None:
        v0, start_kind = np.array(start[0], dtype=np.float64).reshape((NV, 1)), \
            'handed over by the caller (the N=1 headline\'s Stokes state)'
    elif args.start == 'stokes' and NP > args.dense_max:
        # (the initial Stokes solve of this script runs with the dense Schur
        # inverse: NP^2 entries)
        v0, start_kind = np.zeros((NV, 1)), (
            'rest (no Stokes start beyond --dense-max pressure dofs)')
    elif args.start == 'stokes':
        v0 = np.zeros((NV, 1))
        if rank == 0:
            v0, _, stokes = initial_state(
                sm, rhsd, lambda Fm, Jm: saddle.SaddleSystem(Fm, Jm,
                                                             device=device))
        if dist is not None:
            import torch
            t = torch.from_numpy(np.ascontiguousarray(v0.reshape(-1)))
            if not one_gpu:
                t = t.cuda()
            dist.broadcast(t, 0)
            v0 = t.cpu().numpy().reshape((NV, 1)).copy()
        start_kind = ('steady Stokes solution (snu:903-907), solved '
                      'un-partitioned on rank 0 and broadcast')
    else:
        v0, start_kind = np.zeros((NV, 1)), 'rest'

    with stdout_to_stderr():
        if one_gpu:
            comm_obj = dcomm.Comm.gloo(device)
        elif dist is not None:
            comm_obj = dcomm.Comm.rccl_from_torch(device)
        else:
            comm_obj = dcomm.Comm.rccl(device, 1, 0, dcomm.rccl_unique_id())
    # rank-local construction (`dns_saddle_create_rows`): the library is handed
    # this rank's rows of F, JT, J and R1 only (the assembler of this script is
    # replicated: the rows are cut out of its matrices)
    by_rows = args.construction == 'rows' or (
        args.construction == 'auto' and world > 1)
    import resource
    rss_before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1024.
    t_setup = time.perf_counter()
    if by_rows:
        system = saddle.SaddleSystem.from_rows_of(F, J, comm_obj,
                                                  device=device)
    else:
        system = saddle.SaddleSystem(F, J, device=device)
        system.set_comm(comm_obj)
    if prols is not None:
        system.set_schur_mg(prols)
    # bandwidth regime: the partitioned solve needs the explicit polynomial
    # matrix; degree 8 from 1e6 unknowns on (refined_bench.py's setting)
    fhat = 'explicit' if (args.fhat == 'auto' and NV > 200000) else args.fhat
    # (bandwidth regime: the measured degree / drop-tolerance optimum,
    # profiles/r04_gc_pareto; the flags still decide at the reference sizes)
    tuned = saddle.streaming_precond_defaults(NV + NP)
    big = NV + NP >= 100000
    cheb = tuned['cheb_degree'] if big else args.cheb
    drop = tuned['drop_tol'] if big else args.drop
    pkw = dict(cheb_degree=cheb, schur=schur_kind, fhat=fhat,
               fp32_store=bool(args.fp32), drop_tol=drop,
               factorization=args.fact)
    system.setup_precond(**pkw)
    _capi.device_synchronize(device)
    t_setup = time.perf_counter() - t_setup
    rss_setup = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1024.
    cvop = convection.ConvectionP2.from_taylor_hood(
        th, inv, femp['dbcinds'], femp['dbcvals'], device=device)
    nfc = cvop.apply(v0, scale=-1.0)
    cf = saddle.ImexStepper.coeffs(a_c=1., a_p=0., cn_c=1.5*dt, cn_o=-.5*dt,
                                   pscale=-1./dt,
                                   extrapolate=(tuned['extrapolate'] if big
                                                else args.extrap),
                                   carry_residual=bool(args.carry))

    def options(graph):
        return saddle.solve_opts(method='gmres', rtol=args.rtol, maxiter=400,
                                 restart=60, check_every=args.check_every,
                                 use_graph=graph, reorth=args.reorth)
    opts = options(not args.eager)

    class new_stepper(object):
        """a stepper at the start state with a convection operator of its own
        (a partitioned stepper restricts its operator to the rank's cells)"""

        def __init__(self, sysh):
            self.cv = convection.ConvectionP2.from_taylor_hood(
                th, inv, femp['dbcinds'], femp['dbcvals'], device=device)
            self.st = saddle.ImexStepper(sysh, R1)
            self.st.set_state(v0, nfc_c=nfc, nfc_o=nfc)
            self.st.set_rhs(dt*rhsd['fv'], rhsd['fp'])
            self.st.set_convection(self.cv, scale=-1.0)
            self.run, self.get_state = self.st.run, self.st.get_state

        last_run = property(lambda self: self.st.last_run)

        def close(self):
            self.st.close()
            self.cv.close()

    # ---- what is compared: the whole run, or its first steps
    full = reference is not None or (NV + NP) <= args.parity_full_max
    compared = None
    if not full:
        stp = new_stepper(system)
        stp.run(PARITY_PREFIX_STEPS, cf, opts)
        compared = stp.get_state()            # (collective gather)
        stp.close()
    stp = new_stepper(system)
    if args.spinup > 0:
        stp.run(args.spinup, cf, opts)
    stp.run(args.warmup, cf, opts)
    c0 = comm_obj.stats()
    barrier()
    t0 = time.perf_counter()
    _, its, lst = stp.run(args.steps, cf, opts)
    barrier()
    wall = time.perf_counter() - t0
    c1 = comm_obj.stats()
    run_record = dict(stp.last_run)
    if full:
        compared = stp.get_state()
    # ---- device time per collective: a window of plain launches behind the
    # timed one (events cannot time calls inside a replayed graph)
    ntime = max(1, min(args.steps, 20))
    comm_obj.set_timing(True)
    barrier()
    t0 = time.perf_counter()
    stp.run(ntime, cf, options(False))
    barrier()
    wall_eager = time.perf_counter() - t0
    coll_time = comm_obj.timing()
    comm_obj.set_timing(False)
    for rec in coll_time.values():
        rec['us_per_time_step'] = 1e3*rec['device_ms']/ntime
    MAX = None if dist is None else dist.ReduceOp.MAX
    SUM = None if dist is None else dist.ReduceOp.SUM
    wall = over_ranks([wall], MAX)[0]
    info = system.precond_info()       # the nnz are those of THIS rank's rows
    glob = over_ranks([info['nnz_K'], info['nnz_Gc'], info['nnz_JG']], SUM)
    mbytes = system.device_matrix_bytes()
    mb = over_ranks([mbytes], MAX)[0]
    host_bytes = [int(x) for x in over_ranks(
        [float(x) for x in system.host_matrix_bytes()], MAX)]
    t_setup = over_ranks([t_setup], MAX)[0]
    ginfo = dict(info, nnz_K=int(glob[0]), nnz_Gc=int(glob[1]),
                 nnz_JG=int(glob[2]))
    roof = perfmodel.step_roofline(
        ginfo, int(R1.nnz), int(th.mesh.ncells), its/float(args.steps),
        1e3*wall/args.steps, peak_GBs=world*HBM_PEAK_GBS)
    roof.pop('ops', None)

    # ---- the proof: the same steps from the same state, un-partitioned
    parity = None
    if rank == 0 and not args.no_parity:
        nsteps_cmp = (args.spinup + args.warmup + args.steps) if full \
            else PARITY_PREFIX_STEPS
        if reference is not None:
            v_ref, p_ref = reference
            how = 'the un-partitioned headline run of this process'
        else:
            ref_sys = saddle.SaddleSystem(F, J, device=device)
            if prols is not None:
                ref_sys.set_schur_mg(prols)
            ref_sys.setup_precond(**pkw)
            rst = new_stepper(ref_sys)
            # (solved two digits tighter than the timed run: the distance is
            # then the PARTITIONED run's own error against the trajectory,
            # not the sum of two inexact runs' errors)
            ropts = saddle.solve_opts(
                method='gmres', rtol=min(args.rtol, 1e-12), maxiter=400,
                restart=60, check_every=args.check_every,
                use_graph=not args.eager, reorth=args.reorth)
            if full:
                if args.spinup > 0:
                    rst.run(args.spinup, cf, ropts)
                rst.run(args.warmup, cf, ropts)
                rst.run(args.steps, cf, ropts)
            else:
                rst.run(PARITY_PREFIX_STEPS, cf, ropts)
            v_ref, p_ref = rst.get_state()
            rst.close()
            ref_sys.close()
            how = ('an un-partitioned handle on rank 0\'s GPU, same matrices, '
                   'same preconditioner settings, same calls, rtol 1e-12')
        mn = lambda x: float(np.sqrt((x.T @ (M @ x)).item()))
        v_cmp, p_cmp = compared
        ev = mn(v_cmp - v_ref)/max(mn(v_ref), 1e-300)
        ep = float(np.linalg.norm(p_cmp - p_ref)
                   / max(np.linalg.norm(p_ref), 1e-300))
        parity = dict(v_rel_Mnorm=ev, p_rel_l2=ep, steps=nsteps_cmp,
                      tol=PARITY_TOL, ok=bool(ev <= PARITY_TOL
                                              and ep <= PARITY_TOL),
                      window='whole run (spin-up + warm-up + timed steps)'
                      if full else 'first {0} steps from the start state '
                      '(a run of their own before the timed one)'.format(
                          PARITY_PREFIX_STEPS),
                      against=how)
    barrier()
    res = dict(
        steps_per_s=args.steps/wall, ms_per_step=1e3*wall/args.steps,
        steps=args.steps, warmup=args.warmup, spinup=args.spinup,
        ranks=world, level=args.level, refine=args.refine, dt=dt,
        NV=int(NV), NP=int(NP), unknowns=int(NV + NP),
        rows_per_rank=(NV + NP)/float(world),
        dof_steps_per_s=(NV + NP)*args.steps/wall, schur=schur_kind,
        krylov_iters_per_step=its/float(args.steps),
        true_relres_last=lst['true_relres'],
        start_state=start_kind, initial_stokes=stokes,
        parity=parity, carry_residual=bool(args.carry),
        cheb_degree=cheb, drop_tol=drop, fhat=fhat,
        run_record=run_record,
        graph_replay=bool(not args.eager and not one_gpu and
                          os.environ.get('DNS_DIST_GRAPH', '1') != '0'),
        collectives_timed_window={k: int(c1[k] - c0[k]) for k in c1},
        collectives_device_time=dict(
            coll_time, steps=ntime, ms_per_step_plain_launches=(
                1e3*wall_eager/ntime),
            what='{0} further steps with plain launches instead of graph '
            'replay (rank 0): HIP events on the launch stream around every '
            'collective; a call\'s time includes the wait for the slowest '
            'peer; halo_exchange = the ncclSend/Recv group without its pack '
            '/ unpack kernels'.format(ntime)),
        matrix_bytes_per_rank_max=int(mb), precond_setup_s=t_setup,
        construction=dict(
            zip(('host_matrix_bytes_kept', 'host_matrix_bytes_setup'),
                host_bytes),
            kind='rows: every rank hands over its rows of F, JT, J, R1 only '
            '(dns_saddle_create_rows)' if by_rows else
            'whole matrices on every rank (dns_saddle_create + set_comm)',
            create_and_setup_s=t_setup,
            process_peak_rss_mb=dict(before_create=rss_before,
                                     after_setup=rss_setup),
            host_threads=os.environ.get('DNS_HOST_THREADS'),
            what='max over the ranks; host bytes = matrix copies the handle '
            'keeps / matrices alive at the end of the explicit set-up; the '
            'peak resident set of rank 0\'s process (it holds this script\'s '
            'whole assembled matrices as well) before the handle is created '
            'and after its set-up'),
        backend='gloo, host staged (one-GPU rehearsal)' if one_gpu
        else 'RCCL', roofline_step=roof,
        what='one simulation; every rank holds its row blocks of K, Fh^-1, '
             'J Fh^-1 (and of the dense Schur inverse); halo entries by '
             'index lists (Send/Recv), one all-reduce per Arnoldi step')
    if parity is not None and not parity['ok']:
        res['error'] = ('parity: the partitioned run is {0:.2e} (v, M-norm) / '
                        '{1:.2e} (p) away from the un-partitioned run of the '
                        'same steps (tolerance {2:g})'.format(
                            parity['v_rel_Mnorm'], parity['p_rel_l2'],
                            PARITY_TOL))
    stp.close()
    cvop.close()
    if not by_rows:
        system.set_comm(None)
    system.close()
    comm_obj.close()
    return res


def partitioned_child(args, world, rank, local_rank):
    """`--partitioned-only`: the process of one rank of a partitioned run"""
    if args.dry_run:
        # (tests: rendezvous of the child ranks over gloo, canned figures)
        import torch.distributed as dist
        with stdout_to_stderr():
            dist.init_process_group('gloo')
            dist.barrier()
        res = dict(steps_per_s=100.0/(1 + args.refine), ms_per_step=10.0,
                   NV=2*args.level, NP=args.level, unknowns=3*args.level,
                   rows_per_rank=3.*args.level/world, level=args.level,
                   refine=args.refine, krylov_iters_per_step=1.0,
                   collectives_timed_window=dict(allreduce=1), dry_run=True,
                   start_state=args.start,
                   parity=dict(v_rel_Mnorm=0.0, p_rel_l2=0.0, ok=True,
                               tol=PARITY_TOL, dry_run=True),
                   roofline_step=dict(achieved=1.0, frac=1e-4))
        if rank == 0:
            print(json.dumps(res))
            sys.stdout.flush()
        dist.destroy_process_group()
        return res
    dist, one_gpu = None, False
    import faulthandler
    # say where it hangs, shortly before the parent's time limit strikes
    faulthandler.dump_traceback_later(max(30., args.partitioned_timeout - 15.),
                                      exit=True)
    if world > 1:
        import datetime
        import torch
        import torch.distributed as dist
        # rehearsal of the N > 1 code path on a ONE-GPU box: every rank on
        # device 0 and the host-staged gloo communicator (RCCL refuses two
        # ranks per device)
        one_gpu = os.environ.get('DNS_BENCH_REHEARSE_ONE_GPU') == '1'
        if one_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        with stdout_to_stderr():
            dist.init_process_group(
                'gloo' if one_gpu else 'nccl',
                timeout=datetime.timedelta(seconds=args.partitioned_timeout))
            dist.barrier()
    from dolfin_navier_scipy_amd import _capi
    if _capi.device_count() <= local_rank:
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    res = partitioned_run(args, world, rank, local_rank, dist, one_gpu)
    if rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()
    return res


def child_env(**override):
    """environment of a child run: the launcher's rendezvous variables only.
    The TORCHELASTIC_* variables must not travel -- with
    TORCHELASTIC_USE_AGENT_STORE a process group expects the launcher's agent
    to host the store on MASTER_PORT, which is not true on the child's port
    (rank 0 has to host it itself)."""
    env = {k: v for k, v in os.environ.items()
           if not k.startswith('TORCHELASTIC_')}
    env.update({k: str(v) for k, v in override.items()})
    return env


def run_child(cmd, env, timeout, want_result):
    """a child process under a time limit; its last JSON line (or an error).
    The child's stderr goes straight to this process's stderr (a run that is
    killed has then said where it was)."""
    import subprocess
    try:
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE,
                                 stderr=None, start_new_session=True)
        try:
            cout, _ = child.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, 9)         # exactly the group started above
            cout, _ = child.communicate()
            res = dict(error='no result within {0:g} s (killed)'
                       .format(timeout))
            # what it had reported before it stopped (the self-test prints a
            # line per primitive: the last one names the primitive that hung)
            part = [ln for ln in (cout or b'').decode(errors='replace')
                    .splitlines() if ln.startswith('{')]
            if part:
                try:
                    res['partial'] = json.loads(part[-1])
                except ValueError:
                    pass
            return res
        lines = [ln for ln in cout.decode(errors='replace').splitlines()
                 if ln.startswith('{')]
        if lines:
            return json.loads(lines[-1])
        if not want_result and child.returncode == 0:
            return None
        return dict(error='child exited with code {0} and no result (its '
                    'stderr is in this run\'s stderr)'.format(
                        child.returncode))
    except Exception as exc:
        return dict(error=str(exc))


SELFTEST_LEGS = [(name, graph)
                 for name in ('allreduce', 'sendrecv_ring',
                              'allgather_unequal', 'allgather_equal')
                 for graph in (False, True)]


def selftest_child(args, world, rank, local_rank):
    """`--selftest-only`: first contact of the ranks over RCCL, `dns_comm_*`
    only -- no matrix, no solver.  Rendezvous over gloo (CPU), then
    ncclCommInitRank, then every primitive of the partitioned path once as
    plain launches and once captured in a hipGraph.  Rank 0 prints the record
    again after every step (the parent keeps the last line of a child it had
    to kill: the first primitive that is missing is the one that hung);
    every rank says on stderr where it is."""
    rec = dict(world=world, primitives={}, complete=False, stage='start')

    def report(stage):
        rec['stage'] = stage
        sys.stderr.write('[rccl self-test] rank {0}: {1}\n'.format(rank, stage))
        sys.stderr.flush()
        if rank == 0:
            print(json.dumps(rec))
            sys.stdout.flush()
    import faulthandler
    faulthandler.dump_traceback_later(max(10., args.selftest_timeout - 5.),
                                      exit=True)
    if args.dry_run:
        import torch.distributed as dist
        with stdout_to_stderr():
            dist.init_process_group('gloo')
            dist.barrier()
        for name, graph in SELFTEST_LEGS:
            rec['primitives'][name + ('_graph' if graph else '_eager')] = \
                dict(ok=True, us_per_call=1.0)
        rec.update(complete=True, dry_run=True)
        report('done')
        dist.destroy_process_group()
        return rec
    t0 = time.time()
    import torch.distributed as dist
    one_gpu = os.environ.get('DNS_BENCH_REHEARSE_ONE_GPU') == '1'
    if world > 1 or 'MASTER_PORT' in os.environ:
        with stdout_to_stderr():
            dist.init_process_group('gloo')
            dist.barrier()
    else:
        dist = None
    rec['rendezvous_s'] = round(time.time() - t0, 3)
    report('rendezvous (gloo) done')
    from dolfin_navier_scipy_amd import _capi, comm as dcomm
    device = 0 if one_gpu else local_rank
    if _capi.device_count() <= device:
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    t0 = time.time()
    with stdout_to_stderr():
        if one_gpu and world > 1:
            comm_obj = dcomm.Comm.gloo(device)
            rec['backend'] = 'gloo-staged callbacks (one-GPU rehearsal)'
        elif dist is not None:
            comm_obj = dcomm.Comm.rccl_from_torch(device)
            rec['backend'] = 'rccl'
        else:
            comm_obj = dcomm.Comm.rccl(device, 1, 0, dcomm.rccl_unique_id())
            rec['backend'] = 'rccl'
    rec['comm_init_s'] = round(time.time() - t0, 3)
    report('communicator created')
    allok = True
    for name, graph in SELFTEST_LEGS:
        key = name + ('_graph' if graph else '_eager')
        if graph and rec['backend'] != 'rccl':
            rec['primitives'][key] = dict(skipped='host-callback communicator')
            continue
        report('entering ' + key)
        try:
            with stdout_to_stderr():
                one = comm_obj.selftest(name, graph=graph, count=4, reps=20)
        except Exception as exc:
            one = dict(ok=False, error=str(exc))
        one['us_per_call'] = (round(one['us_per_call'], 2)
                              if 'us_per_call' in one else None)
        rec['primitives'][key] = one
        allok = allok and bool(one.get('ok'))
    rec['gather_forms'] = comm_obj.gather_forms()
    rec['complete'] = True
    rec['ok'] = allok
    report('done')
    comm_obj.close()
    if dist is not None:
        dist.destroy_process_group()
    return rec


def visible_gpu_count():
    """GPUs this process could open, WITHOUT a HIP call (the launcher must not
    touch the GPU before it starts its rank processes): the KFD topology's
    nodes with SIMDs, cut down by *_VISIBLE_DEVICES; None when unknown"""
    import glob
    nodes = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    if not nodes:
        return None
    count = 0
    for path in nodes:
        try:
            with open(path) as fh:
                for line in fh:
                    if line.startswith('simd_count'):
                        count += int(line.split()[1]) > 0
        except (OSError, ValueError):
            return None
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES',
                'CUDA_VISIBLE_DEVICES'):
        val = os.environ.get(var)
        if val is not None:
            count = min(count, len([x for x in val.split(',') if x.strip()]))
    return count


def self_launch(args):
    """`bench.py --gpus N` (N > 1) started WITHOUT a launcher: this process
    starts the N rank processes itself -- children with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, spawned before anything here has touched the
    GPU (no torch import, no HIP call: the launcher only waits) -- hands
    rank 0's JSON line through and exits non-zero if there is none.  The
    one-GPU line must never be printed for `--gpus N`."""
    import socket
    import subprocess
    world = args.gpus
    if not args.dry_run and os.environ.get('DNS_BENCH_REHEARSE_ONE_GPU') != '1':
        have = visible_gpu_count()
        if have is not None and have < world:
            sys.stderr.write('bench.py --gpus {0}: only {1} GPU(s) visible on '
                             'this box -- no line printed\n'.format(world, have))
            raise SystemExit(2)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    sys.stderr.write('bench.py --gpus {0} without a launcher: starting {0} '
                     'rank processes (rendezvous 127.0.0.1:{1})\n'
                     .format(world, port))
    sys.stderr.flush()
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(world):
        env = child_env(RANK=r, LOCAL_RANK=r, WORLD_SIZE=world,
                        LOCAL_WORLD_SIZE=world, MASTER_ADDR='127.0.0.1',
                        MASTER_PORT=port, DNS_BENCH_SELF_LAUNCHED=1)
        procs.append(subprocess.Popen(
            cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
            stderr=None, start_new_session=True))
    limit = args.time_budget + 600.
    t0, out0 = time.time(), b''
    try:
        out0, _ = procs[0].communicate(timeout=limit)
        for pr in procs[1:]:
            pr.wait(timeout=max(5., limit - (time.time() - t0)))
    except subprocess.TimeoutExpired:
        sys.stderr.write('bench.py: rank processes still running after {0:g} '
                         's, killed\n'.format(limit))
    finally:
        for pr in procs:
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, 9)    # exactly the groups started above
                except OSError:
                    pass
                pr.wait()
    lines = [ln for ln in out0.decode(errors='replace').splitlines()
             if ln.startswith('{')]
    codes = [pr.returncode for pr in procs]
    if not lines or any(codes):
        sys.stderr.write('bench.py --gpus {0}: rank exit codes {1}, {2} JSON '
                         'line(s)\n'.format(world, codes, len(lines)))
    if not lines:
        raise SystemExit(3)
    print(lines[-1])
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(4)
    return json.loads(lines[-1])


def multi_gpu_main(args, world, rank, local_rank):
    """`--gpus N`, N > 1 (launched by torch.distributed.run, one rank per GPU).

    The parent ranks only coordinate (gloo group, no GPU context): every
    measured run lives in a child process per rank with its own rendezvous and
    a time limit."""
    import torch
    import torch.distributed as dist
    with stdout_to_stderr():
        dist.init_process_group('gloo')
        dist.barrier()
    base_port = int(os.environ.get('MASTER_PORT', '29500'))
    common = ['--Re', str(args.Re), '--cheb', str(args.cheb),
              '--extrap', str(args.extrap), '--fp32',
              str(args.fp32), '--drop', str(args.drop), '--fhat', args.fhat,
              '--fact', args.fact, '--reorth', str(args.reorth),
              '--check-every', str(args.check_every), '--carry',
              str(args.carry), '--parity-full-max', str(args.parity_full_max)]
    if args.eager:
        common.append('--eager')
    if args.dry_run:
        common.append('--dry-run')

    def fresh_port(fallback):
        # a port the OS hands out on rank 0, told to every rank through the
        # parents' gloo group.  The probing socket is bound with SO_REUSEADDR
        # and stays open for the whole life of the child run: it never
        # listens, so the child's TCPStore -- which sets SO_REUSEADDR as well
        # -- binds the same port, while nobody else (gloo's own ephemeral
        # sockets were the original trouble) is handed it in between; `held`
        # is closed by the caller when the children are gone
        import socket
        t = torch.zeros(1, dtype=torch.int64)
        held = None
        if rank == 0:
            try:
                held = socket.socket()
                held.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                held.bind(('127.0.0.1', 0))
                t[0] = held.getsockname()[1]
            except OSError:
                held = None
                t[0] = fallback
                sys.stderr.write('bench: no ephemeral port, falling back to '
                                 '{0}\n'.format(fallback))
        dist.broadcast(t, 0)
        return int(t.item()), held

    def partitioned(port_offset, level, refine, nts, start, timeout=None,
                    rtol=None):
        timeout = args.partitioned_timeout if timeout is None else timeout
        rtol = args.rtol if rtol is None else rtol
        cmd = [sys.executable, os.path.abspath(__file__), '--partitioned-only',
               '--gpus', str(world), '--steps', str(args.steps), '--warmup',
               str(args.warmup), '--spinup', str(args.spinup), '--level',
               str(level), '--refine', str(refine), '--nts', str(nts),
               '--dense-max', str(args.dense_max), '--partitioned-timeout',
               str(timeout), '--start', start, '--rtol', str(rtol)] + common
        tries = [0]

        def attempt(extra_env):
            dist.barrier()
            port, held = fresh_port(base_port + port_offset + 5*tries[0])
            tries[0] += 1
            e2 = child_env(MASTER_PORT=port, MASTER_ADDR='127.0.0.1')
            e2.update(extra_env)
            try:
                res = run_child(cmd, e2, timeout, rank == 0)
            finally:
                if held is not None:
                    held.close()
            # a rank whose child failed makes the run a failure for everybody
            bad = torch.tensor([1 if (res is not None and 'error' in res)
                                else 0])
            dist.all_reduce(bad)
            if int(bad.item()) and rank == 0 and 'error' not in res:
                res = dict(error='the child of another rank failed',
                           partial=res)
            return res, int(bad.item())
        res, bad = attempt({})
        # once more with plain launches instead of replayed graphs (the
        # captured RCCL calls are the one part of this path no one-GPU box can
        # exercise with more than one rank) -- unless the self-test has put
        # the legs on plain launches already, or the budget does not hold a
        # second run of this leg
        again = bad and not args.eager and \
            os.environ.get('DNS_DIST_GRAPH') != '0'
        if again and remaining() < 0.5*timeout + 60.:
            again = False
            if rank == 0 and isinstance(res, dict):
                res['second_attempt'] = 'skipped: --time-budget'
        if again:
            first = res
            res, bad = attempt({'DNS_DIST_GRAPH': '0'})
            if rank == 0 and isinstance(res, dict):
                res['first_attempt_with_graphs'] = first
        return res

    t_start = time.time()

    def remaining():
        """seconds of --time-budget left, the same figure on every rank"""
        t = torch.tensor([args.time_budget - (time.time() - t_start)],
                         dtype=torch.float64)
        dist.broadcast(t, 0)
        return float(t.item())

    # first contact: a short child per rank that drives dns_comm_* only
    def selftest():
        cmd = [sys.executable, os.path.abspath(__file__), '--selftest-only',
               '--gpus', str(world), '--selftest-timeout',
               str(args.selftest_timeout)]
        if args.dry_run:
            cmd.append('--dry-run')
        dist.barrier()
        port, held = fresh_port(base_port + 3)
        try:
            res = run_child(cmd, child_env(MASTER_PORT=port,
                                           MASTER_ADDR='127.0.0.1'),
                            args.selftest_timeout + 10., rank == 0)
        finally:
            if held is not None:
                held.close()
        bad = torch.tensor([1 if (res is not None and 'error' in res) else 0])
        dist.all_reduce(bad)
        if int(bad.item()) and rank == 0 and 'error' not in res:
            res = dict(error='the self-test child of another rank failed',
                       partial=res)
        # what the legs behind it may rely on, agreed by all ranks
        flags = torch.zeros(2, dtype=torch.int64)
        if rank == 0:
            rec = res.get('partial', res) if isinstance(res, dict) else {}
            prim = rec.get('primitives', {}) if isinstance(rec, dict) else {}
            eager = [k for k in prim if k.endswith('_eager')]
            graph = [k for k in prim if k.endswith('_graph')
                     and 'skipped' not in prim[k]]
            flags[0] = int(len(eager) == 4
                           and all(prim[k].get('ok') for k in eager))
            flags[1] = int(len(graph) == 4
                           and all(prim[k].get('ok') for k in graph))
        dist.broadcast(flags, 0)
        return res, bool(flags[0].item()), bool(flags[1].item())

    rccl_selftest, eager_ok, graph_ok = selftest()
    one_gpu_rehearsal = os.environ.get('DNS_BENCH_REHEARSE_ONE_GPU') == '1'
    if not graph_ok and not one_gpu_rehearsal and not args.eager:
        # captured collectives did not pass first contact: plain launches for
        # every partitioned leg (no attempt is spent on finding that out again)
        os.environ['DNS_DIST_GRAPH'] = '0'
        sys.stderr.write('bench: rank {0}: RCCL self-test did not pass its '
                         'graph legs, DNS_DIST_GRAPH=0\n'.format(rank))

    skipped = {}

    def leg(name, expected_s, run):
        """a secondary leg runs only while its expected cost fits the budget"""
        left = remaining()
        if left < expected_s:
            skipped[name] = ('skipped: {0:.0f} s of --time-budget left, the '
                             'leg is expected to take {1:.0f} s'
                             .format(left, expected_s))
            return dict(error=skipped[name])
        return run(left)

    level, refine = weak_ladder(world)
    # dt follows the mesh width: halved per refinement, and once more on the
    # level-3 mesh (explicit convection: dt=1/512 at Re=100 is past its
    # stability limit there -- the run blows up within 200 steps)
    nts_w = args.nts*2**refine*(2 if level >= 3 else 1)
    # latency-regime legs start from the steady Stokes state like the N=1
    # headline; the bandwidth ladder starts from rest on every N (its N=1
    # point, `refined_bench.run`, does too)
    if eager_ok or one_gpu_rehearsal or args.dry_run:
        weak = partitioned(17, level, refine, nts_w, 'stokes')
    else:
        weak = dict(error='not started: the RCCL self-test failed '
                    '(config.rccl_selftest)')
    # (only rank 0 holds the children's records: its verdict for everybody)
    ok_t = torch.tensor([1 if (isinstance(weak, dict) and 'error' not in weak)
                         else 0])
    dist.broadcast(ok_t, 0)
    headline_ok = bool(ok_t.item())
    # the same loop in the bandwidth regime (>= 7e5 rows per rank)
    bandwidth = None
    if not args.no_bandwidth and headline_ok:
        blevel, brefine = bandwidth_ladder(world)
        nts_b = args.nts*2**brefine*(2 if blevel >= 3 else 1)
        bandwidth = leg('weak_scaling_bandwidth', 300.,
                        lambda left: partitioned(
                            41, blevel, brefine, nts_b, 'rest',
                            timeout=min(1.5*args.partitioned_timeout,
                                        max(60., left - 90.))))
    elif not args.no_bandwidth:
        bandwidth = dict(error='not started: the headline leg failed')
    strong = None
    if not args.no_strong and (level, refine) != (args.level, 0) \
            and headline_ok:
        strong = leg('strong_scaling', 90.,
                     lambda left: partitioned(
                         29, args.level, 0, args.nts, 'stokes',
                         timeout=min(args.partitioned_timeout,
                                     max(60., left - 60.))))
    elif not args.no_strong and headline_ok:
        strong = 'identical to the headline run (same mesh)'

    # ensemble: every rank advances its own copy of the N=1 workload on its own
    # GPU -- a single-GPU run of this script per rank (own timed window of
    # exactly --steps steps each; the slowest rank counts)
    ensemble = None
    if not args.no_ensemble and headline_ok and remaining() < 75.:
        ensemble = dict(error='skipped: --time-budget spent', steps_per_s=None)
    elif not args.no_ensemble:
        env = {k: v for k, v in child_env().items()
               if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR',
                            'MASTER_PORT', 'GROUP_RANK', 'LOCAL_WORLD_SIZE',
                            'ROLE_RANK', 'ROLE_WORLD_SIZE', 'GROUP_WORLD_SIZE',
                            'ROLE_NAME')}
        one_gpu = os.environ.get('DNS_BENCH_REHEARSE_ONE_GPU') == '1'
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1',
               '--steps', str(args.steps), '--warmup', str(args.warmup),
               '--spinup', str(args.spinup), '--level', str(args.level),
               '--nts', str(args.nts), '--device',
               '0' if one_gpu else str(local_rank), '--no-cpu', '--no-refined',
               '--no-picard', '--roofline-refine', '0', '--no-force-dist',
               '--no-window-400', '--rtol', str(args.rtol)] + common
        dist.barrier()
        one = run_child(cmd, env, args.partitioned_timeout, True)
        ms = torch.tensor([one.get('ms_per_step', float('inf'))
                           if 'error' not in one else float('inf')],
                          dtype=torch.float64)
        dist.all_reduce(ms, op=dist.ReduceOp.MAX)
        slowest = float(ms.item())
        ensemble = dict(
            steps_per_s=(world*1e3/slowest if np.isfinite(slowest) else None),
            ms_per_step_slowest_rank=slowest if np.isfinite(slowest) else None,
            krylov_iters_per_step=(
                one.get('config', {}).get('krylov_iters_per_step')
                if 'error' not in one else None),
            error=one.get('error'),
            what='{0} independent simulations of the N=1 workload, one per '
                 'GPU, no data-path collective (Stokes start like the N=1 '
                 'headline)'.format(world))
    dist.barrier()

    out = None
    if rank == 0:
        ok = isinstance(weak, dict) and 'error' not in weak
        if ok:
            value, ms = weak['steps_per_s'], weak['ms_per_step']
            mode = ('ONE simulation, row-partitioned over {0} ranks: per-rank '
                    'row blocks, halo index lists (Send/Recv), one '
                    'all-reduce per Arnoldi step'.format(world))
            workload = ('cylinderwake N={0} refined {1}x Re={2:g} CNAB '
                        'dt=1/{3} Taylor-Hood NV={4} NP={5} (weak-scaling '
                        'ladder: {6:.0f} rows per rank); convection on the '
                        'device every step; state: steady Stokes solution '
                        'advanced {7} untimed spin-up steps'.format(
                            level, refine, args.Re, nts_w, weak['NV'],
                            weak['NP'], weak['rows_per_rank'], args.spinup))
            rs = weak.get('roofline_step') or {}
            roofline = dict(bound='hbm', achieved=rs.get('achieved'),
                            peak=world*HBM_PEAK_GBS, unit='GB/s',
                            frac=rs.get('frac'), traffic=None,
                            kernel='(whole CNAB step of the partitioned run: '
                            'published op list x measured Krylov steps / '
                            'measured time, all ranks together)', step=rs)
        else:
            # the partitioned run failed: say so, report the ensemble
            sps = (ensemble or {}).get('steps_per_s')
            value = sps if sps else 0.0
            ms = world*1e3/sps if sps else None
            mode = ('FALLBACK -- the row-partitioned headline run failed '
                    '(config.weak_scaling.error); value is the ensemble of '
                    '{0} independent simulations'.format(world))
            workload = ('cylinderwake N={0} Re={1:g} CNAB dt=1/{2}, one '
                        'independent simulation per GPU'.format(
                            args.level, args.Re, args.nts))
            roofline = None
        out = dict(
            metric='timesteps/sec, 2D cylinder wake Re={0:g} (CNAB step: device '
                   'convection + rhs SpMV + preconditioned Krylov saddle solve '
                   '+ p rescale)'.format(args.Re),
            value=value, unit='timesteps/s', n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=ms, higher_is_better=True,
            scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
            config=dict(workload=workload, parallelism=mode,
                        collectives=(weak.get('collectives_timed_window')
                                     if ok else None),
                        weak_scaling=weak, strong_scaling=strong,
                        weak_scaling_bandwidth=bandwidth,
                        ensemble=ensemble, rccl_selftest=rccl_selftest,
                        time_budget_s=args.time_budget,
                        elapsed_s=round(time.time() - t_start, 1),
                        spinup_steps=args.spinup,
                        method='gmres', cheb_degree=args.cheb,
                        factorization=args.fact, drop_tol=args.drop,
                        rtol=args.rtol,
                        krylov_iters_per_step=(
                            weak.get('krylov_iters_per_step') if ok else None),
                        note='no scaling curve is claimed here: the driver '
                        'computes it from the per-N lines'),
            roofline=roofline, cpu_baseline=None,
            # the partitioned run against an un-partitioned run of the same
            # steps from the same state on rank 0's GPU (partitioned_run)
            parity=(weak.get('parity') if isinstance(weak, dict) else None))
        line, _ = compact_line(out)
        print(json.dumps(line))
        sys.stdout.flush()
    dist.destroy_process_group()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=40)
    ap.add_argument('--spinup', type=int, default=256,
                    help='untimed steps between the impulsive start (Stokes '
                    'state) and the warm-up: the timed window then samples the '
                    'developed run, not the first instants of the start-up '
                    'transient (reported separately in config.early_transient)')
    ap.add_argument('--level', type=int, default=2, help='mesh level N')
    ap.add_argument('--Re', type=float, default=100.)
    ap.add_argument('--nts', type=int, default=512, help='dt = 1/nts')
    ap.add_argument('--method', default='gmres')
    ap.add_argument('--scheme', default='cnab', choices=['cnab', 'sbdf2'],
                    help='the resident loop that is timed: CNAB (tiu:104-143, '
                    'BASELINE headline) or SBDF2 (tiu:320-353)')
    ap.add_argument('--cheb', type=int, default=DEFAULTS['cheb'])
    ap.add_argument('--rtol', type=float, default=DEFAULTS['rtol'])
    ap.add_argument('--carry', type=int, default=1,
                    help='1: the residual of every solve is carried into the '
                    'next right-hand side (dns_imex_coeffs.carry_residual)')
    ap.add_argument('--extrap', type=int, default=DEFAULTS['extrap'],
                    help='warm start: 0 none, 1 linear, 2 quadratic, 3 cubic, '
                    '4 quartic')
    ap.add_argument('--fp32', type=int, default=DEFAULTS['fp32'],
                    help='store the explicit preconditioner matrices in fp32')
    ap.add_argument('--drop', type=float, default=DEFAULTS['drop'],
                    help='relative drop tolerance of the explicit polynomial')
    ap.add_argument('--fact', default=DEFAULTS['fact'],
                    help="block structure of the preconditioner: "
                    "'triangular' or 'full' (block LDU)")
    ap.add_argument('--reorth', type=int, default=DEFAULTS['reorth'],
                    help='1: Gram-Schmidt applied twice (CGS2), 0: once, '
                    '2: once, folded into the head kernel of the next step')
    ap.add_argument('--fhat', default='auto',
                    help="F^-1 approximation: 'cheb' recurrence, 'explicit' "
                    "polynomial matrix, 'auto'")
    ap.add_argument('--check-every', type=int, default=2)
    ap.add_argument('--eager', action='store_true',
                    help='plain launches instead of hipGraph replay')
    ap.add_argument('--roofline-refine', type=int, default=4,
                    help='red refinements of the mesh for the HBM roofline '
                    'SpMV (0 disables)')
    ap.add_argument('--profile-step', action='store_true',
                    help='profiling aid (rocprofv3 --kernel-trace --stats): '
                    'only the timed CNAB loop runs -- start from rest instead '
                    'of the Stokes solve, no secondary legs, no CPU leg -- so '
                    'that the kernel table is the table of the step')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-refined', action='store_true',
                    help='skip the secondary refined-mesh figures')
    ap.add_argument('--no-picard', action='store_true',
                    help='skip the secondary Newton/Picard sweep figures')
    ap.add_argument('--no-strong', action='store_true',
                    help='N>1: skip the secondary strong-scaling leg')
    ap.add_argument('--no-bandwidth', action='store_true',
                    help='N > 1: skip the bandwidth-regime weak-scaling leg')
    ap.add_argument('--no-ensemble', action='store_true',
                    help='N>1: skip the secondary ensemble leg')
    ap.add_argument('--refine', type=int, default=0,
                    help='red refinements of the mesh')
    ap.add_argument('--dense-max', type=int, default=12000,
                    help='(partitioned runs) largest pressure space with the '
                    'dense Schur inverse (held by rows per rank); beyond it '
                    'the multigrid Schur block')
    ap.add_argument('--device', type=int, default=0,
                    help='HIP device of a single-GPU run')
    ap.add_argument('--start', default='stokes', choices=['stokes', 'rest'],
                    help='(partitioned runs) start state: the steady Stokes '
                    'solution like the N=1 headline, or rest')
    ap.add_argument('--parity-full-max', type=int, default=200000,
                    help='(partitioned runs) up to this many unknowns the '
                    'WHOLE run is repeated un-partitioned on rank 0 for the '
                    'parity record; beyond it the first {0} steps'.format(
                        PARITY_PREFIX_STEPS))
    ap.add_argument('--construction', default='auto',
                    choices=['auto', 'rows', 'whole'],
                    help='partitioned runs: hand the library this rank\'s '
                    'rows only (dns_saddle_create_rows) or whole matrices; '
                    'auto = rows on more than one rank')
    ap.add_argument('--no-parity', action='store_true',
                    help='(partitioned runs, profiling aid) no un-partitioned '
                    'reference run: the kernel table of a profiled child is '
                    'then the table of the partitioned path alone')
    ap.add_argument('--no-window-400', dest='window_400',
                    action='store_false',
                    help='skip the second, longer timed window '
                    '(config.window_400)')
    ap.add_argument('--partitioned-only', action='store_true',
                    help='(internal) one row-partitioned simulation on the '
                    'ranks of this launch; prints its figures')
    ap.add_argument('--partitioned-timeout', type=float, default=420.,
                    help='time limit [s] of a row-partitioned child run')
    ap.add_argument('--dry-run', action='store_true',
                    help='(tests) the child processes of a multi-rank launch '
                    'report canned figures without touching a GPU: exercises '
                    'the launch / rendezvous / JSON plumbing on CPU')
    ap.add_argument('--no-force-dist', dest='force_dist',
                    action='store_false',
                    help='skip the run of the N=1 workload through the '
                    'row-partitioned code path on one RCCL rank '
                    '(config.row_partitioned: the code the N > 1 lines time)')
    ap.add_argument('--selftest-only', action='store_true',
                    help='(internal) first contact of the ranks of this '
                    'launch over RCCL: dns_comm_* only, a record per primitive')
    ap.add_argument('--selftest-timeout', type=float, default=120.,
                    help='time limit [s] of the RCCL self-test child')
    ap.add_argument('--time-budget', type=float, default=660.,
                    help='N > 1: wall-clock budget [s] of the whole line; the '
                    'self-test and the headline leg always run, a secondary '
                    'leg only while its expected cost still fits')
    ap.set_defaults(force_dist=True, window_400=True)
    args = ap.parse_args()

    internal = args.partitioned_only or args.selftest_only
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1 and not internal:
        # started without a launcher: start the rank processes here
        return self_launch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and not internal:
        sys.stderr.write('bench.py: --gpus {0} but the launcher started {1} '
                         'rank(s) (WORLD_SIZE): refusing to print a line for '
                         'another N\n'.format(args.gpus, world))
        raise SystemExit(2)
    # `kill -USR1 <pid>` (or `timeout -s USR1`): the Python stacks on stderr
    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, all_threads=True)
    if args.selftest_only:
        return selftest_child(args, world, rank, local_rank)
    if args.partitioned_only:
        return partitioned_child(args, world, rank, local_rank)
    if world > 1:
        return multi_gpu_main(args, world, rank, local_rank)
    if args.dry_run:
        print(json.dumps(dict(ms_per_step=1.0, dry_run=True,
                              config=dict(krylov_iters_per_step=1.0))))
        return None
    device = args.device

    from dolfin_navier_scipy_amd import saddle, _capi
    from dolfin_navier_scipy_amd import lin_alg_utils as lau
    if _capi.device_count() <= device:
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')

    dt = 1./args.nts
    if args.profile_step:
        args.roofline_refine = 0
    hbm_roof = hbm_roofline(saddle, args, dt, device) if rank == 0 \
        else (None, None, None)
    femp, sm, rhsd = build_problem(N=args.level, Re=args.Re,
                                   refine=args.refine)
    M, A, J = sm['M'], sm['A'], sm['J']
    prols = None
    if args.refine > 0:
        # refined mesh (partitioned-only child runs): the pressure space is
        # too large for the dense Schur inverse -> multigrid Schur block
        from dolfin_navier_scipy_amd.fem import (
            cylinder_mesh_hierarchy, pressure_prolongations, TaylorHood)
        hier = cylinder_mesh_hierarchy(N=args.level, refine=args.refine)
        spaces = [TaylorHood(m) for m, _ in hier][::-1]
        prols = pressure_prolongations(spaces, [p for _, p in hier][::-1])
    NP, NV = J.shape
    th, inv = femp['V'], femp['invinds']

    def factory(F, Jm):
        return saddle.SaddleSystem(F, Jm, device=device)

    if args.profile_step:
        args.no_cpu = args.no_refined = args.no_picard = True
        args.roofline_refine = 0
    if args.refine > 0 or args.profile_step:
        v0, pt0, st0 = np.zeros((NV, 1)), None, None     # start from rest
    else:
        v0, pt0, st0 = initial_state(sm, rhsd, factory)
    vfull = np.zeros((th.vdim, 1))
    vfull[inv] = v0
    vfull[femp['dbcinds'], 0] = femp['dbcvals']
    nfc = -th.convection_vec(vfull)[inv, :]      # snu:1136-1140

    F, R1, cfd, gdt = scheme_setup(args.scheme, M, A, dt)
    t_setup = time.perf_counter()
    system = factory(F, J)
    mode, scaling = ('single', 'weak')
    schur_kind = 'dense'
    if prols is not None:
        system.set_schur_mg(prols)
        schur_kind = 'mg'
    system.setup_precond(cheb_degree=args.cheb, schur=schur_kind,
                         fhat=args.fhat, fp32_store=bool(args.fp32),
                         drop_tol=args.drop, factorization=args.fact)
    _capi.device_synchronize(device)
    t_setup = time.perf_counter() - t_setup
    from dolfin_navier_scipy_amd import convection

    def conv_host(v):
        full = np.zeros((th.vdim, 1))
        full[inv] = v
        full[femp['dbcinds'], 0] = femp['dbcvals']
        return -th.convection_vec(full)[inv, :]          # snu:1136-1140

    cf = saddle.ImexStepper.coeffs(pscale=-1./dt, extrapolate=args.extrap,
                                   carry_residual=bool(args.carry), **cfd)
    opts = saddle.solve_opts(method=args.method, rtol=args.rtol, maxiter=400,
                             restart=60, check_every=args.check_every,
                             use_graph=not args.eager,
                             reorth=args.reorth)

    def barrier():
        _capi.device_synchronize(device)

    def timed_run(with_convection, nsteps=None, nwarm=None, spinup=None):
        nsteps = args.steps if nsteps is None else nsteps
        nwarm = args.warmup if nwarm is None else nwarm
        spinup = args.spinup if spinup is None else spinup
        stp = saddle.ImexStepper(system, R1)
        stp.set_state(v0, v_p=(v0 if args.scheme == 'sbdf2' else None),
                      nfc_c=nfc, nfc_o=nfc)
        stp.set_rhs(gdt*rhsd['fv'], rhsd['fp'])
        cvop = convection.ConvectionP2.from_taylor_hood(
            th, inv, femp['dbcinds'], femp['dbcvals'], device=device)
        stp.set_convection(cvop, scale=-1.0)
        if spinup > 0:
            stp.run(spinup, cf, opts)
        stp.run(nwarm, cf, opts)
        barrier()
        t0 = time.perf_counter()
        dev_s, its, lst = stp.run(nsteps, cf, opts)
        barrier()
        wl = time.perf_counter() - t0
        vv, pp = stp.get_state()
        lst = dict(lst, run_record=dict(stp.last_run))
        stp.close()
        cvop.close()
        return wl, dev_s, its, lst, vv, pp

    # headline: the complete time step, convection evaluated on the device
    wall, dev_s, iters, last, v_gpu, p_gpu = timed_run(True)
    if args.profile_step:
        wall_et, iters_et = wall, iters
    else:
        # secondary: the same window right behind the impulsive start (Stokes
        # state, no spin-up): the solves need about two Krylov steps there
        wall_et, _, iters_et, _, _, _ = timed_run(True, spinup=0)
    # a second, longer window of the same loop (the driver's 20 steps pay one
    # batch-begin kernel, the first graph launch and one host poll per call)
    window_400 = None
    if args.window_400 and not args.profile_step and args.steps != 400:
        w4, _, i4, l4, _, _ = timed_run(True, nsteps=400, nwarm=40)
        window_400 = dict(steps=400, warmup=40, steps_per_s=400/w4,
                          ms_per_step=1e3*w4/400,
                          krylov_iters_per_step=i4/400.,
                          true_relres_last=l4['true_relres'],
                          run_record=l4['run_record'])

    # self-test of the multi-GPU code path on one rank (RCCL communicator of
    # size 1): the same partitioned run the N > 1 headline times
    partitioned = None
    if args.force_dist and not args.profile_step:
        try:
            same = args.scheme == 'cnab' and args.refine == 0
            partitioned = partitioned_run(
                args, 1, 0, device, None, False, start=(v0,),
                reference=(v_gpu, p_gpu) if same else None)
        except Exception as exc:          # reported, never fatal
            partitioned = dict(error=str(exc))

    out = None
    if rank == 0:
        value = world*args.steps/wall
        from dolfin_navier_scipy_amd import perfmodel
        step_roof = perfmodel.step_roofline(
            system.precond_info(), int(R1.nnz), int(th.mesh.ncells),
            iters/float(args.steps), 1e3*wall/args.steps,
            peak_GBs=HBM_PEAK_GBS)
        roof = None if args.profile_step else roofline_spmv(
            saddle, saddle_csr(F, J), 200,
            'K at the benchmark size (cache resident)', variants=('vector',))
        roof_hbm, traffic, attain = hbm_roof
        main_roof = roof_hbm if roof_hbm is not None else roof
        if main_roof is None:          # --profile-step: the step's own figure
            main_roof = dict(achieved=step_roof['achieved'],
                             kernel='(whole CNAB step, see roofline.step)')
        roofline = dict(bound='hbm', achieved=main_roof['achieved'],
                        peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=main_roof['achieved']/HBM_PEAK_GBS,
                        traffic=traffic,
                        traffic_source=traffic_source(traffic),
                        kernel=main_roof['kernel'], detail=main_roof,
                        attainable_GBs=attain,
                        # like by like: PHYSICAL bytes (PMC traffic) per second
                        # over the physical read rate of a plain streaming
                        # kernel; `achieved` counts the algorithmic bytes
                        physical_GBs=(traffic/main_roof['avg_us']/1e3
                                      if traffic else None),
                        frac_of_attainable_read=(
                            traffic/main_roof['avg_us']/1e3/attain['read']
                            if (attain and traffic) else None),
                        at_benchmark_size=roof,
                        # the whole time step against the same peak: published
                        # op list x measured Krylov steps / measured time
                        step=step_roof)
        picard = None
        if world == 1 and not args.eager and not args.no_picard:
            picard = picard_sweep_figures(femp, sm, rhsd, v0, dt, device)
        refined = None
        bw_base = None
        if world == 1 and not args.eager and not args.no_refined:
            # bandwidth regime, end to end: the same CNAB loop on the mesh
            # refined twice (n = 173k, multigrid Schur block, dt/4) next to
            # the prefactored SuperLU step on the host
            import refined_bench
            refined = refined_bench.run(refine=2, nts=4*args.nts, nsteps=200,
                                        with_cpu=not args.no_cpu, Re=args.Re)
            roofline['step_refined'] = refined.pop('roofline_step', None)
            # N=1 point of the bandwidth-regime weak ladder (n = 693k): the
            # un-partitioned loop and the SAME run through the partitioned code
            # path on one RCCL rank (what the N > 1 lines time)
            lvl, ref = BANDWIDTH_LADDER[1]
            bw_base = refined_bench.run(refine=ref, nts=args.nts*2**ref,
                                        nsteps=100, with_cpu=False, Re=args.Re)
            bw_base['roofline_step'].pop('ops', None)
            if args.force_dist or os.environ.get('DNS_BENCH_BW_DIST', '1') == '1':
                import copy
                a2 = copy.copy(args)
                a2.level, a2.refine, a2.nts = lvl, ref, args.nts*2**ref
                a2.steps, a2.warmup, a2.spinup = 100, 20, 256
                a2.dense_max = 6000
                a2.start = 'rest'        # like refined_bench.run beside it
                try:
                    bw_base['partitioned_one_rank'] = partitioned_run(
                        a2, 1, 0, device, None, False)
                except Exception as exc:     # reported, never fatal
                    bw_base['partitioned_one_rank'] = dict(error=str(exc))
        cpu = None
        parity = None
        if not args.no_cpu and world == 1:     # rank 0 at N=1 only
            nall = args.spinup + args.warmup + args.steps
            cpu, v_cpu, p_cpu = cpu_baseline(
                sm, rhsd, v0, nfc, dt, conv_host, nall, scheme=args.scheme)
            cpu['other_lines'] = cpu_baseline_extras(sm, rhsd, v0, nfc, dt)
            # same steps, same nonlinear trajectory on both sides
            mn = lambda x: float(np.sqrt((x.T @ (M @ x)).item()))
            parity = dict(
                v_rel_Mnorm=mn(v_gpu - v_cpu)/mn(v_cpu),
                p_rel_l2=float(np.linalg.norm(p_gpu - p_cpu)
                               / np.linalg.norm(p_cpu)),
                steps=args.spinup + args.warmup + args.steps)
        out = dict(
            metric='timesteps/sec, 2D cylinder wake Re={0:g} ({1} step: device '
                   'convection + rhs SpMV + preconditioned Krylov saddle solve '
                   '+ p rescale)'
                   .format(args.Re, args.scheme.upper()),
            value=value, unit='timesteps/s', n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=1e3*wall/args.steps,
            higher_is_better=True, scaling=scaling, vs_baseline=None,
            dtype='f64', data='synthetic',
            config=dict(workload=(
                            'cylinderwake N={0} Re={1:g} {6} dt=1/{2} '
                            'Taylor-Hood NV={3} NP={4}; convection N(v)v '
                            'evaluated on the device every step; state: Stokes '
                            'solution advanced {5} untimed spin-up steps'
                            .format(args.level, args.Re, args.nts, NV, NP,
                                    args.spinup, args.scheme.upper())),
                        window_400=window_400,
                        spinup_steps=args.spinup,
                        early_transient=dict(
                            steps_per_s=world*args.steps/wall_et,
                            krylov_iters_per_step=iters_et/float(args.steps),
                            what='the same window with no spin-up (steps '
                            '{0}..{1} behind the impulsive start)'.format(
                                args.warmup, args.warmup + args.steps)),
                        parallelism=mode, collectives=None,
                        row_partitioned=partitioned,
                        scheme=args.scheme,
                        carry_residual=bool(args.carry),
                        weak_scaling_bandwidth_base=bw_base,
                        method=args.method, cheb_degree=args.cheb,
                        factorization=args.fact, drop_tol=args.drop,
                        schur='dense', rtol=args.rtol,
                        launch='eager' if args.eager else 'hipGraph',
                        krylov_iters_per_step=iters/float(args.steps),
                        true_relres_last=last['true_relres'],
                        timed_run_record=last['run_record'],
                        newton_picard_sweeps=picard,
                        refined_mesh=refined,
                        device_ms_per_step=1e3*dev_s/args.steps,
                        precond_setup_s=t_setup,
                        initial_stokes=st0,
                        device=_capi.device_name(device)),
            roofline=roofline, cpu_baseline=cpu, parity=parity)
        line, _ = compact_line(out)
        print(json.dumps(line))
    system.close()
    lau.clear_cache()
    return out


if __name__ == '__main__':
    main()
