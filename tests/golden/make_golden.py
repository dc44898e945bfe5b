"""Record golden vectors by executing the REFERENCE's own integrators.

Run in the build container only (the reference tree does not exist on the GPU
box and nothing of it is copied):

    python tests/golden/make_golden.py

`/root/reference/dolfin_navier_scipy/time_int_utils.py` is loaded by file path
(the package itself cannot be imported: its `__init__` needs dolfin).  Its one
un-vendored dependency, `sadptprj_riclyap_adi.lin_alg_utils`, is satisfied by
`oracle/saddle_oracle.py` (a sparse direct solve, see the header there).  The
reference functions are then run on the seeded scenarios of
`tests/scenarios.py`; inputs and outputs are written to
`tests/golden/imex_<scheme>_<variant>_s<seed>.npz`:

  inputs : trange; inivel; inip (the CSR triplets of M, A, J, shared by all
           scenarios, go to `imex_toy_system.npz`)
  outputs: times, velocities (with BCs), pressures of every `savevp` call,
           final `v, p`, `ffflag`; and the sequence of `f_vdp` outputs that
           were fed (so a device loop can be replayed from a table)

and, in `lau_calls.npz`, every call the reference integrators made to the
boundary itself while they ran -- `lau.solve_sadpnt_smw(amat=, jmat=, jmatT=,
rhsv=, rhsp=)` at tiu:402,466 (Heun start) and tiu:605 (`return_alu`) -- as
`(amat, rhsv, rhsp) -> vp` pairs ("per-call pairs for the
boundary itself"); `tests/test_gpu_saddle.py` replays them through the drop-in
`lin_alg_utils` on the GPU.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

REFTIU = '/root/reference/dolfin_navier_scipy/time_int_utils.py'


LAU_CALLS = []          # (tag, amat, rhsv, rhsp, vp) of the reference's calls
CALL_TAG = ['']


def recording_lau():
    """`oracle/saddle_oracle.py` behind a stub that records what the
    reference's integrators hand to / get from the boundary"""
    from oracle import saddle_oracle
    mod = types.ModuleType('recording_lau')

    def solve_sadpnt_smw(amat=None, jmat=None, rhsv=None, rhsp=None, **kw):
        out = saddle_oracle.solve_sadpnt_smw(amat=amat, jmat=jmat, rhsv=rhsv,
                                             rhsp=rhsp, **kw)
        sol = out[0] if isinstance(out, tuple) else out
        NP = jmat.shape[0]
        LAU_CALLS.append((CALL_TAG[0], amat.tocsr(), np.array(rhsv),
                          np.zeros((NP, 1)) if rhsp is None
                          else np.array(rhsp), np.array(sol)))
        return out
    mod.solve_sadpnt_smw = solve_sadpnt_smw
    mod.app_prj_via_sadpnt = saddle_oracle.app_prj_via_sadpnt
    return mod


def load_reference_tiu():
    saddle_oracle = recording_lau()
    pkg = types.ModuleType('sadptprj_riclyap_adi')
    pkg.lin_alg_utils = saddle_oracle
    sys.modules['sadptprj_riclyap_adi'] = pkg
    sys.modules['sadptprj_riclyap_adi.lin_alg_utils'] = saddle_oracle
    spec = importlib.util.spec_from_file_location('reference_tiu', REFTIU)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def csr_pack(prefix, mat):
    mat = mat.tocsr()
    mat.sort_indices()
    return {prefix+'_data': mat.data, prefix+'_indices': mat.indices,
            prefix+'_indptr': mat.indptr, prefix+'_shape': np.array(mat.shape)}


def main():
    import scenarios
    reftiu = load_reference_tiu()
    prob = scenarios.toy_problem()
    sysout = {}
    for name in 'MAJ':
        sysout.update(csr_pack(name, prob['smc'][name]))
    np.savez_compressed(os.path.join(HERE, 'imex_toy_system.npz'), **sysout)
    for scheme in ('cnab', 'sbdf2'):
        for seed, variant in enumerate(scenarios.VARIANTS):
            kw, rec, aux = scenarios.build(variant=variant, seed=seed,
                                           prob=prob)
            fed = []
            inner_fvdp = kw['f_vdp']

            def f_vdp(vfull, _f=inner_fvdp):
                out = _f(vfull)
                fed.append(out.reshape(-1).copy())
                return out
            kw['f_vdp'] = f_vdp
            CALL_TAG[0] = '{0}_{1}_s{2}'.format(scheme, variant, seed)
            if scheme == 'sbdf2':
                kw.pop('f_tvdp', None)
                v, p, ff = reftiu.sbdftwo(**kw)
            else:
                v, p, ff = reftiu.cnab(**kw)
            times, vels, prss = rec.arrays()
            out = dict(times=times, vels=vels, prss=prss, vfinal=v, pfinal=p,
                       ffflag=np.array(ff), trange=kw['trange'],
                       inivel=kw['inivel'], inip=kw['inip'],
                       fvdp_fed=np.array(fed), cfv=aux['cfv'], cfp=aux['cfp'])
            fn = os.path.join(HERE, 'imex_{0}_{1}_s{2}.npz'.format(
                scheme, variant, seed))
            np.savez_compressed(fn, **out)
            print(fn, vels.shape, 'ffflag', ff, '|v|', np.linalg.norm(v))

    # semi_implicit_euler (tiu:566-635): nonlinear rhs callback, data subset
    kw, rec, aux = scenarios.build(variant='plain', seed=3, prob=prob)
    trange = kw['trange']
    data_trange = trange[::3]
    fvdp, cfv, appnd = kw['f_vdp'], aux['cfv'], kw['appndbcs']

    def rhsv(t, vvec):
        return cfv + fvdp(appnd(vvec.reshape((-1, 1)), []))
    CALL_TAG[0] = 'sie_plain_s3'
    vlist = reftiu.semi_implicit_euler(
        iniv=kw['inivel'], jmat=kw['J'], mmat=kw['M'], amat=kw['A'],
        rhsv=rhsv, trange=trange, data_trange=data_trange, fp=aux['cfp'])
    out = dict(trange=trange, data_trange=data_trange, inivel=kw['inivel'],
               vlist=np.array([np.asarray(v).reshape(-1) for v in vlist]),
               cfv=cfv, cfp=aux['cfp'])
    fn = os.path.join(HERE, 'imex_sie_plain_s3.npz')
    np.savez_compressed(fn, **out)
    print(fn, out['vlist'].shape)

    # the boundary calls recorded on the way (all on the toy system's J)
    calls = {'ncalls': np.array(len(LAU_CALLS))}
    for k, (tag, amat, rv, rp, vp) in enumerate(LAU_CALLS):
        amat.sort_indices()
        calls['tag_{0}'.format(k)] = np.array(tag)
        calls['amat_data_{0}'.format(k)] = amat.data
        calls['amat_indices_{0}'.format(k)] = amat.indices
        calls['amat_indptr_{0}'.format(k)] = amat.indptr
        calls['rhsv_{0}'.format(k)] = rv
        calls['rhsp_{0}'.format(k)] = rp
        calls['vp_{0}'.format(k)] = vp
    fn = os.path.join(HERE, 'lau_calls.npz')
    np.savez_compressed(fn, **calls)
    print(fn, len(LAU_CALLS), 'calls')


if __name__ == '__main__':
    main()
