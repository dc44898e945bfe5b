"""Algorithmic bytes of one time step -- the published op list behind
`bench.py`'s `roofline.step` ("the builder must publish
the per-iteration op list so the figure is reproducible").

Conventions of SURVEY 8d: fp64 values, int32 indices, every array touched
once, `x` read once:

    CSR SpMV  y = A x, A r x c : 12 nnz + 4 (r+1) + 8 c + 8 r
    axpy 24 n, dot 16 n, nrm2 8 n; fused variants count each vector once

The counts are ALGORITHMIC (what the mathematics of the step needs at fp64 /
int32), not what the kernels move: the explicit preconditioner matrices are
stored with fp32 values and partly 16-bit column offsets, so the kernels move
less -- `roofline.step.bytes_moved_estimate` carries that second figure.
"""

__all__ = ['spmv_bytes', 'krylov_iteration_ops', 'cnab_step_ops',
           'step_roofline']


def spmv_bytes(nnz, rows, cols, val_bytes=8, idx_bytes=4):
    return (val_bytes + idx_bytes)*nnz + 4*(rows + 1) + 8*cols + 8*rows


def _schur_ops(info, moved):
    """ops of one application of the Schur block `zp = -Sh^-1 tau`"""
    NP = info['NP']
    fp32 = info['fp32_store'] and moved
    if info['schur'] == 'dense':
        vb = 4 if fp32 else 8
        return [('Sh^-1 dense GEMV (NP x NP)', 1, vb*NP*NP + 16*NP)]
    if info['schur'] == 'jacobi':
        return [('Sh^-1 diagonal', 1, 24*NP)]
    ops = []
    nu = info.get('mg_nu', 2)
    lv = info['mg_levels']
    for l, lev in enumerate(lv[:-1]):
        n, nc = lev['n'], lv[l + 1]['n']
        sp = spmv_bytes(lev['nnz_S'], n, n)
        # a damped-Jacobi sweep = SpMV(S) + b, dinv, x_old read, x_new written
        ops.append(('MG level {0}: Jacobi sweeps (pre+post)'.format(l),
                    2*nu, sp + 32*n))
        ops.append(('MG level {0}: residual b - S x'.format(l), 1, sp + 8*n))
        ops.append(('MG level {0}: restriction P^T r'.format(l), 1,
                    spmv_bytes(lev['nnz_P'], nc, n)))
        ops.append(('MG level {0}: prolongation x += P e'.format(l), 1,
                    spmv_bytes(lev['nnz_P'], n, nc) + 8*n))
    ncoarse = lv[-1]['n']
    # (a half-precision inverse, dense_half.hpp, is counted as stored in both
    # columns: it is dense at that size only because it is half)
    cb = info.get('mg_coarse_val_bytes', 8)
    vb = 2 if cb == 2 else (4 if (fp32 and cb == 4) else 8)
    ops.append(('MG coarsest: dense inverse GEMV', 1,
                vb*ncoarse*ncoarse + 16*ncoarse))
    ncyc = info.get('mg_cycles', 1)
    if ncyc > 1:
        # two cycles as two Richardson steps: everything above once more,
        # the residual in - S_0 x_1 between them and the combination
        ops = [(name, cnt*ncyc, b) for name, cnt, b in ops]
        n0 = lv[0]['n']
        ops.append(('MG: residual between the cycles, SpMV(S_0)', ncyc - 1,
                    spmv_bytes(lv[0]['nnz_S'], n0, n0) + 8*n0))
        ops.append(('MG: x_1 + a_2 V(r)', ncyc - 1, 24*n0))
    return ops


def krylov_iteration_ops(info, j=0, moved=False):
    """ops of Arnoldi step `j` (0-based) of the right-preconditioned GMRES with
    the block-LDU preconditioner; `moved`: count the bytes the kernels move
    (fp32 values / 16-bit column offsets where they are used) instead of the
    algorithmic fp64/int32 bytes"""
    NV, NP = info['NV'], info['NP']
    n = NV + NP
    ops = []
    big = n > 400000          # streaming kernels (16-bit column offsets)
    if info['nnz_JG']:
        ib = 2 if (moved and big) else 4
        ops.append(('tau = r_p - (J Fh^-1) r_v   SpMV(JG)', 1,
                    spmv_bytes(info['nnz_JG'], NP, NV, idx_bytes=ib) + 8*NP))
    # head: V_j = (w - sum h_i V_i)/||.||  (reads w and j basis vectors,
    # writes V_j) + the Schur block
    ops.append(('head: V_j from w and V_0..j-1 (fused Gram-Schmidt)', 1,
                8*n*(j + 2)))
    ops += _schur_ops(info, moved)
    if info['nnz_Gc']:
        vb = 4 if (moved and info['fp32_store']) else 8
        ib = 2 if (moved and NV > 350000) else 4
        ops.append(('zv = Gc [V_j,v; zp]   SpMV(Gc), explicit polynomial '
                    'Fh^-1 and -Fh^-1 JT', 1,
                    spmv_bytes(info['nnz_Gc'], NV, n, val_bytes=vb,
                               idx_bytes=ib)))
    else:
        deg = info['cheb_degree']
        ops.append(('zv: Chebyshev recurrence, SpMV(JT) + (deg-1) SpMV(F)', 1,
                    spmv_bytes(info['nnz_J'], NV, NP)
                    + (deg - 1)*(spmv_bytes(info['nnz_F'], NV, NV) + 32*NV)))
    ib = 2 if (moved and big) else 4
    ops.append(('w = K z   SpMV(K) fused with the j+1 dots <V_i, w>, <w, w>',
                1, spmv_bytes(info['nnz_K'], n, n, idx_bytes=ib)
                + 8*n*(j + 1)))
    return ops


def cnab_step_ops(info, nnz_R1, ncells, iters, moved=False):
    """ops of one resident CNAB step (tiu:104-143) with `iters` Krylov steps"""
    NV, NP = info['NV'], info['NP']
    n = NV + NP
    ops = [
        ('convection N(v)v: P2 element kernel (12 dof gathers, 12 stores, '
         'cellmap/glam/area per cell)', 1, ncells*(12*4 + 6*8 + 8 + 24*8)),
        ('convection gather into nfc (inverted index)', 1,
         12*ncells*(4 + 8) + 4*(NV + 1) + 8*NV),
        ('rhs_v = R1 v_c + cn_c nfc_c + cn_o nfc_o + g   SpMV(R1) + 3 '
         'vectors', 1, spmv_bytes(nnz_R1, NV, NV) + 24*NV),
        ('warm start x0 = quartic extrapolation of 5 solutions', 1, 8*n*6),
        ('r = b - K x0, ||r||^2, ||b||^2   SpMV(K) + 2 vectors', 1,
         spmv_bytes(info['nnz_K'], n, n) + 16*n),
    ]
    full, frac = int(iters), iters - int(iters)
    for j in range(full):
        for name, cnt, b in krylov_iteration_ops(info, j, moved):
            ops.append(('it {0}: {1}'.format(j, name), cnt, b))
    if frac > 1e-12:
        for name, cnt, b in krylov_iteration_ops(info, full, moved):
            ops.append(('it {0} (x{1:.3f}): {2}'.format(full, frac, name),
                        cnt*frac, b))
    ops.append(('x = x0 + Z y (kept Z_j = P^-1 V_j), p rescale', 1,
                8*n*(max(1., iters) + 2)))
    return ops


def step_roofline(info, nnz_R1, ncells, iters, ms_per_step, peak_GBs=8000.,
                  regime=None):
    """the `roofline.step` block of the bench line"""
    if info.get('mg_cycles', 1) > 1 and \
            iters > info.get('mg_two_cycle_maxc', 1000) + 0.5:
        # (the second V-cycle runs in cycles of at most that many columns)
        info = dict(info, mg_cycles=1)
    ops = cnab_step_ops(info, nnz_R1, ncells, iters, moved=False)
    total = sum(c*b for _, c, b in ops)
    moved = sum(c*b for _, c, b in
                cnab_step_ops(info, nnz_R1, ncells, iters, moved=True))
    secs = 1e-3*ms_per_step
    n = info['NV'] + info['NP']
    if regime is None:
        # resident set of a step: K, Gc, JG, Sinv + ~20 vectors
        regime = ('cache resident (L2 / 256 MB Infinity Cache): bound by '
                  'dependent-launch and memory latency, not HBM'
                  if total < 128e6 else
                  'HBM bound (working set leaves the caches)')
    return dict(bound='hbm', unit='GB/s', peak=peak_GBs,
                bytes_per_step=total, bytes_moved_estimate=moved,
                krylov_iters_per_step=iters, ms_per_step=ms_per_step,
                achieved=total/secs/1e9, frac=total/secs/1e9/peak_GBs,
                achieved_moved=moved/secs/1e9, unknowns=int(n),
                regime=regime,
                ops=[dict(op=name, count=round(c, 4), bytes=int(b))
                     for name, c, b in ops])
