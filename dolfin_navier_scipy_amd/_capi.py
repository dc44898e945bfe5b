"""ctypes binding of `include/dns_amd.h` (the C-ABI of the gfx950 library).

The shared library is built in-tree by `__graft_entry__.build()` /
`dolfin_navier_scipy_amd.build.build_library()`; there is NO CPU fallback:
importing this module without the library raises, and any device failure
surfaces as a Python exception (the reference catches none from its linear
algebra layer either).
"""
import ctypes as ct
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, 'csrc', 'libdnsamd.so')

DNS_OK, DNS_NOT_CONVERGED, DNS_BREAKDOWN = 0, 1, 2
DNS_ERR_HIP, DNS_ERR_BAD_ARGUMENT, DNS_ERR_NOT_READY, DNS_ERR_COMM = 3, 4, 5, 6
DNS_ERR_HOST = 7
DNS_METHOD_GMRES, DNS_METHOD_BICGSTAB = 0, 1
DNS_SCHUR_DENSE, DNS_SCHUR_JACOBI = 0, 1
DNS_SPMV_VECTOR, DNS_SPMV_STREAM = 0, 1
DNS_FHAT_CHEB, DNS_FHAT_EXPLICIT, DNS_FHAT_AUTO = 0, 1, 2


class DnsError(RuntimeError):
    """a call into the device library failed"""

    def __init__(self, status, message):
        RuntimeError.__init__(self, 'dns_amd status {0}: {1}'.format(
            status, message))
        self.status = status


class NotConverged(DnsError):
    pass


class Breakdown(DnsError):
    pass


c_double_p = ct.POINTER(ct.c_double)
c_int32_p = ct.POINTER(ct.c_int32)


class dns_csr(ct.Structure):
    _fields_ = [('nrows', ct.c_int32), ('ncols', ct.c_int32),
                ('nnz', ct.c_int64), ('rowptr', c_int32_p),
                ('colidx', c_int32_p), ('vals', c_double_p)]


class dns_precond_opts(ct.Structure):
    _fields_ = [('cheb_degree', ct.c_int32), ('schur', ct.c_int32),
                ('fhat', ct.c_int32), ('fp32_store', ct.c_int32),
                ('eig_lo_safety', ct.c_double), ('eig_hi_safety', ct.c_double),
                ('eig_lo', ct.c_double), ('eig_hi', ct.c_double),
                ('drop_tol', ct.c_double), ('factorization', ct.c_int32),
                ('pad', ct.c_int32)]


class dns_solve_opts(ct.Structure):
    _fields_ = [('method', ct.c_int32), ('restart', ct.c_int32),
                ('maxiter', ct.c_int32), ('reorth', ct.c_int32),
                ('rtol', ct.c_double), ('atol', ct.c_double),
                ('check_every', ct.c_int32), ('use_graph', ct.c_int32)]


class dns_solve_stats(ct.Structure):
    _fields_ = [('iters', ct.c_int32), ('status', ct.c_int32),
                ('spmv_count', ct.c_int32), ('restarts', ct.c_int32),
                ('bnorm', ct.c_double), ('est_relres', ct.c_double),
                ('true_relres', ct.c_double), ('device_seconds', ct.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class dns_imex_coeffs(ct.Structure):
    _fields_ = [('a_c', ct.c_double), ('a_p', ct.c_double),
                ('cn_c', ct.c_double), ('cn_o', ct.c_double),
                ('pscale', ct.c_double), ('extrapolate_x0', ct.c_int32),
                ('carry_residual', ct.c_int32)]


# every symbol include/dns_amd.h declares: name -> (restype, argtypes)
_VP = ct.c_void_p
ALLREDUCE_CB = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int32)
ALLTOALLV_CB = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, c_int32_p,
                            c_int32_p, ct.c_void_p, c_int32_p, c_int32_p,
                            ct.c_int32)
ALLGATHERV_CB = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, c_int32_p,
                             ct.c_int32)
UNIQUE_ID_BYTES = 128
SIGNATURES = {
    'dns_version': (ct.c_int, []),
    'dns_status_string': (ct.c_char_p, [ct.c_int]),
    'dns_last_error': (ct.c_char_p, []),
    'dns_device_count': (ct.c_int, [ct.POINTER(ct.c_int)]),
    'dns_device_name': (ct.c_int, [ct.c_int, ct.c_char_p, ct.c_size_t]),
    'dns_device_synchronize': (ct.c_int, [ct.c_int]),
    'dns_saddle_create': (ct.c_int, [ct.c_int, ct.POINTER(dns_csr),
                                     ct.POINTER(dns_csr), ct.POINTER(dns_csr),
                                     ct.POINTER(_VP)]),
    'dns_saddle_create_rows': (ct.c_int, [ct.c_int, _VP, ct.c_int32,
                                          ct.c_int32, ct.POINTER(dns_csr),
                                          ct.POINTER(dns_csr),
                                          ct.POINTER(dns_csr),
                                          ct.POINTER(_VP)]),
    'dns_saddle_destroy': (None, [_VP]),
    'dns_saddle_update_values': (ct.c_int, [_VP, c_double_p]),
    'dns_saddle_setup_precond': (ct.c_int, [_VP,
                                            ct.POINTER(dns_precond_opts)]),
    'dns_saddle_solve': (ct.c_int, [_VP, c_double_p, c_double_p, c_double_p,
                                    c_double_p, ct.POINTER(dns_solve_opts),
                                    ct.POINTER(dns_solve_stats)]),
    'dns_saddle_solve_multi': (ct.c_int, [_VP, ct.c_int32, c_double_p,
                                          c_double_p, c_double_p, ct.c_int32,
                                          c_double_p,
                                          ct.POINTER(dns_solve_opts),
                                          ct.POINTER(dns_solve_stats)]),
    'dns_saddle_residual_history_col': (ct.c_int, [_VP, ct.c_int32, c_double_p,
                                                   ct.c_int32, c_int32_p]),
    'dns_saddle_residual_history': (ct.c_int, [_VP, c_double_p, ct.c_int32,
                                               c_int32_p]),
    'dns_saddle_apply': (ct.c_int, [_VP, c_double_p, c_double_p]),
    'dns_saddle_apply_precond': (ct.c_int, [_VP, c_double_p, c_double_p]),
    'dns_saddle_cheb_bounds': (ct.c_int, [_VP, c_double_p, c_double_p]),
    'dns_saddle_precond_info': (ct.c_int, [_VP, ct.c_int32,
                                           ct.POINTER(ct.c_int64),
                                           c_int32_p]),
    'dns_saddle_probe': (ct.c_int, [_VP, ct.c_int32, ct.c_int32, ct.c_int32,
                                    c_double_p]),
    'dns_comm_unique_id': (ct.c_int, [ct.c_char_p]),
    'dns_comm_create_rccl': (ct.c_int, [ct.c_int, ct.c_int32, ct.c_int32,
                                        ct.c_char_p, ct.POINTER(_VP)]),
    'dns_comm_create_callbacks': (ct.c_int, [ct.c_int, ct.c_int32, ct.c_int32,
                                             ALLREDUCE_CB, ALLGATHERV_CB, _VP,
                                             ct.POINTER(_VP)]),
    'dns_comm_destroy': (None, [_VP]),
    'dns_comm_stats': (ct.c_int, [_VP, ct.POINTER(ct.c_int64),
                                  ct.POINTER(ct.c_int64)]),
    'dns_saddle_host_bytes': (ct.c_int, [_VP, ct.POINTER(ct.c_int64),
                                         ct.POINTER(ct.c_int64)]),
    'dns_saddle_set_comm': (ct.c_int, [_VP, _VP]),
    'dns_partition_range': (ct.c_int, [ct.c_int32, ct.c_int32, ct.c_int32,
                                       c_int32_p, c_int32_p]),
    'dns_device_read': (ct.c_int, [ct.c_int, _VP, _VP, ct.c_size_t]),
    'dns_device_write': (ct.c_int, [ct.c_int, _VP, _VP, ct.c_size_t]),
    'dns_default_precond_opts': (None, [ct.POINTER(dns_precond_opts)]),
    'dns_default_solve_opts': (None, [ct.POINTER(dns_solve_opts)]),
    'dns_imex_create': (ct.c_int, [_VP, ct.POINTER(dns_csr),
                                   ct.POINTER(_VP)]),
    'dns_imex_create_rows': (ct.c_int, [_VP, ct.POINTER(dns_csr),
                                        ct.POINTER(_VP)]),
    'dns_imex_destroy': (None, [_VP]),
    'dns_imex_set_state': (ct.c_int, [_VP, c_double_p, c_double_p, c_double_p,
                                      c_double_p, c_double_p]),
    'dns_imex_set_rhs': (ct.c_int, [_VP, c_double_p, c_double_p]),
    'dns_imex_step': (ct.c_int, [_VP, c_double_p,
                                 ct.POINTER(dns_imex_coeffs),
                                 ct.POINTER(dns_solve_opts),
                                 ct.POINTER(dns_solve_stats)]),
    'dns_imex_run': (ct.c_int, [_VP, ct.c_int32, ct.POINTER(dns_imex_coeffs),
                                ct.POINTER(dns_solve_opts),
                                ct.POINTER(dns_solve_stats), c_double_p,
                                ct.POINTER(ct.c_int64)]),
    'dns_imex_get_state': (ct.c_int, [_VP, c_double_p, c_double_p]),
    'dns_imex_vnorm': (ct.c_int, [_VP, c_double_p]),
    'dns_imex_run_info': (ct.c_int, [_VP, c_int32_p, c_int32_p, c_int32_p,
                                     c_int32_p]),
    'dns_imex_step_counters': (ct.c_int, [_VP, ct.POINTER(ct.c_int64)]),
    'dns_conv_create_p2': (ct.c_int, [ct.c_int, ct.c_int32, c_int32_p,
                                      c_double_p, c_double_p, ct.c_int32,
                                      ct.c_int32, c_int32_p, ct.c_int32,
                                      c_int32_p, c_double_p, ct.POINTER(_VP)]),
    'dns_conv_destroy': (None, [_VP]),
    'dns_conv_set_dbcvals': (ct.c_int, [_VP, c_double_p]),
    'dns_conv_apply': (ct.c_int, [_VP, c_double_p, ct.c_double, c_double_p]),
    'dns_imex_set_convection': (ct.c_int, [_VP, _VP, ct.c_double]),
    'dns_spmv': (ct.c_int, [ct.c_int, ct.POINTER(dns_csr), c_double_p,
                            c_double_p, ct.c_double, ct.c_double, ct.c_int32]),
    'dns_dot': (ct.c_int, [ct.c_int, ct.c_int64, c_double_p, c_double_p,
                           c_double_p]),
    'dns_axpy': (ct.c_int, [ct.c_int, ct.c_int64, ct.c_double, c_double_p,
                            c_double_p]),
    'dns_gemv': (ct.c_int, [ct.c_int, ct.c_int32, c_double_p, c_double_p,
                            c_double_p, ct.c_double]),
    'dns_dense_inverse': (ct.c_int, [ct.c_int, ct.c_int32, c_double_p]),
    'dns_spmv_pair': (ct.c_int, [ct.c_int, ct.POINTER(dns_csr), ct.c_int32,
                                 c_double_p, c_double_p, ct.c_int32,
                                 ct.c_int32, c_double_p,
                                 ct.POINTER(ct.c_int64)]),
    'dns_spmv_bench': (ct.c_int, [ct.c_int, ct.POINTER(dns_csr), ct.c_int32,
                                  ct.c_int32, ct.c_int32, c_double_p,
                                  c_double_p]),
    'dns_saddle_set_schur_mg': (ct.c_int, [ct.c_void_p, ct.c_int32,
                                           ct.POINTER(dns_csr), ct.c_int32]),
    'dns_saddle_set_option': (ct.c_int, [ct.c_void_p, ct.c_char_p,
                                         ct.c_double]),
    'dns_conv_bind_pattern': (ct.c_int, [ct.c_void_p, ct.POINTER(dns_csr)]),
    'dns_conv_assemble': (ct.c_int, [ct.c_void_p, c_double_p, ct.c_int32,
                                     c_double_p, c_double_p, c_double_p]),
    'dns_trap_create': (ct.c_int, [ct.c_void_p, ct.c_void_p, c_double_p,
                                   c_double_p, ct.c_int32,
                                   ct.POINTER(ct.c_void_p)]),
    'dns_trap_destroy': (None, [ct.c_void_p]),
    'dns_trap_set_rhs': (ct.c_int, [ct.c_void_p, c_double_p, c_double_p]),
    'dns_trap_traj_write': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32,
                                       c_double_p]),
    'dns_trap_traj_export_async': (ct.c_int, [ct.c_void_p, ct.c_int32,
                                              ct.c_int32, ct.c_int32,
                                              c_double_p]),
    'dns_trap_traj_export_wait': (ct.c_int, [ct.c_void_p]),
    'dns_trap_traj_read': (ct.c_int, [ct.c_void_p, ct.c_int32, ct.c_int32,
                                      c_double_p]),
    'dns_trap_start': (ct.c_int, [ct.c_void_p, c_double_p, ct.c_int32]),
    'dns_trap_step': (ct.c_int, [ct.c_void_p, ct.c_double, ct.c_int32,
                                 ct.c_int32, ct.c_int32, ct.c_int32,
                                 ct.c_int32, ct.POINTER(dns_solve_opts),
                                 ct.POINTER(dns_solve_stats)]),
    'dns_trap_run': (ct.c_int, [ct.c_void_p, ct.c_double, ct.c_int32,
                                ct.c_int32, ct.c_int32, ct.c_int32,
                                ct.c_int32, ct.POINTER(dns_solve_opts)]),
    'dns_trap_get_state': (ct.c_int, [ct.c_void_p, c_double_p, c_double_p]),
    'dns_trap_update_norm': (ct.c_int, [ct.c_void_p, c_double_p]),
    'dns_trap_set_pipeline': (ct.c_int, [ct.c_void_p, ct.c_int32]),
    'dns_trap_set_oversolve': (ct.c_int, [ct.c_void_p, ct.c_double]),
    'dns_trap_poll_ext': (ct.c_int, [ct.c_void_p, ct.POINTER(ct.c_int32),
                                     ct.POINTER(ct.c_double)]),
    'dns_trap_poll': (ct.c_int, [ct.c_void_p, ct.POINTER(ct.c_int32),
                                 ct.POINTER(ct.c_int32),
                                 ct.POINTER(ct.c_int32),
                                 ct.POINTER(ct.c_int32)]),
    'dns_trap_checkpoint': (ct.c_int, [_VP]),
    'dns_trap_restore': (ct.c_int, [_VP, ct.c_int32]),
    'dns_hbm_probe': (ct.c_int, [ct.c_int, ct.c_int64, ct.c_int32, ct.c_int32,
                                 c_double_p]),
    'dns_imex_set_rhs_table': (ct.c_int, [_VP, ct.c_int32, c_double_p,
                                          c_double_p]),
    'dns_imex_table_position': (ct.c_int, [_VP, c_int32_p, c_int32_p]),
    'dns_conv_set_dbc_table': (ct.c_int, [_VP, ct.c_int32, c_double_p]),
    'dns_conv_set_dbc_row': (ct.c_int, [_VP, ct.c_int32]),
    'dns_conv_assemble2': (ct.c_int, [_VP, c_double_p, c_double_p, c_double_p,
                                      ct.c_int32, c_double_p, c_double_p,
                                      c_double_p]),
    'dns_trap_set_tables': (ct.c_int, [_VP, c_double_p, c_double_p,
                                       c_double_p]),
    'dns_trap_step_fb': (ct.c_int, [_VP, ct.c_double, ct.c_int32, ct.c_int32,
                                    ct.c_int32, ct.c_int32, ct.c_int32,
                                    ct.POINTER(dns_solve_opts),
                                    ct.POINTER(dns_solve_stats), ct.c_int32,
                                    c_double_p, c_double_p, c_double_p]),
    'dns_comm_set_alltoallv_cb': (ct.c_int, [_VP, ALLTOALLV_CB]),
    'dns_comm_stats2': (ct.c_int, [_VP, ct.POINTER(ct.c_int64)]),
    'dns_comm_set_timing': (ct.c_int, [_VP, ct.c_int]),
    'dns_comm_selftest': (ct.c_int, [_VP, ct.c_int32, ct.c_int32, ct.c_int32,
                                     ct.c_int32, ct.POINTER(ct.c_int32),
                                     ct.POINTER(ct.c_double)]),
    'dns_comm_gather_forms': (ct.c_int, [_VP, ct.POINTER(ct.c_int64)]),
    'dns_comm_set_gather_form': (ct.c_int, [_VP, ct.c_int32]),
    'dns_comm_timing': (ct.c_int, [_VP, ct.POINTER(ct.c_double),
                                   ct.POINTER(ct.c_int64)]),
    'dns_halo_lists': (ct.c_int, [ct.POINTER(dns_csr), ct.c_int32, ct.c_int32,
                                  ct.c_int32, ct.c_int32, c_int32_p,
                                  ct.c_int32, c_int32_p, c_int32_p,
                                  ct.c_int64, ct.POINTER(ct.c_int64)]),
    'dns_saddle_device_bytes': (ct.c_int, [_VP, ct.POINTER(ct.c_int64)]),
    'dns_op_create': (ct.c_int, [ct.c_int, ct.POINTER(dns_csr),
                                 ct.POINTER(_VP)]),
    'dns_op_destroy': (None, [_VP]),
    'dns_op_apply': (ct.c_int, [_VP, c_double_p, c_double_p, ct.c_double,
                                ct.c_double]),
    'dns_bcmap_create': (ct.c_int, [ct.c_int, ct.c_int32, ct.c_int32,
                                    c_int32_p, ct.c_int32, c_int32_p,
                                    ct.POINTER(_VP)]),
    'dns_bcmap_destroy': (None, [_VP]),
    'dns_bc_scatter': (ct.c_int, [_VP, c_double_p, c_double_p, c_double_p]),
}

_lib = None


def load_library(path=None):
    """load (once) the in-tree shared library and declare all prototypes"""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = LIBPATH if path is None else path
    if not os.path.exists(path):
        raise ImportError(
            'the gfx950 library {0} is missing -- run '
            '`python -c "import __graft_entry__ as g; g.build()"` at the repo '
            'root (there is no CPU fallback)'.format(path))
    lib = ct.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if a symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    if status == DNS_OK:
        return
    msg = load_library().dns_last_error().decode('utf-8', 'replace')
    if status == DNS_NOT_CONVERGED:
        raise NotConverged(status, msg)
    if status == DNS_BREAKDOWN:
        raise Breakdown(status, msg)
    raise DnsError(status, msg)


def dptr(arr):
    return arr.ctypes.data_as(c_double_p) if arr is not None else None


def as_f64(arr, size=None):
    out = np.ascontiguousarray(np.asarray(arr, dtype=np.float64).reshape(-1))
    if size is not None and out.size != size:
        raise ValueError('expected {0} entries, got {1}'.format(size,
                                                                out.size))
    return out


class CsrView(object):
    """keeps the int32/float64 arrays of a SciPy matrix alive next to the
    `dns_csr` struct that points into them"""

    def __init__(self, mat):
        import scipy.sparse as sps
        mat = sps.csr_matrix(mat)
        if not mat.has_canonical_format:
            mat = mat.copy()
            mat.sum_duplicates()
        self.shape = mat.shape
        self.indptr = np.ascontiguousarray(mat.indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(mat.indices, dtype=np.int32)
        self.data = np.ascontiguousarray(mat.data, dtype=np.float64)
        self.struct = dns_csr(
            nrows=mat.shape[0], ncols=mat.shape[1], nnz=self.data.size,
            rowptr=self.indptr.ctypes.data_as(c_int32_p),
            colidx=self.indices.ctypes.data_as(c_int32_p),
            vals=self.data.ctypes.data_as(c_double_p))

    def byref(self):
        return ct.byref(self.struct)


def device_count():
    n = ct.c_int(0)
    try:
        check(load_library().dns_device_count(ct.byref(n)))
    except DnsError:
        return 0
    return n.value


def device_synchronize(device=0):
    check(load_library().dns_device_synchronize(device))


def device_name(device=0):
    buf = ct.create_string_buffer(256)
    check(load_library().dns_device_name(device, buf, 256))
    return buf.value.decode()
