"""Parity of the standalone HIP kernels (through the C-ABI) with NumPy/SciPy."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def sad():
    from dolfin_navier_scipy_amd import saddle, _capi
    assert _capi.device_count() > 0, 'HIP device required for -m gpu tests'
    return saddle


def _rand_csr(rng, nrows, ncols, density):
    mat = sps.random(nrows, ncols, density=density, random_state=rng,
                     format='csr', dtype=np.float64)
    mat.data = rng.uniform(-1, 1, mat.nnz)
    return mat


@pytest.mark.parametrize('variant', ['vector', 'stream', 'stream16'])
@pytest.mark.parametrize('shape,density', [((1, 1), 1.0), ((7, 5), 0.5),
                                           ((300, 300), 0.02),
                                           ((1000, 1300), 0.03),
                                           ((257, 4000), 0.9),   # long rows
                                           ((5000, 64), 0.001)])  # empty rows
def test_spmv_random(sad, variant, shape, density):
    rng = np.random.default_rng(shape[0]*7 + shape[1])
    A = _rand_csr(rng, shape[0], shape[1], density)
    x = rng.standard_normal(shape[1])
    y = sad.spmv(A, x, variant=variant)
    ref = A @ x
    scale = np.abs(A).dot(np.abs(x)).max() + 1e-300
    assert np.abs(y - ref).max() <= 1e-13*scale


@pytest.mark.parametrize('variant', ['vector', 'stream', 'stream16'])
def test_spmv_alpha_beta_and_fem_matrix(sad, variant, toy_prob):
    rng = np.random.default_rng(1)
    A = toy_prob['smc']['A']
    x = rng.standard_normal(A.shape[1])
    b = rng.standard_normal(A.shape[0])
    y = sad.spmv(A, x, y=b, alpha=-0.5, beta=2.0, variant=variant)
    ref = -0.5*(A @ x) + 2.0*b
    assert np.abs(y - ref).max() <= 1e-13*np.abs(ref).max()


def test_spmv_empty_matrix(sad):
    A = sps.csr_matrix((10, 12))
    for variant in ('vector', 'stream', 'stream16'):
        y = sad.spmv(A, np.ones(12), variant=variant)
        assert np.array_equal(y, np.zeros(10))


def test_spmv_linearity_full_size(sad):
    """size-independent property at the benchmark size (cylinder N=2)"""
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=80)
    A = (sm['M'] + 1e-3*sm['A']).tocsr()
    rng = np.random.default_rng(2)
    x1, x2 = rng.standard_normal((2, A.shape[1]))
    for variant in ('vector', 'stream', 'stream16'):
        y1 = sad.spmv(A, x1, variant=variant)
        y2 = sad.spmv(A, x2, variant=variant)
        y12 = sad.spmv(A, 2*x1 - 3*x2, variant=variant)
        assert np.abs(y12 - (2*y1 - 3*y2)).max() <= 1e-12*np.abs(y12).max()
        assert np.abs(y1 - A @ x1).max() <= 1e-13*np.abs(y1).max()


@pytest.mark.parametrize('n', [0, 1, 63, 64, 65, 1000, 100003])
def test_dot_axpy(sad, n):
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal((2, n)) if n else (np.zeros(0), np.zeros(0))
    d = sad.dot(x, y)
    assert abs(d - float(x @ y)) <= 1e-12*max(1.0, np.abs(x*y).sum())
    z = sad.axpy(0.75, x, y)
    assert np.allclose(z, y + 0.75*x, rtol=0, atol=1e-15*max(1, n))


@pytest.mark.parametrize('n', [1, 5, 64, 130, 401])
def test_gemv_and_dense_inverse(sad, n):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = B @ B.T + n*np.eye(n)
    x = rng.standard_normal(n)
    y = sad.gemv(A, x, alpha=-1.0)
    assert np.abs(y + A @ x).max() <= 1e-12*np.abs(A @ x).max()
    Ai = sad.dense_inverse(A)
    assert np.abs(Ai @ A - np.eye(n)).max() <= 1e-10


def test_dense_inverse_zero_pivot_raises(sad):
    from dolfin_navier_scipy_amd import _capi
    A = np.zeros((4, 4))
    with pytest.raises(_capi.Breakdown):
        sad.dense_inverse(A)


def test_spmv_stream16_windows_and_raw_blocks(sad):
    """16-bit column offsets: a saddle-type matrix whose rows reach two far
    apart column windows (second base), and a matrix with columns scattered
    over more than two 32768-wide windows (blocks fall back to the raw 32-bit
    indices)"""
    rng = np.random.default_rng(5)
    n = 150000
    rows = np.repeat(np.arange(n), 6)
    near = (rows + rng.integers(-40, 41, rows.size)) % n
    far = 100000 + (rows // 3 + rng.integers(0, 30, rows.size)) % 40000
    cols = np.where(rng.random(rows.size) < 0.7, near, far)
    A = sps.csr_matrix((rng.standard_normal(rows.size), (rows, cols)),
                       shape=(n, n))
    A.sum_duplicates()
    rb, cb = rng.integers(0, 40000, 1600000), rng.integers(0, 200000, 1600000)
    B = sps.csr_matrix((rng.standard_normal(rb.size), (rb, cb)),
                       shape=(40000, 200000))
    B.sum_duplicates()
    for mat in (A, B):
        x = rng.standard_normal(mat.shape[1])
        ref = mat @ x
        got = sad.spmv(mat, x, variant='stream16')
        assert np.abs(got - ref).max() <= 1e-12*np.abs(ref).max()
        got = sad.spmv(mat, x, y=np.ones(mat.shape[0]), alpha=0.5, beta=-2.0,
                       variant='stream16')
        assert np.abs(got - (0.5*ref - 2.0)).max() <= 1e-12*np.abs(ref).max()


def _pair_k(rng, nodes, npres, dens=0.05, coupled=True):
    """synthetic K = [[F, JT], [J, 0]] with 2 x nodes velocity dofs: ragged
    rows, empty rows, node blocks that are full (`coupled`) or diagonal, and
    one-sided J / JT entries"""
    Fs = (_rand_csr(rng, nodes, nodes, dens) + sps.identity(nodes)).tocoo()
    rows = np.concatenate([2*Fs.row, 2*Fs.row + 1])
    cols = np.concatenate([2*Fs.col, 2*Fs.col + 1])
    vals = np.concatenate([Fs.data, 1.5*Fs.data])
    if coupled:
        rows = np.concatenate([rows, 2*Fs.row, 2*Fs.row + 1])
        cols = np.concatenate([cols, 2*Fs.col + 1, 2*Fs.col])
        vals = np.concatenate([vals, -.3*Fs.data, .7*Fs.data])
    F = sps.coo_matrix((vals, (rows, cols)), shape=(2*nodes, 2*nodes)).tocsr()
    J = _rand_csr(rng, npres, 2*nodes, dens)       # x / y entries independent
    if npres > 3:
        J = J.tolil()
        J[1, :] = 0
        J = J.tocsr()
        J.eliminate_zeros()
    K = sps.bmat([[F, J.T], [J, None]], format='csr')
    K.sort_indices()
    return K, 2*nodes


@pytest.mark.parametrize('nodes,npres,coupled', [(1, 1, True), (5, 3, False),
                                                 (200, 40, True),
                                                 (3000, 400, True),
                                                 (3000, 400, False),
                                                 (700, 0, True)])
def test_spmv_pair_format_matches_scipy(sad, nodes, npres, coupled):
    rng = np.random.default_rng(nodes + npres)
    K, nv = _pair_k(rng, nodes, npres, dens=min(0.5, 12./nodes),
                    coupled=coupled)
    x = rng.standard_normal(K.shape[1])
    y = sad.spmv_pair(K, nv, x)
    ref = K @ x
    assert np.abs(y - ref).max() <= 1e-13*max(1., np.abs(ref).max())


def test_spmv_pair_format_refuses_what_it_cannot_hold(sad):
    from dolfin_navier_scipy_amd import _capi
    rng = np.random.default_rng(8)
    K, nv = _pair_k(rng, 50, 11)         # (an odd number of pressure dofs is
    x = rng.standard_normal(K.shape[1])  # fine: pressure is not paired)
    y = sad.spmv_pair(K, nv, x)
    assert np.abs(y - K @ x).max() <= 1e-13*np.abs(y).max()
    with pytest.raises(_capi.DnsError):
        sad.spmv_pair(K, nv - 1, x)      # odd number of velocity dofs
    pp = K.tolil()
    pp[nv, nv + 3] = 1.0                 # pressure-pressure entry
    with pytest.raises(_capi.DnsError):
        sad.spmv_pair(pp.tocsr(), nv, x)


def test_spmv_pair_format_on_the_cylinder_system(sad):
    """the K of the cylinder wake on the mesh refined once: the
    result equals the CSR product to rounding, the format is smaller than the
    16-bit CSR (dense 2x2 node blocks: 8.5 bytes per non-zero)"""
    from dolfin_navier_scipy_amd.fem import get_sysmats
    femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=100,
                                 refine=1)
    F = (sm['M'] + .5/1024*sm['A']).tocsr()
    J = sm['J']
    K = sps.bmat([[F, J.T], [J, None]], format='csr')
    K.sort_indices()
    rng = np.random.default_rng(4)
    x = rng.standard_normal(K.shape[1])
    y, secs, fbytes = sad.spmv_pair(K, F.shape[0], x, reps=3)
    ref = K @ x
    assert np.abs(y - ref).max() <= 1e-13*np.abs(ref).max()
    assert fbytes < 10*K.nnz             # < 8 B value + 2 B offset per non-zero
