import sys, time
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spsla
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from dolfin_navier_scipy_amd.fem import get_sysmats
import krylov_model as km
femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=100)
M, A, J = sm['M'].tocsr(), sm['A'].tocsr(), sm['J'].tocsr()
NP, NV = J.shape
dt = 1./512
F = (M + .5*dt*A).tocsr()
K = km.saddle(F, J).tocsr()
prec = km.BlockFullPrecond(F, J, degree=6)
S64 = prec.Sinv.copy()
rng = np.random.default_rng(0)
# a warm-start-like residual: smooth velocity part, no pressure part
x = np.concatenate([np.sin(0.01*np.arange(NV)), np.zeros(NP)])
r = np.concatenate([(M @ rng.standard_normal(NV))*1e-9, np.zeros(NP)])
def onestep(Sinv):
    prec.Sinv = Sinv
    z = prec.apply(r)
    w = K @ z
    al = (w @ r)/(w @ w)
    return np.linalg.norm(r - al*w)/np.linalg.norm(r)
def rnd(Sv, dtype):
    sc = np.abs(Sv).max(axis=1, keepdims=True)
    return (Sv/sc).astype(dtype).astype(np.float64)*sc
print('fp64', onestep(S64))
print('fp32', onestep(S64.astype(np.float32).astype(np.float64)))
print('fp16 rowscaled', onestep(rnd(S64, np.float16)))
# bf16: truncate mantissa of fp32 to 8 bits
def bf16(a):
    b = a.astype(np.float32).view(np.uint32)
    b = ((b + 0x8000) & 0xFFFF0000).astype(np.uint32)
    return b.view(np.float32).astype(np.float64)
print('bf16', onestep(bf16(S64)))
