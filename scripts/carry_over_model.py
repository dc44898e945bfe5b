"""CPU prototype: does carrying the velocity residual of step k into the rhs of
step k+1 stop the linear drift of a one-Krylov-step-per-time-step CNAB run?"""
import sys, time
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spsla
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from dolfin_navier_scipy_amd.fem import get_sysmats
import krylov_model as km
Re = float(sys.argv[1]) if len(sys.argv) > 1 else 100
NST = int(sys.argv[2]) if len(sys.argv) > 2 else 400
femp, sm, rhsd = get_sysmats(problem='cylinderwake', N=2, Re=Re)
th, inv = femp['V'], femp['invinds']
M, A, J = sm['M'].tocsr(), sm['A'].tocsr(), sm['J'].tocsr()
NP, NV = J.shape
dt = 1./512
F = (M + .5*dt*A).tocsr(); R1 = (M - .5*dt*A).tocsr()
K = km.saddle(F, J).tocsc()
lu = spsla.splu(K)
K = K.tocsr()
t0 = time.time()
prec = km.BlockFullPrecond(F, J, degree=6)
print('prec', time.time()-t0)
Ks = sps.bmat([[A, J.T], [J, None]], format='csc')
vp0 = spsla.spsolve(Ks, np.concatenate([rhsd['fv'][:, 0], rhsd['fp'][:, 0]]))
dbcinds, dbcvals = femp['dbcinds'], femp['dbcvals']
def conv(v):
    full = np.zeros((th.vdim, 1)); full[inv, 0] = v; full[dbcinds, 0] = dbcvals
    return -th.convection_vec(full)[inv, 0]
fv, fp = rhsd['fv'][:, 0], rhsd['fp'][:, 0]
def mn(x): return np.sqrt(x @ (M @ x))
def run(mode):
    # mode: 'exact', 'one', 'comp'
    x = np.concatenate([vp0[:NV], np.zeros(NP)])
    hist = [x.copy()]
    nfo = conv(x[:NV]); out = []
    rprev = np.zeros(NV)
    for k in range(NST):
        nfc = conv(x[:NV])
        b = np.concatenate([R1 @ x[:NV] + dt*fv + 1.5*dt*nfc - .5*dt*nfo, fp])
        if mode == 'comp':
            b[:NV] += rprev
        nfo = nfc
        if mode == 'exact' or len(hist) < 5:
            xn = lu.solve(b)
        else:
            h = hist
            x0 = 5*h[-1] - 10*h[-2] + 10*h[-3] - 5*h[-4] + h[-5]
            xn, hh, its = km.gmres(K, b, prec, x0=x0, rtol=1e-10, maxiter=20, restart=20)
            if k % 50 == 0: print(mode, k, "its", its, "relres", hh[-1]/np.linalg.norm(b), hh[0]/np.linalg.norm(b))
        rprev = (b - K @ xn)[:NV]
        x = xn
        hist.append(x.copy()); hist = hist[-5:]
        out.append(x.copy())
    return out
ex = run('exact')
for mode in ('one', 'comp'):
    o = run(mode)
    for k in range(49, NST, 50):
        ev = mn(o[k][:NV]-ex[k][:NV])/mn(ex[k][:NV])
        ep = np.linalg.norm(o[k][NV:]-ex[k][NV:])/np.linalg.norm(ex[k][NV:])
        print(mode, k+1, 'v %.3e p %.3e' % (ev, ep))
