"""Executable NumPy model of the DEVICE algorithm (test infrastructure).

Mirrors, operation for operation, what `dolfin_navier_scipy_amd/csrc` runs on
the GPU: Jacobi-Chebyshev approximation of `F^-1`, dense Schur-complement
inverse built from that same polynomial, block upper-triangular right
preconditioner, GMRES with twice-applied classical Gram-Schmidt and Givens
rotations, and BiCGStab.  Used by the tests to cross-check iteration counts and
residual histories of the HIP path, and to choose parameters on the CPU.
It is NOT a fallback: nothing in the package imports it.
"""
import numpy as np
import scipy.sparse as sps


def power_bounds(F, dinv, nit_max=30, nit_min=50, seed=7):
    """eigenvalue bounds of `D^-1 F` by (shifted) power iterations"""
    n = F.shape[0]
    # deterministic start vector, same formula as the device code
    idx = np.arange(n, dtype=np.float64)
    x = 1.0 + 0.5*np.sin(0.37*idx + seed)
    lam = 0.
    for _ in range(nit_max):
        y = dinv*(F @ x)
        lam = np.linalg.norm(y)/np.linalg.norm(x)
        x = y/np.linalg.norm(y)
    lmax = lam
    x = 1.0 + 0.5*np.cos(0.61*idx + seed)
    shift = 1.05*lmax
    mu = 0.
    for _ in range(nit_min):
        y = shift*x - dinv*(F @ x)
        mu = np.linalg.norm(y)/np.linalg.norm(x)
        x = y/np.linalg.norm(y)
    lmin = shift - mu
    return lmin, lmax


class ChebJacobi(object):
    """`x ~= F^-1 b`: `degree` terms of the Chebyshev iteration on `D^-1 F`
    started at zero (Saad, Iterative Methods, Alg. 12.1)"""

    def __init__(self, F, degree=4, lmin=None, lmax=None, safety=(0.9, 1.05)):
        self.F = sps.csr_matrix(F)
        self.dinv = 1.0/self.F.diagonal()
        if lmin is None or lmax is None:
            lmin, lmax = power_bounds(self.F, self.dinv)
            lmin, lmax = safety[0]*lmin, safety[1]*lmax
        self.lmin, self.lmax = lmin, lmax
        self.degree = degree
        self.theta = 0.5*(lmax + lmin)
        self.delta = 0.5*(lmax - lmin)

    def coeffs(self):
        """`(c1_k, c2_k)` with `d_{k+1} = c1_k d_k + c2_k r_{k+1}`"""
        sigma1 = self.theta/self.delta
        rho = 1.0/sigma1
        out = []
        for _ in range(self.degree - 1):
            rho_new = 1.0/(2*sigma1 - rho)
            out.append((rho_new*rho, 2*rho_new/self.delta))
            rho = rho_new
        return out

    def apply(self, b):
        r = self.dinv*b
        d = r/self.theta
        x = np.zeros_like(b)
        for c1, c2 in self.coeffs():
            x = x + d
            r = r - self.dinv*(self.F @ d)
            d = c1*d + c2*r
        return x + d


class BlockTriPrecond(object):
    """`P = [[Fh, JT], [0, -Sh]]`, `Sh = J Fh^-1 JT` formed densely"""

    def __init__(self, F, J, JT=None, degree=4, cheb=None, schur_inv=None):
        self.J = sps.csr_matrix(J)
        self.JT = sps.csr_matrix(J.T) if JT is None else sps.csr_matrix(JT)
        self.NP, self.NV = self.J.shape
        self.cheb = ChebJacobi(F, degree=degree) if cheb is None else cheb
        if schur_inv is None:
            JTd = np.asarray(self.JT.todense())
            FiJT = np.stack([self.cheb.apply(JTd[:, c])
                             for c in range(self.NP)], axis=1)
            S = self.J @ FiJT
            S = 0.5*(S + S.T)
            schur_inv = np.linalg.inv(S)
        self.Sinv = schur_inv

    def apply(self, r):
        rv, rp = r[:self.NV], r[self.NV:]
        zp = -(self.Sinv @ rp)
        zv = self.cheb.apply(rv - self.JT @ zp)
        return np.concatenate([zv, zp])


class BlockFullPrecond(BlockTriPrecond):
    """`P = [[Fh, JT], [J, 0]]` exactly, as a block LDU product (the device's
    `DNS_FACT_FULL`): `zp = -Sh^-1 (rp - J Fh^-1 rv)`, `zv = Fh^-1 (rv - JT zp)`"""

    def apply(self, r):
        rv, rp = r[:self.NV], r[self.NV:]
        zp = -(self.Sinv @ (rp - self.J @ self.cheb.apply(rv)))
        zv = self.cheb.apply(rv - self.JT @ zp)
        return np.concatenate([zv, zp])


def gmres(K, b, prec, x0=None, rtol=1e-10, atol=0., restart=60, maxiter=400,
          reorth=True):
    """right-preconditioned GMRES(restart), CGS2; returns x, history"""
    n = b.size
    x = np.zeros(n) if x0 is None else x0.copy()
    bnorm = np.linalg.norm(b)
    tol = max(rtol*bnorm, atol)
    hist = []
    its = 0
    while True:
        r = b - K @ x
        beta = np.linalg.norm(r)
        hist.append(beta)
        if beta <= tol or its >= maxiter:
            break
        m = restart
        V = np.zeros((m+1, n))
        H = np.zeros((m+1, m))
        cs, sn, g = np.zeros(m), np.zeros(m), np.zeros(m+1)
        V[0] = r/beta
        g[0] = beta
        jend = 0
        for j in range(m):
            w = K @ prec.apply(V[j])
            h = V[:j+1] @ w
            w = w - V[:j+1].T @ h
            if reorth:
                h2 = V[:j+1] @ w
                w = w - V[:j+1].T @ h2
                h = h + h2
            hn = np.linalg.norm(w)
            H[:j+1, j] = h
            H[j+1, j] = hn
            V[j+1] = w/hn
            for i in range(j):
                t = cs[i]*H[i, j] + sn[i]*H[i+1, j]
                H[i+1, j] = -sn[i]*H[i, j] + cs[i]*H[i+1, j]
                H[i, j] = t
            den = np.hypot(H[j, j], H[j+1, j])
            cs[j], sn[j] = H[j, j]/den, H[j+1, j]/den
            H[j, j] = den
            H[j+1, j] = 0.
            g[j+1] = -sn[j]*g[j]
            g[j] = cs[j]*g[j]
            its += 1
            jend = j + 1
            hist.append(abs(g[j+1]))
            if abs(g[j+1]) <= tol or its >= maxiter:
                break
        y = np.linalg.solve(np.triu(H[:jend, :jend]), g[:jend])
        x = x + prec.apply(V[:jend].T @ y)
    return x, np.array(hist), its


def bicgstab(K, b, prec, x0=None, rtol=1e-10, atol=0., maxiter=400):
    n = b.size
    x = np.zeros(n) if x0 is None else x0.copy()
    bnorm = np.linalg.norm(b)
    tol = max(rtol*bnorm, atol)
    r = b - K @ x
    rhat = r.copy()
    rho = alpha = omega = 1.
    p = np.zeros(n)
    v = np.zeros(n)
    hist = [np.linalg.norm(r)]
    its = 0
    while hist[-1] > tol and its < maxiter:
        rho_new = rhat @ r
        beta = (rho_new/rho)*(alpha/omega)
        p = r + beta*(p - omega*v)
        y = prec.apply(p)
        v = K @ y
        alpha = rho_new/(rhat @ v)
        s = r - alpha*v
        z = prec.apply(s)
        t = K @ z
        omega = (t @ s)/(t @ t)
        x = x + alpha*y + omega*z
        r = s - omega*t
        rho = rho_new
        its += 1
        hist.append(np.linalg.norm(r))
    return x, np.array(hist), its


def saddle(F, J, JT=None):
    JT = sps.csr_matrix(J.T) if JT is None else JT
    return sps.bmat([[F, JT], [J, None]], format='csr')
