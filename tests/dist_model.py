"""NumPy model of the ROW-PARTITIONED solve (test infrastructure).

Mirrors what `csrc/` does with a communicator attached: every rank keeps all
vectors in full, computes only its block of rows of each operator apply
(K, explicit Fh^-1 matrix, dense Sh^-1), all-gathers the blocks, and sums the
Gram-Schmidt dots / norms it took over its own rows by an all-reduce.  The
collectives are injected, so the same code runs serially (identity
collectives) and under `torch.distributed` (gloo) in the CPU tests.
"""
import numpy as np

import krylov_model as km


def starts_for(n, nranks):
    # (even chunks: the two velocity dofs of a node stay on one rank)
    chunk = ((n + nranks - 1)//nranks + 1) & ~1
    return [min(n, r*chunk) for r in range(nranks + 1)]


class SerialComm(object):
    rank, nranks = 0, 1

    def allreduce(self, buf):
        return buf

    def allgatherv(self, buf, starts):
        return buf


class PartitionedSaddle(object):
    def __init__(self, F, J, prec, comm):
        self.K = km.saddle(F, J).tocsr()
        self.prec, self.comm = prec, comm
        self.NP, self.NV = J.shape
        self.n = self.NV + self.NP
        P = comm.nranks
        self.st_n = starts_for(self.n, P)
        self.st_v = starts_for(self.NV, P)
        self.st_p = starts_for(self.NP, P)
        r = comm.rank
        self.Kloc = self.K[self.st_n[r]:self.st_n[r+1], :]
        self.Sloc = prec.Sinv[self.st_p[r]:self.st_p[r+1], :]
        self.JT = prec.JT

    def apply_K(self, z):
        r = self.comm.rank
        w = np.zeros(self.n)
        w[self.st_n[r]:self.st_n[r+1]] = self.Kloc @ z
        return self.comm.allgatherv(w, self.st_n)

    def apply_prec(self, rvec):
        r = self.comm.rank
        rv, rp = rvec[:self.NV], rvec[self.NV:]
        zp = np.zeros(self.NP)
        zp[self.st_p[r]:self.st_p[r+1]] = -(self.Sloc @ rp)
        zp = self.comm.allgatherv(zp, self.st_p)
        # Fh^-1: every rank could apply its rows of the explicit matrix; the
        # Chebyshev recurrence is row-local given the gathered direction
        zv = self.prec.cheb.apply(rv - self.JT @ zp)
        return np.concatenate([zv, zp])

    def dots_owned(self, V, w):
        """Gram-Schmidt dots over this rank's rows, then all-reduce"""
        r = self.comm.rank
        a, b = self.st_n[r], self.st_n[r+1]
        h = V[:, a:b] @ w[a:b]
        return self.comm.allreduce(h.copy())

    def gmres(self, b, x0=None, rtol=1e-10, restart=60, maxiter=400):
        n = self.n
        x = np.zeros(n) if x0 is None else x0.copy()
        bnorm = np.sqrt(self.dots_owned(b[None, :], b)[0])
        tol = rtol*bnorm
        its = 0
        hist = []
        while True:
            r = b - self.apply_K(x)
            beta = np.sqrt(self.dots_owned(r[None, :], r)[0])
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
                w = self.apply_K(self.apply_prec(V[j]))
                h = self.dots_owned(V[:j+1], w)
                w = w - V[:j+1].T @ h
                hn = np.linalg.norm(w)          # full vector, every rank alike
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
            x = x + self.apply_prec(V[:jend].T @ y)
        return x, np.array(hist), its
