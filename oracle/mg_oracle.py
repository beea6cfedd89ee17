"""ORACLE (test infrastructure) - host restatement of the multigrid V-cycle and the flexible GMRES of
nupgcm_amd/csrc/mg.hip, in numpy / scipy.

This is NEW work with no counterpart in the reference (its only non-diagonal preconditioner is the experimental
BlockDiagonalPreconditioner of /root/reference/src/preconditioners.jl:53-125): there is nothing of the reference to pin it
against - "parity unpinned" in that sense - and it exists so that the tests can check the device implementation operation by
operation (same operators, same order of updates) and not only through the solutions it produces, which ARE checked against
the fixture-pinned direct solve.  Only tests/ may import it.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


class Level:
    def __init__(self, A, nu, Dinv, P=None):
        self.A = sp.csr_matrix(A)
        self.nu = nu
        self.G, self.D = self.A[:nu, nu:], self.A[nu:, :nu]
        self.Dinv = sp.csr_matrix(Dinv)
        self.S = sp.csr_matrix(self.D @ self.Dinv @ self.G)
        self.sdinv = 1.0 / self.S.diagonal()
        self.P = P


def smooth(l: Level, x, b, nsteps, omega, jw, sweeps):
    """Braess-Sarazin steps: [w Dh, G; D, 0] [du; dp] = r with S dp = D Dh^-1 r_u - w r_p relaxed by damped Jacobi"""
    nu = l.nu
    for _ in range(nsteps):
        r = b - l.A @ x
        t = l.Dinv @ r[:nu]
        rhs = l.D @ t - omega * r[nu:]
        dp = jw * l.sdinv * rhs
        for _k in range(1, sweeps):
            dp = dp + jw * l.sdinv * (rhs - l.S @ dp)
        du = l.Dinv @ (r[:nu] - l.G @ dp) / omega
        x = x + np.concatenate([du, dp])
    return x


def vcycle(levels, lev, b, omega=2.5, jw=0.7, sweeps=3, nu1=2, nu2=2, coarse=20):
    l = levels[lev]
    x = np.zeros_like(b)
    if lev == 0:
        return smooth(l, x, b, coarse, omega, jw, sweeps)
    x = smooth(l, x, b, nu1, omega, jw, sweeps)
    r = b - l.A @ x
    x = x + l.P @ vcycle(levels, lev - 1, l.P.T @ r, omega, jw, sweeps, nu1, nu2, coarse)
    return smooth(l, x, b, nu2, omega, jw, sweeps)


def fgmres(A, b, M, x0=None, m=20, scale=1.0, atol=1e-6, rtol=1e-6, itmax=0):
    """right-preconditioned restarted FGMRES(m); stop when scale ||r|| <= atol + rtol scale ||r0||.
    Returns x, dict(niter, residuals (scaled estimates), solved)."""
    n = len(b)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=float)
    itmax = itmax or 2 * n
    r = b - A @ x
    beta = np.linalg.norm(r)
    r0 = scale * beta
    eps = atol + rtol * r0
    hist = [r0]
    it = 0
    solved = r0 <= eps
    while not solved and it < itmax:
        V = np.zeros((m + 1, n))
        Z = np.zeros((m, n))
        H = np.zeros((m + 1, m))
        g = np.zeros(m + 1)
        cs, sn = np.zeros(m), np.zeros(m)
        V[0] = r / beta
        g[0] = beta
        k = 0
        for j in range(m):
            if it >= itmax:
                break
            Z[j] = M(V[j]) if M is not None else V[j]
            w = A @ Z[j]
            h1 = V[:j + 1] @ w
            w = w - V[:j + 1].T @ h1
            h2 = V[:j + 1] @ w
            w = w - V[:j + 1].T @ h2
            H[:j + 1, j] = h1 + h2
            hn = np.linalg.norm(w)
            H[j + 1, j] = hn
            for i in range(j):
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            d = np.hypot(H[j, j], H[j + 1, j])
            cs[j], sn[j] = H[j, j] / d, H[j + 1, j] / d
            H[j, j], H[j + 1, j] = d, 0.0
            g[j + 1] = -sn[j] * g[j]
            g[j] = cs[j] * g[j]
            it += 1
            k = j + 1
            hist.append(scale * abs(g[j + 1]))
            if hist[-1] <= eps:
                break
            V[j + 1] = w / hn
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
        x = x + Z[:k].T @ y
        r = b - A @ x
        beta = np.linalg.norm(r)
        solved = scale * beta <= eps
    return x, dict(niter=it, residuals=hist, solved=bool(solved), rnorm=scale * beta)
