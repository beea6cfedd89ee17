"""ORACLE (test infrastructure) - CPU restatement of ILU(0) and of the preconditioned CG the reference wraps around it:

    P_prec = KrylovPreconditioners.kp_ilu0(P)
    CgPreconditioner(P, P_prec; ldiv=true, label="P-block", itmax=100)            /root/reference/src/preconditioners.jl:101-107
    Krylov.krylov_solve!(workspace, matrix, x, workspace.x, M=preconditioner, ldiv=true, itmax=itmax)        :24-37

The algorithm itself lives in third-party dependencies that are not vendored under /root/reference: KrylovPreconditioners.jl
0.3.7 (Manifest.toml:692-696), whose CUDA extension calls cuSPARSE csrilu02 (zero fill-in incomplete LU on the matrix's own
pattern, no pivoting) and csrsv2 (sparse triangular solves; L with a unit diagonal), and Krylov.jl's `cg`.  Restated from the
published definition (Saad, Iterative Methods for Sparse Linear Systems, 2nd ed., Alg. 10.4 "ILU(0), IKJ variant"):

    for i = 2..n:  for k in pattern(i), k < i (ascending):  a_ik /= a_kk ;  for j in pattern(i), j > k:  a_ij -= a_ik a_kj (if (k, j) in pattern)

PINNING: the reference holds no test, fixture or golden vector for this path (its test suite never builds a
BlockDiagonalPreconditioner) and the dependency cannot run here - "parity unpinned" against reference OUTPUT.  What pins the
restatement instead is the defining property, checked in tests/test_oracle_ilu0.py: on a pattern closed under elimination (no
fill is dropped) ILU(0) is the exact LU factorisation (compared with scipy's dense LU without pivoting on diagonally dominant
matrices), and on a general pattern (L U - A) vanishes on A's pattern (Saad, Prop. 10.2).
Pure-Python loops: small matrices only.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def ilu0(A):
    """values of the ILU(0) factors in A's (sorted CSR) pattern: strictly lower part = L (unit diagonal implied), rest = U"""
    A = sp.csr_matrix(A).astype(np.float64)
    A.sort_indices()
    rp, ci, v = A.indptr, A.indices, A.data.copy()
    n = A.shape[0]
    diag = np.full(n, -1, dtype=np.int64)
    for i in range(n):
        for k in range(rp[i], rp[i + 1]):
            if ci[k] == i:
                diag[i] = k
    if (diag < 0).any():
        raise ValueError("ilu0: a row has no diagonal entry")
    for i in range(n):
        pos = {int(ci[k]): k for k in range(rp[i], rp[i + 1])}
        for kk in range(rp[i], diag[i]):
            k = int(ci[kk])
            v[kk] /= v[diag[k]]
            lik = v[kk]
            for jj in range(diag[k] + 1, rp[k + 1]):
                p = pos.get(int(ci[jj]))
                if p is not None:
                    v[p] -= lik * v[jj]
    return sp.csr_matrix((v, ci.copy(), rp.copy()), shape=A.shape)


def split(LU):
    """(L with its unit diagonal, U) as CSR matrices"""
    L = sp.tril(LU, -1, format="csr") + sp.identity(LU.shape[0], format="csr")
    U = sp.triu(LU, 0, format="csr")
    return sp.csr_matrix(L), sp.csr_matrix(U)


def solve(LU, r):
    """z = U^-1 L^-1 r"""
    L, U = split(LU)
    t = spla.spsolve_triangular(L, r, lower=True, unit_diagonal=True)
    return spla.spsolve_triangular(U, t, lower=False)


def levels(A):
    """number of levels of the lower / upper dependency graphs of A's pattern (what a level-scheduled solve launches)"""
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    rp, ci = A.indptr, A.indices
    lo, up = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    for i in range(n):
        c = ci[rp[i]:rp[i + 1]]
        c = c[c < i]
        lo[i] = lo[c].max() + 1 if len(c) else 0
    for i in range(n - 1, -1, -1):
        c = ci[rp[i]:rp[i + 1]]
        c = c[c > i]
        up[i] = up[c].max() + 1 if len(c) else 0
    return int(lo.max()) + 1, int(up.max()) + 1


def pcg(A, b, LU, x0=None, atol=1e-6, rtol=1e-6, itmax=0):
    """Krylov.jl cg(A, b, x0; M = (LU), ldiv = true): returns (x, iterations, solved, [sqrt(r'z) history])"""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    if itmax <= 0:
        itmax = 2 * n
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    r = b - A @ x
    z = solve(LU, r)
    p = z.copy()
    gamma = float(r @ z)
    rn0 = np.sqrt(max(gamma, 0.0))
    eps = atol + rtol * rn0
    hist = [rn0]
    it = 0
    solved = rn0 <= eps
    while not solved and it < itmax:
        Ap = A @ p
        pAp = float(p @ Ap)
        if not pAp > 0.0:
            break
        alpha = gamma / pAp
        x += alpha * p
        r -= alpha * Ap
        z = solve(LU, r)
        g2 = float(r @ z)
        it += 1
        rn = np.sqrt(max(g2, 0.0))
        hist.append(rn)
        if rn <= eps:
            solved = True
            break
        p = z + (g2 / gamma) * p
        gamma = g2
    return x, it, solved, hist
