"""ORACLE (test infrastructure) - restatement of the two Krylov.jl 0.10.6 solvers the reference drives.

Krylov.jl is an un-vendored dependency of the reference (Manifest.toml pins 0.10.6); its call site is
/root/reference/src/iterative_solvers.jl:58 with the workspaces and keyword arguments of
/root/reference/src/inversion.jl:74-94 (GMRES, memory=20, restart=true, atol=rtol=1e-6, itmax=0 => 2n) and
/root/reference/src/evolution.jl:118-126 (CG).  What is restated is the published algorithm:

  gmres : left-preconditioned restarted GMRES(m), *modified* Gram-Schmidt, Givens QR of the Hessenberg matrix kept as
          packed upper-triangular R, residual estimate |zeta_{k+1}|, stop when ||M r|| <= atol + rtol ||M r0||, warm start
          solves A dx = b - A x0 and returns x0 + dx, true residual recomputed at every restart.
  cg    : preconditioned CG, stop when sqrt(r'z) <= atol + rtol sqrt(r0'z0), warm start likewise.

Iteration-level behaviour of Krylov.jl is NOT pinned by any fixture of the reference (no test records iteration counts
or residual histories) - "parity unpinned" for iteration counts; the *solutions* are pinned through the direct-solve
fixtures to within the solver tolerance (test K5).
"""
from __future__ import annotations

import numpy as np


def sym_givens(a, b):
    """[c s; s -c] [a; b] = [rho; 0] (real case)."""
    if b == 0.0:
        c = 1.0 if a == 0.0 else np.sign(a)
        return c, 0.0, abs(a)
    if a == 0.0:
        return 0.0, np.sign(b), abs(b)
    if abs(b) > abs(a):
        t = a / b
        s = np.sign(b) / np.sqrt(1.0 + t * t)
        c = s * t
        return c, s, b / s
    t = b / a
    c = np.sign(a) / np.sqrt(1.0 + t * t)
    s = c * t
    return c, s, a / c


def gmres(A, b, x0=None, M=None, memory=20, atol=1e-6, rtol=1e-6, itmax=0, restart=True):
    """Returns x, stats(dict: solved, niter, residuals).  M: vector (diagonal inverse action) or scalar or None."""
    n = len(b)
    apply_M = (lambda v: v.copy()) if M is None else (lambda v: M * v)
    x = np.zeros(n)
    if x0 is not None:
        dx0 = np.array(x0, dtype=float)
        w = b - A @ dx0
        x += dx0
    else:
        w = b.copy()
    r0 = apply_M(w)
    beta = np.linalg.norm(r0)
    rnorm = beta
    hist = [beta]
    eps_ = atol + rtol * rnorm
    if beta == 0.0:
        return x, dict(solved=True, niter=0, residuals=hist)
    if itmax == 0:
        itmax = 2 * n
    inner_itmax = itmax
    btol = np.finfo(float).eps ** 0.75
    it = 0
    npass = 0
    solved = rnorm <= eps_
    tired = it >= itmax
    breakdown = False
    mem = memory
    V = np.zeros((mem, n))
    while not (solved or tired or breakdown):
        c = np.zeros(mem)
        s = np.zeros(mem)
        R = np.zeros(mem * (mem + 1) // 2)
        z = np.zeros(mem)
        V[:] = 0.0
        dx = np.zeros(n)
        if npass >= 1:
            w = b - A @ x
            r0 = apply_M(w)
        beta = np.linalg.norm(r0)
        z[0] = beta
        V[0] = r0 / beta
        npass += 1
        inner = 0
        nr = 0
        inner_tired = False
        while not (solved or inner_tired or breakdown):
            inner += 1
            w = A @ V[inner - 1]
            q = apply_M(w)
            for i in range(inner):
                R[nr + i] = V[i] @ q
                q -= R[nr + i] * V[i]
            hbis = np.linalg.norm(q)
            for i in range(inner - 1):
                tmp = c[i] * R[nr + i] + s[i] * R[nr + i + 1]
                R[nr + i + 1] = s[i] * R[nr + i] - c[i] * R[nr + i + 1]
                R[nr + i] = tmp
            c[inner - 1], s[inner - 1], R[nr + inner - 1] = sym_givens(R[nr + inner - 1], hbis)
            zeta = s[inner - 1] * z[inner - 1]
            z[inner - 1] = c[inner - 1] * z[inner - 1]
            rnorm = abs(zeta)
            hist.append(rnorm)
            nr += inner
            solved = (rnorm <= eps_) or (rnorm + 1.0 <= 1.0)
            breakdown = hbis <= btol
            inner_tired = inner >= min(mem, inner_itmax)
            if not (solved or inner_tired or breakdown):
                V[inner] = q / hbis
                z[inner] = zeta
        y = z.copy()
        for i in range(inner - 1, -1, -1):
            pos = nr + i - inner
            for j in range(inner - 1, i, -1):
                y[i] -= R[pos] * y[j]
                pos -= j
            y[i] = 0.0 if abs(R[pos]) <= btol else y[i] / R[pos]
        for i in range(inner):
            dx += y[i] * V[i]
        x += dx
        inner_itmax -= inner
        it += inner
        tired = it >= itmax
    return x, dict(solved=bool(solved), niter=it, residuals=hist)


def cg(A, b, x0=None, M=None, atol=1e-6, rtol=1e-6, itmax=0):
    n = len(b)
    apply_M = (lambda v: v.copy()) if M is None else (lambda v: M * v)
    x = np.zeros(n)
    if x0 is not None:
        dx0 = np.array(x0, dtype=float)
        r = b - A @ dx0
    else:
        dx0 = None
        r = b.copy()
    z = apply_M(r)
    p = z.copy()
    gamma = r @ z
    rnorm = np.sqrt(gamma)
    hist = [rnorm]
    if gamma == 0.0:
        if dx0 is not None:
            x += dx0
        return x, dict(solved=True, niter=0, residuals=hist)
    if itmax == 0:
        itmax = 2 * n
    eps_ = atol + rtol * rnorm
    solved = rnorm <= eps_
    it = 0
    pnorm2 = gamma
    tiny = np.finfo(float).eps
    while not (solved or it >= itmax):
        Ap = A @ p
        pAp = p @ Ap
        if pAp <= tiny * pnorm2:
            break                       # zero / negative curvature: Krylov.jl stops here (status only)
        alpha = gamma / pAp
        x += alpha * p
        r -= alpha * Ap
        z = apply_M(r)
        gamma_next = r @ z
        rnorm = np.sqrt(gamma_next)
        hist.append(rnorm)
        solved = (rnorm <= eps_) or (rnorm + 1.0 <= 1.0)
        if not solved:
            beta = gamma_next / gamma
            pnorm2 = gamma_next + beta * beta * pnorm2
            gamma = gamma_next
            p = z + beta * p
        it += 1
    if dx0 is not None:
        x += dx0
    return x, dict(solved=bool(solved), niter=it, residuals=hist)
