"""IterativeSolverToolkit / iterative_solve! - mirrors /root/reference/src/iterative_solvers.jl:1-68.

On GPU() the Krylov workspaces are the device-resident solvers of libnupgcm_hip.so (one C call per solve, x warm-started
because `x` aliases `workspace.x` exactly as in the reference, src/iterative_solvers.jl:26-29)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .architectures import DeviceCSR, DeviceVector


class Diagonal:
    """Diagonal(v): the preconditioner type of src/inversion.jl:54 and src/evolution.jl:149,167 (`M` is applied with
    mul!, so it holds the INVERSE action).  `scalar` is set when every entry is the same number - the device then folds
    it into the SpMV epilogue without streaming a vector."""

    def __init__(self, diag=None, scalar=None, n=None):
        self.diag, self.scalar = diag, scalar
        self.n = n if n is not None else (len(diag) if diag is not None else None)

    def kind(self):
        if self.scalar is not None:
            return L.NPG_PRECOND_SCALAR, float(self.scalar), None
        return L.NPG_PRECOND_DIAG, 0.0, self.diag.h

    def __repr__(self):
        return f"Diagonal(scalar={self.scalar})" if self.scalar is not None else f"Diagonal({self.diag!r})"


class LU:
    """`P = lu(A)` of the reference's CPU() path (src/inversion.jl:55-58, src/evolution.jl:150-153: UMFPACK through
    SparseArrays) - here SuperLU through scipy, the sparse direct solver this image has.  iterative_solve! then takes the
    `P <: Factorization` branch: x = P \\ y (src/iterative_solvers.jl:42-47)."""

    def __init__(self, A: DeviceCSR):
        import scipy.sparse.linalg as spla
        self.n = A.shape[0]
        self.factor = spla.splu(A.to_scipy_csc())

    def solve(self, y):
        return self.factor.solve(y)

    def __repr__(self):
        return f"LU({self.n}x{self.n}, SuperLU)"


class _Workspace:
    def __init__(self):
        self.stats = None

    def history(self):
        raise NotImplementedError


class GmresWorkspace(_Workspace):
    """Krylov.GmresWorkspace(n, n, VT; memory) at src/inversion.jl:84"""

    def __init__(self, ctx, n, memory=20):
        super().__init__()
        h = C.c_void_p()
        L.check(L.lib().npg_gmres_create(ctx.h, int(n), int(memory), C.byref(h)))
        self.h, self.ctx, self.n, self.memory = h, ctx, n, memory
        self.x = DeviceVector(ctx, n)          # workspace.x .= 0 (src/inversion.jl:85)

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_gmres_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def solve(self, A: DeviceCSR, y: DeviceVector, x: DeviceVector, P, atol=1e-6, rtol=1e-6, itmax=0,
              reorth_eta=0.1, **_ignored):
        """reorth_eta: take the second Gram-Schmidt pass when ||w - V h|| < eta ||w||.  Measured on the bowl meshes, GMRES
        iteration counts are identical for eta in {0, 0.05, 0.3, 0.707}: the DGKS value 0.707 keeps the basis orthogonal to
        machine precision, which restarted GMRES(20) at rtol 1e-6 does not need; 0.1 keeps the safety net."""
        kind, s, dh = (L.NPG_PRECOND_NONE, 0.0, None) if P is None else P.kind()
        st = L.SolveStats()
        L.check(L.lib().npg_gmres_solve(self.h, A.h, kind, s, dh, y.h, x.h, float(atol), float(rtol), int(itmax),
                                        float(reorth_eta), C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def history(self):
        buf = np.empty(int(self.stats["niter"]) + 1 if self.stats else 1)
        k = L.lib().npg_gmres_history(self.h, L.ptr(buf), buf.size)
        return buf[:max(k, 0)]

    def set_split(self, mode):
        """-1: by size (default), 0: fused Arnoldi kernel, 1: split kernels + column-major basis (npg_gmres_set_split)"""
        L.check(L.lib().npg_gmres_set_split(self.h, int(mode)))

    def set_basis(self, bits):
        """stored Krylov basis of the split organisation: 64, 32 (compressed basis), 0 = by tolerance (npg_gmres_set_basis)"""
        L.check(L.lib().npg_gmres_set_basis(self.h, int(bits)))

    def set_gather(self, mode):
        """Arnoldi kernel's SpMV input from the fp32 gather-layout copy of the Krylov vector (npg_gmres_set_gather):
        -1 = default (on where it applies: one GPU, fp32-stored basis, node-blocked matrix), 0 = off, 1 = on where it applies,
        2 = on, without the matrix's windowed tile set (csrc/spmv_window.h)"""
        L.check(L.lib().npg_gmres_set_gather(self.h, int(mode)))

    def set_profile(self, on=True):
        """eager launches with HIP events around every Arnoldi (SpMV) kernel; see npg_gmres_set_profile"""
        L.check(L.lib().npg_gmres_set_profile(self.h, int(bool(on))))

    def get_profile(self):
        """(total milliseconds, launches) of the Arnoldi kernel since set_profile(True)"""
        ms, n = C.c_double(), C.c_int64()
        L.check(L.lib().npg_gmres_get_profile(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class MgsGmresWorkspace(_Workspace):
    """Krylov.jl's gmres! in ITS OWN order of operations on device vectors: left-preconditioned restarted GMRES(memory) with
    MODIFIED Gram-Schmidt - one dot product and one axpy per basis column, Givens rotations and the triangular solve on the host -
    statement for statement what the reference runs through Krylov.gmres!(workspace, A, y, x; M = P, ldiv = false, restart = true,
    atol, rtol, itmax, history) (src/iterative_solvers.jl:56-58; Krylov.jl v0.10's gmres.jl, restated in oracle/krylov_oracle.py).
    Every dot is a host round trip: this is the solver for parity studies - on the reference's bowl system its residual history
    is the oracle's to rounding over the first restart cycles and its iteration count the oracle's to 0.3 % (5 182 against 5 167:
    two roundings of one recurrence; tests/test_gpu_parity.py) -, not for production: GmresWorkspace (classical Gram-Schmidt with a
    selective second pass, device-resident restart cycles) is 50-100 times faster per iteration and takes 3 % fewer iterations
    there."""

    def __init__(self, ctx, n, memory=20):
        super().__init__()
        self.ctx, self.n, self.memory = ctx, n, memory
        self.x = DeviceVector(ctx, n)
        self.V = [DeviceVector(ctx, n) for _ in range(memory)]
        self.w, self.q, self.dx = DeviceVector(ctx, n), DeviceVector(ctx, n), DeviceVector(ctx, n)
        self._hist = np.zeros(0)

    @staticmethod
    def _sym_givens(a, b):
        """Krylov.jl's sym_givens (krylov_utils.jl): (c, s, rho) with [c s; s -c] [a; b] = [rho; 0]"""
        if b == 0.0:
            return (1.0 if a >= 0.0 else -1.0) if a != 0.0 else 1.0, 0.0, abs(a)
        if a == 0.0:
            return 0.0, (1.0 if b >= 0.0 else -1.0), abs(b)
        if abs(b) > abs(a):
            t = a / b
            s = (1.0 if b >= 0.0 else -1.0) / np.sqrt(1.0 + t * t)
            return s * t, s, b / s
        t = b / a
        c = (1.0 if a >= 0.0 else -1.0) / np.sqrt(1.0 + t * t)
        return c, c * t, a / c

    def solve(self, A: DeviceCSR, y: DeviceVector, x: DeviceVector, P, atol=1e-6, rtol=1e-6, itmax=0, **_ignored):
        n, mem = self.n, self.memory
        if P is not None and not isinstance(P, Diagonal):
            raise TypeError("MgsGmresWorkspace: P must be a Diagonal (or None)")

        def apply_M(src, dst):
            if P is None:
                dst.copy_from(src)
            elif P.scalar is not None:
                dst.axpby(float(P.scalar), src, 0.0)
            else:
                dst.mul(P.diag, src)

        w, q, dx, V = self.w, self.q, self.dx, self.V
        # x = dx0 is the warm start (workspace.x, src/iterative_solvers.jl:56): w = b - A dx0
        w.copy_from(y)
        A.mul(x, w, alpha=-1.0, beta=1.0)
        apply_M(w, q)                               # r0
        beta = q.norm()
        rnorm = rnorm0 = beta
        hist = [beta]
        eps_ = atol + rtol * rnorm
        stats = dict(solved=1, niter=0, npass=0, status=1, nreorth=0, nflagged=0, rnorm0=rnorm0, rnorm=rnorm, seconds=0.0)
        if beta == 0.0:
            self._hist, self.stats = np.asarray(hist), stats
            return stats
        if itmax == 0:
            itmax = 2 * n
        inner_itmax = itmax
        btol = np.finfo(float).eps ** 0.75
        it = npass = 0
        solved, tired, breakdown = rnorm <= eps_, it >= itmax, False
        while not (solved or tired or breakdown):
            c, s, z = np.zeros(mem), np.zeros(mem), np.zeros(mem)
            R = np.zeros(mem * (mem + 1) // 2)
            if npass >= 1:
                w.copy_from(y)
                A.mul(x, w, alpha=-1.0, beta=1.0)
                apply_M(w, q)
            beta = q.norm()
            z[0] = beta
            V[0].axpby(1.0 / beta, q, 0.0)
            npass += 1
            inner = nr = 0
            inner_tired = False
            while not (solved or inner_tired or breakdown):
                inner += 1
                A.mul(V[inner - 1], w)
                apply_M(w, q)
                for i in range(inner):
                    R[nr + i] = V[i].dot(q)
                    q.axpby(-R[nr + i], V[i], 1.0)
                hbis = q.norm()
                for i in range(inner - 1):
                    tmp = c[i] * R[nr + i] + s[i] * R[nr + i + 1]
                    R[nr + i + 1] = s[i] * R[nr + i] - c[i] * R[nr + i + 1]
                    R[nr + i] = tmp
                c[inner - 1], s[inner - 1], R[nr + inner - 1] = self._sym_givens(R[nr + inner - 1], hbis)
                zeta = s[inner - 1] * z[inner - 1]
                z[inner - 1] = c[inner - 1] * z[inner - 1]
                rnorm = abs(zeta)
                hist.append(rnorm)
                nr += inner
                solved = (rnorm <= eps_) or (rnorm + 1.0 <= 1.0)
                breakdown = hbis <= btol
                inner_tired = inner >= min(mem, inner_itmax)
                if not (solved or inner_tired or breakdown):
                    V[inner].axpby(1.0 / hbis, q, 0.0)
                    z[inner] = zeta
            yv = z.copy()
            for i in range(inner - 1, -1, -1):
                pos = nr + i - inner
                for j in range(inner - 1, i, -1):
                    yv[i] -= R[pos] * yv[j]
                    pos -= j
                yv[i] = 0.0 if abs(R[pos]) <= btol else yv[i] / R[pos]
            dx.fill(0.0)
            for i in range(inner):
                dx.axpby(yv[i], V[i], 1.0)
            x.axpby(1.0, dx, 1.0)
            inner_itmax -= inner
            it += inner
            tired = it >= itmax
        stats.update(solved=int(bool(solved)), niter=it, npass=npass, status=1 if solved else (2 if tired else 3), rnorm=rnorm)
        self._hist, self.stats = np.asarray(hist), stats
        return stats

    def history(self):
        return self._hist


class CgWorkspace(_Workspace):
    """Krylov.CgWorkspace(n, n, VT) at src/evolution.jl:120"""

    def __init__(self, ctx, n):
        super().__init__()
        h = C.c_void_p()
        L.check(L.lib().npg_cg_create(ctx.h, int(n), C.byref(h)))
        self.h, self.ctx, self.n = h, ctx, n
        self.x = DeviceVector(ctx, n)

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_cg_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def solve(self, A: DeviceCSR, y: DeviceVector, x: DeviceVector, P, atol=1e-6, rtol=1e-6, itmax=0, **_ignored):
        kind, s, dh = (L.NPG_PRECOND_NONE, 0.0, None) if P is None else P.kind()
        st = L.SolveStats()
        L.check(L.lib().npg_cg_solve(self.h, A.h, kind, s, dh, y.h, x.h, float(atol), float(rtol), int(itmax),
                                     C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def history(self):
        buf = np.empty(int(self.stats["niter"]) + 1 if self.stats else 1)
        k = L.lib().npg_cg_history(self.h, L.ptr(buf), buf.size)
        return buf[:max(k, 0)]


class IterativeSolverToolkit:
    """src/iterative_solvers.jl:1-9,26-29: {A, P, x, y, workspace, kwargs, label}; x just points to workspace.x."""

    def __init__(self, A, P, y, workspace, kwargs, label):
        self.A, self.P, self.y, self.workspace = A, P, y, workspace
        self.x = workspace.x
        self.kwargs, self.label = dict(kwargs), label

    def __repr__(self):
        return (f"IterativeSolverToolkit:\n├── A: {self.A!r}\n├── P: {self.P!r}\n├── x: {self.x!r}\n├── y: {self.y!r}\n"
                f"├── workspace: {type(self.workspace).__name__}\n├── kwargs: {self.kwargs}\n└── label: \"{self.label}\"")


def iterative_solve(solver: IterativeSolverToolkit):
    """iterative_solve!(solver) - src/iterative_solvers.jl:31-68, all three branches: a factorisation as P -> x = P \\ y (:42-47,
    CPU() with fixed coefficients); CPU() and fewer than 300 000 rows -> x = A \\ y (:49-55); otherwise the Krylov solve (:58) -
    on GPU() the device-resident solvers of libnupgcm_hip.so, on CPU() Krylov.jl's methods restated in libnupgcm_host.so."""
    if hasattr(solver, "solve"):            # distributed.DistributedSolverToolkit: row-block solve + all-gather
        solver.solve()
        return solver
    if not isinstance(solver.A, DeviceCSR):
        raise TypeError("iterative_solve: A must be a DeviceCSR (GPU(): in HBM; CPU(): a handle of the host library)")
    on_host = solver.A.ctx.device < 0
    if isinstance(solver.P, LU):
        import time
        t0 = time.perf_counter()
        solver.x.upload(solver.P.solve(solver.y.to_host()))
        solver.workspace.stats = dict(solved=1, niter=0, npass=0, status=1, nreorth=0, nflagged=0, rnorm0=0.0, rnorm=0.0,
                                      seconds=time.perf_counter() - t0, direct=True)
        return solver
    if on_host and solver.A.shape[0] < 300_000:
        import time

        import scipy.sparse.linalg as spla
        t0 = time.perf_counter()
        solver.x.upload(spla.spsolve(solver.A.to_scipy_csc(), solver.y.to_host()))
        solver.workspace.stats = dict(solved=1, niter=0, npass=0, status=1, nreorth=0, nflagged=0, rnorm0=0.0, rnorm=0.0,
                                      seconds=time.perf_counter() - t0, direct=True)
        return solver
    solver.workspace.solve(solver.A, solver.y, solver.x, solver.P, **solver.kwargs)
    return solver
