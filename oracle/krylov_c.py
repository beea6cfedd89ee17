"""ORACLE (test infrastructure) - ctypes wrapper of oracle/krylov_c.c (OpenMP restatement of the host Krylov branch), built
by oracle/Makefile into oracle/_build/liborc_krylov.so.  Used by bench.py's cpu_baseline leg and by tests/ only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liborc_krylov.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_LIB)
        P = C.c_void_p
        _lib.orc_gmres.restype = C.c_int64
        _lib.orc_gmres.argtypes = [C.c_int64, P, P, P, P, P, P, C.c_int, C.c_double, C.c_double, C.c_int64, P,
                                   C.POINTER(C.c_int)]
        _lib.orc_cg.restype = C.c_int64
        _lib.orc_cg.argtypes = [C.c_int64, P, P, P, P, P, P, C.c_double, C.c_double, C.c_int64, C.POINTER(C.c_int)]
    return _lib


def usable_cores():
    """cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands one GPU's job a
    share of the host - 16 cores - while the mask still shows every core)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def set_threads(n):
    lib().orc_set_threads(int(n))
    return int(n)


def _csr(A):
    A = sp.csr_matrix(A)
    return (np.ascontiguousarray(A.indptr, dtype=np.int64), np.ascontiguousarray(A.indices, dtype=np.int32),
            np.ascontiguousarray(A.data, dtype=np.float64))


def _diag(M, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(1.0 if M is None else M, dtype=np.float64), (n,)))


def gmres(A, b, x0=None, M=None, memory=20, atol=1e-6, rtol=1e-6, itmax=0):
    n = len(b)
    rp, ci, v = _csr(A)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    Md = _diag(M, n)
    hist = np.zeros((itmax if itmax else 2 * n) + 2)
    ok = C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    it = lib().orc_gmres(n, p(rp), p(ci), p(v), p(b), p(x), p(Md), int(memory), float(atol), float(rtol), int(itmax), p(hist),
                         C.byref(ok))
    return x, dict(solved=bool(ok.value), niter=int(it), residuals=hist[:it + 1])


def cg(A, b, x0=None, M=None, atol=1e-6, rtol=1e-6, itmax=0):
    n = len(b)
    rp, ci, v = _csr(A)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    Md = _diag(M, n)
    ok = C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    it = lib().orc_cg(n, p(rp), p(ci), p(v), p(b), p(x), p(Md), float(atol), float(rtol), int(itmax), C.byref(ok))
    return x, dict(solved=bool(ok.value), niter=int(it))
