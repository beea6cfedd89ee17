"""CPU: the ILU(0) restatement (oracle/ilu0_oracle.py) against its defining properties - the reference holds no fixture for
this path (src/preconditioners.jl:101-107 hands it to KrylovPreconditioners.jl / cuSPARSE)."""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from oracle import ilu0_oracle as io


def _spd_banded(n, bw, seed):
    rng = np.random.default_rng(seed)
    A = np.zeros((n, n))
    for d in range(1, bw + 1):
        v = rng.uniform(-1.0, 1.0, n - d)
        A += np.diag(v, d) + np.diag(v, -d)
    A += np.diag(np.abs(A).sum(axis=1) + 1.0)
    return A


def test_ilu0_is_lu_on_a_pattern_closed_under_elimination():
    """a full band is closed under elimination: ILU(0) drops nothing and equals LU without pivoting"""
    A = _spd_banded(60, 3, 1)
    LU = io.ilu0(sp.csr_matrix(A))
    L, U = io.split(LU)
    P, Ld, Ud = sla.lu(A)
    assert np.allclose(P, np.eye(60))                       # diagonally dominant: no row exchanges
    assert np.abs(L.toarray() - Ld).max() < 1e-13 and np.abs(U.toarray() - Ud).max() < 1e-12
    r = np.sin(np.arange(60.0))
    assert np.abs(io.solve(LU, r) - np.linalg.solve(A, r)).max() < 1e-12


def test_ilu0_residual_vanishes_on_the_pattern():
    """general pattern: (L U - A)_ij = 0 wherever A holds an entry (Saad, Prop. 10.2); elsewhere the dropped fill shows"""
    rng = np.random.default_rng(3)
    n = 80
    M = sp.random(n, n, density=0.06, random_state=5, format="csr")
    M = M + M.T
    A = sp.csr_matrix(M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 1.0))
    LU = io.ilu0(A)
    L, U = io.split(LU)
    E = (L @ U - A).toarray()
    mask = A.toarray() != 0
    assert np.abs(E[mask]).max() < 1e-13
    assert np.abs(E[~mask]).max() > 1e-6                    # (the pattern is not closed: something was dropped)
    x, it, solved, hist = io.pcg(A, rng.standard_normal(n), LU, atol=1e-10, rtol=1e-10)
    assert solved and it < 30
    lo, up = io.levels(A)
    assert 1 < lo <= n and 1 < up <= n
