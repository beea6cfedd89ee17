"""RCCL call sites - and, with NPG_COMM_TRANSPORT=peer, the peer-window kernels - on ONE GPU (launched by
tests/test_gpu_rccl_selftest.py with NPG_COMM_SELFTEST=1).

RCCL refuses two ranks on one device, so the multi-rank rehearsals use the shared-memory loop-back transport and the
product transport's call sites never ran on a development box.  A ONE-rank RCCL communicator does run them: with
NPG_COMM_SELFTEST=1 csrc/comm.hip skips its single-rank shortcuts and allows a halo plan whose peer is the rank itself,
so ncclCommInitRank, ncclAllReduce, grouped ncclSend/ncclRecv, ncclBroadcast and the distributed GMRES / CG cycles built
on them execute on hardware and are checked against the serial solvers.  With NPG_COMM_TRANSPORT=peer the same sequence runs
on the peer-window transport of comm.hip (the rank maps its own window as its neighbour's): push / flag / unpack /
acknowledge kernels, the one-kernel fold + all-reduce, and the distributed cycle replayed from ONE hipGraph by default."""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg                                          # noqa: E402
from nupgcm_amd import _lib as L                                  # noqa: E402
from nupgcm_amd import distributed                                # noqa: E402
from nupgcm_amd.architectures import comm_unique_id               # noqa: E402


def main():
    assert os.environ.get("NPG_COMM_SELFTEST") == "1" and os.environ.get("NPG_COMM_TRANSPORT", "") != "shm"
    transport = "peer" if os.environ.get("NPG_COMM_TRANSPORT", "") == "peer" else "rccl"
    arch = npg.GPU(0)
    ctx = arch.ctx
    ctx.comm_init(comm_unique_id(), 0, 1)                        # ncclGetUniqueId + ncclCommInitRank
    info = ctx.comm_info()
    assert info["in_cycle_transport"] == transport and info["rccl_ranks"] == (1 if transport == "rccl" else 0), info
    rng = np.random.default_rng(5)

    # ncclAllReduce (one rank: the sum is the input)
    v = rng.standard_normal(31)
    assert np.array_equal(ctx.allreduce_sum(v), v)

    # grouped ncclSend / ncclRecv: the rank is its own neighbour
    n, S = 3000, np.arange(2936, 3000)[::-1].copy()
    plan = dict(peers=np.array([0], np.int32), send_ptr=np.array([0, len(S)], np.int64), send_idx=S.astype(np.int32),
                recv_ptr=np.array([0, len(S)], np.int64))
    halo = distributed.Halo(ctx, n, len(S), plan)
    xh = np.concatenate([rng.standard_normal(n), np.zeros(len(S))])
    x = npg.DeviceVector.from_host(ctx, xh)
    halo.exchange(x)
    ctx.sync()
    assert np.array_equal(x.to_host()[n:], xh[S])

    # ncclBroadcast per segment (the all-gather of the owned slices)
    loc = npg.DeviceVector.from_host(ctx, rng.standard_normal(500))
    full = npg.DeviceVector(ctx, 800)
    full.fill(0.0)
    distributed.allgather_segments(ctx, loc, [(0, 0, 100, 200), (0, 200, 400, 300)], full)
    ctx.sync()
    f, lh = full.to_host(), loc.to_host()
    assert np.array_equal(f[100:300], lh[:200]) and np.array_equal(f[400:700], lh[200:500]) and not f[:100].any()

    # distributed GMRES: the columns in S are served from the ghost segment, which the halo exchange fills with x[S]
    # banded (like an RCM-ordered row block): only the rows next to the block's end read ghost columns
    offs = [-37, -2, -1, 1, 2, 37]
    M = sp.diags([rng.uniform(-1.0, 1.0, n - abs(o)) for o in offs], offs, shape=(n, n), format="csr") \
        + sp.diags(np.linspace(4.0, 9.0, n))
    M = sp.csr_matrix(M)
    Mc = M.tocsc()
    keep = np.ones(n)
    keep[S] = 0.0
    own = sp.csr_matrix(M @ sp.diags(keep))
    own.eliminate_zeros()
    A_loc = sp.hstack([own, Mc[:, S].tocsr()], format="csr")
    A_loc.sort_indices()
    A_ser = npg.DeviceCSR.from_scipy(ctx, M)
    A_dis = npg.DeviceCSR.from_scipy(ctx, A_loc)
    rhs = rng.standard_normal(n)
    y = npg.DeviceVector.from_host(ctx, rhs)
    P = npg.Diagonal(diag=npg.DeviceVector.from_host(ctx, 1.0 / M.diagonal()))
    ws0 = npg.GmresWorkspace(ctx, n, memory=20)
    x0 = npg.DeviceVector(ctx, n)
    x0.fill(0.0)
    st0 = ws0.solve(A_ser, y, x0, P, atol=1e-12, rtol=1e-10, itmax=400, reorth_eta=0.0)
    ws1 = npg.GmresWorkspace(ctx, n, memory=20)
    L.check(L.lib().npg_gmres_set_halo(ws1.h, halo.h))
    x1 = npg.DeviceVector(ctx, n + len(S))
    x1.fill(0.0)
    st1 = ws1.solve(A_dis, y, x1, P, atol=1e-12, rtol=1e-10, itmax=400)
    ctx.sync()
    xs, xd = x0.to_host(), x1.to_host()[:n]
    assert st0["solved"] and st1["solved"], (st0, st1)
    res = np.linalg.norm(M @ xd - rhs) / np.linalg.norm(rhs)
    assert res < 1e-8, res
    assert np.linalg.norm(xd - xs) <= 1e-8 * np.linalg.norm(xs)
    assert abs(st1["niter"] - st0["niter"]) <= 2, (st0["niter"], st1["niter"])
    # the split cycle (what systems of >= 8192 rows per rank use): tiles without ghost columns run beside the exchange,
    # which is enqueued on the plan's own stream; same iterates as with the exchange first
    sols = []
    for ov, graph in ((1, 0), (0, 0), (1, 1), (0, 1)):
        ws2 = npg.GmresWorkspace(ctx, n, memory=20)
        ws2.set_split(1)
        L.check(L.lib().npg_gmres_set_halo(ws2.h, halo.h))
        L.check(L.lib().npg_gmres_set_dist_options(ws2.h, ov, graph))
        x3 = npg.DeviceVector(ctx, n + len(S))
        x3.fill(0.0)
        st3 = ws2.solve(A_dis, y, x3, P, atol=1e-12, rtol=1e-10, itmax=400)
        ctx.sync()
        assert st3["solved"], st3
        sols.append((st3["niter"], x3.to_host()[:n]))
    for it, xv in sols[1:]:       # exchange first / hipGraph replay of the cycle with its RCCL calls: the same iterates
        assert it == sols[0][0] and np.array_equal(xv, sols[0][1]), (it, sols[0][0])
    assert np.linalg.norm(sols[0][1] - xs) <= 1e-8 * np.linalg.norm(xs)

    # distributed CG on an SPD matrix through the same halo
    Sy = sp.csr_matrix(M + M.T + sp.diags(np.full(n, 20.0)))
    Sc = Sy.tocsc()
    own = sp.csr_matrix(Sy @ sp.diags(keep))
    own.eliminate_zeros()
    B_loc = sp.hstack([own, Sc[:, S].tocsr()], format="csr")
    B_loc.sort_indices()
    B_dis = npg.DeviceCSR.from_scipy(ctx, B_loc)
    Pc = npg.Diagonal(diag=npg.DeviceVector.from_host(ctx, 1.0 / Sy.diagonal()))
    cg = npg.CgWorkspace(ctx, n)
    L.check(L.lib().npg_cg_set_halo(cg.h, halo.h))
    x2 = npg.DeviceVector(ctx, n + len(S))
    x2.fill(0.0)
    st2 = cg.solve(B_dis, y, x2, Pc, atol=1e-12, rtol=1e-10, itmax=400)
    ctx.sync()
    xc = x2.to_host()[:n]
    assert st2["solved"], st2
    assert np.linalg.norm(Sy @ xc - rhs) <= 1e-8 * np.linalg.norm(rhs)
    # the cancellation fallback with the cycle replayed from a hipGraph (the captured cycle has the Pythagorean norm baked
    # in: the solver must re-capture, not replay it for ever): A = I + small, as in test_gmres_fast_mode_falls_back...
    nn = 600
    Ai = sp.csr_matrix(sp.eye(nn) + 1e-3 * sp.random(nn, nn, density=0.05, random_state=np.random.default_rng(42), format="csr"))
    bi = np.random.default_rng(43).standard_normal(nn)
    halo0 = distributed.Halo(ctx, nn, 0, dict(peers=np.zeros(0, np.int32), send_ptr=np.zeros(1, np.int64),
                                              send_idx=np.zeros(0, np.int32), recv_ptr=np.zeros(1, np.int64)))
    ws5 = npg.GmresWorkspace(ctx, nn, memory=30)
    L.check(L.lib().npg_gmres_set_halo(ws5.h, halo0.h))
    L.check(L.lib().npg_gmres_set_dist_options(ws5.h, 1, 1))
    st5 = ws5.solve(npg.DeviceCSR.from_scipy(ctx, Ai), npg.DeviceVector.from_host(ctx, bi), ws5.x, None, atol=0.0, rtol=1e-10,
                    itmax=300)
    import scipy.sparse.linalg as spla
    xi = spla.spsolve(Ai.tocsc(), bi)
    assert st5["solved"] == 1 and st5["nflagged"] > 0, st5
    assert np.linalg.norm(ws5.x.to_host() - xi) <= 1e-8 * np.linalg.norm(xi)
    print(f"{transport.upper()} self-test OK: gmres {st0['niter']} / {st1['niter']} iterations (serial / through {transport}), "
          f"cg {st2['niter']}; cancellation fallback under graph replay {st5['niter']} iterations")


if __name__ == "__main__":
    main()
