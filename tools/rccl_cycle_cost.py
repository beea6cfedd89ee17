#!/usr/bin/env python3
"""Per-iteration cost of the distributed GMRES cycle's communication steps on ONE GPU: a one-rank communicator in self-test mode
(NPG_COMM_SELFTEST=1, see tests/rccl_selftest_worker.py) against the serial cycle on the same matrix.
Usage: NPG_COMM_SELFTEST=1 [NPG_COMM_TRANSPORT=peer] python tools/rccl_cycle_cost.py [n]
RCCL (default): ncclAllReduce + grouped ncclSend/ncclRecv to self.  peer: the peer-window kernels of comm.hip with the rank as
its own neighbour (push / flag / unpack / acknowledge, one-kernel fold + all-reduce)."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg                                          # noqa: E402
from nupgcm_amd import _lib as L, distributed                     # noqa: E402
from nupgcm_amd.architectures import comm_unique_id               # noqa: E402

n = min(int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 400000)      # a latency probe: one rank's share of a large system at most
arch = npg.GPU(0)
ctx = arch.ctx
ctx.comm_init(comm_unique_id(), 0, 1)
rng = np.random.default_rng(5)
S = np.arange(n - (256 if n <= 50000 else 30000), n)[::-1].copy()       # ghost columns: a slab interface at the larger sizes
plan = dict(peers=np.array([0], np.int32), send_ptr=np.array([0, len(S)], np.int64), send_idx=S.astype(np.int32),
            recv_ptr=np.array([0, len(S)], np.int64))
halo = distributed.Halo(ctx, n, len(S), plan)
# slowly converging: a shifted 1-D Laplacian-like band + random couplings
# (banded like an RCM-ordered FE row block, ~40 entries per row at the larger sizes so that the SpMV has a rank's weight)
offs = [-1, 0, 1] if n <= 50000 else list(range(-20, 21))
M = sp.diags([(2.0005 if o == 0 else -1.0 / max(1, abs(o)) ** 2) * np.ones(n - abs(o)) for o in offs], offs, shape=(n, n), format="csr")
if n <= 50000:
    M = M + 1e-3 * sp.random(n, n, density=4.0 / n, random_state=1)
M = sp.csr_matrix(M)
Mc = M.tocsc()
keep = np.ones(n)
keep[S] = 0.0
own = sp.csr_matrix(M @ sp.diags(keep))
own.eliminate_zeros()
A_loc = sp.hstack([own, Mc[:, S].tocsr()], format="csr")
A_loc.sort_indices()
A_ser, A_dis = npg.DeviceCSR.from_scipy(ctx, M), npg.DeviceCSR.from_scipy(ctx, A_loc)
y = npg.DeviceVector.from_host(ctx, rng.standard_normal(n))
P = npg.Diagonal(diag=npg.DeviceVector.from_host(ctx, 1.0 / M.diagonal()))
tr = "peer windows" if os.environ.get("NPG_COMM_TRANSPORT") == "peer" else "RCCL"
for label, A, nx, dist_, graph in (("serial (hipGraph cycles)", A_ser, n, False, 0),
                                   (f"through {tr}, one-rank communicator, eager launches", A_dis, n + len(S), True, 0),
                                   (f"through {tr}, one-rank communicator, hipGraph replay", A_dis, n + len(S), True, 1)):
    ws = npg.GmresWorkspace(ctx, n, memory=20)
    if os.environ.get("CYCLE_BASIS"):                     # 32: the fp32-stored basis the reference's tolerance selects (the probe's
        ws.set_basis(int(os.environ["CYCLE_BASIS"]))      # rtol = 1e-14 would pick fp64)
    if dist_:
        L.check(L.lib().npg_gmres_set_halo(ws.h, halo.h))
        L.check(L.lib().npg_gmres_set_dist_options(ws.h, 1, graph))
    for rep in range(2):
        x = npg.DeviceVector(ctx, nx)
        x.fill(0.0)
        ctx.sync()
        t0 = time.perf_counter()
        st = ws.solve(A, y, x, P, atol=1e-30, rtol=1e-14, itmax=2000)
        ctx.sync()
        dt = time.perf_counter() - t0
    print(f"{label:56s}: {st['niter']} iterations, {dt / st['niter'] * 1e6:7.1f} us per iteration")
