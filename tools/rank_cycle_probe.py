#!/usr/bin/env python3
"""What ONE rank of an N-GPU run executes per GMRES inner iteration, measured on one GPU.

Builds rank R's share of the mesh-partitioned inversion system of a bowl workload exactly as partition.partitioned_model does
(NodePartition -> RankLayout -> local element engine -> the rank's rows of A over [owned | ghost] columns, node-blocked with its
windowed tile set, interior / boundary tiles) and runs the DISTRIBUTED restart cycle on it through a one-rank self-test
communicator (NPG_COMM_SELFTEST=1): the halo plan has the rank's real ghost count and a send list of the rank's real send
volume, but its only neighbour is the rank itself, and the all-reduce has nobody to wait for.  The ghost values are therefore
wrong and the iteration does not converge - what is measured is the cost of `its` iterations of the production kernels on
production-shaped data (same tiles, same launches, same bytes), without the wire.

    NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer python tools/rank_cycle_probe.py [workload] [nranks] [rank] [its]

Prints one line per mode (eager / hipGraph replay) and, for comparison, the serial cycle on a mesh of the rank's size when
`--serial-workload W` is given.  Under rocprofv3 --kernel-trace the launches are eager by construction (gmres.hip)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg                                          # noqa: E402
from nupgcm_amd import _lib as L, partition, workloads            # noqa: E402
from nupgcm_amd.architectures import DeviceCSR, DeviceVector, comm_unique_id      # noqa: E402
from nupgcm_amd.assembly import DeviceFE                          # noqa: E402
from nupgcm_amd.distributed import Halo, halo_plan                # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = args[0] if len(args) > 0 else "bowl3D_h0.02"
nranks = int(args[1]) if len(args) > 1 else 8
rank = int(args[2]) if len(args) > 2 else nranks // 2
its = int(args[3]) if len(args) > 3 else 2000
assert os.environ.get("NPG_COMM_SELFTEST") == "1", "run with NPG_COMM_SELFTEST=1 (one-rank communicator, the rank as its own neighbour)"

arch = npg.GPU(0)
ctx = arch.ctx
ctx.comm_init(comm_unique_id(), 0, 1)
t0 = time.time()
fed = workloads.example_fe_data(workloads.bowl_mesh_model(wl))
prm, frc = workloads.example_parameters()
part = partition.NodePartition(fed, nranks)
lay = partition.RankLayout(fed, part, rank)
fe = DeviceFE(ctx, partition.LocalFEData(fed, lay))
fe.set_coeff("nu", frc.nu)
fe.set_coeff("f", prm.f)
rp, ci, shp = lay.local_pattern(fed.pattern_A(structural=False), lay.inv, lay.inv)
A = DeviceCSR.from_pattern(ctx, shp[0], shp[1], rp, ci)
fe.assemble(L.NPG_MAT_A, A, scale=prm.alpha ** 2 * prm.eps ** 2, full_stress=False)
if os.environ.get("NPG_GHOST_NODES", "1") != "0":
    A.set_ghost_nodes(*lay.ghost_nodes(fed))
A.block_nodes(*part.local_nodes(rank))
n_own, n_gh = lay.inv.n_own, len(lay.inv.g_sol)
# the rank's REAL plan (whom it would talk to, how much it would send) - then folded onto itself
ghosts = [partition.RankLayout(fed, part, q).inv.g_sol if abs(q - rank) <= 2 else np.zeros(0, np.int64) for q in range(nranks)]
ghosts[rank] = lay.inv.g_sol
real = halo_plan(rank, lay.inv.owned, lay.owner_inv, ghosts)
n_send = int(real["send_ptr"][-1])
sidx = np.resize(real["send_idx"], n_gh).astype(np.int32) if n_send else np.arange(n_gh, dtype=np.int32) % n_own
plan = dict(peers=np.array([0], np.int32), send_ptr=np.array([0, n_gh], np.int64), send_idx=sidx, recv_ptr=np.array([0, n_gh], np.int64))
halo = Halo(ctx, n_own, n_gh, plan)
wi = A.window_info() if hasattr(A, "window_info") else {}
print(f"{wl}: rank {rank} of {nranks}: {n_own} owned rows, {n_gh} ghost columns, real plan: peers {real['peers'].tolist()} "
      f"sends {n_send} receives {n_gh}; matrix {A.stored_spmv_bytes() / 1e6:.1f} MB as stored, window tiles {wi.get('tiles')}; "
      f"set-up {time.time() - t0:.1f} s", flush=True)
h = fed.mesh.median_edge_length()
rng = np.random.default_rng(7)
y = DeviceVector.from_host(ctx, 1e-3 * rng.standard_normal(n_own))
P = npg.Diagonal(scalar=1.0 / h ** 3, n=n_own)
modes = [("eager launches", 0), ("hipGraph replay", 1)]
for label, graph in modes:
    ws = npg.GmresWorkspace(ctx, n_own, memory=20)
    L.check(L.lib().npg_gmres_set_halo(ws.h, halo.h))
    L.check(L.lib().npg_gmres_set_dist_options(ws.h, -1, graph))
    for rep in range(2):
        x = DeviceVector(ctx, n_own + n_gh)
        x.fill(0.0)
        ctx.sync()
        t1 = time.perf_counter()
        st = ws.solve(A, y, x, P, atol=1e-30, rtol=1e-7, itmax=its)       # (rtol 1e-7: the fp32-stored basis, as at the reference's 1e-6)
        ctx.sync()
        dt = time.perf_counter() - t1
    print(f"distributed cycle, one-rank self-test communicator, {label:16s}: {st['niter']} iterations, "
          f"{dt / max(1, st['niter']) * 1e6:7.2f} us per iteration", flush=True)
