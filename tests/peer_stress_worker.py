"""Stress of the communication layer with N ranks on one GPU (launched by tests/test_gpu_distributed.py through
torch.distributed.run; NPG_COMM_TRANSPORT selects the transport).  Thousands of back-to-back halo exchanges and small
all-reduces WITHOUT host synchronisation in between - the pattern of a replayed Krylov cycle - under deliberately uneven
pacing of the ranks; every exchange and every sum is checked exactly (integer-valued doubles), on the device, by accumulating
the deviation from the expected values into an error vector that must end up identically zero."""
import os
import sys
import time

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg                                     # noqa: E402
from nupgcm_amd import _lib as L                             # noqa: E402
from nupgcm_amd import distributed                           # noqa: E402
from nupgcm_amd.architectures import comm_unique_id          # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    n_own = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    arch = npg.GPU(int(os.environ.get("NPG_FORCE_DEVICE", 0)))
    ctx = arch.ctx
    ids = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ctx.comm_init(ids[0], rank, world)
    rng = np.random.default_rng(7)                           # the same stream on every rank: a global description
    N = world * n_own
    owner = (np.arange(N) // n_own).astype(np.int32)
    owned = np.arange(rank * n_own, (rank + 1) * n_own)
    ghosts_by_rank = []
    for r in range(world):
        others = np.setdiff1d(np.arange(N), np.arange(r * n_own, (r + 1) * n_own))
        g = rng.choice(others, size=min(len(others), 7000 + 900 * r), replace=False)
        ghosts_by_rank.append(g[np.lexsort((g, owner[g]))])
    ghosts = ghosts_by_rank[rank]
    plan = distributed.halo_plan(rank, owned, owner, ghosts_by_rank)
    halo = distributed.Halo(ctx, n_own, len(ghosts), plan)
    base = (np.arange(N) % 1021 + 1).astype(float)           # integer-valued: every product and sum below is exact
    x = npg.DeviceVector(ctx, n_own + len(ghosts))
    x.fill(0.0)
    x_own, x_gh = x.view(0, n_own), x.view(n_own, len(ghosts))
    b_own = npg.DeviceVector.from_host(ctx, base[owned])
    b_gh = npg.DeviceVector.from_host(ctx, base[ghosts])
    err = npg.DeviceVector(ctx, len(ghosts))
    err.fill(0.0)
    v = npg.DeviceVector(ctx, 32)
    vbase = npg.DeviceVector.from_host(ctx, (np.arange(32) + 1.0) * (rank + 1))
    vsum = npg.DeviceVector.from_host(ctx, (np.arange(32) + 1.0) * world * (world + 1) / 2)
    verr = npg.DeviceVector(ctx, 32)
    verr.fill(0.0)
    dist.barrier()
    t0 = time.perf_counter()
    for k in range(1, reps + 1):
        c = float(1 + (k * 7919) % 1000)
        x_own.axpby(c, b_own, 0.0)                           # owned entries of "iteration" k
        halo.exchange(x)
        err.axpby(1.0, x_gh, 1.0)                            # err += ghosts - c * expected   (exactly zero each time)
        err.axpby(-c, b_gh, 1.0)
        if k % 3:                                            # mostly an all-reduce between exchanges (a Krylov step) ...
            v.axpby(c, vbase, 0.0)
            L.check(L.lib().npg_comm_allreduce_vec(ctx.h, v.h))
            verr.axpby(1.0, v, 1.0)
            verr.axpby(-c, vsum, 1.0)
        # ... sometimes two exchanges back to back (restart: x then wt) - the double-buffered window and its acknowledgements
        if (k + rank) % 97 == 0:
            ctx.sync()
            time.sleep(0.003 * (1 + rank))                   # uneven pacing: this rank falls behind, then catches up
    ctx.sync()
    dt = time.perf_counter() - t0
    em, _ = err.maxabs()
    vm, _ = verr.maxabs()
    bad = ctx.allreduce_sum([1.0 if (em != 0.0 or vm != 0.0) else 0.0])[0]
    print(f"rank {rank}: {reps} exchanges of {len(ghosts)} ghosts + {2 * reps // 3} all-reduces in {dt:.2f} s; "
          f"max halo error {em:g}, max all-reduce error {vm:g}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if bad:
        raise SystemExit(3)
    if rank == 0:
        print("STRESS OK", flush=True)


if __name__ == "__main__":
    main()
