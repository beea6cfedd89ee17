"""One rank of the multi-rank rehearsal (launched by tests/test_gpu_distributed.py through torch.distributed.run).

All ranks share GPU 0 (NPG_FORCE_DEVICE=0) and talk through the transport NPG_COMM_TRANSPORT names (comm.hip): `peer` - the
production peer-window kernels over hipIpc mappings of the same device - or `shm`, the host-driven loop-back.  RCCL refuses
two ranks on one device; everything else - partition, local blocks, halo plan, the distributed GMRES/CG kernels and their
collective call sequence - is the production path."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg                                     # noqa: E402
from nupgcm_amd import distributed, workloads                # noqa: E402


def main():
    out, nsteps = sys.argv[1], int(sys.argv[2])
    block_nodes = len(sys.argv) > 3 and sys.argv[3] == "blocks"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    arch = npg.GPU(int(os.environ.get("NPG_FORCE_DEVICE", 0)))
    ctx = arch.ctx
    mode = sys.argv[3] if len(sys.argv) > 3 else "csr"
    channel = mode in ("channel", "pchannel")
    partitioned = mode in ("part", "partcsr", "pchannel")          # mesh-partitioned model (nupgcm_amd.partition)
    if partitioned:
        from nupgcm_amd import partition
    if channel:
        from nupgcm_amd import channel_basin
        mm = channel_basin.channel_basin_model(0.0625, workloads.CB_ALPHA)
        m = (partition.channel_basin_model(arch, mm, dist, element_precision="fp64") if partitioned else
             distributed.channel_basin_model(arch, mm, dist, element_precision="fp64"))
    else:
        mm = workloads.bowl_mesh_model("bowl3D_h0.1")
        m = (partition.example_model(arch, mm, dist, block_nodes=mode == "part") if partitioned else
             distributed.example_model(arch, mm, dist, block_nodes=block_nodes))
    # distributed SpMV: owned rows of A x for a known global x
    s = m.inversion.solver
    part = m.partition
    owned = part.inv_owned(rank)
    N = part.nu + part.np
    xg = np.sin(0.37 * np.arange(N))
    x_loc = npg.DeviceVector(ctx, s.A.shape[1])
    x_loc.view(0, len(owned)).copy_from(npg.DeviceVector.from_host(ctx, xg[owned]))
    s.halo.exchange(x_loc)
    y_loc = s.A.mul(x_loc).to_host()
    ghosts = x_loc.to_host()[len(owned):]
    if partitioned:
        assert np.array_equal(ghosts, xg[m.layout.inv.g_sol])          # the ghosts arrived, in layout order
    if not channel:
        npg.invert(m)
    npg.run(m, n_steps=nsteps)
    ctx.sync()
    peers = np.asarray(s.halo._keep["peers"])
    extra = {}
    if partitioned:
        extra = dict(layout=np.array([m.layout.inv.n_own, len(m.layout.inv.g_sol), len(m.layout.inv.g_ext), m.layout.b.n_own,
                                      len(m.layout.b.g_sol), len(m.layout.b.g_ext), len(m.layout.cells),
                                      m.fe_data.mesh.ncell, m.comm_layout["matrix_bytes"]]))
    u_, p_, b_ = m.state.u, m.state.p, m.state.b          # (collective for the partitioned model)
    ws = getattr(s, "workspace", None)
    if ws is not None and getattr(ws, "stats", None):           # residual history and outcome of the last inversion (probes)
        extra.update(hist_gm=ws.history(), stat_gm=np.array([ws.stats[k] for k in ("status", "npass", "nreorth", "nflagged")]))
    if os.environ.get("NPG_WORKER_DUMP") == "1":                # pieces of the LAST step, for the repeatability probes
        se = m.evolution.solver
        extra.update(hist_cg=se.workspace.history(), rhs_b=se.y.to_host(), rhs_inv=s.y.to_host(),
                     A_evol=se.A.to_scipy_csr().data, x_evol=(se.x_loc if hasattr(se, "x_loc") else se.x).to_host())
    np.savez(f"{out}.rank{rank}.npz", transport=ctx.comm_info()["in_cycle_transport"], peers=peers, **extra, dt=m.timestepper.dt, storage=np.array(s.A.storage()), owned=owned, y_loc=y_loc, n_ghost=len(ghosts), u=u_, p=p_,
             b=b_, gm=[st[1]["niter"] for st in m.stats], cg=[st[0]["niter"] for st in m.stats],
             solved=[bool(st[1]["solved"]) and bool(st[0]["solved"]) for st in m.stats])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
