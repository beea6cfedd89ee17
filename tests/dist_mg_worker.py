"""One rank of the distributed-multigrid rehearsal (launched by tests/test_gpu_distributed.py through torch.distributed.run; all
ranks share GPU 0, NPG_COMM_TRANSPORT selects the transport).  bowl3D h = 0.05 (134 866 unknowns) with its two-level
hierarchy: the finest level row-partitioned on the partitioned mesh, the coarse level (h = 0.1, dense inverse) replicated."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg                                     # noqa: E402
from nupgcm_amd import partition, workloads                  # noqa: E402


def main():
    out, nsteps, label = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    arch = npg.GPU(int(os.environ.get("NPG_FORCE_DEVICE", 0)))
    if label.startswith("channel_basin"):
        # BASELINE configs[4] with converged inversions: x-periodic channel basin, both closures (the eddy closure re-assembles A
        # in the full-stress form at step 10 and the preconditioner follows), two-level hierarchy, finest level partitioned
        hh = float(label.split("_h")[1])
        nlev = int(sys.argv[4]) if len(sys.argv) > 4 else 1              # distributed multigrid levels = refinements of the hierarchy
        models = workloads.channel_basin_hierarchy_models(hh, nlev)
        hier = [workloads.channel_basin_fe_data(mm) for mm in models]
        # (itmax: a preconditioner gone wrong fails the test's `solved` check instead of iterating to 2 N)
        m = partition.channel_basin_model(arch, models[-1], dist, element_precision="fp64", itmax=0 if nlev == 1 else 2000, fe_data=hier[-1],
                                          invert_now=False)
        zl = os.environ.get("NPG_TEST_SMOOTHER", "node") == "zline"          # the z-line blocks, cut at the rank boundaries
        partition.use_multigrid(m, hier, omega=1.7 if zl else 2.0, distributed_levels=nlev, smoother="zline" if zl else "node")
        assert m.verify_transport()
    else:
        nlev = int(sys.argv[4]) if len(sys.argv) > 4 else 1              # distributed multigrid levels (1 or 2)
        hier = [workloads.example_fe_data(m) for m in workloads.bowl_hierarchy_models(label)]
        prm, frc = workloads.example_parameters()
        m = partition.partitioned_model(arch, hier[-1], prm, frc, npg.BDF2(t_start=0.0, t_stop=1e9, dt=1e-3), dist)
        partition.use_multigrid(m, hier, distributed_levels=nlev)
        assert m.verify_transport()
    npg.invert(m)
    npg.run(m, n_steps=nsteps)
    arch.ctx.sync()
    u, p, b = m.state.u, m.state.p, m.state.b
    np.savez(f"{out}.rank{rank}.npz", u=u, p=p, b=b, its=[s[1]["niter"] for s in m.stats],
             solved=[bool(s[1]["solved"]) and bool(s[0]["solved"]) for s in m.stats], rn=[s[1]["rnorm"] for s in m.stats],
             mg=np.array([m.inversion.solver.P.layout_mg[k] for k in ("ghost_u", "ghost_p", "S_nnz")]),
             mg2=np.array([getattr(m.inversion.solver.P, "layout_mg2", {}).get(k, 0) for k in ("rows", "ghost_x", "ghost_P", "ghost_R")]),
             precond=repr(m.inversion.solver.P),
             devplan=all(lv.st.get("dev") is not None for lv in m.inversion.solver.P._lv))      # refreshes stay on the device
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
