#!/usr/bin/env python3
"""bench.py - nuPGCM hot path on MI355X: timesteps/sec of the evolve! + invert! loop and the inversion SpMV's achieved
bandwidth on a bowl3D mesh (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload bowl3D_h0.02]

One process per GPU (torch.distributed.run launches N of them; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).
A step = one timestep of /root/reference/src/model.jl:128-209 (advection assembly + CG evolution solve + GMRES inversion)
with the parameters of examples/bowl_mixing.jl, state resident in HBM.  The timed region is bracketed by barrier +
device sync on both sides and the maximum over ranks is reported; rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     - the dominant kernel (the fused Arnoldi kernel: CSR SpMV of A_inversion + Gram-Schmidt dots), timed live
                 with HIP events on the library's stream in a profile pass right after the timed region
  cpu_baseline - the CPU oracle (a port of the reference recipe) timed on this host's cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured streaming copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("NPG_BENCH_WORKLOAD", "bowl3D_h0.02"))
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--reorth-eta", type=float, default=None, help="override the GMRES second-pass threshold")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true",
                    help="skip the HIP-event profile pass (use when the whole run is under rocprofv3)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU budget of the cpu_baseline sample")
    return ap.parse_args()


def pmc_traffic(workload, kernel="k_gmres_arnoldi"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r01_pmc.json, made by
    tools/pmc_probe.py + tools/pmc_summary.py on this workload; FETCH_SIZE / WRITE_SIZE in separate passes, gfx950
    correction applied as described in profiles/README.md).  None when no counters were collected for the workload."""
    p = os.path.join(ROOT, "profiles", "r01_pmc.json")
    if not os.path.exists(p):
        return None
    rec = json.load(open(p)).get(workload, {}).get(kernel)
    return None if rec is None else rec["traffic_bytes_per_launch"]


def cpu_baseline(workload, mesh_model, dt, its_per_step, budget, A_host=None, h=None):
    """The oracle on this host's cores, bounded sample.  Meshes small enough for a sparse LU run the reference's CPU()
    path proper (direct solves, src/iterative_solvers.jl:42-55); larger ones time the host Krylov branch
    (src/iterative_solvers.jl:58 via InversionToolkit(CPU(), A, Diagonal(1/h^3), B, b)) per GMRES iteration and scale by
    the iterations a step needs."""
    from threadpoolctl import threadpool_limits

    from oracle import krylov_oracle as ko
    from oracle import recipe as rc
    with threadpool_limits(limits=1):
        if A_host is not None and A_host.shape[0] > 40000:
            # large mesh: the oracle's dense-per-cell numpy assembly would need tens of GB; time the oracle's GMRES on the
            # same matrix (assembled by the product, downloaded) - the matrix is an input here, the solver is the port
            y = np.sin(np.arange(A_host.shape[0], dtype=float)) * 1e-3
            t0 = time.perf_counter()
            its = 0
            while time.perf_counter() - t0 < budget:
                _, st = ko.gmres(A_host, y, M=1 / h ** 3, itmax=20)
                its += st["niter"]
            per_it = (time.perf_counter() - t0) / its
            return dict(value=1.0 / (per_it * max(its_per_step, 1)), unit="timesteps/s", cores=1, kind="port",
                        sample=f"{its} GMRES(20) iterations of the oracle's host Krylov path (scipy CSR SpMV + numpy "
                               f"MGS, the reference's large-system branch src/iterative_solvers.jl:58) on the {workload} "
                               f"inversion matrix: {per_it * 1e3:.0f} ms/iteration, scaled to the {its_per_step:.0f} "
                               f"iterations a timestep took on the GPU; evolution solve and assembly not included")
        S = rc.setup("example", model=mesh_model, dt=dt)
        N = S.A.shape[0]
        if N <= 40000:
            timer = {}
            rc.run(S, 1, timer=timer)                    # includes the LU factorisations outside the timed loop
            per = max(timer["loop_seconds"], 1e-3)
            n = int(max(2, min(50, budget / per)))
            rc.run(S, n, timer=timer)
            return dict(value=n / timer["loop_seconds"], unit="timesteps/s", cores=1, kind="port",
                        sample=f"{n} timesteps of the oracle's direct-solve path (reference CPU() recipe: advection "
                               f"assembly + 2 sparse-LU solves per step, factorisation excluded) on {workload}")
        h, _ = S.orc.precond_h()
        y = S.B @ (0.01 * np.sin(np.arange(S.B.shape[1]))) + S.b0
        t0 = time.perf_counter()
        its = 0
        while time.perf_counter() - t0 < budget:
            _, st = ko.gmres(S.A, y, M=1 / h ** 3, itmax=20)
            its += st["niter"]
        per_it = (time.perf_counter() - t0) / its
        return dict(value=1.0 / (per_it * max(its_per_step, 1)), unit="timesteps/s", cores=1, kind="port",
                    sample=f"{its} GMRES(20) iterations of the oracle's host Krylov path on {workload} "
                           f"({per_it * 1e3:.1f} ms/iteration), scaled to the {its_per_step:.0f} iterations a timestep took "
                           f"on the GPU; evolution solve and assembly not included")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_dist = bool(os.environ.get("NPG_BENCH_FORCE_DIST")) and world == 1     # 1-rank run of the distributed code path
    if force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
    if world > 1:
        import torch
        import torch.distributed as dist
        dev = int(os.environ.get("NPG_FORCE_DEVICE", local))
        torch.cuda.set_device(dev)
        dist.init_process_group(os.environ.get("NPG_TORCH_BACKEND", "nccl"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")

    import nupgcm_amd as npg
    from nupgcm_amd import workloads

    arch = npg.GPU(int(os.environ.get("NPG_FORCE_DEVICE", local)))     # NPG_FORCE_DEVICE: rehearse N ranks on one GPU
    ctx = arch.ctx
    t_setup = time.time()
    mesh_model = workloads.bowl_mesh_model(a.workload)
    if world > 1 or force_dist:
        from nupgcm_amd import distributed
        model = distributed.example_model(arch, mesh_model, dist, dt=a.dt)
    else:
        kw = {} if a.reorth_eta is None else {"reorth_eta": a.reorth_eta}
        model = workloads.example_model(arch, mesh_model, dt=a.dt, **kw)
    d = model.fe_data.dofs
    npg.invert(model)                                     # examples/bowl_mixing.jl:194
    ctx.sync()
    t_setup = time.time() - t_setup

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    npg.run(model, n_steps=a.warmup)
    barrier()
    t0 = time.perf_counter()
    npg.run(model, n_steps=a.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cpu" if dist.get_backend() == "gloo" else f"cuda:{torch.cuda.current_device()}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = model.stats[-a.steps:]
    gm_its = [s[1]["niter"] for s in stats]
    cg_its = [s[0]["niter"] for s in stats]

    # ---- profile pass: HIP events around every Arnoldi (SpMV) kernel of one more timestep -------------------------
    A = model.inversion.solver.A
    N, nnz = A.shape[0], A.nnz
    ws = model.inversion.solver.workspace
    ms_total, launches = 0.0, 0
    if not a.no_profile_pass:
        ws.set_profile(True)
        npg.run(model, n_steps=1)
        ms_total, launches = ws.get_profile()
        ws.set_profile(False)
    alg_bytes = 12 * nnz + 4 * (N + 1) + 16 * N          # CSR fp64/int32 SpMV on the stored (numeric) pattern
    roofline = None
    if launches > 0:
        avg_ms = ms_total / launches
        ach = alg_bytes / (avg_ms * 1e-3) / 1e9
        roofline = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                        traffic=pmc_traffic(a.workload), kernel="k_gmres_arnoldi (Givens prologue + CSR-stream SpMV"
                        + (")" if N >= 8192 else " + fused Gram-Schmidt dots)"),
                        avg_launch_us=avg_ms * 1e3, launches=launches, algorithmic_bytes_per_launch=alg_bytes,
                        cache_resident=bool(alg_bytes < 256 * 2 ** 20))
        # what actually moved, next to the algorithmic figure: bytes as laid out in HBM (node-block records stream 20 B
        # where CSR streams 60) and, where a PMC measurement is on file, the measured traffic
        stored = A.stored_spmv_bytes() + 2 * 8 * N           # SpMV bytes (matrix, x, y) + wt re-read + new basis column
        roofline["stored_bytes_per_launch"] = int(stored)
        real = roofline["traffic"] or stored
        roofline["real_GBps"] = real / (avg_ms * 1e-3) / 1e9
        roofline["real_frac"] = roofline["real_GBps"] / HBM_PEAK_GBS
    # stand-alone SpMV kernel (same tiles, no Krylov epilogue) for reference
    x = npg.DeviceVector.from_host(ctx, np.sin(np.arange(A.shape[1], dtype=float)))
    y = npg.DeviceVector(ctx, N)
    for _ in range(3):
        A.mul(x, y)
    reps = 20
    ctx.timer_start()
    for _ in range(reps):
        A.mul(x, y)
    spmv_ms = ctx.timer_stop() / reps

    out = {
        "metric": "timesteps/sec (evolve!+invert! loop; inversion SpMV GB/s in 'roofline')",
        "value": a.steps / elapsed, "unit": "timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{a.workload}: example parameters of examples/bowl_mixing.jl (eps=0.2, alpha=1/2, "
                               f"mu_rho=1, N2=2, BDF2 dt={a.dt:g}), full evolve!+invert! timestep loop",
                   "tets": int(model.fe_data.mesh.ncell), "nu": int(d.nu), "np": int(d.np), "nb": int(d.nb),
                   "N_inversion": int(N), "nnz_A": int(nnz), "node_block_storage": bool(getattr(A, "paired", False)),
                   "full_nodes": int(d.n_full), "surface_nodes": int(d.n_surf), "gmres_iterations_per_step": gm_its,
                   "cg_iterations_per_step": cg_its, "gmres_second_gs_passes_per_step": [s[1]["nreorth"] for s in stats],
                   "gmres_memory": 20, "atol": 1e-6, "rtol": 1e-6,
                   "setup_seconds": round(t_setup, 1), "parallelism": f"row-partitioned x{world}" if world > 1 else "1 GPU"},
        "roofline": roofline,
        "spmv_standalone": {"avg_launch_us": spmv_ms * 1e3, "GBps": alg_bytes / (spmv_ms * 1e-3) / 1e9},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        big = N > 40000
        A_host = None
        if big:
            # the solver's A may be stored by node blocks; the oracle gets a plain CSR copy assembled afresh
            A_plain = A if not getattr(A, "paired", False) else npg.build_A_inversion(arch, model.fe_data, model.params,
                                                                                      model.forcings.nu)
            A_host = A_plain.to_scipy_csr()
            del A_plain
        out["cpu_baseline"] = cpu_baseline(a.workload, mesh_model, a.dt, float(np.mean(gm_its)), a.cpu_seconds,
                                           A_host=A_host,
                                           h=model.fe_data.mesh.median_edge_length() if big else None)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
