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
    ap.add_argument("--preconditioner", default="diagonal", choices=["diagonal", "multigrid", "dense_inverse"],
                    help="inversion preconditioner: the reference's Diagonal(1/h^3) (default; the headline configuration) or "
                         "the multigrid V-cycle behind flexible GMRES (refined bowl meshes)")
    ap.add_argument("--reorth-eta", type=float, default=None, help="override the GMRES second-pass threshold")
    ap.add_argument("--extrapolate-guess", type=int, nargs="?", const=1, default=0,
                    help="start each inversion from 2 x_{n-1} - x_{n-2} instead of the reference's warm start x_{n-1} "
                         "(model.extrapolate_guess; NOT the reference's recipe: the run is labelled in `config`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-multigrid", action="store_true", help="skip the extra multigrid-preconditioned run")
    ap.add_argument("--no-profile-pass", action="store_true",
                    help="skip the HIP-event profile pass (use when the whole run is under rocprofv3)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU budget of the cpu_baseline sample")
    return ap.parse_args()


PMC_FILE = "profiles/r05_pmc.json"


def pmc_traffic(workload, kernel="k_gmres_arnoldi", stored=None):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes (a builder run, not this run:
    profiles/r05_pmc.json, made by `tools/prof.sh pmc` = tools/pmc_probe.py + tools/pmc_summary.py on this workload; FETCH_SIZE /
    WRITE_SIZE in separate passes, gfx950 correction applied as described in profiles/README.md) - the line says so in
    `traffic_source`.  The figure belongs to ONE layout of the matrix: when the record carries `stored_bytes_per_launch` and the
    layout that is running lays out a different number of bytes (> 0.5 %), the figure is refused (returns None with the reason)
    and the roofline is priced in `stored_bytes_per_launch` instead.  Returns (bytes or None, reason or None)."""
    p = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(p):
        return None, f"{PMC_FILE} is missing"
    rec = json.load(open(p)).get(workload, {}).get(kernel)
    if rec is None:
        return None, f"no counters on file for {kernel} on {workload}"
    on_file = rec.get("stored_bytes_per_launch")
    if stored is not None and on_file is not None and abs(on_file - stored) > 0.005 * stored:
        return None, (f"{PMC_FILE} was measured on a layout of {on_file} stored bytes per launch, the layout running now has "
                      f"{int(stored)}: the committed traffic figure does not describe this kernel")
    return rec["traffic_bytes_per_launch"], None


def cpu_baseline(workload, mesh_model, dt, gm_its, cg_its, budget, A_host=None, h=None, ncell=None):
    """The reference's CPU path on THIS host's cores, bounded sample (a reported baseline, not the target).

    Meshes small enough for a sparse LU run the reference's CPU() path proper: the oracle's direct-solve recipe (advection
    assembly + two sparse-LU back-substitutions per step, src/iterative_solvers.jl:42-55; numpy + SuperLU, effectively one
    core).  Larger ones cannot be factorised (fill-in: ten minutes and 7.5 GB already at 135 k unknowns), so what is timed
    is the host Krylov branch the reference takes there (src/iterative_solvers.jl:58 through
    InversionToolkit(CPU(), A, Diagonal(1/h^3), B, b)): oracle/krylov_c.c, an OpenMP restatement of Krylov.jl's GMRES(20) /
    CG on ALL cores, per iteration on the workload's own matrices, times the iterations a timestep needs (the GPU run's
    counts - same algorithm, same stopping rule), plus the advection assembly measured per cell on the base mesh."""
    import scipy.sparse as sp

    from oracle import krylov_c as kc
    from oracle import recipe as rc
    visible = cores = kc.usable_cores()             # affinity mask and cgroup quota
    if cores > 32 and not os.environ.get("NPG_CPU_CORES"):
        cores = 16          # no quota visible, but a one-GPU job's share of the box is 16 cores: do not oversubscribe it
                            # (`cores` in the JSON is the thread count actually used, `cores_visible` what the OS showed)
    cores = int(os.environ.get("NPG_CPU_CORES", cores))
    kc.set_threads(cores)
    if A_host is None:
        S = rc.setup("example", model=mesh_model, dt=dt)
        timer = {}
        rc.run(S, 1, timer=timer)                    # includes the LU factorisations outside the timed loop
        per = max(timer["loop_seconds"], 1e-3)
        n = int(max(2, min(50, budget / per)))
        rc.run(S, n, timer=timer)
        return dict(value=n / timer["loop_seconds"], unit="timesteps/s", cores=1, host_cores=cores, cores_visible=visible, kind="port", threads=1,
                    timesteps_sampled=int(n), seconds_per_timestep=timer["loop_seconds"] / n, extrapolated=False,
                    sample=f"{n} timesteps of the oracle's direct-solve path (reference CPU() recipe: advection "
                           f"assembly + 2 sparse-LU solves per step, factorisation excluded; numpy + SuperLU, one core) "
                           f"on {workload}")
    n = A_host.shape[0]
    y = np.sin(np.arange(n, dtype=float)) * 1e-3
    t0 = time.perf_counter()
    its = 0
    while time.perf_counter() - t0 < 0.6 * budget:
        _, st = kc.gmres(A_host, y, M=1 / h ** 3, itmax=40)
        its += st["niter"]
    per_gm = (time.perf_counter() - t0) / its
    # evolution CG and advection assembly on the base mesh of the hierarchy (cost per row / per cell carries over)
    S = rc.setup("example", model=mesh_model, dt=dt)
    Ab = (S.M + S.theta("BDF2") * (S.Kh + S.Kv)).tocsr()
    rhs = S.M @ np.ones(Ab.shape[0])
    t1 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t1 < 0.1 * budget:
        _, sc = kc.cg(Ab, rhs, M=1 / Ab.diagonal(), itmax=10)
        reps += sc["niter"]
    per_cg_row = (time.perf_counter() - t1) / reps / Ab.shape[0]
    b0 = np.sin(np.arange(S.orc.sp.nb, dtype=float))
    u0 = np.cos(np.arange(S.orc.sp.nu, dtype=float))
    t2 = time.perf_counter()
    S.orc.advection_rhs(b0, b0, u0, u0, dt, "BDF2")
    per_cell = (time.perf_counter() - t2) / len(S.orc.topo.cells)
    nb_big = int(round(Ab.shape[0] * ncell / len(S.orc.topo.cells)))
    step = per_gm * np.mean(gm_its) + per_cg_row * nb_big * np.mean(cg_its) + per_cell * ncell
    return dict(value=1.0 / step, unit="timesteps/s", cores=cores, cores_visible=visible, kind="port", threads=cores,
                gmres_ms_per_iteration=per_gm * 1e3, gmres_iterations_sampled=int(its), gmres_iterations_per_timestep=float(np.mean(gm_its)),
                cg_ms_per_iteration=per_cg_row * nb_big * 1e3, advection_seconds_per_timestep=per_cell * ncell, seconds_per_timestep=float(step),
                extrapolated=True,
                sample=f"host Krylov branch of the reference (src/iterative_solvers.jl:58) restated in C + OpenMP on {cores} "
                       f"cores: {its} GMRES(20) iterations on the {workload} inversion matrix = {per_gm * 1e3:.1f} ms/iteration "
                       f"x {np.mean(gm_its):.0f} iterations per timestep (the GPU run's count: same algorithm and stopping "
                       f"rule), + Jacobi-CG {per_cg_row * nb_big * 1e3:.2f} ms/iteration x {np.mean(cg_its):.0f} and the "
                       f"advection assembly {per_cell * ncell:.2f} s (numpy, per-cell cost measured on the base mesh)")


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
    channel = a.workload.startswith("channel_basin")
    if channel:
        # BASELINE.json configs[4]: "channel_basin_h<h>[_dirichlet]" - scratch/run.jl on the structured-to-tet periodic mesh
        from nupgcm_amd import channel_basin
        parts = a.workload.split("_")
        hh = float(parts[2][1:])
        mesh_model = channel_basin.channel_basin_model(hh, workloads.CB_ALPHA)
    else:
        mesh_model = workloads.bowl_mesh_model(a.workload)
    if channel:
        surf = "dirichlet" if a.workload.endswith("dirichlet") else "flux"
        if world > 1 or force_dist:
            from nupgcm_amd import partition                        # mesh, matrices and state partitioned over the ranks
            model = partition.channel_basin_model(arch, mesh_model, dist, surface=surf)
        elif a.preconditioner == "multigrid":
            # converged inversions instead of run.jl's 1000-iteration cap: V-cycle over a 3-level refinement hierarchy whose
            # finest mesh has the requested spacing
            model = workloads.channel_basin_model(arch, h=hh, levels=int(os.environ.get("NPG_CB_LEVELS", 2)), surface=surf, itmax=0)
        else:
            model = workloads.channel_basin_model(arch, mesh_model=mesh_model, surface=surf)
    elif world > 1 or force_dist:
        from nupgcm_amd import partition
        if a.preconditioner == "multigrid":     # finest level row-partitioned, coarser levels replicated
            model = partition.example_model(arch, a.workload, dist, dt=a.dt, preconditioner="multigrid")
        else:
            model = partition.example_model(arch, mesh_model, dist, dt=a.dt)
    else:
        kw = {} if a.reorth_eta is None else {"reorth_eta": a.reorth_eta}
        if a.preconditioner in ("multigrid", "dense_inverse"):
            model = workloads.example_model(arch, a.workload, dt=a.dt, preconditioner=a.preconditioner, **kw)
        else:
            model = workloads.example_model(arch, mesh_model, dt=a.dt, **kw)
    transport_check = None
    if hasattr(model, "verify_transport"):
        # end-to-end check of halo + all-reduce on this hardware before anything is timed; if the peer windows fail it, fall
        # back on RCCL for the in-cycle traffic (auto transport) and build the model again
        transport_check = "passed"
        if not model.verify_transport():
            was = ctx.comm_info()["in_cycle_transport"]
            if was != "peer" or os.environ.get("NPG_COMM_TRANSPORT") == "peer":
                # (NPG_COMM_TRANSPORT=peer creates no RCCL communicator: there is nothing to fall back on)
                raise SystemExit(f"bench: the {was} transport failed the halo / all-reduce check"
                                 + (" and NPG_COMM_TRANSPORT=peer leaves no RCCL communicator to fall back on" if was == "peer" else ""))
            if rank == 0:
                print("[bench] peer-window transport failed its end-to-end check: falling back on RCCL", file=sys.stderr)
            import gc
            del model
            gc.collect()
            ctx.disable_peer()
            from nupgcm_amd import partition
            model = (partition.channel_basin_model(arch, mesh_model, dist, surface=surf) if channel else
                     partition.example_model(arch, a.workload if a.preconditioner == "multigrid" else mesh_model, dist, dt=a.dt,
                                             preconditioner=a.preconditioner if a.preconditioner == "multigrid" else "diagonal"))
            if not model.verify_transport():
                raise SystemExit("bench: the RCCL transport failed the halo / all-reduce check too")
            transport_check = "peer failed, rccl passed"
    if a.extrapolate_guess:
        model.extrapolate_guess = a.extrapolate_guess            # 1: linear, 2: quadratic
    d = model.fe_data.dofs
    if not channel:
        npg.invert(model)                                 # examples/bowl_mixing.jl:194 (the channel model is built inverted)
    ctx.sync()
    t_setup = time.time() - t_setup

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            if dist.get_backend() == "nccl":
                import torch
                torch.cuda.synchronize()
            ctx.sync()

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
    N, nnz = A.shape[0], A.nnz                            # this rank's row block when distributed
    N_glob, nnz_glob = int(d.nu + d.np), int(nnz)
    if dist is not None and world > 1:
        nnz_glob = int(round(ctx.allreduce_sum([float(nnz)])[0]))
    ws = model.inversion.solver.workspace
    ms_total, launches = 0.0, 0
    if not a.no_profile_pass and hasattr(ws, "set_profile"):
        ws.set_profile(True)
        npg.run(model, n_steps=1)
        ms_total, launches = ws.get_profile()
        ws.set_profile(False)
    alg_bytes = 12 * nnz + 4 * (N + 1) + 16 * N          # CSR fp64/int32 SpMV on the stored (numeric) pattern
    roofline = None
    if launches > 0:
        avg_ms = ms_total / launches
        # bytes the kernel's layout says must move per launch: the matrix in the form the Arnoldi kernel streams it (the
        # windowed tile set where it has one, csrc/spmv_window.h), its input gathered once, + wt re-read + w and the new basis column
        winfo = A.window_info() if hasattr(A, "window_info") and getattr(A, "paired", False) else {"tiles": 0}
        windowed = bool(winfo["tiles"]) and os.environ.get("NPG_GMRES_WINDOW", "1") != "0"
        stored = (winfo["bytes"] if windowed else A.stored_spmv_bytes()) + 2 * 8 * N
        measured, refused = pmc_traffic(a.workload, stored=stored if windowed else None) if world == 1 else (None, "one GPU only")
        if not windowed and measured is not None:
            measured, refused = None, "the committed counters are for the windowed-tile instance; this run's Arnoldi kernel uses another"
        real = measured or stored
        real_gbps = real / (avg_ms * 1e-3) / 1e9
        csr_gbps = alg_bytes / (avg_ms * 1e-3) / 1e9
        # `achieved` / `frac`: bytes that really move (PMC-measured where a measurement of this round's kernel is on file, else
        # the stored bytes) over the live HIP-event launch time, against the 8 TB/s HBM spec.  The rate in the ALGORITHMIC bytes
        # of SURVEY 8(d) - what a plain-CSR SpMV of the same matrix would have had to stream in that time - is a separate key.
        roofline = dict(bound="hbm", achieved=real_gbps, peak=HBM_PEAK_GBS, unit="GB/s", frac=real_gbps / HBM_PEAK_GBS,
                        scope="one GPU" if world == 1 else f"rank 0's row block ({N} of {N_glob} rows), one of {world} GPUs",
                        # (the committed counters are for the whole matrix on one GPU: no figure for a rank's row block)
                        traffic=measured,
                        traffic_source=(f"{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a builder run of this "
                                        "round's kernel on this workload, not measured in this run)" if measured is not None else None),
                        bytes_priced="traffic (PMC)" if measured is not None else f"stored_bytes_per_launch ({refused})",
                        kernel="k_gmres_arnoldi (Givens prologue + record-stream SpMV"
                        + (", windowed tiles" if windowed else "") + (")" if N >= 8192 else " + fused Gram-Schmidt dots)"),
                        avg_launch_us=avg_ms * 1e3, launches=launches, stored_bytes_per_launch=int(stored),
                        algorithmic_bytes_per_launch=alg_bytes, csr_equivalent_GBps=csr_gbps,
                        csr_equivalent_frac=csr_gbps / HBM_PEAK_GBS,
                        cache_resident=bool(alg_bytes < 256 * 2 ** 20),
                        window=({k: winfo[k] for k in ("tiles", "block_tiles", "distinct")} if windowed else None))
        roofline["note"] = ("`achieved` / `frac` price the launch in the bytes that really move; the kernel streams the matrix "
                            "in record form, fewer bytes than the 12 B per stored entry of a CSR SpMV - `csr_equivalent_GBps` "
                            "is the rate in those algorithmic bytes of SURVEY 8(d) and may exceed the HBM peak")
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

    # the inversion SpMV AS THE SOLVER RUNS IT, stand-alone: the gather-layout product on the windowed tiles (npg_spmv_gather32 - the
    # tile code of the Arnoldi kernel without its Givens prologue and its epilogue's wt / w / basis-column traffic), priced in the
    # bytes of the windowed set as stored.  Wall clock over back-to-back launches (the entry point synchronises once at the end).
    spmv_g32 = None
    if rank == 0 and world == 1 and getattr(A, "paired", False) and hasattr(A, "window_info"):
        wi = A.window_info()
        if wi.get("tiles"):
            A.mul_gather32(x, y, windowed=True, reps=3)
            t0g = time.perf_counter()
            A.mul_gather32(x, y, windowed=True, reps=200)
            g_ms = 1e3 * (time.perf_counter() - t0g) / 200
            spmv_g32 = {"avg_launch_us": g_ms * 1e3, "stored_bytes_per_launch": int(wi["bytes"]), "GBps": wi["bytes"] / (g_ms * 1e-3) / 1e9,
                        "frac_of_8TBps": wi["bytes"] / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "launches": 200,
                        "kernel": "k_spmv_g32 (npg_spmv_gather32): fp32 gather-layout input, fp64 products and sums, windowed tiles",
                        "note": "the SpMV of the Arnoldi kernel's instance alone; wall clock / 200 back-to-back launches including one "
                                "allocation and one fill of the gather-layout copy (~0.1 us per launch)"}

    # the same product on a plain-CSR copy of the matrix (what the reference's cuSPARSE path streams): real bytes = algorithmic
    # bytes there, so this is the kernel's bandwidth figure without the node-block storage's byte savings
    spmv_plain = None
    if getattr(A, "paired", False) and rank == 0 and world == 1:
        Ap = npg.build_A_inversion(arch, model.fe_data, model.params, model.forcings.nu)
        for _ in range(3):
            Ap.mul(x, y)
        ctx.timer_start()
        for _ in range(reps):
            Ap.mul(x, y)
        pms = ctx.timer_stop() / reps
        spmv_plain = {"avg_launch_us": pms * 1e3, "GBps": alg_bytes / (pms * 1e-3) / 1e9,
                      "frac_of_8TBps": alg_bytes / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": "stand-alone k_spmv on the plain-CSR layout: every algorithmic byte really moves; slower in time "
                              "than the node-block layout above, which moves 0.53x the bytes"}
        del Ap

    out = {
        "metric": "timesteps/sec (evolve!+invert! loop; inversion SpMV GB/s in 'roofline')",
        "value": a.steps / elapsed, "unit": "timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("f64" if channel or a.preconditioner != "diagonal" or N < 8192 else
                  "f64 arithmetic; fp32-STORED Krylov basis and fp32 gather-layout copy of the SpMV input (the library's default at "
                  "rtol >= 1e-7; products, sums, x += V y and the restart residual are fp64 - the all-fp64 figure is `fp64_basis`)"),
        "data": "synthetic",
        "config": {"workload": (f"{a.workload}: production parameters of scratch/run.jl (alpha=1/8, f=y, P1 buoyancy, nu(x) -> "
                                f"full-stress form, convection + eddy closures, wind, BDF1 with the adaptive CFL step, "
                                f"GMRES itmax=1000) on the x-periodic structured-to-tet channel-basin mesh; element kernels "
                                f"{model.evolution.fe.precision}-local / fp64-accumulate, fp64 solves" if channel else
                                f"{a.workload}: example parameters of examples/bowl_mixing.jl (eps=0.2, alpha=1/2, "
                                f"mu_rho=1, N2=2, BDF2 dt={a.dt:g}), full evolve!+invert! timestep loop"),
                   "tets": int(model.fe_data.mesh.ncell), "nu": int(d.nu), "np": int(d.np), "nb": int(d.nb),
                   "N_inversion": N_glob, "nnz_A": nnz_glob, "node_block_storage": bool(getattr(A, "paired", False)),
                   "full_nodes": int(d.n_full), "surface_nodes": int(d.n_surf), "gmres_iterations_per_step": gm_its,
                   "cg_iterations_per_step": cg_its, "gmres_second_gs_passes_per_step": [s[1]["nreorth"] for s in stats],
                   # scaled residual each inversion STARTS from (warm start = previous solution): its growth over the first
                   # steps is what drives the iteration count up as the flow spins up
                   "gmres_initial_residual_per_step": [float(f"{s[1]['rnorm0']:.4g}") for s in stats],
                   "inversion_seconds_per_step": [round(s[1]["seconds"], 4) for s in stats],
                   "gmres_memory": 20, "atol": 1e-6, "rtol": 1e-6, "gmres_itmax": model.inversion.solver.kwargs["itmax"],
                   "initial_guess": {0: "x_{n-1} (the reference's warm start)",
                                     1: "2 x_{n-1} - x_{n-2} (extrapolated: not the reference's recipe)",
                                     2: "3 x_{n-1} - 3 x_{n-2} + x_{n-3} (extrapolated: not the reference's recipe)"}[
                                         int(getattr(model, "extrapolate_guess", 0) or 0)],
                   "all_solved": all(s[1]["solved"] == 1 for s in stats), "preconditioner": repr(model.inversion.solver.P),
                   "setup_seconds": round(t_setup, 1),
                   "parallelism": ("1 GPU" if world == 1 else
                                   f"mesh, matrices and state partitioned x{world} (node-aligned, one ghost-cell layer)")},
        "roofline": roofline,
        # stand-alone k_spmv (ordinary tiles, fp64 gathers): priced in the bytes its layout moves; the CSR-equivalent rate (the
        # algorithmic bytes of SURVEY 8(d) over the same time) is labelled as such and may exceed the HBM peak
        "spmv_standalone": {"avg_launch_us": spmv_ms * 1e3, "stored_bytes_per_launch": int(A.stored_spmv_bytes() + 8 * A.shape[1] + 8 * N),
                            "GBps": (A.stored_spmv_bytes() + 8 * A.shape[1] + 8 * N) / (spmv_ms * 1e-3) / 1e9,
                            "frac_of_8TBps": (A.stored_spmv_bytes() + 8 * A.shape[1] + 8 * N) / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "csr_equivalent_GBps": alg_bytes / (spmv_ms * 1e-3) / 1e9,
                            "note": "GBps = bytes of the stored layout (records) per launch time; csr_equivalent_GBps = what a plain-CSR "
                                    "SpMV of the same matrix would have had to stream in that time (not a physical rate)"},
        "spmv_gather32_standalone": spmv_g32,
        "spmv_plain_csr": spmv_plain,
    }
    # ---- the same loop with the multigrid-preconditioned inversion (new work; the headline above stays the reference's
    # configuration): refined bowl meshes on one GPU
    if (rank == 0 and world == 1 and not channel and a.preconditioner == "diagonal" and not a.no_multigrid
            and workloads.BOWL_MESHES[a.workload][1] >= 1):
        t_mg = time.time()
        mg = workloads.example_model(arch, a.workload, dt=a.dt, preconditioner="multigrid", fine_fe_data=model.fe_data)
        npg.invert(mg)
        ctx.sync()
        t_mg = time.time() - t_mg
        npg.run(mg, n_steps=max(a.warmup, 3))            # the extrapolated initial guess needs two steps of history to settle
        ctx.sync()
        k = max(a.steps, 10)
        t0 = time.perf_counter()
        npg.run(mg, n_steps=k)
        ctx.sync()
        el = time.perf_counter() - t0
        st = mg.stats[-k:]
        out["multigrid"] = {
            "what": "the same timestep loop with the inversion preconditioned by a geometric multigrid V-cycle behind "
                    "flexible GMRES(20), same stopping rule (csrc/mg.hip)",
            "value": k / el, "unit": "timesteps/s", "steps": k, "ms_per_step": 1e3 * el / k,
            "speedup_vs_headline": (k / el) / out["value"],
            "fgmres_iterations_per_step": [x[1]["niter"] for x in st], "all_solved": all(x[1]["solved"] == 1 for x in st),
            "inversion_ms_per_step": [round(1e3 * x[1]["seconds"], 2) for x in st],
            "preconditioner": repr(mg.inversion.solver.P), "setup_seconds": round(t_mg, 1)}
        # roofline of the path itself: ONE application of the V-cycle (what every outer iteration pays), timed live on the library's
        # stream over 20 back-to-back applications, against the bytes its operators lay out (npg_precond_cycle_bytes: every product of
        # the cycle with the matrix in the form its kernel reads + its vectors; counted by the library when the cycle was captured)
        Pmg = mg.inversion.solver.P
        if hasattr(Pmg, "cycle_bytes") and Pmg.cycle_bytes() > 0:
            rr = npg.DeviceVector.from_host(ctx, np.sin(np.arange(N_glob, dtype=float)) * 1e-3)
            zz = npg.DeviceVector(ctx, N_glob)
            for _ in range(3):
                Pmg.apply(rr, zz)
            ctx.timer_start()
            for _ in range(20):
                Pmg.apply(rr, zz)
            cyc_ms = ctx.timer_stop() / 20
            cb = Pmg.cycle_bytes()
            its_mg = [x[1]["niter"] for x in st]
            out["multigrid"]["roofline"] = {
                "bound": "hbm on the finest level, launch latency on the coarse levels", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": cb / (cyc_ms * 1e-3) / 1e9, "frac": cb / (cyc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "cycle_ms": cyc_ms, "stored_bytes_per_cycle": int(cb), "traffic": None,
                "kernel": "one V(2,2) cycle = every launch of csrc/mg.hip's mg_cycle; dominant kernels: the finest level's products with A "
                          "(k_spmv_g32e on the windowed tiles for residuals, k_spmv in the smoother) and the coarsest level's dense fp16 "
                          "inverse (k_dense_gemv_part8h) - positions and times: profiles/r04_multigrid_cycle.txt",
                "ms_per_outer_iteration": 1e3 * sum(x[1]["seconds"] for x in st) / max(1, sum(its_mg)),
                "note": "bytes as laid out (not PMC-measured); an outer FGMRES iteration = this cycle + one product with A + two "
                        "Gram-Schmidt passes"}
        del mg
    # ---- the reference's solver configuration with ONE change outside the solver: each inversion starts from the extrapolation
    # 2 x_{n-1} - x_{n-2} of the last two solutions instead of x_{n-1} (model.extrapolate_guess; the answer moves within the
    # solver tolerance only).  Same steps as the headline, on a fresh model.
    if (rank == 0 and world == 1 and not channel and a.preconditioner == "diagonal" and not a.no_multigrid
            and not a.extrapolate_guess):
        t_x = time.time()
        ex = workloads.example_model(arch, mesh_model, dt=a.dt, fine_fe_data=model.fe_data)
        ex.extrapolate_guess = True
        npg.invert(ex)
        ctx.sync()
        t_x = time.time() - t_x
        npg.run(ex, n_steps=a.warmup)
        ctx.sync()
        t0 = time.perf_counter()
        npg.run(ex, n_steps=a.steps)
        ctx.sync()
        el = time.perf_counter() - t0
        st = ex.stats[-a.steps:]
        out["extrapolated_guess"] = {
            "what": "the headline configuration (GMRES(20), Diagonal(1/h^3), same stopping rule, same steps) with every "
                    "inversion started from 2 x_{n-1} - x_{n-2} instead of the reference's warm start x_{n-1}",
            "value": a.steps / el, "unit": "timesteps/s", "steps": a.steps, "ms_per_step": 1e3 * el / a.steps,
            "speedup_vs_headline": (a.steps / el) / out["value"],
            "gmres_iterations_per_step": [x[1]["niter"] for x in st], "all_solved": all(x[1]["solved"] == 1 for x in st),
            "gmres_initial_residual_per_step": [float(f"{x[1]['rnorm0']:.4g}") for x in st], "setup_seconds": round(t_x, 1)}
        del ex
    # ---- the like-for-like precision: the headline loop with the Krylov basis and the SpMV input stored in fp64 throughout, as
    # Krylov.jl's CuVector{Float64} workspaces are (src/inversion.jl:84)
    if rank == 0 and world == 1 and not channel and a.preconditioner == "diagonal" and not a.no_multigrid and N >= 8192:
        t_x = time.time()
        f64 = workloads.example_model(arch, mesh_model, dt=a.dt, fine_fe_data=model.fe_data)
        if a.extrapolate_guess:
            f64.extrapolate_guess = a.extrapolate_guess
        w64 = f64.inversion.solver.workspace
        w64.set_basis(64)
        w64.set_gather(0)
        npg.invert(f64)
        ctx.sync()
        t_x = time.time() - t_x
        k = max(1, min(a.steps, 5))
        npg.run(f64, n_steps=a.warmup)
        ctx.sync()
        t0 = time.perf_counter()
        npg.run(f64, n_steps=k)
        ctx.sync()
        el = time.perf_counter() - t0
        st = f64.stats[-k:]
        w64.set_profile(True)
        npg.run(f64, n_steps=1)
        ms64, l64 = w64.get_profile()
        w64.set_profile(False)
        out["fp64_basis"] = {
            "what": "the headline loop with the Krylov basis stored in fp64 and the SpMV input gathered from the fp64 vector "
                    "(npg_gmres_set_basis(64), npg_gmres_set_gather(0)): every stored number fp64, as in the reference",
            "value": k / el, "unit": "timesteps/s", "steps": k, "warmup": a.warmup, "ms_per_step": 1e3 * el / k,
            "gmres_iterations_per_step": [x[1]["niter"] for x in st], "all_solved": all(x[1]["solved"] == 1 for x in st),
            "arnoldi_avg_launch_us": (1e3 * ms64 / l64) if l64 else None, "setup_seconds": round(t_x, 1)}
        if l64:
            # the all-fp64 instance runs on the ORDINARY tiles with fp64 gathers: its own PMC passes (profiles/r05_pmc.json)
            A64 = f64.inversion.solver.A
            st64 = A64.stored_spmv_bytes() + 8 * A64.shape[1] + 3 * 8 * N        # matrix + wt gathered once + wt row, w, fp64 basis column
            tr64, why64 = pmc_traffic(a.workload, "k_gmres_arnoldi_fp64")
            b64 = tr64 or st64
            out["fp64_basis"]["roofline"] = {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": b64 / (ms64 / l64 * 1e-3) / 1e9,
                "frac": b64 / (ms64 / l64 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr64, "stored_bytes_per_launch": int(st64),
                "bytes_priced": "traffic (PMC)" if tr64 is not None else f"stored_bytes_per_launch ({why64})",
                "avg_launch_us": 1e3 * ms64 / l64, "launches": l64,
                "kernel": "k_gmres_arnoldi<8, false, 0, false, 0, true> (fp64 basis, fp64 gathers, ordinary record tiles)",
                "csr_equivalent_GBps": alg_bytes / (ms64 / l64 * 1e-3) / 1e9}
        del f64
    # ---- small meshes (the reference's own): the explicit inverse in HBM instead of latency-bound Krylov iterations
    if rank == 0 and world == 1 and not channel and a.preconditioner == "diagonal" and not a.no_multigrid and N <= 40000:
        t_d = time.time()
        dm = workloads.example_model(arch, mesh_model, dt=a.dt, preconditioner="dense_inverse")
        npg.invert(dm)
        ctx.sync()
        t_d = time.time() - t_d
        npg.run(dm, n_steps=max(a.warmup, 1))
        ctx.sync()
        k = max(a.steps, 20)
        t0 = time.perf_counter()
        npg.run(dm, n_steps=k)
        ctx.sync()
        el = time.perf_counter() - t0
        st = dm.stats[-k:]
        out["dense_inverse"] = {
            "what": "the same timestep loop with P = A^-1 held explicitly in HBM (n^2 doubles; rocSOLVER getrf + getri at "
                    "set-up, hand-written GEMV per application) behind flexible GMRES, same stopping rule",
            "value": k / el, "unit": "timesteps/s", "steps": k, "ms_per_step": 1e3 * el / k,
            "speedup_vs_headline": (k / el) / out["value"], "fgmres_iterations_per_step": [x[1]["niter"] for x in st][:8],
            "all_solved": all(x[1]["solved"] == 1 for x in st), "preconditioner": repr(dm.inversion.solver.P),
            "setup_seconds": round(t_d, 1)}
        del dm
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not channel:
        big = N > 40000
        A_host = None
        base_model = mesh_model
        if big:
            # the solver's A may be stored by node blocks; the oracle gets a plain CSR copy assembled afresh
            A_plain = A if not getattr(A, "paired", False) else npg.build_A_inversion(arch, model.fe_data, model.params,
                                                                                      model.forcings.nu)
            A_host = A_plain.to_scipy_csr()
            del A_plain
            base_model = workloads.bowl_hierarchy_models(a.workload)[0]
        out["cpu_baseline"] = cpu_baseline(a.workload, base_model, a.dt, gm_its, cg_its, a.cpu_seconds, A_host=A_host,
                                           h=model.fe_data.mesh.median_edge_length() if big else None,
                                           ncell=int(model.fe_data.mesh.ncell))
    if dist is not None:
        # evidence of the communicator the timed region used: ncclCommCount / rank / device as RCCL reports them, the
        # transport of the in-cycle halo and all-reduce, and every rank's share of the system
        import resource
        mine = dict(ctx.comm_info(), **getattr(model, "comm_layout", {}),
                    host_maxrss_mb=round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1),
                    setup_seconds=round(t_setup, 1))
        infos = [None] * world
        dist.all_gather_object(infos, mine)
        out["comm"] = {"rccl_ranks": infos[0].get("rccl_ranks"), "in_cycle_transport": infos[0].get("in_cycle_transport"),
                       "transport_check": transport_check,
                       "cycle_replayed_from_hipgraph": bool(infos[0].get("in_cycle_transport") == "peer"
                                                           and os.environ.get("NPG_DIST_GRAPH", "1") != "0"),
                       "ranks": infos}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
