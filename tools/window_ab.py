#!/usr/bin/env python3
"""A/B of the windowed tile set (csrc/spmv_window.h) against the ordinary tiles on the assembled A_inversion of a bowl mesh:
the stand-alone gather-layout product (npg_spmv_gather32) on both tile sets, timed with the wall clock over `reps` back-to-back
launches, and checked against the plain-CSR product of the fp32-rounded vector.  Extra arguments KEY=VALUE,KEY=VALUE ... are
environment settings (NPG_SPMV_WLANES, NPG_WIN_BYTES ...) under which the matrix is blocked once more and timed again.
Usage: python tools/window_ab.py [workload] [reps] [ENV=V,ENV=V ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "bowl3D_h0.02"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
arch = npg.GPU(0)
fed = workloads.example_fe_data(workloads.bowl_mesh_model(wl))
prm, frc = workloads.example_parameters()
A0 = npg.build_A_inversion(arch, fed, prm, frc.nu)
N = A0.shape[0]
rng = np.random.default_rng(0)
xh = rng.standard_normal(N)
x = npg.DeviceVector.from_host(arch.ctx, xh)
want = A0.mul(npg.DeviceVector.from_host(arch.ctx, xh.astype(np.float32).astype(np.float64))).to_host()
y = npg.DeviceVector(arch.ctx, N)


def timed(A, win, by, name):
    A.mul_gather32(x, y, windowed=win, reps=3)
    err = np.linalg.norm(y.to_host() - want) / np.linalg.norm(want)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        A.mul_gather32(x, y, windowed=win, reps=reps)
        best = min(best, (time.perf_counter() - t0) / reps)
    print(f"{name:40s}: {best * 1e6:8.1f} us per product  ({by / best / 1e9:7.1f} GB/s of its {by / 1e6:.1f} MB)  relerr {err:.1e}", flush=True)


for k, cfg in enumerate([""] + sys.argv[3:]):
    env = dict(kv.split("=") for kv in cfg.split(",") if kv)
    os.environ.update(env)
    A = npg.build_A_inversion(arch, fed, prm, frc.nu)
    assert A.block_nodes(fed.dofs.n_full, fed.dofs.n_surf)
    info = A.window_info()
    if k == 0:
        print(f"{wl}: N={N} storage={A.storage()} 28-byte records={A.coupling_records()} ordinary bytes={A.stored_spmv_bytes() / 1e6:.1f} MB", flush=True)
        timed(A, False, A.stored_spmv_bytes(), "ordinary tiles")
    print(f"[{cfg or 'default'}] window set: {info}", flush=True)
    if info["tiles"]:
        timed(A, True, info["bytes"], f"windowed [{cfg or 'default'}]")
    for key in env:
        del os.environ[key]
    del A
