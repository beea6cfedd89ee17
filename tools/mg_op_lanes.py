"""Lanes per row of the multigrid cycle's operators (npg_csr_set_lanes): every operator of the finest level (and the transfers to
the level below) timed stand-alone at 4 / 8 / 16 / 32 lanes.   python3 tools/mg_op_lanes.py [workload=bowl3D_h0.02]"""
import sys

import numpy as np

import nupgcm_amd as npg
from nupgcm_amd import workloads

wl = sys.argv[1] if len(sys.argv) > 1 else "bowl3D_h0.02"
arch = npg.GPU()
ctx = arch.ctx
m = workloads.example_model(arch, wl, dt=1e-3, preconditioner="multigrid")
P = m.inversion.solver.P
top = len(P.levels) - 1
ops = P.ops[top]
Pd, Rd = [k for k in P._keep if k is not None][-2:]
for name, M in (("G", ops.G), ("D", ops.D), ("Dinv", ops.Dinv), ("S", ops.S), ("P", Pd), ("R", Rd)):
    x = npg.DeviceVector.from_host(ctx, np.sin(np.arange(M.shape[1], dtype=float)))
    y = npg.DeviceVector(ctx, M.shape[0])
    res = []
    for lanes in (0, 4, 8, 16, 32):
        M.set_lanes(lanes)
        for _ in range(3):
            M.mul(x, y)
        ctx.timer_start()
        for _ in range(20):
            M.mul(x, y)
        res.append(f"{lanes or 'default'}: {1e3 * ctx.timer_stop() / 20:.1f}")
    M.set_lanes(0)
    print(f"{name:5s} {M.shape[0]:8d} x {M.shape[1]:8d}  nnz {M.nnz:9d} ({M.nnz / M.shape[0]:.1f} per row)   us  " + "  ".join(res), flush=True)
