#!/usr/bin/env python3
"""Where a CSR-stream tile spends its time: cycle stamps inside the product SpMV tile loop (tools/tune/spmv_variants.hip,
npg_spmv_phase_cycles), averaged per tile, for 1..3 workgroups per CU, with and without the x gathers.
Usage: python tools/spmv_phases.py [workload] [--plain]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import _lib as L  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402

plain = "--plain" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = args[0] if args else "bowl3D_h0.02"
arch = npg.GPU(0)
fed = workloads.example_fe_data(workloads.bowl_mesh_model(wl))
prm, frc = workloads.example_parameters()
A = npg.build_A_inversion(arch, fed, prm, frc.nu)
if not plain:
    assert A.block_nodes(fed.dofs.n_full, fed.dofs.n_surf)
N = A.shape[0]
x = npg.DeviceVector.from_host(arch.ctx, np.sin(np.arange(N, dtype=float)))
y = npg.DeviceVector(arch.ctx, N)
L.lib()      # the product library first: the harness links against it
_tune = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune", "libnupgcm_tune.so"))   # tuning harness (make -C tools/tune)
fn = _tune.npg_spmv_phase_cycles
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
fn.restype = C.c_int
print(f"{wl}: N={N} storage={A.storage()} stored bytes={A.stored_spmv_bytes() / 1e6:.0f} MB; cycles per tile (s_memtime, 100 MHz ticks x?)")
print("wg/CU gather |   desc  stream+gather+prod  barrier1  seg.sums  barrier2+out |  total   tiles")
for bpc in (1, 2, 3):
    for gather in (1, 0):
        out = (C.c_ulonglong * 7)()
        fn(A.h, x.h, y.h, bpc, gather, out)   # warm-up
        rc = fn(A.h, x.h, y.h, bpc, gather, out)
        assert rc == 0, L.lib().npg_last_error().decode()
        nt = out[6]
        v = [out[i] / nt for i in range(5)]
        print(f"  {bpc}     {gather}    | {v[0]:7.0f} {v[1]:12.0f} {v[2]:14.0f} {v[3]:9.0f} {v[4]:10.0f}      | {sum(v):7.0f} {nt:7d}")
