#!/usr/bin/env python3
"""Times the stand-alone CSR-stream SpMV variants of tools/tune/spmv_variants.hip on the assembled A_inversion of a bowl mesh.
Usage: python tools/spmv_tune.py [workload] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import _lib as L  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "bowl3D_h0.02"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
arch = npg.GPU(0)
fed = workloads.example_fe_data(workloads.bowl_mesh_model(wl))
prm, frc = workloads.example_parameters()
A = npg.build_A_inversion(arch, fed, prm, frc.nu)
N, nnz = A.shape[0], A.nnz
if "--paired" in sys.argv:
    sys.argv.remove("--paired")
    assert A.block_nodes(fed.dofs.n_full, fed.dofs.n_surf)
    print("node-block storage:", A.storage(), "stored bytes", A.stored_spmv_bytes())
alg = 12 * nnz + 4 * (N + 1) + 16 * N
x = npg.DeviceVector.from_host(arch.ctx, np.sin(np.arange(N, dtype=float)))
yref = A.mul(x).to_host()
y = npg.DeviceVector(arch.ctx, N)
L.lib()      # the product library first: the harness links against it
_tune = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune", "libnupgcm_tune.so"))   # tuning harness (make -C tools/tune)
fn = _tune.npg_spmv_variant
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
fn.restype = C.c_int
names = {0: "NT512 T4096 U4", 1: "NT512 T4096 U8", 2: "NT1024 T8192 U8", 3: "NT256 T2048 U8", 4: "NT512 T8192 U8",
         5: "NT256 T4096 U8", 6: "NT1024 T4096 U4", 7: "wide NT512 T4096 U2x4", 8: "wide NT256 T4096 U2x8",
         9: "wide NT1024 T8192 U2x4", 10: "wide NT256 T2048 U2x4",
         11: "wide NT1024 T4096 U2x2", 12: "wide+nt NT1024 T8192", 13: "wide+nt NT512 T4096", 14: "wide NT1024 T8192 U2x2",
         15: "wide NT512 T8192 U2x4",
         30: "product NT512 T4096 U4 wpe6", 31: "product NT512 T2048 U2 wpe6", 32: "product NT1024 T4096 U2 wpe8",
         40: "product, gathers removed (diagnostic)", 42: "product, z gather of the records removed (diagnostic)",
         50: "wide NT512 T5824 UP4 UC3 (2 WG/CU)", 51: "wide NT512 T8192 UP6 UC4", 52: "wide NT512 T9216 UP6 UC4",
         53: "wide NT512 T8192 UP6 UC3", 54: "wide NT1024 T9216 UP3 UC2",
         43: "T4800: gathers from an LDS stage filled per tile (diagnostic)", 44: "T4800 product",
         45: "T4800 gathers removed (diagnostic)",
         60: "packed NT512 T2048 nt (3 WG/CU)", 61: "packed NT512 T2048 (3 WG/CU)", 62: "packed NT512 T2560 nt (2 WG/CU)",
         63: "packed NT256 T1024 nt (6 WG/CU)", 64: "packed NT1024 T2560 nt (1 WG/CU)",
         65: "packed NT256 T2048 nt (3 WG/CU)", 66: "packed wave tiles NT64 T256", 67: "packed wave tiles NT64 T384",
         68: "packed NT128 T512", 69: "packed wave tiles NT64 T512",
         70: "product tiles, column-sorted + slot index", 71: "product tiles, slot index only (unsorted)",
         72: "packed + column-sorted NT512 T1792", 73: "packed + column-sorted NT256 T896", 74: "packed NT512 T1792",
         75: "packed, gathers a tile ahead NT512 T1280 (3 WG/CU)", 76: "packed, gathers a tile ahead NT256 T640 (6 WG/CU)",
         77: "packed, gathers a tile ahead NT512 T1664 (2-3 WG/CU)",
         80: "product tiles, CU-local tile queues (adjacent tiles per CU)",
         81: "product tiles, the 3 workgroups of a CU on adjacent tiles, global order kept",
         82: "product kernel rebuilt in the harness (reference for 83-85)", 83: "gathers wrapped into 16 KB of x (all L1 hits; diagnostic)",
         84: "gathers wrapped into 256 KB of x (L2 hits; diagnostic)", 85: "gathers wrapped into 4 MB of x (diagnostic)",
         86: "T5120 tiles, gathers from global (reference for 87, 88)", 87: "T5120 tiles, gathers served by LDS reads (diagnostic)",
         88: "T5120 tiles, gathers replaced by arithmetic (diagnostic)",
         90: "x windows in LDS, T3072 tiles", 91: "same kernel, no windows (all gathers global)"}
print(f"{wl}: N={N} nnz={nnz} algorithmic bytes={alg / 1e6:.1f} MB")
import time
for _ in range(3):
    A.mul(x, y)
arch.ctx.sync()
t0 = time.perf_counter()
for _ in range(50):
    A.mul(x, y)
arch.ctx.sync()
print(f"product k_spmv (A.mul, wall over 50 calls): {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us")
vs = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else range(16)
for v in vs:
    for bpc in [int(b) for b in os.environ.get('BPC', '2,3,4,5,6,8').split(',')]:
        ms = C.c_double()
        y.fill(0.0)
        rc = fn(A.h, x.h, y.h, v, bpc, reps, C.byref(ms))
        if rc != 0:
            print(v, bpc, "error", L.lib().npg_last_error().decode())
            continue
        err = np.linalg.norm(y.to_host() - yref) / np.linalg.norm(yref)
        print(f"variant {v:2d} [{names[v]:24s}] blocks/CU<={bpc}: {ms.value * 1e3:8.1f} us  {alg / ms.value / 1e6:7.0f} GB/s  "
              f"relerr {err:.1e}")
