#!/usr/bin/env python3
"""A short, fixed amount of hot-path work for counter collection: assembles A_inversion of a bowl mesh and runs exactly
`cycles` GMRES(20) restart cycles (itmax = 20*cycles) plus a few stand-alone SpMVs.  Run it under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and again with WRITE_SIZE)  --output-format csv -d <dir> -- python3 tools/pmc_probe.py
and post-process with tools/pmc_summary.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "bowl3D_h0.02"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 2
arch = npg.GPU(0)
fed = workloads.example_fe_data(workloads.bowl_mesh_model(wl))
prm, frc = workloads.example_parameters()
A = npg.build_A_inversion(arch, fed, prm, frc.nu)
N = A.shape[0]
nnz_csr = A.nnz
paired = N >= 100000 and A.block_nodes(fed.dofs.n_full, fed.dofs.n_surf)   # as InversionToolkit (nupgcm_amd/inversion.py)
npairs, npe, nnz_rem = A.storage()
h = fed.mesh.median_edge_length()
y = npg.DeviceVector.from_host(arch.ctx, np.sin(np.arange(N, dtype=float)) * 1e-3)
ws = npg.GmresWorkspace(arch.ctx, N, memory=20)
if os.environ.get("PMC_FP64"):          # the all-fp64 instance (bench.py's `fp64_basis` object): fp64 basis, fp64 gathers
    ws.set_basis(64)
    ws.set_gather(0)
st = ws.solve(A, y, ws.x, npg.Diagonal(scalar=1 / h ** 3), itmax=20 * cycles)
x = npg.DeviceVector.from_host(arch.ctx, np.cos(np.arange(N, dtype=float)))
out = npg.DeviceVector(arch.ctx, N)
for _ in range(5):
    A.mul(x, out)
arch.ctx.sync()
print(f"{wl}: N={N} nnz={nnz_csr} algorithmic SpMV bytes={12 * nnz_csr + 4 * (N + 1) + 16 * N} iterations={st['niter']} "
      f"node_blocks={bool(paired)} records={npe} coupling_records={A.coupling_records()} csr_entries={nnz_rem} "
      f"stored SpMV bytes={A.stored_spmv_bytes()}")
