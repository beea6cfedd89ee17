"""ILU(0) of the reference's P-block (friction-only velocity block, src/preconditioners.jl:74-80,101-107) at size: levels of the
analysis, set-up and application times, and the block's CG with the factors against the Jacobi vector.
    python3 tools/ilu_probe.py [workload=bowl3D_h0.05]"""
import sys
import time

import numpy as np

import nupgcm_amd as npg
from nupgcm_amd import workloads

wl = sys.argv[1] if len(sys.argv) > 1 else "bowl3D_h0.05"
arch = npg.GPU()
ctx = arch.ctx
prm, frc = workloads.example_parameters()
fed = workloads.example_fe_data(workloads.bowl_mesh_model(wl))
d = fed.dofs
t0 = time.time()
P = npg.BlockDiagonalPreconditioner(arch, prm, fed, u_itmax=100, p_itmax=0, atol=1e-6, rtol=1e-6, u_precond="ilu0")
ctx.sync()
t_set = time.time() - t0
M, A = P.ilu, P.ilu.A
t0 = time.time()
M.refactor()
ctx.sync()
t_fac = time.time() - t0
n = A.shape[0]
print(f"{wl}: friction block {n} x {n}, {A.nnz} entries ({A.nnz / n:.1f} per row); levels lower / upper {M.levels}; "
      f"set-up (assembly + analysis + factorisation) {t_set:.2f} s, factorisation alone {1e3 * t_fac:.1f} ms", flush=True)
r = npg.DeviceVector.from_host(ctx, np.cos(np.arange(n) * 0.11))
z = npg.DeviceVector(ctx, n)
for _ in range(2):
    M.ldiv(r, z)
ctx.sync()
t0 = time.time()
for _ in range(5):
    M.ldiv(r, z)
ctx.sync()
t_app = (time.time() - t0) / 5
print(f"z = U^-1 L^-1 r: {1e3 * t_app:.2f} ms per application = {1e6 * t_app / sum(M.levels):.2f} us per level launch (hipGraph replay)", flush=True)
x = npg.DeviceVector(ctx, n)
st = M.cg(A, r, x, atol=1e-6, rtol=1e-6, itmax=100)
print(f"CG with the ILU(0) factors: {st['niter']} iterations, solved {st['solved']}, {1e3 * st['seconds']:.1f} ms", flush=True)
jac = npg.DeviceVector.from_host(ctx, 1.0 / A.to_scipy_csr().diagonal())
ws = npg.CgWorkspace(ctx, n)
x2 = npg.DeviceVector(ctx, n)
st2 = ws.solve(A, r, x2, npg.Diagonal(jac), atol=1e-6, rtol=1e-6, itmax=0)
st2 = ws.solve(A, r, x2.fill(0.0), npg.Diagonal(jac), atol=1e-6, rtol=1e-6, itmax=0)
print(f"CG with the Jacobi vector:  {st2['niter']} iterations, solved {st2['solved']}, {1e3 * st2['seconds']:.1f} ms", flush=True)
