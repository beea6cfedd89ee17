"""channel basin at production size, step by step: outer iterations, solved flags, dt - until a blow-up or `steps` timesteps.
    python3 tools/cb_long.py [steps=45] [h=0.01]    (NPG_MG_SMOOTHER / NPG_MG_OMEGA / NPG_MG_PARAMS / NPG_MG_MIXED select the cycle)"""
import sys
import time

import nupgcm_amd as npg
from nupgcm_amd import workloads

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 45
h = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
arch = npg.GPU()
m = workloads.channel_basin_model(arch, h=h, levels=2, surface="flux", itmax=0)
print(repr(m.inversion.solver.P), flush=True)
t0 = time.time()
for k in range(steps):
    try:
        npg.run(m, n_steps=1)
    except npg.BlowUp as e:
        print(f"step {k}: {e}", flush=True)
        break
    s = m.stats[-1]
    print(f"step {k}: its {s[1]['niter']} solved {s[1]['solved']} rnorm0 {s[1]['rnorm0']:.3e} cg {s[0]['niter']} dt {m.timestepper.dt:.4e} "
          f"max|u| {float(abs(m.state.u).max()):.3e}  t {time.time() - t0:.1f}s", flush=True)
