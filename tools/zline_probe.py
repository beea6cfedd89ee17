"""z-line against node-block smoothing on the channel basin (BASELINE configs[4] recipe): outer iterations and time per timestep.
    python3 tools/zline_probe.py [h=0.03125] [levels=2] [steps=12]      (NPG_MG_OMEGA overrides the Braess-Sarazin scaling)"""
import sys
import time

import nupgcm_amd as npg
from nupgcm_amd import workloads

h = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03125
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
arch = npg.GPU()
for smoother, omega in (("node", 2.0), ("zline", 2.0), ("zline", 1.5)):
    t0 = time.time()
    m = workloads.channel_basin_model(arch, h=h, levels=levels, itmax=600, precond_kw=dict(smoother=smoother, omega=omega))
    arch.ctx.sync()
    ts = time.time() - t0
    t0 = time.time()
    npg.run(m, n_steps=steps)
    arch.ctx.sync()
    el = time.time() - t0
    st = m.stats[-steps:]
    print(f"{smoother:6s} omega {omega}: set-up {ts:.1f} s, {1e3 * el / steps:.1f} ms per timestep, outer iterations {[s[1]['niter'] for s in st]}, "
          f"all solved {all(s[1]['solved'] == 1 for s in st)}, inversion ms {[round(1e3 * s[1]['seconds'], 1) for s in st]}", flush=True)
    print("   ", repr(m.inversion.solver.P), flush=True)
    del m
