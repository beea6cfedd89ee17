"""channel basin with a deeper hierarchy (levels = 3: coarsest level small enough for the exact dense solve): set-up stages with a
watchdog stack dump, then a few timesteps.    python3 tools/cb_levels_probe.py [levels=3] [steps=12] [h=0.01]"""
import faulthandler
import sys
import time

faulthandler.dump_traceback_later(150, repeat=True, file=sys.stderr)
import os                                    # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg                      # noqa: E402
from nupgcm_amd import workloads              # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
h = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
t0 = time.time()
arch = npg.GPU()
m = workloads.channel_basin_model(arch, h=h, levels=levels, surface="flux", itmax=0)
print(f"set-up {time.time() - t0:.1f} s; {m.inversion.solver.P!r}", flush=True)
for k in range(steps):
    t1 = time.time()
    npg.run(m, n_steps=1)
    s = m.stats[-1]
    print(f"step {k}: its {s[1]['niter']} solved {s[1]['solved']} cg {s[0]['niter']} dt {m.timestepper.dt:.3e} {1e3 * (time.time() - t1):.0f} ms", flush=True)
