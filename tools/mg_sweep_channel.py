#!/usr/bin/env python3
"""Parameter sweep of the multigrid-preconditioned inversion on the channel-basin mesh (BASELINE configs[4]; anisotropic cells:
62-79 outer iterations with the bowl's settings): iterations and milliseconds per COLD solve for V/W-cycle and smoother settings.
Usage: python tools/mg_sweep_channel.py [h] ['<json list of parameter dicts>'] [itmax]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402

h = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
arch = npg.GPU(0)
t0 = time.time()
m = workloads.channel_basin_model(arch, h=h, levels=2, itmax=0)
print(f"channel_basin h={h}: set-up {time.time() - t0:.1f} s, levels {m.inversion.solver.P.levels}", flush=True)
s = m.inversion.solver
P = s.P
combos = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [dict()]
s.kwargs["itmax"] = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for kw in combos:
    P.set_params(**kw)
    s.x.fill(0.0)
    npg.invert(m)
    st = s.workspace.stats
    hist = s.workspace.history()
    rate = (hist[-1] / hist[0]) ** (1.0 / max(len(hist) - 1, 1))
    print(f"{kw}: solved={st['solved']} its={st['niter']} {1e3 * st['seconds']:.1f} ms  ({1e3 * st['seconds'] / max(st['niter'], 1):.2f} ms/it)"
          f"  mean factor per iteration {rate:.3f}", flush=True)
