#!/usr/bin/env python3
"""Parameter sweep of the multigrid-preconditioned inversion on a refined bowl mesh: iterations and milliseconds per solve
(cold start, reference stopping rule) for V-cycle / smoother settings.  Usage: python tools/mg_sweep.py [workload]"""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "bowl3D_h0.02"
arch = npg.GPU(0)
t0 = time.time()
m = workloads.example_model(arch, wl, preconditioner="multigrid")
prm = m.params
npg.set_b(m, lambda x: 0.1 * np.exp(-(x[..., 2] + prm.H(x)) / (0.1 * prm.alpha)))
print(f"{wl}: set-up {time.time() - t0:.1f} s, levels {m.inversion.solver.P.levels}", flush=True)
s = m.inversion.solver
P = s.P
import json
combos = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [dict()]
s.kwargs["itmax"] = int(sys.argv[3]) if len(sys.argv) > 3 else 150
from nupgcm_amd import _lib as L  # noqa: E402
for kw in combos:
    kw = dict(kw)
    L.check(L.lib().npg_precond_mg_set_coarse_dense(P.h, int(kw.pop("coarse_dense", 0))))
    mixed = int(kw.pop("mixed", 0))
    L.check(L.lib().npg_precond_mg_set_mixed(P.h, mixed))
    P.set_params(**kw)
    s.x.fill(0.0)
    npg.invert(m)
    st = s.workspace.stats
    h = s.workspace.history()
    rate = (h[-1] / h[0]) ** (1.0 / max(len(h) - 1, 1))
    print("   history every 10:", " ".join(f"{v / h[0]:.1e}" for v in h[::10]), flush=True)
    kw["mixed"] = mixed
    print(f"{kw}: solved={st['solved']} its={st['niter']} {1e3 * st['seconds']:.1f} ms  ({1e3 * st['seconds'] / max(st['niter'], 1):.2f} ms/it)"
          f"  residual x{h[-1] / h[0]:.2e}, mean factor per iteration {rate:.3f}", flush=True)
