#!/usr/bin/env python3
"""BASELINE configs[4] at production size with the multigrid hierarchy: timesteps with different Braess-Sarazin scalings omega
(and smoothing counts) on ONE model - 6 steps each, iterations and ms per step.  Usage: python tools/mg_channel_omega.py [h]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402

h = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
combos = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [{}, {"omega": 1.8}, {"omega": 1.5}]
arch = npg.GPU(0)
t0 = time.time()
m = workloads.channel_basin_model(arch, h=h, levels=2, itmax=0)
P = m.inversion.solver.P
print(f"channel_basin h={h}: set-up {time.time() - t0:.1f} s, levels {[lv['n'] for lv in P.levels]}", flush=True)
npg.run(m, n_steps=2)
for kw in combos:
    P.set_params(**kw)
    n0 = len(m.stats)
    arch.ctx.sync()
    t0 = time.perf_counter()
    npg.run(m, n_steps=6)
    arch.ctx.sync()
    el = time.perf_counter() - t0
    st = m.stats[n0:]
    print(f"{kw}: iterations {[s[1]['niter'] for s in st]} solved {all(s[1]['solved'] == 1 for s in st)}  {1e3 * el / 6:.1f} ms per timestep", flush=True)
