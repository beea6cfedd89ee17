"""Is the single-GPU channel-basin run (the `ref` of tests/test_gpu_distributed.py) reproducible inside one process?
Runs the model three times (a bowl model is built and stepped between runs, to dirty the allocator) and prints a digest of
(u, b, dt) per run.  Usage on the GPU box: python3 tools/channel_determinism_probe.py"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import channel_basin, workloads  # noqa: E402

arch = npg.GPU()
mm = channel_basin.channel_basin_model(0.0625, workloads.CB_ALPHA)
runs = []
for k in range(3):
    ref = workloads.channel_basin_model(arch, mesh_model=mm, element_precision="fp64")
    npg.run(ref, n_steps=11)
    u, b = np.array(ref.state.u), np.array(ref.state.b)
    runs.append((u, b, ref.timestepper.dt))
    print(k, hashlib.sha1(u.tobytes()).hexdigest()[:12], hashlib.sha1(b.tobytes()).hexdigest()[:12], repr(ref.timestepper.dt),
          [s[0]["niter"] for s in ref.stats][-3:], flush=True)
    del ref
    other = workloads.example_model(arch, "bowl3D_h0.1") if hasattr(workloads, "example_model") else None
    if other is not None:
        npg.run(other, n_steps=2)
    del other
for k in (1, 2):
    du = np.linalg.norm(runs[k][0] - runs[0][0]) / np.linalg.norm(runs[0][0])
    db = np.linalg.norm(runs[k][1] - runs[0][1]) / np.linalg.norm(runs[0][1])
    print(f"run {k} vs run 0: rel(u) = {du:.3e}, rel(b) = {db:.3e}, dt equal: {runs[k][2] == runs[0][2]}")
