"""Which part of a distributed run stops repeating bit for bit when ANOTHER process keeps the GPU busy?
(tests/test_gpu_distributed.py failed once on a marginal tolerance; alone on the card every run repeats exactly.)
The parent holds a GPU context and keeps stepping a bowl model, as the pytest process's serial reference leaves the card;
each configuration below is then run `reps` times and the digests of (u, b) are compared.

  python3 tools/dist_repeat_probe.py [reps] [config ...]      configs: see CONFIGS"""
import hashlib
import os
import subprocess
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402
from tests import test_gpu_distributed as T  # noqa: E402

CONFIGS = {
    # name: (worker mode, transport, extra environment)
    "serial": None,
    "stress": None,
    "shm": ("channel", "shm", {}),
    "peer_eager_nooverlap": ("channel", "peer", {"NPG_DIST_GRAPH": "0", "NPG_HALO_OVERLAP": "0"}),
    "peer_eager": ("channel", "peer", {"NPG_DIST_GRAPH": "0"}),
    "peer_nooverlap": ("channel", "peer", {"NPG_HALO_OVERLAP": "0"}),
    "peer": ("channel", "peer", {}),
    "ppeer": ("pchannel", "peer", {}),
    "pshm": ("pchannel", "shm", {}),
    "blocks_peer": ("blocks", "peer", {"NPG_GMRES_SPLIT": "1"}),
    "blocks_shm": ("blocks", "shm", {"NPG_GMRES_SPLIT": "1"}),
}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    names = sys.argv[2:] or list(CONFIGS)
    arch = npg.GPU()
    stop = []
    if os.environ.get("PROBE_IDLE") != "1":
        m = workloads.example_model(arch, "bowl3D_h0.05")

        def spin():
            while not stop:
                npg.run(m, n_steps=1)

        th = threading.Thread(target=spin)
        th.start()
    for name in names:
        if name == "serial":
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "channel_determinism_probe.py")], capture_output=True,
                               text=True, timeout=600)
            print("serial:", r.stdout.strip().splitlines()[-2:], r.stderr[-300:] if r.returncode else "", flush=True)
            continue
        if name == "stress":
            env = dict(os.environ, NPG_COMM_TRANSPORT="peer", NPG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                       NPG_PEER_TIMEOUT_S="60")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr",
                   "127.0.0.1", "--master-port", str(T._free_port()), os.path.join(ROOT, "tests", "peer_stress_worker.py"), "3000"]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            print("stress: rc", r.returncode, [ln for ln in r.stdout.splitlines() if "rank" in ln or "STRESS" in ln], flush=True)
            continue
        if name in ("mg", "bench"):
            env = dict(os.environ, NPG_COMM_TRANSPORT="peer", NPG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                       NPG_PEER_TIMEOUT_S="60", OMP_NUM_THREADS="2", NPG_TORCH_BACKEND="gloo")
            got = []
            for k in range(reps):
                head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr",
                        "127.0.0.1", "--master-port", str(T._free_port())]
                if name == "mg":
                    out = f"/tmp/rep_mg_{k}"
                    r = subprocess.run(head + [os.path.join(ROOT, "tests", "dist_mg_worker.py"), out, "3", "bowl3D_h0.05"], env=env,
                                       capture_output=True, text=True, timeout=900)
                    assert r.returncode == 0, r.stderr[-2000:]
                    z = np.load(out + ".rank0.npz")
                    got.append((hashlib.sha1(z["u"].tobytes() + z["b"].tobytes()).hexdigest()[:12], list(map(int, z["its"]))))
                else:
                    r = subprocess.run(head + [os.path.join(ROOT, "bench.py"), "--gpus", "3", "--workload", "bowl3D_h0.04", "--steps",
                                               "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
                    assert r.returncode == 0, r.stderr[-2000:]
                    import json
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    got.append((d["config"]["gmres_iterations_per_step"], d["config"]["gmres_initial_residual_per_step"],
                                d["comm"]["transport_check"], round(d["ms_per_step"], 1)))
                print("   ", name, k, got[-1], flush=True)
            print(f"{name}: {len(set(map(str, [g[:2] for g in got])))} distinct result(s) in {reps} runs", flush=True)
            continue
        mode, transport, extra = CONFIGS[name]
        os.environ.update(extra)
        seen, hists, pieces = {}, {}, {}
        for k in range(reps):
            out = f"/tmp/rep_{name}_{k}"
            T._launch(3, out, int(os.environ.get("PROBE_STEPS", 11)), mode, transport)
            z = [np.load(f"{out}.rank{r}.npz") for r in range(3)]
            dig = hashlib.sha1(z[0]["u"].tobytes() + z[0]["b"].tobytes()).hexdigest()[:12]
            if dig not in seen:
                print("   ", name, k, dig, "cg", list(map(int, z[0]["cg"])), "gm", list(map(int, z[0]["gm"]))[-3:], "dt", float(z[0]["dt"]), flush=True)
            seen.setdefault(dig, z[0]["u"].copy())
            if "hist_gm" in z[0]:
                hists.setdefault(dig, (z[0]["hist_gm"].copy(), z[0]["stat_gm"].copy()))
            if "hist_cg" in z[0] and dig not in pieces:
                pieces[dig] = [{k: zz[k].copy() for k in ("hist_cg", "rhs_b", "rhs_inv", "A_evol", "x_evol")} for zz in z]
        for key in extra:
            del os.environ[key]
        keys = list(seen)
        for d in list(pieces)[1:]:
            for r in range(3):
                a, b = pieces[list(pieces)[0]][r], pieces[d][r]
                msg = []
                for k in a:
                    if a[k].shape != b[k].shape:
                        msg.append(f"{k}: shapes {a[k].shape} {b[k].shape}")
                    elif not np.array_equal(a[k], b[k]):
                        bad = np.nonzero(a[k] != b[k])[0]
                        msg.append(f"{k}: {len(bad)} of {a[k].size} differ, first at {int(bad[0])} ({a[k][bad[0]]!r} vs {b[k][bad[0]]!r}), "
                                   f"last at {int(bad[-1])}, rel {np.linalg.norm(a[k] - b[k]) / np.linalg.norm(a[k]):.2e}")
                print(f"    pieces {d} vs {list(pieces)[0]}, rank {r}: " + ("; ".join(msg) or "all equal"), flush=True)
        for d in list(hists)[1:]:
            h0, h1 = hists[list(hists)[0]][0], hists[d][0]
            n = min(len(h0), len(h1))
            bad = np.nonzero(h0[:n] != h1[:n])[0]
            print(f"    residual history {d} vs {list(hists)[0]}: lengths {len(h1)} {len(h0)}, first differing iteration "
                  f"{int(bad[0]) if len(bad) else None}, values there {h0[bad[0]] if len(bad) else ''} {h1[bad[0]] if len(bad) else ''}; "
                  f"stats {hists[d][1]} {hists[list(hists)[0]][1]}", flush=True)
        spread = max([float(np.linalg.norm(seen[d] - seen[keys[0]]) / np.linalg.norm(seen[keys[0]])) for d in keys[1:]] or [0.0])
        print(f"{name}: {len(keys)} distinct result(s) in {reps} runs, largest rel(u) between them {spread:.3e}", flush=True)
    stop.append(1)


if __name__ == "__main__":
    main()
