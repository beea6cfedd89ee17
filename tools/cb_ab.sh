#!/bin/bash
# channel-basin multigrid bench under different environments on ONE box: tools/cb_ab.sh NAME "ENV.." "ENV.." -> gpurun_out/NAME_<k>.json
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
name=$1; shift
k=0
for envs in "$@"; do
    env $envs timeout -k 10 ${CB_TIMEOUT:-500} python3 bench.py --workload ${CB_WORKLOAD:-channel_basin_h0.01} --preconditioner multigrid --steps ${CB_STEPS:-20} --warmup 5 --no-cpu-baseline > gpurun_out/${name}_$k.json 2> gpurun_out/${name}_$k.err || { echo "run $k ($envs) failed"; tail -5 gpurun_out/${name}_$k.err; exit 1; }    # (a failed GPU step ends the call: nothing else is started on that box)
    python3 - "$envs" gpurun_out/${name}_$k.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
c = d["config"]
inv = c["inversion_seconds_per_step"]
print(f"{sys.argv[1] or '(default)':44s} ms/step {d['ms_per_step']:.1f}  its {c['gmres_iterations_per_step']}  inversion ms/iteration "
      f"{1e3 * sum(inv) / sum(c['gmres_iterations_per_step']):.2f}  solved {c['all_solved']}  set-up {c['setup_seconds']} s", flush=True)
PY
    k=$((k+1))
done
