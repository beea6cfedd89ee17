#!/bin/bash
# A/B of the multigrid timestep loop on ONE box: tools/mg_ab.sh NAME "ENV1=.. ENV2=.." "ENV=.." ...  (one bench run per quoted
# environment, alternating is up to the caller) -> gpurun_out/NAME_<k>.json, one summary line each
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
name=$1; shift
k=0
for envs in "$@"; do
    env $envs timeout -k 10 400 python3 bench.py --preconditioner multigrid --steps ${MG_STEPS:-20} --warmup 3 --no-cpu-baseline --no-profile-pass > gpurun_out/${name}_$k.json 2> gpurun_out/${name}_$k.err || { echo "run $k failed"; tail -5 gpurun_out/${name}_$k.err; exit 1; }
    python3 - "$envs" gpurun_out/${name}_$k.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
c = d["config"]
inv = c["inversion_seconds_per_step"]
print(f"{sys.argv[1] or '(default)':40s} ms/step {d['ms_per_step']:.2f}  its {c['gmres_iterations_per_step']}  inversion ms/iteration "
      f"{1e3 * sum(inv) / sum(c['gmres_iterations_per_step']):.3f}  solved {c['all_solved']}")
PY
    k=$((k+1))
done
