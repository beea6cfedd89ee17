#!/bin/bash
# Arnoldi-kernel A/B on ONE box: tools/k1_ab.sh NAME "ENV.." "ENV.." ... -> one line per environment (profile-pass launch time, ms/step)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
name=$1; shift
k=0
for envs in "$@"; do
    env $envs timeout -k 10 300 python3 bench.py --steps ${K1_STEPS:-4} --warmup 2 --no-cpu-baseline --no-multigrid > gpurun_out/${name}_$k.json 2> gpurun_out/${name}_$k.err || { echo "run $k failed"; tail -5 gpurun_out/${name}_$k.err; exit 1; }
    python3 - "$envs" gpurun_out/${name}_$k.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[1] or '(default)':32s} Arnoldi {r['avg_launch_us']:.1f} us  frac {r['frac']:.3f}  ms/step {d['ms_per_step']:.1f}  its {d['config']['gmres_iterations_per_step']}", flush=True)
PY
    k=$((k+1))
done
