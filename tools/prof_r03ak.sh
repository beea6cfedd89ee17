#!/bin/bash
# round-3 session AK: lanes per row / node in the segmented sums, now that block tiles sum node-wise
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ak
mkdir -p $O
for l in 8 4 16; do
NPG_SPMV_LANES=$l timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid > $O/bench_l$l.json 2> $O/bench_l$l.err
python3 -c "
import json
d=json.loads(open('$O/bench_l$l.json').read().strip().splitlines()[-1]); print('lanes=$l K1', round(d['roofline']['avg_launch_us'],1), 'spmv', round(d['spmv_standalone']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],1), 'its', d['config']['gmres_iterations_per_step'])" | tee -a $O/summary.txt
done
