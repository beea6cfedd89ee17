#!/bin/bash
# round-3 session AD: the Arnoldi kernel's SpMV input from an fp32 gather-layout copy (one 16-byte gather per node record)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ad
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gather_layout or node_block or compressed or split or full_size" > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
for g in 1 0; do
NPG_GMRES_XG=$g timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid > $O/bench_g$g.json 2> $O/bench_g$g.err
python3 -c "
import json
d=json.loads(open('$O/bench_g$g.json').read().strip().splitlines()[-1]); print('gather=$g K1', round(d['roofline']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],1), 'its', d['config']['gmres_iterations_per_step'], d['config']['all_solved'])" | tee -a $O/summary.txt
done
