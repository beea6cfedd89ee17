#!/bin/bash
# round-3 session AJ: node-wise segmented sums in all-record block tiles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03aj
mkdir -p $O
timeout -k 10 300 python3 tools/spmv_phases.py bowl3D_h0.02 > $O/phases.txt 2>&1
tail -7 $O/phases.txt | tee -a $O/summary.txt
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_precond.py -q -m gpu -x -k "gather_layout or node_block or compressed or split or full_size or mixed or linear" > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid > $O/bench.json 2> $O/bench.err
python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('K1', round(d['roofline']['avg_launch_us'],1), 'spmv', round(d['spmv_standalone']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],1), 'its', d['config']['gmres_iterations_per_step'], d['config']['all_solved'])" | tee -a $O/summary.txt
