#!/bin/bash
# round-3 session AF: column records {m, a_x, a_y, a_z} for the block rows' entries outside the block columns
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03af
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_precond.py -q -m gpu -x -k "gather_layout or node_block or compressed or split or full_size or mixed or linear" > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
for g in 1 0 1 0; do
NPG_SPMV_COLUMN_RECORDS=$g timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid > $O/bench_c$g.json 2> $O/bench_c$g.err
python3 -c "
import json
d=json.loads(open('$O/bench_c$g.json').read().strip().splitlines()[-1]); print('column records=$g K1', round(d['roofline']['avg_launch_us'],1), 'spmv', round(d['spmv_standalone']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],1), 'its', d['config']['gmres_iterations_per_step'], d['config']['all_solved'], 'stored', d['roofline']['stored_bytes_per_launch'])" | tee -a $O/summary.txt
done
