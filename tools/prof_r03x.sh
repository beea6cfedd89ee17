#!/bin/bash
# round-3 session X: coupling records {c, d_x, d_y, d_z} for the divergence rows as a tile class of their own - parity tests,
# then the default bench with and without them; determinism probe of the single-GPU channel run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03x
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_precond.py -q -m gpu -x -k "node_block or linear or storage or mixed or split or gmres" > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
for c in 1 0; do
NPG_SPMV_COUPLING=$c timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid > $O/bench_c$c.json 2> $O/bench_c$c.err
python3 -c "
import json
d=json.loads(open('$O/bench_c$c.json').read().strip().splitlines()[-1]); print('coupling=$c K1', round(d['roofline']['avg_launch_us'],1), 'spmv', round(d['spmv_standalone']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],1), 'its', d['config']['gmres_iterations_per_step'], 'stored', d['roofline']['stored_bytes_per_launch'])" | tee -a $O/summary.txt
done
timeout -k 10 300 python3 tools/channel_determinism_probe.py > $O/determinism.txt 2>&1
tail -5 $O/determinism.txt | tee -a $O/summary.txt
