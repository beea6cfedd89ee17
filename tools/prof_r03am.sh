#!/bin/bash
# round-3 session AM: fp32 dense inverse applied with 16-byte loads and a pipelined loop - multigrid tests and bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03am
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_precond.py -q -m gpu -x > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/K -- python3 bench.py --preconditioner multigrid --steps 3 --warmup 1 --no-cpu-baseline > $O/K.out 2> $O/K.err
f=$(find $O/K -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/K_kernel_stats.csv
rm -rf $O/K
grep "gemv" $O/K_kernel_stats.csv | cut -d, -f1-4 | tee -a $O/summary.txt
timeout -k 10 400 python3 bench.py --preconditioner multigrid --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_mg.json 2> $O/bench_mg.err
python3 -c "
import json
d=json.loads(open('$O/bench_mg.json').read().strip().splitlines()[-1]); print('multigrid ms', round(d['ms_per_step'],2), d['config']['gmres_iterations_per_step'], d['config']['all_solved'])" | tee -a $O/summary.txt
