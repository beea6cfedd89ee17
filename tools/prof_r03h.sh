#!/bin/bash
# round-3 session H: compressed-basis GMRES - the parity tests that touch GMRES, then the default bench with fp64 and fp32 stored basis
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_parity.txt 2>&1
echo "pytest parity rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED|^E  " $O/pytest_parity.txt | head -20
for b in 64 32; do
NPG_GMRES_BASIS=$b timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-multigrid > $O/bench_basis$b.json 2> $O/bench_basis$b.err
echo "bench basis $b rc=$? $(python3 -c "
import json
d=json.loads(open('$O/bench_basis$b.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), d['config']['gmres_iterations_per_step'], d['config']['all_solved'], round(d['roofline']['avg_launch_us'],1))
")" | tee -a $O/summary.txt
done
NPG_GMRES_TRACE=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/trace32.json 2> $O/trace32.err
grep "npg gmres" $O/trace32.err | tail -2 | tee -a $O/summary.txt
NPG_GMRES_BASIS=64 NPG_GMRES_TRACE=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/trace64.json 2> $O/trace64.err
grep "npg gmres" $O/trace64.err | tail -2 | tee -a $O/summary.txt
cat $O/summary.txt
