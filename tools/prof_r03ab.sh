#!/bin/bash
# round-3 session AB: software-pipelined row kernels (dots / orthogonalisation over the fp32-stored basis)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ab
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gmres or split or compressed or K5 or full_size or invert or 50_steps" > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/K -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/K.out 2> $O/K.err
echo "bench under tracer rc=$?" | tee -a $O/summary.txt
f=$(find $O/K -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/K_kernel_stats.csv
rm -rf $O/K
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid > $O/bench.json 2> $O/bench.err
python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('K1', round(d['roofline']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],1), 'its', d['config']['gmres_iterations_per_step'])" | tee -a $O/summary.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_rccl_selftest.py -q -m gpu -x > $O/pytest_dist.txt 2>&1
echo "pytest dist rc=$? $(grep -E 'passed|failed' $O/pytest_dist.txt | tail -1)" | tee -a $O/summary.txt
