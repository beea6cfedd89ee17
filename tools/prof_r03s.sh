#!/bin/bash
# round-3 session S: embedded 2-D meshes on the device; the whole GPU suite; one-GPU iteration time after the revert
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "embedded_2d" > $O/pytest_2d.txt 2>&1
echo "pytest 2d rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED|^E  " $O/pytest_2d.txt | head -20
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "pytest gpu rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED" $O/pytest_gpu.txt | head -20
NPG_GMRES_TRACE=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/trace.json 2> $O/trace.err
grep "npg gmres" $O/trace.err | tail -1 | tee -a $O/summary.txt
