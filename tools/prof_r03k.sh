#!/bin/bash
# round-3 session K: full GPU test-suite on the round's code, then the driver-shaped default bench, PMC passes, cycle cost
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03k
mkdir -p $O
timeout -k 10 1150 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "pytest gpu rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED|^E  " $O/pytest_gpu.txt | head -30
