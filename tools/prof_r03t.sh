#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03t
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "embedded_2d" > $O/pytest_2d.txt 2>&1
echo "pytest 2d rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED|^E  " $O/pytest_2d.txt | head -20
