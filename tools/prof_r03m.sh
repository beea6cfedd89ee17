#!/bin/bash
# round-3 session M: distributed multigrid rehearsal
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_distributed.py -q -m gpu -k "multigrid" -x > $O/pytest_dmg.txt 2>&1
echo "pytest dmg rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED|^E  " $O/pytest_dmg.txt | head -30
tail -40 $O/pytest_dmg.txt | grep -v "^$" | tail -30
