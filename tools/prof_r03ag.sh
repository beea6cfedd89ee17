#!/bin/bash
# round-3 session AG: full GPU suite on the code with column records + gather-layout input
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ag
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^FAILED" $O/pytest.txt | tee -a $O/summary.txt
