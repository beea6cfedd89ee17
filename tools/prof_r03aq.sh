#!/bin/bash
# round-3 session AQ: full GPU suite with full node records in, incl. the eddy re-assembly behind a record-form companion
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03aq
mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "closures_in_the_timestep or full_node" > $O/pytest_a.txt 2>&1
echo "closures rc=$? $(grep -E 'passed|failed' $O/pytest_a.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^E  " $O/pytest_a.txt | head -5 | tee -a $O/summary.txt
timeout -k 10 800 python3 -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^FAILED" $O/pytest.txt | tee -a $O/summary.txt
