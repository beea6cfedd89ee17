#!/bin/bash
# round-3 session Z: full GPU suite after the coupling records and the halo epoch fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03z
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x -s > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
grep "channel, 3 ranks" $O/pytest.txt | tee -a $O/summary.txt
