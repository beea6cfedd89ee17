#!/bin/bash
# round-3 session AO: stability - the whole GPU suite twice in a row on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ao
mkdir -p $O
for i in 1 2; do
timeout -k 10 550 python3 -m pytest tests -q -m gpu > $O/pytest_$i.txt 2>&1
echo "run $i rc=$? $(grep -E 'passed|failed' $O/pytest_$i.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^FAILED" $O/pytest_$i.txt | tee -a $O/summary.txt
done
