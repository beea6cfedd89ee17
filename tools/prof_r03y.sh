#!/bin/bash
# round-3 session Y: repeatability of the distributed rehearsals while a fourth process keeps the GPU busy
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
timeout -k 10 1000 python3 tools/dist_repeat_probe.py 3 mg bench > $O/matrix6.txt 2> $O/matrix6.err
echo "rc=$?" >> $O/matrix6.txt
cat $O/matrix6.txt | tee -a $O/summary.txt
