#!/bin/bash
# round-3 session Y: repeatability of the distributed rehearsals while a fourth process keeps the GPU busy
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
timeout -k 10 1000 python3 tools/dist_repeat_probe.py 4 stress shm peer_eager_nooverlap peer ppeer pshm > $O/matrix5.txt 2> $O/matrix5.err
cat $O/matrix5.txt | tee -a $O/summary.txt
