#!/bin/bash
# round-3 session O: communication overhead of the distributed cycle at one rank's share of the bench system (270 k rows, split organisation)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03o
mkdir -p $O
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer timeout -k 10 300 python3 tools/rccl_cycle_cost.py 270000 > $O/cc_peer.txt 2>&1
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer NPG_HALO_OVERLAP=0 timeout -k 10 300 python3 tools/rccl_cycle_cost.py 270000 > $O/cc_peer_nooverlap.txt 2>&1
NPG_COMM_SELFTEST=1 timeout -k 10 300 python3 tools/rccl_cycle_cost.py 270000 > $O/cc_rccl.txt 2>&1
grep -H iteration $O/cc_*.txt
