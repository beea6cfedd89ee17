#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03r
mkdir -p $O
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer timeout -k 10 300 python3 tools/rccl_cycle_cost.py 270000 > $O/cc_peer_default.txt 2>&1
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer NPG_HALO_DIRECT=1 timeout -k 10 300 python3 tools/rccl_cycle_cost.py 270000 > $O/cc_peer_direct.txt 2>&1
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer NPG_HALO_OVERLAP=0 timeout -k 10 300 python3 tools/rccl_cycle_cost.py 270000 > $O/cc_peer_nooverlap.txt 2>&1
grep -H iteration $O/cc_*.txt | tee -a $O/summary.txt
NPG_GMRES_TRACE=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/trace.json 2> $O/trace.err
grep "npg gmres" $O/trace.err | tail -1 | tee -a $O/summary.txt
