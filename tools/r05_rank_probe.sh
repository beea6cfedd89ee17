#!/bin/bash
# rank 4 of 8 of the bench system on one GPU: per-iteration cost of the distributed cycle (self-test communicator), per-launch table
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out; mkdir -p $O
name=${1:-r05_rank}
export NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer
timeout -k 10 400 python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 2000 > $O/${name}_probe.txt 2> $O/${name}_probe.err || { tail -20 $O/${name}_probe.err; exit 1; }
cat $O/${name}_probe.txt
tools/prof.sh trace ${name}_tr python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 400 || exit 1
cat $O/${name}_tr.out
