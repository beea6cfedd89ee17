#!/bin/bash
# the end ranks of the 8-, 4- and 2-rank partitions after the split rule took the exchange's size into account
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r05_end_ranks.txt
for spec in "8 0" "8 7" "4 0" "4 3" "2 0" "2 1"; do
    set -- $spec
    NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer NPG_HALO_OVERLAP_VERBOSE=1 timeout -k 10 200 python3 tools/rank_cycle_probe.py bowl3D_h0.02 $1 $2 400 >> gpurun_out/r05_end_ranks.txt 2> gpurun_out/r05_end_ranks.err || { echo "rank $2 of $1 failed"; tail -5 gpurun_out/r05_end_ranks.err; exit 1; }
    grep -h "halo overlap" gpurun_out/r05_end_ranks.err | tail -1
    tail -3 gpurun_out/r05_end_ranks.txt | cut -c1-250
done
