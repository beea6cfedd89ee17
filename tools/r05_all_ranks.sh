#!/bin/bash
# round 5: every rank's share of the 8-, 4- and 2-rank partitions of the bench system built and run through the one-rank self-test
# communicator (ghost nodes as record columns, windowed tiles, halo plan of the real size) - what the driver's multi-GPU run sets up
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r05_all_ranks.txt
for spec in "8 0" "8 1" "8 2" "8 3" "8 5" "8 6" "8 7" "4 0" "4 2" "4 3" "2 0" "2 1"; do
    set -- $spec
    NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer timeout -k 10 200 python3 tools/rank_cycle_probe.py bowl3D_h0.02 $1 $2 400 >> gpurun_out/r05_all_ranks.txt 2> gpurun_out/r05_all_ranks.err || { echo "rank $2 of $1 failed"; tail -5 gpurun_out/r05_all_ranks.err; exit 1; }
    tail -3 gpurun_out/r05_all_ranks.txt | cut -c1-250
done
