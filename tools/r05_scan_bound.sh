#!/bin/bash
# round 5: what could a register segmented scan buy at most?  the stand-alone windowed product with (a) its own baseline, (b) no
# segmented sums, (c) products never reaching LDS (no product slots, no sums, one barrier less) - results of (b), (c) are meaningless
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r05_scan_bound.txt
for d in 256 2 32 256 32; do
    echo "NPG_WIN_DIAG=$d" >> gpurun_out/r05_scan_bound.txt
    NPG_WIN_DIAG=$d timeout -k 10 300 python3 tools/window_ab.py bowl3D_h0.02 200 2>> gpurun_out/r05_scan_bound.err | grep -E "windowed|ordinary" >> gpurun_out/r05_scan_bound.txt || { echo "diag $d failed"; tail -5 gpurun_out/r05_scan_bound.err; exit 1; }
    tail -2 gpurun_out/r05_scan_bound.txt
done
