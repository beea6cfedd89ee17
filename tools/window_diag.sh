#!/bin/bash
# timing diagnostics of the windowed tile set (NPG_WIN_DIAG, csrc/spmv_window.h): bit 1 = no gathers, 2 = no segmented sums, 4 = no record
# loads, 8 = the rows behind the block rows skipped, 16 = the block tiles skipped, 64 = staggered start (combinations as instantiated
# in csr.hip).  Usage: tools/window_diag.sh "8 16" [ENV=V,...]
for d in ${1:-0 1 2 3 4 7 8 16}; do echo "== NPG_WIN_DIAG=$d"; NPG_WIN_DIAG=$d timeout -k 10 200 python tools/window_ab.py bowl3D_h0.02 200 $2 2>&1 | grep "windowed"; done
