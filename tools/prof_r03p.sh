#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt -- python3 tools/rccl_cycle_cost.py 270000 > $O/cc.txt 2> $O/cc.err
f=$(find $O/pt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
rm -rf $O/pt
head -14 $O/kernel_stats.csv | cut -c1-140
