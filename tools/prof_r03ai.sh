#!/bin/bash
# round-3 session AI: where a Krylov iteration's time goes at ONE RANK's share of the rows (270 k): kernel trace of the cycle-cost probe
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ai
mkdir -p $O
export NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer CYCLE_BASIS=32
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/K -- python3 tools/rccl_cycle_cost.py 270000 > $O/K.out 2> $O/K.err
f=$(find $O/K -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/K_kernel_stats.csv
rm -rf $O/K
cat $O/K.out
head -16 $O/K_kernel_stats.csv | cut -d, -f1-4
