#!/bin/bash
# round-3 session AL: kernel trace of the multigrid-preconditioned bench on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03al
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/K -- python3 bench.py --preconditioner multigrid --steps 3 --warmup 1 --no-cpu-baseline > $O/K.out 2> $O/K.err
echo "rc=$?" | tee -a $O/summary.txt
f=$(find $O/K -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/K_kernel_stats.csv
rm -rf $O/K
tail -1 $O/K.out | cut -c1-300
