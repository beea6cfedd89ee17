#!/bin/bash
# round-2 profiling, third session: kernel statistics of the multigrid-preconditioned bench in its final form (SpMV epilogue
# fusion, dense fp32 inverse on the coarsest level, extrapolated initial guess); the traced process launches eagerly
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof_r02c
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/M -- python3 bench.py --preconditioner multigrid --steps 10 --warmup 3 --no-cpu-baseline > $O/M.out 2> $O/M.err
echo "multigrid bench under tracer rc=$?" | tee -a $O/summary.txt
f=$(find $O/M -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/M_kernel_stats.csv
rm -rf $O/M
head -14 $O/M_kernel_stats.csv | cut -c1-150
tail -c 600 $O/M.out
