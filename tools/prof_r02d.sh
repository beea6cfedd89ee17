#!/bin/bash
# round-2 profiling, final session: the default bench command under the kernel tracer (a traced process launches the
# Krylov cycles eagerly by construction, profiles/README.md)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof_r02d
mkdir -p $O
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/K -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --no-multigrid > $O/K.out 2> $O/K.err
echo "bench under tracer rc=$?" | tee -a $O/summary.txt
f=$(find $O/K -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/K_kernel_stats.csv
rm -rf $O/K
head -8 $O/K_kernel_stats.csv | cut -c1-160; tail -1 $O/K.out | cut -c1-400
