#!/bin/bash
# round-2 profiling session: (A) multigrid-preconditioned bench under the kernel tracer, (B) the headline bench with eager
# launches under the tracer, (C) the headline bench with hipGraph relaunches under the tracer (the run that died in round 1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof_r02
mkdir -p $O
env | grep -i -E "rocp|ld_preload" > $O/env_outside.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/A -- python3 bench.py --preconditioner multigrid --steps 3 --warmup 1 --no-cpu-baseline > $O/A.out 2> $O/A.err
echo "A rc=$?" | tee -a $O/summary.txt
export NPG_GMRES_EAGER=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/B -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass > $O/B.out 2> $O/B.err
echo "B (eager) rc=$?" | tee -a $O/summary.txt
unset NPG_GMRES_EAGER
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/C -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass > $O/C.out 2> $O/C.err
echo "C (graph) rc=$?" | tee -a $O/summary.txt
for d in A B C; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/A $O/B $O/C
tail -3 $O/*.err
