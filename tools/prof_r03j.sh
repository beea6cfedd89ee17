#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
for b in 32 64; do
NPG_GMRES_BASIS=$b NPG_GMRES_TRACE=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/trace$b.json 2> $O/trace$b.err
echo "basis $b: $(grep 'npg gmres' $O/trace$b.err | tail -1)" | tee -a $O/summary.txt
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/traced.out 2> $O/traced.err
f=$(find $O/pt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_basis32.csv
rm -rf $O/pt
head -9 $O/kernel_stats_basis32.csv | cut -c1-120
