#!/bin/bash
# round-3 measurements on the code with the coupling records: the driver-shaped default bench; the same command under the
# kernel tracer (eager by construction); FETCH_SIZE / WRITE_SIZE of the Krylov kernels (separate --pmc passes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03aa
mkdir -p $O
timeout -k 10 700 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "default bench rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/K -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/K.out 2> $O/K.err
echo "bench under tracer rc=$?" | tee -a $O/summary.txt
f=$(find $O/K -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/K_kernel_stats.csv
rm -rf $O/K
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/F -- python3 tools/pmc_probe.py bowl3D_h0.02 2 > $O/F.out 2> $O/F.err
echo "FETCH rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/W -- python3 tools/pmc_probe.py bowl3D_h0.02 2 > $O/W.out 2> $O/W.err
echo "WRITE rc=$?" | tee -a $O/summary.txt
python3 tools/pmc_summary.py $O/F $O/W > $O/pmc_summary.txt 2>&1
rm -rf $O/F $O/W
cat $O/summary.txt; grep -E "arnoldi|k_spmv|residual" $O/pmc_summary.txt; tail -1 $O/F.out
