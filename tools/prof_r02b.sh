#!/bin/bash
# round-2 profiling, second session: (1) the bench command itself under the kernel tracer with NO variable set (a traced
# process launches eagerly by construction), (2)+(3) FETCH_SIZE / WRITE_SIZE of the GMRES kernels on the current bench mesh,
# separate --pmc passes with --kernel-trace only
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/prof_r02b
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/K -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass > $O/K.out 2> $O/K.err
echo "bench under tracer rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/F -- python3 tools/pmc_probe.py bowl3D_h0.02 2 > $O/F.out 2> $O/F.err
echo "FETCH rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/W -- python3 tools/pmc_probe.py bowl3D_h0.02 2 > $O/W.out 2> $O/W.err
echo "WRITE rc=$?" | tee -a $O/summary.txt
f=$(find $O/K -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/K_kernel_stats.csv
python3 tools/pmc_summary.py $O/F $O/W > $O/pmc_summary.txt 2>&1
rm -rf $O/K $O/F $O/W
cat $O/summary.txt; tail -2 $O/K.out | cut -c1-300; cat $O/pmc_summary.txt | grep -E "arnoldi|k_spmv|residual" ; tail -1 $O/F.out
