#!/bin/bash
# round-3 session D: (1) partitioned-model + transport tests; (2) library-free reproducer of the rocprofv3 graph fault
# (graph launches whose AQL packets wrap the queue ring); (3) bench rehearsals of the mesh-partitioned model, 2 and 4 ranks on
# one device through the peer windows; (4) kernel statistics of the default bench (eager under the tracer) with and without
# the reversed orthogonalisation sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_rccl_selftest.py tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest_dist.txt 2>&1
echo "pytest dist rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_dist.txt
for nk in 62 61 37; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt_probe_$nk -- tools/graph_probe cycle 3000 $nk > $O/probe_wrap_$nk.txt 2>&1
  echo "probe cycle 3000 relaunches x $nk kernels traced rc=$?" | tee -a $O/summary.txt
  rm -rf $O/pt_probe_$nk
done
timeout -k 10 60 tools/graph_probe cycle 3000 61 > $O/probe_wrap_61_plain.txt 2>&1
echo "probe cycle 3000 x 61 untraced rc=$?" | tee -a $O/summary.txt
for n in 2 4; do
NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --workload bowl3D_h0.04 --steps 2 --warmup 1 > $O/bench_${n}rank_part.json 2> $O/bench_${n}rank_part.err
echo "$n-rank partitioned bench rehearsal rc=$?" | tee -a $O/summary.txt
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt_def -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/def_traced.out 2> $O/def_traced.err
echo "default traced rc=$?" | tee -a $O/summary.txt
f=$(find $O/pt_def -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/def_kernel_stats.csv
rm -rf $O/pt_def
NPG_ORTH_REVERSE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt_rev -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/rev_traced.out 2> $O/rev_traced.err
echo "reverse-sweep traced rc=$?" | tee -a $O/summary.txt
f=$(find $O/pt_rev -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rev_kernel_stats.csv
rm -rf $O/pt_rev
cat $O/summary.txt
