#!/bin/bash
# round-3 session C: (1) the default bench under rocprofv3 --kernel-trace --stats with hipGraph relaunches (the command line
# that died in round 2) + backtrace handler; (2) the same with the orthogonalisation sweep reversed (Infinity Cache reuse of
# the basis); (3) per-iteration cost of the distributed cycle: serial vs RCCL vs peer windows (one-rank self-test);
# (4) two-rank bench rehearsal on one device through the peer windows
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
export NPG_SEGV_BACKTRACE=1
export NPG_GMRES_EAGER=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt_graph -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/graph_traced.out 2> $O/graph_traced.err
echo "graph traced --stats csv rc=$?" | tee -a $O/summary.txt
f=$(find $O/pt_graph -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/graph_kernel_stats.csv
rm -rf $O/pt_graph
NPG_ORTH_REVERSE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt_rev -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/rev_traced.out 2> $O/rev_traced.err
echo "reverse-sweep traced rc=$?" | tee -a $O/summary.txt
f=$(find $O/pt_rev -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rev_kernel_stats.csv
rm -rf $O/pt_rev
unset NPG_GMRES_EAGER NPG_SEGV_BACKTRACE
for n in 20000 100000; do
NPG_COMM_SELFTEST=1 timeout -k 10 200 python3 tools/rccl_cycle_cost.py $n > $O/cycle_cost_rccl_$n.txt 2>&1
echo "cycle cost rccl $n rc=$?" | tee -a $O/summary.txt
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer timeout -k 10 200 python3 tools/rccl_cycle_cost.py $n > $O/cycle_cost_peer_$n.txt 2>&1
echo "cycle cost peer $n rc=$?" | tee -a $O/summary.txt
done
NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload bowl3D_h0.04 --steps 2 --warmup 1 > $O/bench_2rank_peer.json 2> $O/bench_2rank_peer.err
echo "2-rank bench rehearsal (peer) rc=$?" | tee -a $O/summary.txt
cat $O/summary.txt; cat $O/cycle_cost_*.txt | grep iteration
