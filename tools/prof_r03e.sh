#!/bin/bash
# round-3 session E: debugging the peer transport (stress) and the partitioned model (tests + bisecting bench rehearsals)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
for tr in peer shm; do
NPG_COMM_TRANSPORT=$tr NPG_FORCE_DEVICE=0 NPG_PEER_TIMEOUT_S=30 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29621 tests/peer_stress_worker.py 600 > $O/stress_$tr.txt 2>&1
echo "stress $tr rc=$?" | tee -a $O/summary.txt
grep -E "rank [0-9]:|STRESS|Error|error" $O/stress_$tr.txt | head -8
done
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py -q -m gpu -k "partitioned or same_bits" > $O/pytest_part.txt 2>&1
echo "pytest partitioned rc=$?" | tee -a $O/summary.txt
tail -30 $O/pytest_part.txt
run_bench () {  # name, extra env
  env $2 NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=30 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --workload bowl3D_h0.04 --steps 2 --warmup 1 > $O/bench_$1.json 2> $O/bench_$1.err
  echo "bench 2-rank $1 rc=$?" | tee -a $O/summary.txt
  grep -E "BlowUp|Error" $O/bench_$1.err | head -3
}
run_bench peer_default "NPG_COMM_TRANSPORT=peer"
run_bench peer_nooverlap "NPG_COMM_TRANSPORT=peer NPG_HALO_OVERLAP=0"
run_bench peer_nograph "NPG_COMM_TRANSPORT=peer NPG_DIST_GRAPH=0"
run_bench shm "NPG_COMM_TRANSPORT=shm"
run_bench shm_noblocks "NPG_COMM_TRANSPORT=shm NPG_BLOCK_NODES=0"
cat $O/summary.txt
