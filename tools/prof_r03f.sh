#!/bin/bash
# round-3 session F: bisecting the overlapped halo exchange on the peer transport (2 ranks, bowl3D h = 0.04, partitioned model)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
run_bench () {  # name, extra env
  env $2 NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=30 timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 2 --workload bowl3D_h0.04 --steps 1 --warmup 1 --no-profile-pass > $O/bench_$1.json 2> $O/bench_$1.err
  echo "bench 2-rank $1 rc=$? $(grep -c BlowUp $O/bench_$1.err) $(python3 -c "
import json,sys
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]); print(d['config']['gmres_iterations_per_step'], round(d['ms_per_step'],1))
except Exception as e: print('-')
")" | tee -a $O/summary.txt
}
run_bench dbg1_hostsync_eager "NPG_DIST_GRAPH=0 NPG_HALO_DEBUG=1"
run_bench dbg3_exchange_then_interior_eager "NPG_DIST_GRAPH=0 NPG_HALO_DEBUG=3"
run_bench dbg2_boundary_redoes_all_eager "NPG_DIST_GRAPH=0 NPG_HALO_DEBUG=2"
run_bench dbg2_boundary_redoes_all_graph "NPG_HALO_DEBUG=2"
run_bench reserve0_eager "NPG_DIST_GRAPH=0 NPG_HALO_RESERVE_CUS=0"
run_bench noblocks_eager "NPG_DIST_GRAPH=0 NPG_BLOCK_NODES=0"
cat $O/summary.txt
