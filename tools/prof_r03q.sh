#!/bin/bash
# round-3 session Q: in-place halo consumption by the Arnoldi kernel - tests, cycle cost, bench rehearsal
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03q
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_rccl_selftest.py tests/test_gpu_distributed.py -q -m gpu > $O/pytest_dist.txt 2>&1
echo "pytest dist rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED|^E  " $O/pytest_dist.txt | head -20
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer timeout -k 10 300 python3 tools/rccl_cycle_cost.py 270000 > $O/cc_peer_direct.txt 2>&1
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer NPG_HALO_DIRECT=0 timeout -k 10 300 python3 tools/rccl_cycle_cost.py 270000 > $O/cc_peer_unpack.txt 2>&1
grep -H iteration $O/cc_*.txt | tee -a $O/summary.txt
NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=60 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --workload bowl3D_h0.02 --steps 2 --warmup 1 > $O/bench_h002_2rank.json 2> $O/bench_h002_2rank.err
echo "bench rc=$? $(python3 -c "
import json
d=json.loads(open('$O/bench_h002_2rank.json').read().strip().splitlines()[-1]); print(d['config']['gmres_iterations_per_step'], round(d['ms_per_step'],1), d['comm']['transport_check'])")" | tee -a $O/summary.txt
