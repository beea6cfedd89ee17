#!/bin/bash
# round-3 session V: stability - the distributed tests three times in a row, then a 20-step 3-rank partitioned bench rehearsal
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03v
mkdir -p $O
for i in 1 2 3; do
timeout -k 10 600 python3 -m pytest tests/test_gpu_rccl_selftest.py tests/test_gpu_distributed.py -q -m gpu > $O/pytest_$i.txt 2>&1
echo "pytest round $i rc=$? $(grep -E 'passed|failed' $O/pytest_$i.txt | tail -1)" | tee -a $O/summary.txt
done
NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=60 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29681 bench.py --gpus 3 --workload bowl3D_h0.04 --steps 20 --warmup 5 > $O/bench_3rank.json 2> $O/bench_3rank.err
echo "bench 3-rank 20 steps rc=$? $(python3 -c "
import json
d=json.loads(open('$O/bench_3rank.json').read().strip().splitlines()[-1]); print(d['config']['gmres_iterations_per_step'], round(d['ms_per_step'],1), d['config']['all_solved'], d['comm']['transport_check'])")" | tee -a $O/summary.txt
timeout -k 10 300 python3 bench.py --workload bowl3D_h0.04 --steps 20 --warmup 5 --no-cpu-baseline --no-multigrid > $O/bench_serial_h004.json 2> $O/bench_serial_h004.err
python3 -c "
import json
d=json.loads(open('$O/bench_serial_h004.json').read().strip().splitlines()[-1]); print('serial h0.04', d['config']['gmres_iterations_per_step'], round(d['ms_per_step'],1))" | tee -a $O/summary.txt
