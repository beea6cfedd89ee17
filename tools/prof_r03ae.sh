#!/bin/bash
# round-3 session AE: gather-layout input also in distributed runs - distributed tests, 2-rank rehearsal A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ae
mkdir -p $O
timeout -k 10 700 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_rccl_selftest.py -q -m gpu -x > $O/pytest_dist.txt 2>&1
echo "pytest dist rc=$? $(grep -E 'passed|failed' $O/pytest_dist.txt | tail -1)" | tee -a $O/summary.txt
export NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=60
for g in 1 0; do
NPG_GMRES_XG=$g timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2971$g bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2rank_g$g.json 2> $O/bench_2rank_g$g.err
python3 -c "
import json
d=json.loads(open('$O/bench_2rank_g$g.json').read().strip().splitlines()[-1]); print('2 ranks gather=$g K1', round(d['roofline']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],1), 'its', d['config']['gmres_iterations_per_step'], d['config']['all_solved'])" | tee -a $O/summary.txt
done
