#!/bin/bash
# round-3 session G: the single-stream overlapped exchange - tests, then bench rehearsals of the partitioned model
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_rccl_selftest.py tests/test_gpu_distributed.py -q -m gpu > $O/pytest_dist.txt 2>&1
echo "pytest dist rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|^FAILED|^E  " $O/pytest_dist.txt | head -20
run_bench () {  # name, nranks, workload, extra env
  env $4 NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=60 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port 29651 bench.py --gpus $2 --workload $3 --steps 2 --warmup 1 > $O/bench_$1.json 2> $O/bench_$1.err
  echo "bench $1 rc=$? $(grep -c BlowUp $O/bench_$1.err) $(python3 -c "
import json,sys
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]); print(d['config']['gmres_iterations_per_step'], round(d['ms_per_step'],1), d['comm']['transport_check'])
except Exception as e: print('-')
")" | tee -a $O/summary.txt
}
run_bench h004_2rank 2 bowl3D_h0.04 ""
run_bench h004_2rank_nooverlap 2 bowl3D_h0.04 "NPG_HALO_OVERLAP=0"
run_bench h004_4rank 4 bowl3D_h0.04 ""
run_bench h002_2rank 2 bowl3D_h0.02 ""
cat $O/summary.txt
