#!/bin/bash
# round-3 session N: bench rehearsals of the distributed multigrid (2 ranks on one device, peer windows)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03n
mkdir -p $O
run_bench () {  # name, nranks, workload, steps
  NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=90 timeout -k 10 800 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port 29661 bench.py --gpus $2 --workload $3 --preconditioner multigrid --steps $4 --warmup 3 > $O/bench_$1.json 2> $O/bench_$1.err
  echo "bench $1 rc=$? $(python3 -c "
import json,sys
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]); print(d['config']['gmres_iterations_per_step'], round(d['ms_per_step'],1), d['config']['all_solved'], d['config']['setup_seconds'], d['config']['preconditioner'][:60])
except Exception as e: print('-', e)
")" | tee -a $O/summary.txt
}
run_bench mg_h005_2rank 2 bowl3D_h0.05 6
run_bench mg_h002_2rank 2 bowl3D_h0.02 6
timeout -k 10 400 python3 bench.py --preconditioner multigrid --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_mg_h002_serial.json 2> $O/bench_mg_h002_serial.err
python3 -c "
import json
d=json.loads(open('$O/bench_mg_h002_serial.json').read().strip().splitlines()[-1]); print('serial', d['config']['gmres_iterations_per_step'], round(d['ms_per_step'],1))" | tee -a $O/summary.txt
cat $O/summary.txt; grep -E "Error|error|Traceback" $O/*.err | head
