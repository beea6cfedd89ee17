#!/bin/bash
# round-3 measurements on the final code (gather-layout input, column records): driver-shaped default bench, kernel trace of the bench command,
# 2-rank / 4-rank partitioned rehearsals (peer windows, one device), 2-rank distributed multigrid rehearsal
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ah
mkdir -p $O
timeout -k 10 700 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "default bench rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/K -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/K.out 2> $O/K.err
echo "bench under tracer rc=$?" | tee -a $O/summary.txt
f=$(find $O/K -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/K_kernel_stats.csv
rm -rf $O/K
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/F -- python3 tools/pmc_probe.py bowl3D_h0.02 2 > $O/F.out 2> $O/F.err
echo "FETCH rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/W -- python3 tools/pmc_probe.py bowl3D_h0.02 2 > $O/W.out 2> $O/W.err
echo "WRITE rc=$?" | tee -a $O/summary.txt
python3 tools/pmc_summary.py $O/F $O/W > $O/pmc_summary.txt 2>&1
rm -rf $O/F $O/W
grep -E "arnoldi|k_spmv|residual" $O/pmc_summary.txt; tail -1 $O/F.out
export NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=60
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29701 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err
echo "2-rank rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29702 bench.py --gpus 4 --workload bowl3D_h0.04 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_4rank.json 2> $O/bench_4rank.err
echo "4-rank rc=$?" | tee -a $O/summary.txt
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29703 bench.py --gpus 2 --preconditioner multigrid --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_2rank_mg.json 2> $O/bench_2rank_mg.err
echo "2-rank multigrid rc=$?" | tee -a $O/summary.txt
python3 - <<'PY' | tee -a gpurun_out/r03ah/summary.txt
import json
for f in ("bench_default","bench_2rank","bench_4rank","bench_2rank_mg"):
    try:
        d=json.loads(open(f"gpurun_out/r03ah/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), d["config"]["gmres_iterations_per_step"][:4], d["config"]["all_solved"], (d.get("comm") or {}).get("transport_check"), (d.get("multigrid") or {}).get("ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
