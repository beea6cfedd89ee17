#!/bin/bash
# round 5: the strong-scaling budget at S3b (bowl3D h = 0.0125, 8.97 M unknowns): one GPU's serial cycle and rank 4 of 8's distributed cycle, one box
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python3 bench.py --workload bowl3D_h0.0125 --steps 1 --warmup 1 --no-multigrid --no-cpu-baseline > gpurun_out/r05_s3b_serial.json 2> gpurun_out/r05_s3b_serial.err || { echo "serial bench failed"; tail -5 gpurun_out/r05_s3b_serial.err; exit 1; }
python3 tools/bench_ab_summary.py gpurun_out/r05_s3b_serial.json
NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer timeout -k 10 500 python3 tools/rank_cycle_probe.py bowl3D_h0.0125 8 4 ${S3B_ITS:-600} > gpurun_out/r05_s3b_rank.txt 2> gpurun_out/r05_s3b_rank.err || { echo "rank probe failed"; tail -5 gpurun_out/r05_s3b_rank.err; exit 1; }
cat gpurun_out/r05_s3b_rank.txt
