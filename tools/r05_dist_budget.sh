#!/bin/bash
# Round 5, VERDICT item 2: what one inner iteration costs on a RANK-SIZED block (bowl3D h = 0.04, 263 k rows = an eighth of the bench
# system) - serial cycle and the distributed code path on a one-rank communicator - per launch (rocprofv3, eager) and per iteration
# (graph replay, wall clock).  One gpurun call: every number from the same box.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out; mkdir -p $O
B="bench.py --workload bowl3D_h0.04 --steps 3 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass"
sum() { python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
its = d["config"]["gmres_iterations_per_step"]
print(f"{sys.argv[1]}: ms/step {d['ms_per_step']:.2f} its {its} -> {1e3 * d['ms_per_step'] / (sum(its) / len(its)):.2f} us per inner iteration (whole timestep / iterations)")
PY
}
timeout -k 10 300 python3 $B > $O/r05_h04_serial.json 2> $O/r05_h04_serial.err && sum $O/r05_h04_serial.json || exit 1
NPG_BENCH_FORCE_DIST=1 NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer timeout -k 10 300 python3 $B > $O/r05_h04_dist1.json 2> $O/r05_h04_dist1.err && sum $O/r05_h04_dist1.json || exit 1
tools/prof.sh trace r05_h04_serial_tr python3 $B || exit 1
NPG_BENCH_FORCE_DIST=1 NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer tools/prof.sh trace r05_h04_dist1_tr python3 $B || exit 1
for tr in peer rccl; do
  NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=$tr CYCLE_BASIS=32 timeout -k 10 200 python3 tools/rccl_cycle_cost.py 263000 > $O/r05_cycle_cost_$tr.txt 2>&1 || exit 1
  cat $O/r05_cycle_cost_$tr.txt
done
# the all-fp64 Arnoldi instance's HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes
PMC_FP64=1 tools/prof.sh pmc r05_pmc_fp64 python3 tools/pmc_probe.py bowl3D_h0.02 2
