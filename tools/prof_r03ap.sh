#!/bin/bash
# round-3 session AP: full node records (function-valued viscosity) - parity test, channel-basin A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03ap
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "full_node_records" > $O/pytest.txt 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' $O/pytest.txt | tail -1)" | tee -a $O/summary.txt
grep -E "^E  " $O/pytest.txt | head -8 | tee -a $O/summary.txt
for g in 1 0; do
NPG_PACK_NODES=$g timeout -k 10 500 python3 bench.py --workload channel_basin_h0.01 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cb_p$g.json 2> $O/bench_cb_p$g.err
python3 -c "
import json
d=json.loads(open('$O/bench_cb_p$g.json').read().strip().splitlines()[-1]); r=d['roofline']; print('pack=$g', round(d['ms_per_step'],1), d['config']['gmres_iterations_per_step'], round(r['avg_launch_us'],1), r['stored_bytes_per_launch'], d['config']['setup_seconds'])" | tee -a $O/summary.txt
tail -2 $O/bench_cb_p$g.err | cut -c1-300
done
