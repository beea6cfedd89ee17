#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=15 > $O/r05_gputests.log 2>&1; rc=$?
tail -30 $O/r05_gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $O/r05_bench_short.json 2> $O/r05_bench_short.err || { tail -20 $O/r05_bench_short.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_short.json').read().strip().splitlines()[-1])
for k in ('value', 'ms_per_step'): print(k, d[k])
print('roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'traffic', 'bytes_priced', 'avg_launch_us')})
print('spmv_standalone', d['spmv_standalone'])
print('fp64_basis', {k: v for k, v in d['fp64_basis'].items() if k != 'what'})
print('multigrid', {k: v for k, v in d['multigrid'].items() if k not in ('what',)})
print('cpu_baseline', d.get('cpu_baseline'))
PY
