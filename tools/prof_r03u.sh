#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03u
mkdir -p $O
NPG_GMRES_TRACE=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid > $O/trace.json 2> $O/trace.err
grep "npg gmres" $O/trace.err | tail -1 | tee -a $O/summary.txt
python3 -c "
import json
d=json.loads(open('$O/trace.json').read().strip().splitlines()[-1]); print('K1', round(d['roofline']['avg_launch_us'],1), 'spmv', round(d['spmv_standalone']['avg_launch_us'],1), 'plain', round(d['spmv_plain_csr']['avg_launch_us'],1), 'ms', round(d['ms_per_step'],1))" | tee -a $O/summary.txt
