#!/bin/bash
# Multi-rank rehearsals on ONE device (peer windows over hipIpc mappings of the same GPU; gloo as the launcher's backend).
#   tools/dist_rehearsal.sh NAME NRANKS "ENV=V ..." [bench.py arguments]      -> gpurun_out/NAME.json / .err, one summary line
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
name=$1; n=$2; envs=$3; shift 3
O=gpurun_out; mkdir -p $O
port=$((29600 + RANDOM % 300))
t0=$(date +%s)
env $envs NPG_COMM_TRANSPORT=${NPG_COMM_TRANSPORT:-peer} NPG_FORCE_DEVICE=0 NPG_TORCH_BACKEND=gloo NPG_PEER_TIMEOUT_S=${NPG_PEER_TIMEOUT_S:-60} \
  timeout -k 10 ${REH_TIMEOUT:-600} python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
  --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n "$@" > $O/$name.json 2> $O/$name.err
rc=$?
t1=$(date +%s)
echo "$name: $n ranks rc=$rc wall=$((t1 - t0))s blowups=$(grep -c BlowUp $O/$name.err) $(python3 - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    c = d.get('comm') or {}
    print('its', d['config']['gmres_iterations_per_step'][:4], 'ms/step', round(d['ms_per_step'], 1), 'transport', c.get('in_cycle_transport'), c.get('transport_check'), 'setup_s', d['config']['setup_seconds'])
except Exception as e:
    print('no JSON line:', e)
PY
)"
