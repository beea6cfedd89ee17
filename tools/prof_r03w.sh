#!/bin/bash
# round-3 session W: run-to-run determinism of the distributed channel-basin rehearsals (3 ranks, peer transport)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03w
mkdir -p $O
for mode in channel pchannel blocks part; do
for i in 1 2 3; do
NPG_COMM_TRANSPORT=peer NPG_FORCE_DEVICE=0 NPG_PEER_TIMEOUT_S=60 OMP_NUM_THREADS=2 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29691 tests/dist_rehearsal_worker.py $O/${mode}_$i 11 $mode > $O/${mode}_$i.log 2>&1
done
python3 - <<PY | tee -a $O/summary.txt
import numpy as np
a=[np.load("$O/${mode}_%d.rank0.npz"%i) for i in (1,2,3)]
print("$mode", "u identical:", [bool(np.array_equal(a[0]["u"], x["u"])) for x in a[1:]], "gm", [list(x["gm"][:4]) for x in a], "max rel diff", [float(np.linalg.norm(a[0]["u"]-x["u"])/np.linalg.norm(a[0]["u"])) for x in a[1:]])
PY
done
