#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_channel_basin.py -x -q -k "weak_form" 2>&1 | tail -5 || exit 1
echo "== dist tests, fused all-reduce forced on a shared device"
NPG_AR_FUSED=1 timeout -k 10 800 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_rccl_selftest.py -x -q 2>&1 | tail -8 || exit 1
export NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 1000 2>&1 | grep -v "^\[W\|amdgpu.ids\|^bowl3D" || exit 1; }
run NPG_AR_FUSED=0
run NPG_AR_FUSED=1
NPG_AR_FUSED=1 tools/prof.sh trace r05_rank_fused_tr python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 400 || exit 1
