#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out; mkdir -p $O
echo "== dist tests (ghost nodes on by default)"
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_rccl_selftest.py -x -q 2>&1 | tail -8 || exit 1
export NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer NPG_HALO_OVERLAP_VERBOSE=1
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 1000 > $O/_probe.txt 2>&1; rc=$?; grep -v "^\[W\|amdgpu.ids" $O/_probe.txt; [ $rc -eq 0 ] || exit $rc; }
run NPG_GHOST_NODES=0
run NPG_GHOST_NODES=1
NPG_GHOST_NODES=1 tools/prof.sh trace r05_rank_gn_tr python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 400 || exit 1
