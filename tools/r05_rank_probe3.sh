#!/bin/bash
# A/B of the round-5 changes to the distributed cycle on rank 4 of 8 (one gpurun call = one box)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out; mkdir -p $O
export NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer NPG_HALO_OVERLAP_VERBOSE=1
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 1000 2>&1 | grep -v "^\[W\|amdgpu.ids\|^bowl3D" || exit 1; }
run NPG_PART_INTERIOR_FIRST=0 NPG_HALO_OVERLAP=0 NPG_HALO_CHUNK=1024 NPG_HALO_WG=32 NPG_HALO_WAIT_WG=16
run NPG_PART_INTERIOR_FIRST=0 NPG_HALO_OVERLAP=0
run NPG_PART_INTERIOR_FIRST=1 NPG_HALO_OVERLAP=0
run NPG_PART_INTERIOR_FIRST=1 NPG_HALO_OVERLAP=1
run NPG_PART_INTERIOR_FIRST=1
tools/prof.sh trace r05_rank_if_tr python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 400 || exit 1
