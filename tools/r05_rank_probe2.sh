#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out; mkdir -p $O
export NPG_COMM_SELFTEST=1 NPG_COMM_TRANSPORT=peer NPG_HALO_OVERLAP_VERBOSE=1
for ov in 1 0; do
  echo "== NPG_HALO_OVERLAP=$ov"
  NPG_HALO_OVERLAP=$ov timeout -k 10 300 python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 1000 2>&1 | grep -v "^\[W\|amdgpu.ids" || exit 1
done
NPG_HALO_OVERLAP=0 tools/prof.sh trace r05_rank_ov0_tr python3 tools/rank_cycle_probe.py bowl3D_h0.02 8 4 400 || exit 1
