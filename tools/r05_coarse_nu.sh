#!/bin/bash
# round 5: coefficient restriction on the device - its test, then the channel-basin multigrid bench with both coarse-viscosity rules (one box)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_precond.py -m gpu -x -q -k "coarse_viscosity or closure or zline" > gpurun_out/r05_coarse_nu_tests.log 2>&1 || { tail -30 gpurun_out/r05_coarse_nu_tests.log; exit 1; }
tail -2 gpurun_out/r05_coarse_nu_tests.log
tools/cb_ab.sh r05_coarse_nu_dev "NPG_MG_COARSE_NU=inject" "NPG_MG_COARSE_NU=average" "NPG_MG_COARSE_NU=inject" "NPG_MG_COARSE_NU=average"
