#!/bin/bash
# round 5: the ADVICE-low changes' GPU tests, then the channel-basin multigrid bench with the two coarse-viscosity rules
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_ilu0.py tests/test_gpu_precond.py tests/test_gpu_channel_basin.py -m gpu -x -q > gpurun_out/r05_lows_tests.log 2>&1 || { tail -30 gpurun_out/r05_lows_tests.log; exit 1; }
tail -3 gpurun_out/r05_lows_tests.log
grep -h "npg mg" gpurun_out/r05_lows_tests.log | sort | uniq -c
CB_STEPS=25 tools/cb_ab.sh r05_coarse_nu "NPG_MG_COARSE_NU=inject" "NPG_MG_COARSE_NU=average"
grep -h "npg mg" gpurun_out/r05_coarse_nu_*.err | sort | uniq -c
exit 0
