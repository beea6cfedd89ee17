#!/bin/bash
# round 5: the software-pipelined windowed tiles (records of tile t+1 requested behind the products of tile t): correctness, then K1 A/B on one box
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_julia_mirror.py -m gpu -x -q -k "window or gather or full_size or ghost" > gpurun_out/r05_pipe_tests.log 2>&1 || { tail -30 gpurun_out/r05_pipe_tests.log; exit 1; }
tail -2 gpurun_out/r05_pipe_tests.log
tools/k1_ab.sh r05_pipe "$@" || exit 1
