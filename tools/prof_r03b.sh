#!/bin/bash
# round-3 session B: the rocprofv3 + hipGraph fault at the size where it showed (bowl3D h = 0.02), with the library's backtrace
# handler (NPG_SEGV_BACKTRACE=1: modules + offsets + the mapping at the faulting address); then the peer-transport tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
export NPG_SEGV_BACKTRACE=1
export NPG_GMRES_EAGER=0
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/pt_lib -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/lib_traced_h0.02.out 2> $O/lib_traced_h0.02.err
echo "library graphs traced (h0.02) rc=$?" | tee -a $O/summary.txt
rm -rf $O/pt_lib
unset NPG_GMRES_EAGER NPG_SEGV_BACKTRACE
timeout -k 10 900 python3 -m pytest tests/test_gpu_rccl_selftest.py tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest_dist.txt 2>&1
echo "pytest dist rc=$?" | tee -a $O/summary.txt
tail -30 $O/pytest_dist.txt
