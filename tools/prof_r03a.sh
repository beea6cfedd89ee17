#!/bin/bash
# round-3 session A: (1) the rocprofv3 hipGraphLaunch fault, discriminated (tools/graph_trace_probe.hip) and symbolised;
# (2) why the GMRES iteration count climbs over a 20-step run: the default kernels beside NPG_GMRES_FAST=0
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
for v in memcpy kernel_small kernel_big both cycle; do
  timeout -k 10 90 tools/graph_probe $v 200 > $O/probe_plain_$v.txt 2>&1
  echo "plain $v rc=$?" | tee -a $O/summary.txt
  timeout -k 10 120 rocprofv3 --kernel-trace -d $O/pt_$v -- tools/graph_probe $v 200 > $O/probe_traced_$v.txt 2>&1
  echo "traced $v rc=$?" | tee -a $O/summary.txt
  rm -rf $O/pt_$v
done
# the library's own cycle graphs under the tracer on the smallest mesh, with the backtrace handler
export NPG_SEGV_BACKTRACE=1
export NPG_GMRES_EAGER=0
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/pt_lib -- python3 bench.py --workload bowl3D_h0.1 --steps 2 --warmup 1 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/lib_traced_h0.1.out 2> $O/lib_traced_h0.1.err
echo "library graphs traced (h0.1) rc=$?" | tee -a $O/summary.txt
rm -rf $O/pt_lib
unset NPG_GMRES_EAGER NPG_SEGV_BACKTRACE
# iteration climb
export NPG_GMRES_TRACE=1
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/climb_default.json 2> $O/climb_default.err
echo "climb default rc=$?" | tee -a $O/summary.txt
NPG_GMRES_FAST=0 timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-multigrid --no-profile-pass > $O/climb_fast0.json 2> $O/climb_fast0.err
echo "climb FAST=0 rc=$?" | tee -a $O/summary.txt
cat $O/summary.txt
