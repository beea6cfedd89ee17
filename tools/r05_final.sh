#!/bin/bash
# round 5: the round's final artefacts on one box - smoke(), the default bench line, the rocprofv3 kernel stats of the bench command
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/r05_smoke.log 2>&1 || { tail -20 $O/r05_smoke.log; exit 1; }
tail -2 $O/r05_smoke.log
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $O/r05_bench_default.json 2> $O/r05_bench_default.err || { tail -20 $O/r05_bench_default.err; exit 1; }
python3 tools/bench_ab_summary.py $O/r05_bench_default.json
tools/prof.sh trace r05_bench python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-multigrid
