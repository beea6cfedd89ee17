#!/bin/bash
# One parameterised profiling script (replaces the per-session tools/prof_r0*.sh; the command lines that produced each file under
# profiles/ are recorded in profiles/README.md).  Run on the GPU box through gpurun, from the repository root.
#
#   tools/prof.sh trace NAME  python3 <script> [args]     rocprofv3 --kernel-trace --stats  -> gpurun_out/NAME_kernel_stats.csv
#   tools/prof.sh pmc   NAME  python3 <script> [args]     FETCH_SIZE and WRITE_SIZE in SEPARATE passes (never with other trace
#                                                         domains) -> gpurun_out/NAME_pmc_summary.txt (tools/pmc_summary.py)
#   tools/prof.sh pmcs  NAME "C1 C2 ..." python3 <script> one pass per listed counter -> gpurun_out/NAME_<counter>.csv
# The program itself follows directly (python3 ..., never env / bash -c: the profiler initialises the GPU before the program starts).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mode=$1; name=$2; shift 2
O=gpurun_out
mkdir -p $O
case "$mode" in
trace)
    timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $O/_$name -- "$@" > $O/$name.out 2> $O/$name.err
    echo "trace $name rc=$?"
    f=$(find $O/_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${name}_kernel_stats.csv
    rm -rf $O/_$name
    head -12 $O/${name}_kernel_stats.csv
    ;;
pmc)
    timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/_${name}_F -- "$@" > $O/${name}_F.out 2> $O/${name}_F.err
    echo "FETCH_SIZE pass rc=$?"
    timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/_${name}_W -- "$@" > $O/${name}_W.out 2> $O/${name}_W.err
    echo "WRITE_SIZE pass rc=$?"
    python3 tools/pmc_summary.py $O/_${name}_F $O/_${name}_W > $O/${name}_pmc_summary.txt 2>&1
    rm -rf $O/_${name}_F $O/_${name}_W
    cat $O/${name}_pmc_summary.txt
    ;;
pmcs)
    counters=$1; shift
    for c in $counters; do
        timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/_${name}_$c -- "$@" > $O/${name}_$c.out 2> $O/${name}_$c.err
        echo "$c pass rc=$?"
        f=$(find $O/_${name}_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${name}_$c.csv
        rm -rf $O/_${name}_$c
    done
    ;;
*) echo "usage: tools/prof.sh trace|pmc|pmcs NAME ... python3 <script> [args]"; exit 2 ;;
esac
