#!/bin/bash
# bench.py + rocprofv3 kernel stats of bench.py (no test suite).  usage: gpu_bench.sh <tag>
set -u
tag=${1:-b}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_$tag.log 2> $O/bench_$tag.err; echo "bench rc=$?"; tail -3 $O/bench_$tag.err; tail -1 $O/bench_$tag.log | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$tag -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$tag.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof_$tag -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" 80 > $O/kernel_stats_$tag.csv && head -50 $O/kernel_stats_$tag.csv | cut -c1-200
rm -rf $O/prof_$tag
