#!/bin/bash
# Generic GPU-box visit: full GPU test suite, bench.py, rocprofv3 kernel stats of bench.py.  usage: tools/gpu/visit.sh <tag> [pytest-args...]
set -u
tag=${1:-visit}; shift || true
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -x -rfEs -p no:cacheprovider "$@" ) > $O/pytest_$tag.log 2>&1
echo "pytest rc=$?"; tail -12 $O/pytest_$tag.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_$tag.log 2> $O/bench_$tag.err; echo "bench rc=$?"; tail -1 $O/bench_$tag.log | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$tag -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$tag.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof_$tag -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" 80 > $O/kernel_stats_$tag.csv && head -45 $O/kernel_stats_$tag.csv | cut -c1-230
rm -rf $O/prof_$tag
