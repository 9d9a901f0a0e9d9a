#!/bin/bash
# Round 2, first GPU-box visit: (1) full GPU test suite incl. the opt-in new-script tests and the 2-process multirank tests,
# (2) bench.py (the number), (3) rocprofv3 --kernel-trace --stats of bench.py, (4) rocprofv3 kernel trace of the K1/K3 sweeps.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -rfEs --durations=25 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -45 $O/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-600
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
tail -1 $O/prof_bench.log | cut -c1-300
db=$(find $O/prof_bench -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" 80 > $O/r02_bench_n1_kernel_stats.csv && head -30 $O/r02_bench_n1_kernel_stats.csv
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_sweep -o sweep -- python $GRAFT_REPO_ROOT/tools/sweep_k1k3.py 20 > $GRAFT_REPO_ROOT/$O/r02_sweep_k1k3_events.jsonl 2> $GRAFT_REPO_ROOT/$O/sweep.err; echo "sweep rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof_sweep -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_dump.py "$db" > $O/r02_sweep_k1k3_kernel_trace.csv && wc -l $O/r02_sweep_k1k3_kernel_trace.csv
cat $O/r02_sweep_k1k3_events.jsonl
rm -rf $O/prof_bench/*/*.db $O/prof_sweep/*/*.db 2>/dev/null
ls $O
