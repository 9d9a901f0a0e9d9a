#!/bin/bash
# K3 (fused loss) check: parity tests, then the K1/K3 sweeps under rocprofv3 --kernel-trace (kernel durations, not event brackets).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "loss or capturable" -p no:cacheprovider ) > $O/pytest_k3.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest_k3.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_sweep -o sweep -- python $GRAFT_REPO_ROOT/tools/sweep_k1k3.py 20 > $GRAFT_REPO_ROOT/$O/r02_sweep_k1k3_events.jsonl 2> $GRAFT_REPO_ROOT/$O/sweep.err; echo "sweep rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof_sweep -name '*.db' | head -1); [ -n "$db" ] && python tools/rocpd_dump.py "$db" > $O/r02_sweep_k1k3_kernel_trace.csv && wc -l $O/r02_sweep_k1k3_kernel_trace.csv
grep loss $O/r02_sweep_k1k3_events.jsonl
rm -rf $O/prof_sweep
