#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "kernel_z or full_minibatch or trunk or fcz or 4GiB") > $O/pytest_z.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_z.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_blds.json 2> $O/bench_C_blds.err; echo "bench rc=$?"; cut -c1-600 $O/bench_C_blds.json
