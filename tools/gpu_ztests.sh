#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "bits or trunk or conv1q or full_minibatch") > $O/pytest_z.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_z.log | cut -c1-300
(timeout 900 python -m pytest tests/test_gpu_learner.py -m gpu -q -x) > $O/pytest_l.log 2>&1; echo "pytest learner rc=$?"; tail -4 $O/pytest_l.log | cut -c1-300
for v in 0 1; do
MI355PPO_MASK_BITS=$v timeout 600 python bench.py --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_bits$v.json 2> $O/bench_C_bits$v.err; echo "bench bits=$v rc=$?"; cut -c1-240 $O/bench_C_bits$v.json
done
