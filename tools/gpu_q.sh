#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_cnn.py -q -x -k "conv1q or conv1_fwd or full_minibatch or trunk" -p no:cacheprovider ) > $O/pytest_q.log 2>&1
echo "pytest rc=$?"; tail -25 $O/pytest_q.log
CNNBENCH_ONLY=fwd timeout 300 python tools/cnnbench.py > $O/cnnbench_q.jsonl 2> $O/cnnbench_q.err; echo "cnnbench rc=$?"; tail -30 $O/cnnbench_q.jsonl | cut -c1-300
