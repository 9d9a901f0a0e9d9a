#!/bin/bash
# Border-class rows of kernel Z's data gradients: correctness, then entry-point timings of one minibatch update's launches.
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "dgrad or full_minibatch or trunk or fcz or 4GiB") > $O/pytest_cls.log 2>&1; echo "pytest cls rc=$?"; tail -4 $O/pytest_cls.log | cut -c1-300
for m in 32768 32768 8192 4096 1024; do
  timeout 120 tools/conv_traffic $m 5 2>&1 | head -1 | tee -a $O/cls_timing.jsonl | cut -c1-420
done
