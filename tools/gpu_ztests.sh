#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "kernel_z or bits or trunk or full_minibatch or fcz") > $O/pytest_z.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_z.log | cut -c1-300
(timeout 900 python -m pytest tests/test_gpu_learner.py -m gpu -q -x) > $O/pytest_l.log 2>&1; echo "pytest learner rc=$?"; tail -4 $O/pytest_l.log | cut -c1-300
