#!/bin/bash
# Are the trained parameters bit-identical from process to process -- alone, and with a second process competing for the GPU?
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4det; rm -rf $O; mkdir -p $O
cd $R
for N in 256 128 1024; do
  for rep in 1 2; do timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | tee -a $O/alone.log; done
done
echo "== two at a time"
for N in 256 128; do
  for rep in 1 2; do
    timeout 200 python tools/gpu/determinism.py $N 2>/dev/null > $O/c1.log & timeout 200 python tools/gpu/determinism.py $N 2>/dev/null > $O/c2.log; wait
    cat $O/c1.log $O/c2.log | tee -a $O/concurrent.log
  done
done
echo "== config D test, all-reduce after the backward (MI355PPO_AR_OVERLAP=0), twice"
for rep in 1 2; do
  MI355PPO_AR_OVERLAP=0 timeout 600 python -m pytest tests/test_gpu_multirank.py -q -x -s -k config_d 2>&1 | grep -E "rank 0: (update|values)|passed|failed" | sort -u | head -6 | tee -a $O/pytest_d_nooverlap.log
done
