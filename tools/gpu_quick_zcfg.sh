#!/bin/bash
# Tile-shape A/B of kernel Z (MI355PPO_Z_CFG, gemmz.hip) on one box.
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(MI355PPO_Z_CFG=7 timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "kernel_z or full_minibatch") > $O/pytest_zcfg.log 2>&1; echo "pytest zcfg rc=$?"; tail -3 $O/pytest_zcfg.log | cut -c1-300
for c in 0 7 0 7; do
  MI355PPO_Z_CFG=$c timeout 120 tools/conv_traffic 32768 4 2>&1 | head -1 | sed "s/^/cfg=$c /" | tee -a $O/zcfg2_ab.jsonl | cut -c1-420
done
for m in 8192 4096 1024; do for c in 0 7; do
  MI355PPO_Z_CFG=$c timeout 120 tools/conv_traffic $m 6 2>&1 | head -1 | sed "s/^/cfg=$c /" | tee -a $O/zcfg2_ab.jsonl | cut -c1-420
done; done
