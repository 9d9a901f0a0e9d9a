#!/bin/bash
# Tile-configuration A/B of kernel Z (MI355PPO_Z_CFG, gemmz.hip) on one box: correctness of the alternative configurations, then
# entry-point timings of one minibatch update's launches with each configuration (twice, interleaved).
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(MI355PPO_Z_CFG=15 timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -k "kernel_z or fcz or full_minibatch or fc_kernels or trunk" ) > $O/pytest_zcfg.log 2>&1; echo "pytest zcfg rc=$?"; tail -4 $O/pytest_zcfg.log | cut -c1-300
for c in 0 15 0 15 1 2 4 8; do
  MI355PPO_Z_CFG=$c timeout 120 tools/conv_traffic 32768 4 2>&1 | head -1 | sed "s/^/cfg=$c /" | tee -a $O/zcfg_ab.jsonl | cut -c1-420
done
for m in 8192 4096 1024; do for c in 0 15; do
  MI355PPO_Z_CFG=$c timeout 120 tools/conv_traffic $m 6 2>&1 | head -1 | sed "s/^/cfg=$c /" | tee -a $O/zcfg_ab.jsonl | cut -c1-420
done; done
(time timeout 1500 python -m pytest tests -m gpu -q --durations=6) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log | cut -c1-300
