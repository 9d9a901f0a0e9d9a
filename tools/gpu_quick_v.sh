#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "wgrad or trunk" 2>&1 | tail -8 | cut -c1-300
for r in 1 2; do
MI355PPO_CONV_WGRAD=t CONV_TRAFFIC_CONV_Z=1 timeout 120 tools/conv_traffic 32768 4 > $O/ab_vt_$r.json 2>&1; head -1 $O/ab_vt_$r.json
CONV_TRAFFIC_CONV_Z=1 timeout 120 tools/conv_traffic 32768 4 > $O/ab_vv_$r.json 2>&1; head -1 $O/ab_vv_$r.json
done
for m in 8192 4096 1024; do
MI355PPO_CONV_WGRAD=t CONV_TRAFFIC_CONV_Z=1 timeout 120 tools/conv_traffic $m 4 > $O/ab_vt_m$m.json 2>&1; head -1 $O/ab_vt_m$m.json
CONV_TRAFFIC_CONV_Z=1 timeout 120 tools/conv_traffic $m 4 > $O/ab_vv_m$m.json 2>&1; head -1 $O/ab_vv_m$m.json
done
