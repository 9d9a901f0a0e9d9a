#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_base.json 2>&1
MI355PPO_FCX_COAL=1 timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_fcx_coal.json 2>&1
head -1 $O/conv_traffic_base.json; head -1 $O/conv_traffic_fcx_coal.json
