#!/bin/bash
# conv_traffic A/B on ONE box: f32-pipe kernels (F, X) vs kernel Z, twice each (the boxes of the pool differ by several per cent)
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for r in 1 2; do
CONV_TRAFFIC_FWD_F32=1 CONV_TRAFFIC_FC_X=1 timeout 120 tools/conv_traffic ${1:-32768} 4 > $O/ab_f_$r.json 2>&1; head -1 $O/ab_f_$r.json
CONV_TRAFFIC_CONV_Z=1 timeout 120 tools/conv_traffic ${1:-32768} 4 > $O/ab_z_$r.json 2>&1; head -1 $O/ab_z_$r.json
done
