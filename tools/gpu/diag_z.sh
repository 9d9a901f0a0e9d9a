#!/bin/bash
# Kernel Z with parts compiled out (tools/z_diag_build.py): what is left, timed on the torch-free driver.  Results are garbage.
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
L=cleanrl_amd/csrc/libmi355ppo.so
cp $L /tmp/lib_main.so
for d in 0 1 2 3 4 7 8 16 24 32 64 120 0; do
  if [ $d = 0 ]; then cp /tmp/lib_main.so $L; else cp tools/oldlib/zdiag_$d/libmi355ppo.so $L; fi
  timeout 120 tools/conv_traffic 32768 4 2>&1 | head -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps({'z_diag': $d, **{k:d[k] for k in ('fwd2_us','fwd3_us','dgrad3_us','dgrad2_us','fc_fwd_us','fc_dgrad_us')}}))" | tee -a $O/z_diag.jsonl
done
cp /tmp/lib_main.so $L
