#!/bin/bash
# Kernel Z with the workgroup's B ring (BLDS) against every wave streaming its own B (MI355PPO_Z_BLDS=0): bit-identity of the
# results (the ring only changes where a fragment is read from), then same-box timings of one minibatch update's launches.
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for m in 32768 1000 1; do
  MI355PPO_Z_BLDS=0 timeout 120 tools/conv_traffic $m 1 /tmp/d0_$m.bin > /dev/null 2>&1; echo "base rc=$?"
  MI355PPO_Z_BLDS=1 timeout 120 tools/conv_traffic $m 1 /tmp/d1_$m.bin > /dev/null 2>&1; echo "blds rc=$?"
  cmp /tmp/d0_$m.bin /tmp/d1_$m.bin && echo "images=$m: bit-identical"
done
for rep in 1 2; do
for v in 0 1; do
  for m in 32768 8192 4096 1024; do
    echo -n "blds=$v "; MI355PPO_Z_BLDS=$v timeout 120 tools/conv_traffic $m 5 2>&1 | head -1 | tee -a $O/blds_ab.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], {k:d[k] for k in ('fwd2_us','fwd3_us','dgrad3_us','dgrad2_us','fc_dgrad_us','sum_ms')})"
  done
done
done
