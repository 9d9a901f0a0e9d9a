#!/bin/bash
# ReLU masks as bits against the f32-activation masks (CONV_TRAFFIC_NOBITS=1): bit-identity of every result, then same-box timings.
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for m in 32768 2049 1000 1; do
  CONV_TRAFFIC_NOBITS=1 timeout 120 tools/conv_traffic $m 1 /tmp/d0_$m.bin > /dev/null 2>&1; echo "f32 masks rc=$?"
  timeout 120 tools/conv_traffic $m 1 /tmp/d1_$m.bin > /dev/null 2>&1; echo "bits rc=$?"
  cmp /tmp/d0_$m.bin /tmp/d1_$m.bin && echo "images=$m: bit-identical"
done
for rep in 1 2; do
for v in 1 0; do
  for m in 32768 8192 4096; do
    echo -n "nobits=$v "; if [ $v = 1 ]; then export CONV_TRAFFIC_NOBITS=1; else unset CONV_TRAFFIC_NOBITS; fi
    timeout 120 tools/conv_traffic $m 5 2>&1 | head -1 | tee -a $O/bits_ab.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], {k:d[k] for k in ('fwd1_us','fwd2_us','fwd3_us','fc_dgrad_us','dgrad3_us','dgrad2_us','sum_ms')})"
  done
done
done
