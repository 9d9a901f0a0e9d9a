#!/bin/bash
# Torch-free A/B of conv kernel variants (tools/conv_traffic.cpp): new default weight-gradient kernels (R, T) against the
# previous generation (D, W) at odd / small / non-gathered batches.
set -u
export TMPDIR=/tmp
B=tools/conv_traffic
for cfg in "513 0" "96 1" "3 1" "4097 1" "1024 0"; do
  set -- $cfg; n=$1; noinds=$2
  extra=""; [ "$noinds" = 1 ] && extra="CONV_TRAFFIC_NOINDS=1"
  env $extra MI355PPO_WGRAD=2 MI355PPO_WGRAD_TAPS=0 timeout 30 $B $n 1 /tmp/old.bin > /tmp/old.log 2>&1 || echo "old failed"
  env $extra timeout 30 $B $n 1 /tmp/new.bin > /tmp/new.log 2>&1 || echo "new failed"
  echo "images=$n noinds=$noinds: $(python tools/cmp_f32.py /tmp/old.bin /tmp/new.bin | grep WORST) $(grep -c 'identical=True' <(python tools/cmp_f32.py /tmp/old.bin /tmp/new.bin)) sections identical"
done
tail -2 /tmp/new.log | head -1
