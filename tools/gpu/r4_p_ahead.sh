#!/bin/bash
# round 4: kernel P (zero-extended operand) with dz one / two steps ahead
mkdir -p gpurun_out; out=gpurun_out/r04_kernel_p_ahead_ab.txt; : > $out
for images in 32768 4096; do
  for a in 1 2 1 2; do
    echo -n "images=$images ahead=$a " >> $out
    CONV_TRAFFIC_HASH=1 MI355PPO_P_AHEAD=$a tools/conv_traffic $images 8 2>&1 | grep -i "hash dW1\|^{" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('wgrad1_us', json.loads(l)['wgrad1_us'], end=' ')
    else: print(l.strip(), end=' ')
print()" >> $out
  done
done
cat $out
