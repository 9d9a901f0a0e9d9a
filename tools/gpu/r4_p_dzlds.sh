#!/bin/bash
# round 4: kernel P with dz staged global -> LDS directly (global_load_lds, inline asm, hand-counted waits), 2 / 3 steps ahead
mkdir -p gpurun_out; out=gpurun_out/r04_kernel_p_dz_lds_direct_ab.txt; : > $out
for images in 32768 4096; do
  for d in 0 3 2 0 3 2; do
    echo -n "images=$images dz_lds_steps_ahead=$d " >> $out
    CONV_TRAFFIC_HASH=1 MI355PPO_P_DZLDS=$d timeout 60 tools/conv_traffic $images 8 2>&1 | grep -i "hash dW1\|hash db1\|^{" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('wgrad1_us', json.loads(l)['wgrad1_us'], end=' ')
    else: print(l.strip(), end=' ')
print()" >> $out
  done
done
cat $out
