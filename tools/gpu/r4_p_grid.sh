#!/bin/bash
# round 4: kernel P at small minibatches: 512 workgroups (2,048 partials of 32 KB) vs fewer
mkdir -p gpurun_out; out=gpurun_out/r04_kernel_p_grid_small_batches.txt; : > $out
for images in 4096 8192; do
  for g in 512 384 256 128; do
    echo -n "images=$images grid=$g " >> $out
    MI355PPO_P_GRID=$g MI355PPO_P_GRID_BELOW=100000 tools/conv_traffic $images 20 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad1_us', d['wgrad1_us'])" >> $out
  done
done
cat $out
