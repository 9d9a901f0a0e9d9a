#!/bin/bash
# round 4: 32-row wave tiles for the rollout-sized layer-2 / layer-3 forwards (MI355PPO_Z_SMALL_MT=1) against the 64-row default
mkdir -p gpurun_out
out=gpurun_out/r04_small_tiles_ab.txt; : > $out
for images in 1024 256 128; do
  for mt in 2 1; do
    echo "images=$images small_mt=$mt" >> $out
    CONV_TRAFFIC_NOBITS=1 CONV_TRAFFIC_HASH=1 MI355PPO_Z_SMALL_MT=$mt tools/conv_traffic $images 20 2>&1 | grep -i "conv2_fwd\|conv3_fwd\|hash\|{" | cut -c1-600 >> $out
  done
done
cat $out
