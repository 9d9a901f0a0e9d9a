#!/bin/bash
# round 4: K-split granularity of the rollout-sized FC forward at 128 / 256 / 1024 rows (MI355PPO_FC_SPLIT_WAVES = wave tiles aimed for)
mkdir -p gpurun_out
out=gpurun_out/r04_fcsplit_sweep.txt; : > $out
for images in 128 256 1024; do
  for w in 128 256 512 1024 2048; do
    echo -n "images=$images split_waves=$w " >> $out
    CONV_TRAFFIC_NOBITS=1 MI355PPO_FC_SPLIT_WAVES=$w tools/conv_traffic $images 30 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fc_fwd_us', d['fc_fwd_us'])" >> $out
  done
done
cat $out
