#!/bin/bash
# round 4: K split of the FC forward at minibatch sizes 4,096 / 8,192 rows (configs B / D): whole-K tiles (MI355PPO_FC_SPLIT_BELOW=4096) vs splits
mkdir -p gpurun_out; out=gpurun_out/r04_fc_split_minibatch_ab.txt; : > $out
for images in 4096 8192 12288; do
  echo -n "images=$images whole_K " >> $out
  MI355PPO_FC_SPLIT_BELOW=4096 tools/conv_traffic $images 20 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fc_fwd_us', d['fc_fwd_us'])" >> $out
  for w in 1024 2048 4096; do
    echo -n "images=$images split_waves=$w " >> $out
    MI355PPO_FC_SPLIT_WAVES=$w tools/conv_traffic $images 20 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fc_fwd_us', d['fc_fwd_us'])" >> $out
  done
done
cat $out
