#!/bin/bash
# round 4: the documented A/B switches still work -- the learner's GPU tests (whole-iteration goldens, captured rollout / update) under each
mkdir -p gpurun_out; out=gpurun_out/r04_switches_check.txt; : > $out
for sw in MI355PPO_FUSED_PACKS=0 MI355PPO_FUSED_ACT=0 MI355PPO_P_ZEXT=0 MI355PPO_DIRECT_GRADS=0 MI355PPO_Z_SMALL_MT=2 MI355PPO_FC_SPLIT_BELOW=4096 MI355PPO_MASK_BITS=0 MI355PPO_Z_BLDS=0; do
  echo -n "$sw: " >> $out
  env $sw python -m pytest tests/test_gpu_learner.py -q -x -k "teacher or config_b or captured or rollout_gae" 2>&1 | tail -1 >> $out
done
echo -n "MI355PPO_MLP=torch: " >> $out
MI355PPO_MLP=torch python -m pytest tests/test_gpu_learner.py -q -x -k "continuous or rpo or cartpole" 2>&1 | tail -1 >> $out
cat $out
