#!/bin/bash
# round 4: the fused rollout step behind the trunk -- parity, then A/B at configs C / B / D (MI355PPO_FUSED_ACT=0: the four launches)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cnn.py -q -k "fused_rollout_step or heads_forward" -x 2>&1 | tail -5
python -m pytest tests/test_gpu_learner.py -q -x 2>&1 | tail -5
out=gpurun_out/r04_fused_act_ab.jsonl; : > $out
for rep in 1 2; do
  for cfg in C B; do
    for f in 1 0; do
      MI355PPO_FUSED_ACT=$f python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'config':'$cfg','fused_act':$f,'value':d['value'],'ms_per_step':d['ms_per_step'],'phases':d.get('phases_ms') or d.get('phases')}))" >> $out
    done
  done
done
cat $out
