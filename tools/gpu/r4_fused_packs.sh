#!/bin/bash
# round 4: all weight packs of the NatureCNN agent in one launch (MI355PPO_FUSED_PACKS=0: the 13 launches) -- parity, then A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cnn.py -q -x -k "all_weight_packs or trunk_matches or fused_rollout" 2>&1 | tail -3
python -m pytest tests/test_gpu_learner.py tests/test_gpu_multirank.py -q -x 2>&1 | tail -3
out=gpurun_out/r04_fused_packs_ab.jsonl; : > $out
for rep in 1 2; do
for cfg in C B; do
  for f in 1 0; do
  MI355PPO_FUSED_PACKS=$f python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'config':'$cfg','fused_packs':$f,'value':d['value'],'ms_per_step':d['ms_per_step'],'phases':d.get('phases_ms')}))" >> $out
  done
done
done
cat $out
