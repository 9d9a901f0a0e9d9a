#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cnn.py -q -x -k "fc or fused_rollout or trunk" 2>&1 | tail -3
python -m pytest tests/test_gpu_learner.py tests/test_gpu_multirank.py -q -x 2>&1 | tail -3
out=gpurun_out/r04_fcsplit_bench.jsonl; : > $out
for cfg in B D; do
  for w in 0 1024; do
  MI355PPO_FC_SPLIT_WAVES=$w python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'config':'$cfg','fc_split_waves':$w,'value':d['value'],'ms_per_step':d['ms_per_step'],'phases':d.get('phases_ms')}))" >> $out
  done
done
cat $out
