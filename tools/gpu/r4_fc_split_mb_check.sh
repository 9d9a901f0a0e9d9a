#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cnn.py -q -x -k "fc or trunk or fused_rollout" 2>&1 | tail -3
python -m pytest tests/test_gpu_learner.py -q -x 2>&1 | tail -3
out=gpurun_out/r04_fc_split_minibatch_bench.jsonl; : > $out
for rep in 1 2; do
  for b in 8192 4096; do
  MI355PPO_FC_SPLIT_BELOW=$b python bench.py --config B --steps 8 --warmup 2 --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'config':'B','fc_split_below':$b,'value':d['value'],'ms_per_step':d['ms_per_step'],'phases':d.get('phases_ms')}))" >> $out
  done
done
cat $out
