#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cnn.py -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_learner.py -q -x -k "captured or config_b or teacher" 2>&1 | tail -3
out=gpurun_out/r04_small_tiles_bench.jsonl; : > $out
for cfg in B D; do
  for mt in 1 2; do
  MI355PPO_Z_SMALL_MT=$mt python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'config':'$cfg','small_mt':$mt,'value':d['value'],'ms_per_step':d['ms_per_step'],'phases':d.get('phases_ms')}))" >> $out
  done
done
cat $out
