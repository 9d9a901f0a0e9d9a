#!/bin/bash
# round 4: the stand-in Atari env writes the rollout row's own layout (gather + relayout in one launch) -- tests, then bench C / B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_learner.py -q -x 2>&1 | tail -4
out=gpurun_out/r04_env_rows_bench.jsonl; : > $out
for cfg in C B D; do
  python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'config':'$cfg','value':d['value'],'ms_per_step':d['ms_per_step'],'phases':d.get('phases_ms')}))" >> $out
done
cat $out
