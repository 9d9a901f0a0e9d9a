#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_cnn.py -x -q -k "trunk" 2>&1 | tail -1
for v in 0 1 0 1; do
  MI355PPO_BWD_STREAMS=$v timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STREAMS=$v', round(d['value']), round(d['ms_per_step'],1), d['phases_ms']['update'])"
done
