#!/bin/bash
mkdir -p gpurun_out
for c in 0 1 2 0 1 2; do
  echo "CFG=$c"; MI355PPO_CONV_CFG=$c CNNBENCH_ONLY=fwd timeout 300 python tools/cnnbench.py 1024 32768 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  M%d L%d %8.1f us  %.3f' % (d['M'], d['layer'], d['us'], d['frac']))"
done
