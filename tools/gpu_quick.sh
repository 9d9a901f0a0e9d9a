#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_cnn.py -x -q -k "${KEXPR:-conv1_fwd}" 2>&1 | tail -2
for w in 1 3 1 3; do
echo "CONV1_WG=$w"; MI355PPO_CONV1_WG=$w CNNBENCH_ONLY=fwd timeout 300 python tools/cnnbench.py 32768 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['layer']==1: print('  %-6s L%d %8.1f us  %.3f' % (d['k'], d['layer'], d['us'], d['frac']))"
done
