#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cnn.py -x -q -k "${KEXPR:-wgrad}" 2>&1 | tail -2
for w in 1 2; do
timeout 300 python tools/cnnbench.py 32768 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['k']=='wgrad': print('  %-6s L%d %8.1f us  %.3f' % (d['k'], d['layer'], d['us'], d['frac']))"
done
