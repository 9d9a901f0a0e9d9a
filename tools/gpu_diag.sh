#!/bin/bash
for d in 3 11 3 11; do
  echo "DIAG=$d"; MI355PPO_CONV_DIAG=$d timeout 300 python tools/cnnbench.py 32768 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['k']=='wgrad': print('  L%d %8.1f us  %.3f' % (d['layer'], d['us'], d['frac']))"
done
