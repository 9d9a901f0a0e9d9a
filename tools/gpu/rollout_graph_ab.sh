#!/bin/bash
# Rollout captured as T graphs of one step / graphs of several steps / one graph: bit-identity test, then bench.py config C, B alternating.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rgraph; rm -rf $O; mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_gpu_learner.py -q -x -k "captured_rollout" 2>&1 | tail -3
for rep in 1 2; do for per in 1 16 128; do for c in C B; do
  echo -n "config $c steps_per_graph=$per: "
  timeout 200 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive --steps 6 --warmup 2 --rollout-steps-per-graph $per 2>/dev/null | sed "s/^{/{\"steps_per_graph\": $per, /" | tee -a $O/rollout_graph_ab.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['phases_ms']['rollout_incl_gae'],2), d['final_loss'])"
done; done; done
