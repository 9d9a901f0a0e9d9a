#!/bin/bash
# PPOLearner.update_async (diagnostics resolved one iteration late): its GPU test, the tests that drive bench.py, then bench.py
# config C with and without --sync-metrics, each also with a 25 ms host stall injected into every update (what a busy box does to
# the issuing thread), alternating, same box.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/async; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_learner.py tests/test_gpu_multirank.py -q -x -k "update_async or captured_rollout or bench or multirank or two_ranks or config_b" 2>&1 | tail -4
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['phases_ms'], d['final_loss'])"; }
for rep in 1 2; do
  for mode in "" "--sync-metrics"; do
    for stall in 0 25; do
      echo -n "mode='$mode' stall=$stall: "
      timeout 300 python bench.py --no-cpu-baseline --no-pcie-inclusive --steps 8 --warmup 3 $mode --inject-host-stall-ms $stall 2>/dev/null | sed "s/^{/{\"mode\": \"${mode:-async}\", \"stall_ms\": $stall, /" | tee -a $O/async_ab.jsonl | show
    done
  done
done
for c in B D E; do echo -n "config $c async: "; timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | tee -a $O/async_cfg.jsonl | show; echo -n "config $c sync: "; timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive --sync-metrics 2>/dev/null | tee -a $O/async_cfg.jsonl | show; done
