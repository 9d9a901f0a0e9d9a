#!/bin/bash
# Round 4, visit 4: parameter gradients written straight into the flat buffer (no AccumulateGrad adds): tests, then A/B at C / B.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v4; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_multirank.py tests/test_gpu_cnn.py -q -x -k "not beyond_4GiB" 2>&1 | tail -5 | tee $O/pytest.log
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items() if k!='note'}, 'frac', r.get('frac'), 'hbm_frac', d.get('hbm_frac'))"; }
for rep in 1 2; do
  for c in C B; do
    for v in 1 0; do
      echo -n "$c direct_grads=$v: "; MI355PPO_DIRECT_GRADS=$v timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive --steps 6 --warmup 2 2>$O/err.log | sed "s/^{/{\"direct_grads\": $v, /" | tee -a $O/bench_direct_ab.jsonl | show
    done
  done
done
tail -3 $O/err.log
