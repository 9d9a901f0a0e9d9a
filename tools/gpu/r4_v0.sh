#!/bin/bash
# Round 4, first visit: the full-size whole-iteration tests of configs C and D, the update-graph tests (continuous path for the
# first time on hardware, Categorical at config B's shape), then eager vs --update-graphs at configs C / B / E on the same box.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v0; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_multirank.py -q -x -k "captured_update or config_c or config_d" 2>&1 | tail -15 | tee $O/pytest.log
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(round(d['value']), round(d['ms_per_step'],2), d['phases_ms'], 'frac', r.get('frac'), 'hbm_frac', d.get('hbm_frac'), 'itTF', d.get('iteration_f32_equiv_TFLOPs'))"; }
for rep in 1 2; do
  for mode in "" "--update-graphs"; do
    echo -n "C mode='$mode': "
    timeout 300 python bench.py --no-cpu-baseline --no-pcie-inclusive --steps 6 --warmup 2 $mode 2>$O/err_c.log | tee -a $O/bench_c_ab.jsonl | show
  done
done
for c in B E; do
  for mode in "" "--update-graphs"; do
    echo -n "$c mode='$mode': "
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive $mode 2>$O/err_$c.log | tee -a $O/bench_${c}_ab.jsonl | show
  done
done
tail -5 $O/err_*.log
