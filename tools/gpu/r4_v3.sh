#!/bin/bash
# Round 4, visit 3: after the heads-backward fix and the whole-update replay path.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v3; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_mlp.py -q -x -s 2>&1 | grep -E "config E|update |passed|failed|Error|error" | cut -c1-400 | head -20 | tee $O/pytest_mlp.log
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -5 | tee $O/pytest_learner.log
for rep in 1 2; do
  timeout 600 python -m pytest tests/test_gpu_multirank.py -q -x -s -k config_d 2>&1 | grep -E "rank 0: (update|values)|passed|failed" | sort -u | head -6 | tee -a $O/pytest_d.log
done
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items() if k!='note'}, 'frac', r.get('frac'), 'us', r.get('avg_launch_us'), 'hbm_frac', d.get('hbm_frac'))"; }
for rep in 1 2; do
  echo -n "E fused+graphs: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive 2>$O/err_e.log | tee -a $O/bench_e.jsonl | show
done
echo -n "E fused eager update: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-update-graphs 2>>$O/err_e.log | tee -a $O/bench_e.jsonl | show
for c in C B D; do
  for mode in "" "--no-update-graphs"; do
    echo -n "$c mode='$mode': "
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive --steps 6 --warmup 2 $mode 2>$O/err_$c.log | tee -a $O/bench_${c}_ab.jsonl | show
  done
done
tail -n 3 $O/err_*.log
