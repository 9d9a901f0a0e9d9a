#!/bin/bash
# Round 4, visit 2: K7 after the alignment fix (flat parameter buffers pack without padding); run-to-run determinism and all
# measurements of the config C / D whole-iteration tests; bench E.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v2; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_mlp.py -q -x -s 2>&1 | tail -30 | tee $O/pytest_mlp.log
timeout 600 python -m pytest tests/test_gpu_learner.py -q -x -k "captured or rpo or continuous or cartpole or agent_api" 2>&1 | tail -15 | tee $O/pytest_learner.log
for rep in 1 2; do
  timeout 600 python -m pytest tests/test_gpu_multirank.py -q -x -s -k config_d 2>&1 | grep -E "rank [01]: |passed|failed" | sort -u | head -12 | tee -a $O/pytest_d.log
  timeout 600 python -m pytest tests/test_gpu_learner.py -q -x -s -k config_c 2>&1 | grep -E "update |values |passed|failed" | head -8 | tee -a $O/pytest_c.log
done
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items() if k!='note'}, 'frac', r.get('frac'), 'us', r.get('avg_launch_us'), 'hbm_frac', d.get('hbm_frac'))"; }
for rep in 1 2; do
  echo -n "E fused+graphs: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive 2>$O/err_e.log | tee -a $O/bench_e.jsonl | show
  echo -n "E fused eager update: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-update-graphs 2>>$O/err_e.log | tee -a $O/bench_e.jsonl | show
done
echo -n "E fused, all eager: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-update-graphs --no-rollout-graphs 2>>$O/err_e.log | tee -a $O/bench_e.jsonl | show
tail -n 5 $O/err_*.log
