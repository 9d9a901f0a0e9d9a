#!/bin/bash
# Round 4, visit 1: the fused MLP kernel family K7 for the first time on hardware (tests/test_gpu_mlp.py), the update graphs with the
# optimizer step inside, config D's whole-iteration numbers under the three kernel sets (is the update-16 norm deviation a bias of
# the six-pair bf16 split or trajectory sensitivity?), then bench lines: E (fused, graphs; A/B against MI355PPO_MLP=torch), C, B.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v1; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_mlp.py -q -x -s 2>&1 | tail -40 | tee $O/pytest_mlp.log
timeout 600 python -m pytest tests/test_gpu_learner.py -q -x -k "captured or rpo or continuous or cartpole or agent_api" 2>&1 | tail -15 | tee $O/pytest_learner.log
for v in "" "MI355PPO_BF16_PAIRS=9" "MI355PPO_CONV=f MI355PPO_CONV_WGRAD=t MI355PPO_FC_WGRAD=y"; do
  echo "== config D whole iteration, kernel set: '${v:-default}'"
  env $v timeout 600 python -m pytest tests/test_gpu_multirank.py -q -x -k config_d 2>&1 | grep -E "update 16|update 8|passed|failed" | head -6 | tee -a $O/pytest_d.log
done
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items() if k!='note'}, 'frac', r.get('frac'), 'us', r.get('avg_launch_us'), 'hbm_frac', d.get('hbm_frac'))"; }
for rep in 1 2; do
  echo -n "E fused+graphs: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive 2>$O/err_e.log | tee -a $O/bench_e.jsonl | show
  echo -n "E fused eager update: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-update-graphs 2>>$O/err_e.log | tee -a $O/bench_e.jsonl | show
done
echo -n "E fused, all eager: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-update-graphs --no-rollout-graphs 2>>$O/err_e.log | tee -a $O/bench_e.jsonl | show
echo -n "E torch MLP (round 3 arrangement): "; MI355PPO_MLP=torch timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-update-graphs --steps 2 --warmup 1 2>>$O/err_e.log | tee -a $O/bench_e.jsonl | show
for c in C B D; do
  for mode in "" "--no-update-graphs"; do
    echo -n "$c mode='$mode': "
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive --steps 6 --warmup 2 $mode 2>$O/err_$c.log | tee -a $O/bench_${c}_ab.jsonl | show
  done
done
tail -n 5 $O/err_*.log
