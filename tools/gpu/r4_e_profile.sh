#!/bin/bash
# Config E on the fused MLP kernels: tests, bench lines (graphs / eager), rocprofv3 kernel statistics of both.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_mlp.py -q -x -s 2>&1 | grep -E "config E|update |passed|failed|Error|error" | head -20 | tee $O/pytest_mlp.log
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items() if k!='note'}, 'frac', r.get('frac'), 'us', r.get('avg_launch_us'))"; }
echo -n "E fused+graphs: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive 2>$O/err_e.log | tee -a $O/bench_e.jsonl | show
echo -n "E fused eager update: "; timeout 300 python bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-update-graphs 2>>$O/err_e.log | tee -a $O/bench_e.jsonl | show
cd /tmp
for mode in graphs eager; do
  flag=""; [ $mode = eager ] && flag="--no-update-graphs --no-rollout-graphs"
  rm -rf /tmp/prof_$mode
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o e -- python $R/bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-kernel-timing --steps 2 --warmup 1 $flag > $O/bench_e_rocprof_$mode.json 2>$O/rocprof_$mode.err
  f=$(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_cfgE_$mode.csv && head -14 $f | cut -c1-200
  ls /tmp/prof_$mode/* | head -5
done
tail -3 $O/err_e.log
