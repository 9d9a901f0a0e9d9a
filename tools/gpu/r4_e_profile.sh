#!/bin/bash
# Config E on the fused MLP kernels: tests, bench lines (graphs / eager), rocprofv3 kernel statistics of both.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; rm -rf $O; mkdir -p $O
cd $R
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
  db=$(find /tmp/prof_$mode -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py "$db" 40 > $O/kernel_stats_cfgE_$mode.csv && head -16 $O/kernel_stats_cfgE_$mode.csv | cut -c1-200
  [ -n "$db" ] && python $R/tools/rocpd_gaps_all.py "$db" 2>/dev/null | sed -n "2,9p;/idle by/,\$p" | head -40
done
tail -3 $O/err_e.log
