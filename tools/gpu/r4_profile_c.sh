#!/bin/bash
# round 4: rocprofv3 kernel statistics of the default bench command (config C) and its bench line without the profiler
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; rm -rf $O; mkdir -p $O
cd /tmp
for cfg in ${CFGS:-C}; do
  rm -rf /tmp/prof_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o c -- python $R/bench.py --config $cfg --no-cpu-baseline --no-pcie-inclusive --steps 3 --warmup 2 > $O/bench_cfg${cfg}_under_rocprof.json 2>$O/rocprof_$cfg.err
  db=$(find /tmp/prof_$cfg -name '*.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_stats.py "$db" 60 > $O/kernel_stats_cfg$cfg.csv && head -45 $O/kernel_stats_cfg$cfg.csv | cut -c1-170
  [ -n "$db" ] && python $R/tools/rocpd_gaps_all.py "$db" 2>/dev/null | sed -n "1,12p" 
done
