#!/bin/bash
# round 6: rocprofv3 kernel statistics of the default bench command (config C) -> gpurun_out/$TAG/kernel_stats_cfgC.csv (+ the profiled line)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r6prof}; mkdir -p $O
cd /tmp; rm -rf /tmp/prof_C
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_C -o bench -- python $R/bench.py --config C --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_C.log 2>&1; echo "prof rc=$?"
db=$(find /tmp/prof_C -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py "$db" 80 > $O/kernel_stats_cfgC.csv
grep '^{' $O/prof_C.log | tail -1 > $O/bench_cfgC_profiled.json
head -40 $O/kernel_stats_cfgC.csv | cut -c1-70,150-215
