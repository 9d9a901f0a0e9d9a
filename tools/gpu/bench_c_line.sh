#!/bin/bash
# bench.py config C (the full line) + its rocprofv3 kernel statistics, B / D lines.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cline; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_cfgC.json 2> $O/bench_cfgC.err; echo "bench C rc=$?"; cut -c1-250 $O/bench_cfgC.json
cd /tmp; rm -rf /tmp/prof_C
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_C -o bench -- python $R/bench.py --config C --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_C.log 2>&1; echo "prof C rc=$?"
db=$(find /tmp/prof_C -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py "$db" 80 > $O/kernel_stats_cfgC.csv
grep '^{' $O/prof_C.log | tail -1 > $O/bench_cfgC_profiled.json
cd $R
for c in B D; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive > $O/bench_cfg$c.json 2>/dev/null; cut -c1-200 $O/bench_cfg$c.json; done
(lscpu | head -20; rocm-smi --showproductname 2>/dev/null | head -12) > $O/gpu_box_host.txt 2>&1
