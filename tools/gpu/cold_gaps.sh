#!/bin/bash
# The first bench.py run on a fresh box under the kernel trace (cold host: the driver's round-end bench is exactly this), then the
# same command again (warm): where does the GPU idle?   Outputs: gpurun_out/cold/gaps_{cold,warm}.txt + the bench lines.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cold; rm -rf $O; mkdir -p $O
cd /tmp
for v in cold warm; do
  rm -rf /tmp/prof_$v
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$v -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_$v.log 2>&1; echo "prof $v rc=$?"
  grep '^{' $O/prof_$v.log | tail -1 | cut -c1-260
  db=$(find /tmp/prof_$v -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_gaps_all.py "$db" > $O/gaps_$v.txt
  rm -rf /tmp/prof_$v
done
cd $R
timeout 300 python bench.py --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | cut -c1-260
