#!/bin/bash
# Round-3 visit 7: kernel Z integrated into the trunk: full GPU suite, same-box A/B of bench.py (f32-pipe kernels F / X vs kernel Z).
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(time timeout 1500 python -m pytest tests -m gpu -q -x --durations=6) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log | cut -c1-300
CONV_TRAFFIC_CONV_Z=1 timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_z_all.json 2>&1; head -1 $O/conv_traffic_z_all.json
MI355PPO_CONV=f MI355PPO_FC=x timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_ab_f.log 2> $O/bench_C_ab_f.err; tail -1 $O/bench_C_ab_f.log | cut -c1-160
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_ab_z.log 2> $O/bench_C_ab_z.err; tail -1 $O/bench_C_ab_z.log | cut -c1-160
MI355PPO_CONV=f MI355PPO_FC=x timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_ab_f2.log 2> $O/bench_C_ab_f2.err; tail -1 $O/bench_C_ab_f2.log | cut -c1-160
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_ab_z2.log 2> $O/bench_C_ab_z2.err; tail -1 $O/bench_C_ab_z2.log | cut -c1-160
for c in B D; do
  MI355PPO_CONV=f MI355PPO_FC=x timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_${c}_ab_f.log 2>/dev/null; tail -1 $O/bench_${c}_ab_f.log | cut -c1-160
  timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_${c}_ab_z.log 2>/dev/null; tail -1 $O/bench_${c}_ab_z.log | cut -c1-160
done
