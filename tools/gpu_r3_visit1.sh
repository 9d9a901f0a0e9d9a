#!/bin/bash
# Round-3 visit 1: parity suite, matrix-pipe floor, 6-vs-9 term-pair A/B, bench lines + rocprof kernel stats for configs B, C, D, E.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > $O/host.txt 2>&1
timeout 60 tools/mfma_floor > $O/mfma_floor.jsonl 2> $O/mfma_floor.err; echo "floor rc=$?"
for p in 6 9; do
  MI355PPO_BF16_PAIRS=$p timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_pairs$p.json 2>&1
  MI355PPO_BF16_PAIRS=$p timeout 120 python tools/err_pairs.py > $O/err_pairs$p.json 2> $O/err_pairs$p.err
done
CONV_TRAFFIC_FWD_F32=1 timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_f32.json 2>&1
for m in 1024 4096 8192; do MI355PPO_BF16_PAIRS=6 timeout 120 tools/conv_traffic $m 6 > $O/conv_traffic_pairs6_m$m.json 2>&1; CONV_TRAFFIC_FWD_F32=1 timeout 120 tools/conv_traffic $m 6 > $O/conv_traffic_f32_m$m.json 2>&1; done
tail -n 2 $O/conv_traffic_pairs6.json $O/conv_traffic_pairs9.json $O/conv_traffic_f32.json
(time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_C.log 2> $O/bench_C.err; echo "bench C rc=$?"
MI355PPO_FWD23=bf16 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_fwd23bf16.log 2> $O/bench_C_fwd23bf16.err; echo "bench C bf16 rc=$?"
for c in B D E; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$c.log 2> $O/bench_$c.err; echo "bench $c rc=$?"
done
for c in B D E C; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o bench -- python bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_$c.log 2>&1; echo "prof $c rc=$?"
  find $O/prof_$c -name "*kernel_trace.csv" -size +8M -delete
  find $O/prof_$c -name "*.db" -delete
done
ls -la $O | head -50
