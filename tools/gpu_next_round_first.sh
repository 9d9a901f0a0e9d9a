#!/bin/bash
# First GPU-box visit of the next round (everything since profiles/r01_bench_n1_mfma_cnn_v11.json landed without a full bench):
#   1. bench.py (the number), 2. its rocprofv3 kernel stats, 3. the always-on whole-iteration GPU test + the opt-in new-script
#   tests, 4. the staged kernel-T variant's A/B.  ~6-8 GPU-minutes; summaries land in gpurun_out/ -> copy into profiles/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
db=$(ls gpurun_out/prof_bench/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > gpurun_out/bench_kernel_stats.csv
MI355PPO_GPU_EXTENDED=1 timeout 1500 python -m pytest tests/test_zz_gpu_new_scripts.py -q -rxXs > gpurun_out/pytest_new_scripts.log 2>&1; tail -12 gpurun_out/pytest_new_scripts.log
bash tools/gpu_lab.sh 2>&1 | tail -12
