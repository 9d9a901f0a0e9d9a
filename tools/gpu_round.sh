#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprofv3 kernel stats of the bench command.  Run via gpurun from the repo root.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" > gpurun_out/host.txt 2>&1
rocm-smi --showproductname >> gpurun_out/host.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -1 gpurun_out/bench.log
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_bench -name "*kernel_trace.csv" -size +20M -delete
ls -la gpurun_out/prof_bench/* | head
