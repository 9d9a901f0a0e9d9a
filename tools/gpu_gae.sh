#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k gae > gpurun_out/pytest_gae.log 2>&1; tail -3 gpurun_out/pytest_gae.log
for sz in 128x1024 2048x64 128x128; do
  rm -rf gpurun_out/prof_gae_$sz
  KBENCH_GAE_SIZES=$sz timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_gae_$sz -o gae -- python tools/kbench.py gae > gpurun_out/kbench_gae_$sz.jsonl 2>/dev/null
  echo "== $sz"; python tools/rocpd_stats.py gpurun_out/prof_gae_$sz/gae_results.db 40 | grep gae
done
python tools/kbench.py gae > gpurun_out/kbench_gae_sweep.jsonl 2>/dev/null
