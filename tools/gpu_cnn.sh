#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cnn.py -x -q > gpurun_out/pytest_cnn.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_cnn.log
tail -15 gpurun_out/pytest_cnn.log
timeout 600 python tools/cnnbench.py 1024 32768 > gpurun_out/cnnbench.jsonl 2> gpurun_out/cnnbench.err; echo "cnnbench rc=$?"
cat gpurun_out/cnnbench.jsonl; tail -5 gpurun_out/cnnbench.err
