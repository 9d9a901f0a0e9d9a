#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_cnn
timeout 600 rocprofv3 --pmc ${PMC:-SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU} --kernel-trace -d gpurun_out/pmc_cnn -o pmc -- python tools/cnnbench.py 32768 > gpurun_out/pmc_cnn.log 2>&1; echo "pmc rc=$?"
ls -la gpurun_out/pmc_cnn/ | head; tail -3 gpurun_out/pmc_cnn.log
