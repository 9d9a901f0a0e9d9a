#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
CNNBENCH_TORCH=0 timeout 300 python tools/cnnbench.py 1024 32768 > gpurun_out/cnnbench_final.jsonl 2>/dev/null; wc -l gpurun_out/cnnbench_final.jsonl
rm -rf gpurun_out/pmc_cnn
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d gpurun_out/pmc_cnn -o pmc -- python tools/cnnbench.py 32768 > gpurun_out/pmc_cnn.log 2>&1; echo "pmc rc=$?"
