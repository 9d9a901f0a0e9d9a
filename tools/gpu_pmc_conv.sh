#!/bin/bash
# Matrix-pipe occupancy PMC pass over the conv kernels of one minibatch update (torch-free driver).
set -u
export TMPDIR=/tmp
out=gpurun_out/pmc_conv
rm -rf $out; mkdir -p $out
timeout 60 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU \
    --kernel-trace -d $out/busy -o t -- tools/conv_traffic 32768 3 > $out/busy.log 2>&1; echo "busy rc=$?"
db=$(ls $out/busy/*.db | head -1); python tools/rocpd_pmc.py "$db" > $out/busy.csv && rm -f "$db"
timeout 60 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE \
    --kernel-trace -d $out/mem -o t -- tools/conv_traffic 32768 3 > $out/mem.log 2>&1; echo "mem rc=$?"
db=$(ls $out/mem/*.db | head -1); python tools/rocpd_pmc.py "$db" > $out/mem.csv && rm -f "$db"
head -9 $out/busy.csv | cut -c16-60,90-250
