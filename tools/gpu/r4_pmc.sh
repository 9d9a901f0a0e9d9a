#!/bin/bash
# round 4: the busy / mem / lds counter passes of DESIGN 3.2b's table over the final kernel set, and the busy pass with kernel P's
# conversion route (MI355PPO_P_ZEXT=0) for the before / after of the zero-extended operand
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4pmc; rm -rf $O; mkdir -p $O; cd $R
pmc_pass() {   # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    timeout 90 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o t -- tools/conv_traffic 32768 3 > $O/pmc_$name.log 2>&1
    echo "pmc $name rc=$?"
    db=$(ls /tmp/pmc_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmc_$name.csv
    rm -rf /tmp/pmc_$name
}
pmc_pass busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
pmc_pass mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pmc_pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
export MI355PPO_P_ZEXT=0
pmc_pass busy_p_converted SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
pmc_pass mem_p_converted SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE
rm -f $O/pmc_*.log
ls $O
