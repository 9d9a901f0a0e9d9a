#!/bin/bash
# round 6: the five counter passes (fetch / write with the calibration copy, busy, mem, lds) over one minibatch's launches on the torch-free driver
# (`tools/pmc_table.py <prefix>` prints DESIGN 3.2's table from them).  $E: extra environment; $TAG: output directory; $PASSES: subset.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r6pmc}; rm -rf $O; mkdir -p $O; cd $R
T0=$SECONDS
pmc_pass() {   # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    env ${E:-} CONV_TRAFFIC_F16=1 timeout 120 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o t -- tools/conv_traffic ${M:-32768} 3 > $O/pmc_$name.log 2>&1
    echo "pmc $name rc=$? t=$((SECONDS-T0))"
    db=$(ls /tmp/pmc_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmc_$name.csv
    rm -rf /tmp/pmc_$name
}
P=" ${PASSES:-fetch write busy mem lds} "
export CONV_TRAFFIC_CALIB=1
[[ $P == *" fetch "* ]] && pmc_pass fetch FETCH_SIZE
[[ $P == *" write "* ]] && pmc_pass write WRITE_SIZE
unset CONV_TRAFFIC_CALIB
[[ $P == *" busy "* ]] && pmc_pass busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
[[ $P == *" mem "* ]] && pmc_pass mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE
[[ $P == *" lds "* ]] && pmc_pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
rm -f $O/pmc_*.log
ls $O
