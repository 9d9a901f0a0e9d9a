#!/bin/bash
# Round-3 visit 8: after the prune (kernels C, R, X gone): full GPU suite, PMC passes over the final kernel set (Q, Z, V, W, P) on the
# torch-free driver, one bench line.
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(time timeout 1500 python -m pytest tests -m gpu -q -x --durations=6) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log | cut -c1-300
timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_final.json 2>&1; head -1 $O/conv_traffic_final.json | cut -c1-600
pmc_pass() {   # name, counters...
    name=$1; shift
    rm -rf $O/pmc_$name
    timeout 90 rocprofv3 --pmc "$@" --kernel-trace -d $O/pmc_$name -o t -- tools/conv_traffic 32768 3 > $O/pmc_$name.log 2>&1
    echo "pmc $name rc=$?"
    db=$(ls $O/pmc_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmc_$name.csv
    rm -rf $O/pmc_$name
}
pmc_pass busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
pmc_pass mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pmc_pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
pmc_pass cache TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
export CONV_TRAFFIC_CALIB=1
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
unset CONV_TRAFFIC_CALIB
cut -c1-200 $O/pmc_busy.csv | head -30
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_pruned.log 2> $O/bench_C_pruned.err; tail -1 $O/bench_C_pruned.log | cut -c1-1500
