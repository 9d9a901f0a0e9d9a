#!/bin/bash
# Round-3 visit 5: kernel Z (FC forward / data gradient) parity + timing + PMC.
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "fc or pack or trunk" > $O/pytest_v5.log 2>&1; tail -15 $O/pytest_v5.log | cut -c1-300
timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_z.json 2>&1; head -1 $O/conv_traffic_z.json
CONV_TRAFFIC_FC_X=1 timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_x.json 2>&1; head -1 $O/conv_traffic_x.json
MI355PPO_BF16_PAIRS=9 timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_z9.json 2>&1; head -1 $O/conv_traffic_z9.json
rm -rf $O/pmc_busy3
timeout 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $O/pmc_busy3 -o t -- tools/conv_traffic 32768 3 > $O/pmc_busy3.log 2>&1
db=$(ls $O/pmc_busy3/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py "$db" zgemm > $O/pmc_busy3.csv; rm -rf $O/pmc_busy3
rm -rf $O/pmc_mem3
timeout 90 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mem3 -o t -- tools/conv_traffic 32768 3 > $O/pmc_mem3.log 2>&1
db=$(ls $O/pmc_mem3/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py "$db" zgemm > $O/pmc_mem3.csv; rm -rf $O/pmc_mem3
cat $O/pmc_busy3.csv $O/pmc_mem3.csv | cut -c1-300
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_z.log 2> $O/bench_C_z.err; tail -1 $O/bench_C_z.log | cut -c1-200
