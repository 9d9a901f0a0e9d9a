#!/bin/bash
# Round-3 visit 6: kernel Z on the convolutions: parity + timing at several sizes.
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "kernel_z or fcz or pack or full_minibatch" > $O/pytest_v6.log 2>&1; tail -12 $O/pytest_v6.log | cut -c1-300
for m in 32768 8192 4096 1024; do
  timeout 120 tools/conv_traffic $m 4 /tmp/dump_f_$m.bin > $O/conv_traffic_f_$m.json 2>&1
  CONV_TRAFFIC_CONV_Z=1 timeout 120 tools/conv_traffic $m 4 /tmp/dump_z_$m.bin > $O/conv_traffic_convz_$m.json 2>&1
  head -1 $O/conv_traffic_f_$m.json; head -1 $O/conv_traffic_convz_$m.json
  python tools/cmp_f32.py /tmp/dump_f_$m.bin /tmp/dump_z_$m.bin > $O/cmp_f_z_$m.txt 2>&1; tail -2 $O/cmp_f_z_$m.txt
done
rm -rf $O/pmc_busy4
CONV_TRAFFIC_CONV_Z=1 timeout 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $O/pmc_busy4 -o t -- tools/conv_traffic 32768 3 > $O/pmc_busy4.log 2>&1
db=$(ls $O/pmc_busy4/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py "$db" z_kernel > $O/pmc_busy4.csv; rm -rf $O/pmc_busy4
cat $O/pmc_busy4.csv | cut -c1-60,150-400
