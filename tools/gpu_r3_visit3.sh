#!/bin/bash
# Round-3 visit 3: VALU dependency-chain cost beside MFMAs; kernels X / C with the staged fixed-position split vs the adaptive one.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 60 tools/mfma_floor > $O/mfma_floor2.jsonl 2> $O/mfma_floor2.err; echo "floor rc=$?"
for sp in adaptive fixed; do
  MI355PPO_BF16_SPLIT=$sp timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic_split_$sp.json 2>&1
  MI355PPO_BF16_SPLIT=$sp timeout 120 python tools/err_pairs.py > $O/err_split_$sp.json 2> $O/err_split_$sp.err
done
tail -n 2 $O/conv_traffic_split_adaptive.json $O/conv_traffic_split_fixed.json
rm -rf $O/pmc_busy2
timeout 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $O/pmc_busy2 -o t -- tools/conv_traffic 32768 3 > $O/pmc_busy2.log 2>&1
db=$(ls $O/pmc_busy2/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmc_busy2.csv; rm -rf $O/pmc_busy2
timeout 600 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_multirank.py -m gpu -q -x > $O/pytest_v3.log 2>&1; tail -3 $O/pytest_v3.log
