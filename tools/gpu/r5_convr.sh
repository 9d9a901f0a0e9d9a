#!/bin/bash
# Kernel R (convr.hip) against kernel Z on the torch-free conv driver: output hashes (must agree bit for bit) and per-launch times,
# MI355PPO_CONV_R=0 (kernel Z) vs default, at the minibatch size and at sizes with a partial last image group.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5convr; rm -rf $O; mkdir -p $O; cd $R
for M in ${SIZES:-32768 1027 61}; do
  for mode in 0 1; do
    MI355PPO_CONV_R=$mode CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 4 > $O/run_${M}_$mode.log 2>&1; echo "M=$M R=$mode rc=$?"
  done
  diff <(grep '^hash' $O/run_${M}_0.log) <(grep '^hash' $O/run_${M}_1.log) > $O/hashdiff_$M.txt && echo "M=$M: hashes identical" || { echo "M=$M: HASHES DIFFER"; cat $O/hashdiff_$M.txt; }
  for mode in 0 1; do grep '^{' $O/run_${M}_$mode.log | sed "s/^{/{\"conv_r\": $mode, /" | tee -a $O/ab.jsonl | cut -c1-420; done
done
for i in 1 2; do for mode in 0 1; do
  MI355PPO_CONV_R=$mode CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic 32768 4 2>&1 | grep '^{' | sed "s/^{/{\"conv_r\": $mode, /" | tee -a $O/ab.jsonl | cut -c1-420
done; done
