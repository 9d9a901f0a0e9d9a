#!/bin/bash
# Kernel R's forwards with the next group's source requested evenly over the k-loop (default) against two loads per step (MI355PPO_R_SPREAD=0):
# output hashes (must agree bit for bit), per-launch times alternating, the s_memtime stamps of the default, bench lines alternating.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5spread; rm -rf $O; mkdir -p $O; cd $R
for M in 32768 1027 61; do
  for mode in 0 1; do
    MI355PPO_R_SPREAD=$mode CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 4 > $O/run_${M}_$mode.log 2>&1; echo "M=$M spread=$mode rc=$?"
  done
  diff <(grep '^hash' $O/run_${M}_0.log) <(grep '^hash' $O/run_${M}_1.log) > $O/hashdiff_$M.txt && echo "M=$M: hashes identical ($(grep -c '^hash' $O/run_${M}_1.log) tensors)" || { echo "M=$M: HASHES DIFFER"; cat $O/hashdiff_$M.txt; }
done
for M in 32768 1024 4096; do for i in 1 2; do for mode in 0 1; do
  MI355PPO_R_SPREAD=$mode CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 6 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(json.dumps({'spread': $mode, 'images': j['images'], 'fwd2_us': j['fwd2_us'], 'fwd3_us': j['fwd3_us'], 'dgrad3_us': j['dgrad3_us'], 'dgrad2_us': j['dgrad2_us'], 'sum_ms': j['sum_ms']}))" | tee -a $O/ab.jsonl
done; done; done
MI355PPO_R_TRACE=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic 32768 1 2>&1 >/dev/null | grep r_trace | awk '!seen[$0]++' > $O/trace_32768_spread.txt
grep -A10 "fwd_packed" $O/trace_32768_spread.txt | grep "k-steps\|visit 1 wave 0\|visit 2 wave 0" | cut -c1-330
if [ "${BENCH:-1}" = 1 ]; then
for i in 1 2; do for mode in 1 0; do
  MI355PPO_R_SPREAD=$mode timeout 300 python bench.py --no-cpu-baseline --no-pcie-inclusive --no-kernel-timing 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(json.dumps({'spread': $mode, 'value': j['value'], 'ms_per_step': j['ms_per_step'], 'phases_ms': {k: v for k, v in j['phases_ms'].items() if k != 'note'}}))" | tee -a $O/bench_ab.jsonl
done; done
fi
