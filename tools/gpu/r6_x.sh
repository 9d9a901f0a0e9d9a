#!/bin/bash
# round 6: generic A/B of two environment settings on the torch-free driver (env assignments in $A and $B; hashes of every tensor of a
# minibatch must agree at $HSIZES; alternating timing runs at $SIZES, all eleven launches printed); optionally a pytest selection first ($PYT)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r6x}; rm -rf $O; mkdir -p $O; cd $R
if [ -n "${PYT:-}" ]; then timeout ${PYT_TIMEOUT:-900} python -m pytest $PYT -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt; fi
for M in ${HSIZES:-1027 32768}; do
  env $A CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 2 2>&1 | grep '^hash' > $O/ha_$M.txt
  env $B CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 2 2>&1 | grep '^hash' > $O/hb_$M.txt
  diff $O/ha_$M.txt $O/hb_$M.txt > $O/hdiff_$M.txt && echo "M=$M: hashes identical ($(wc -l < $O/ha_$M.txt))" || { echo "M=$M: HASHES DIFFER"; cat $O/hdiff_$M.txt; }
done
for M in ${SIZES:-32768}; do for i in $(seq 1 ${REPS:-3}); do for mode in A B; do
  if [ $mode = A ]; then e="$A"; else e="$B"; fi
  env $e CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 6 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$mode', j['images'], ' '.join('%s %.1f' % (k[:-3], j[k]) for k in ('fwd1_us','fwd2_us','fwd3_us','fc_fwd_us','fc_dgrad_us','fc_wgrad_us','wgrad3_us','dgrad3_us','wgrad2_us','dgrad2_us','wgrad1_us') if k in j), 'sum', j['sum_ms'])" | tee -a $O/ab.txt
done; done; done
