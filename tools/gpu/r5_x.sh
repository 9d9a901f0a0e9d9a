#!/bin/bash
# scratch A/B on the torch-free driver: env assignments in $A and $B, sizes in $SIZES; hashes must agree
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5x; rm -rf $O; mkdir -p $O; cd $R
for M in ${HSIZES:-1027 32768}; do
  env $A CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 2 2>&1 | grep '^hash' > $O/ha_$M.txt
  env $B CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 2 2>&1 | grep '^hash' > $O/hb_$M.txt
  diff $O/ha_$M.txt $O/hb_$M.txt > /dev/null && echo "M=$M: hashes identical ($(wc -l < $O/ha_$M.txt))" || echo "M=$M: HASHES DIFFER"
done
for M in ${SIZES:-32768}; do for i in 1 2 3; do for mode in A B; do
  if [ $mode = A ]; then e="$A"; else e="$B"; fi
  env $e CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 6 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$mode', j['images'], ' '.join('%s %.1f' % (k[:-3], j[k]) for k in ('fwd2_us','fwd3_us','dgrad3_us','dgrad2_us','wgrad3_us','wgrad2_us','wgrad1_us','fwd1_us')), 'sum', j['sum_ms'])" | tee -a $O/ab.txt
done; done; done

