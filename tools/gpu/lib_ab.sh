#!/bin/bash
# Same-box A/B of VARIANT builds of the library (tools/build_variant.py -> tools/oldlib/<name>/libmi355ppo.so) against the in-tree one on the
# torch-free driver: hashes of every tensor of a minibatch (bit-identity), then alternating timing runs of the eleven launches.
#   VARIANTS="ntst ntld" [SIZES="32768"] [REPS=3] [PROBE=1] tools/gpu/lib_ab.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-libab}; rm -rf $O; mkdir -p $O; cd $R
if [ "${PROBE:-0}" = 1 ]; then timeout 300 tools/hbm_probe | tee $O/hbm_probe.jsonl | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%-24s grid %6d  avg %7.1f us  %5.0f GB/s (best %5.0f)' % (j['kernel'], j['grid'], j['avg_us'], j['GBps_avg'], j['GBps_best']))"; fi
libdir() { if [ $1 = base ]; then echo $R/cleanrl_amd/csrc; else echo $R/tools/oldlib/$1; fi; }
for M in ${HSIZES:-1027 32768}; do
  for v in base $VARIANTS; do
    LD_LIBRARY_PATH=$(libdir $v) CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 2 2>&1 | grep '^hash' > $O/h_${v}_$M.txt
    [ $v = base ] || { diff $O/h_base_$M.txt $O/h_${v}_$M.txt > /dev/null && echo "M=$M $v: hashes identical ($(wc -l < $O/h_base_$M.txt))" || echo "M=$M $v: HASHES DIFFER"; }
  done
done
for M in ${SIZES:-32768}; do for i in $(seq 1 ${REPS:-3}); do for v in base $VARIANTS; do
  LD_LIBRARY_PATH=$(libdir $v) CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $M 6 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('%-8s' % '$v', j['images'], ' '.join('%s %.1f' % (k[:-3], j[k]) for k in ('fwd1_us','fwd2_us','fwd3_us','fc_fwd_us','fc_dgrad_us','fc_wgrad_us','wgrad3_us','dgrad3_us','wgrad2_us','dgrad2_us','wgrad1_us') if k in j), 'sum', j['sum_ms'])" | tee -a $O/ab.txt
done; done; done
