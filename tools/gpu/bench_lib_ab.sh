#!/bin/bash
# bench.py (config C, the driver's step counts, no CPU / host-env legs) with the in-tree library and tools/oldlib/<V> (default: head, the previous
# commit's) swapped in turn on one box: the whole-iteration counterpart of lib_ab.sh
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-benchab}; rm -rf $O; mkdir -p $O; cd $R
V=${V:-head}; L=$R/cleanrl_amd/csrc/libmi355ppo.so; cp $L /tmp/lib_tree.so
for i in $(seq 1 ${REPS:-3}); do for v in tree $V; do
  if [ $v = tree ]; then cp /tmp/lib_tree.so $L; else cp tools/oldlib/$V/libmi355ppo.so $L; fi
  timeout 300 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print('$v', round(j['value']), 'ms', round(j['ms_per_step'],2), 'rollout', round(j['phases_ms']['rollout_incl_gae'],2), 'update', round(j['phases_ms']['update'],2), 'dgrad2', round(k['conv2_dgrad@32768']['avg_us'],1), 'Q', round(k['conv1_fwd@32768']['avg_us'],1), 'gae_hbm', round(k['gae_hbm_regime']['frac'],3))" | tee -a $O/bench_ab.txt
done; done
cp /tmp/lib_tree.so $L
