#!/bin/bash
# A prefetched two k-steps ahead (MI355PPO_Z_PF=2) vs one: bit-identity (hashes), per-launch times, the f16x2 tests, bench A/B
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5pf; rm -rf $O; mkdir -p $O; cd $R
for pf in 1 2 1 2; do
  CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 MI355PPO_Z_PF=$pf timeout 120 tools/conv_traffic 32768 4 > $O/t_$pf.txt 2>&1
  grep '^{' $O/t_$pf.txt | sed "s/^{/{\"pf\": $pf, /" >> $O/pf_ab.jsonl
done
cut -c1-420 $O/pf_ab.jsonl
grep '^hash' $O/t_1.txt | md5sum; grep '^hash' $O/t_2.txt | md5sum
for pf in 1 2; do CONV_TRAFFIC_F16=1 MI355PPO_Z_PF=$pf timeout 60 tools/conv_traffic 1024 6 2>&1 | grep '^{' | cut -c1-400; done
MI355PPO_Z_PF=2 timeout 600 python -m pytest tests/test_gpu_f16x2.py -q -p no:cacheprovider 2>&1 | tail -3
for pf in 1 2 1 2; do
  MI355PPO_Z_PF=$pf timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-pcie-inclusive --no-kernel-timing 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'pf': $pf, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'phases': d.get('phases_ms')}))" | tee -a $O/bench_pf_ab.jsonl
done
