#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pab2; mkdir -p $O; L=$R/cleanrl_amd/csrc/libmi355ppo.so
cd $R
cp $L /tmp/lib_new.so
for m in 32768 2049 3; do
  cp tools/oldlib/base/libmi355ppo.so $L; timeout 120 tools/conv_traffic $m 1 /tmp/d0_$m.bin > /dev/null 2>&1; echo "base rc=$?"
  cp /tmp/lib_new.so $L; timeout 120 tools/conv_traffic $m 1 /tmp/d1_$m.bin > /dev/null 2>&1; echo "new rc=$?"
  cmp /tmp/d0_$m.bin /tmp/d1_$m.bin && echo "images=$m: bit-identical" || python tools/cmp_f32.py /tmp/d0_$m.bin /tmp/d1_$m.bin | tail -2
done
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], {k:v for k,v in d.items() if k in ('wgrad1_us','sum_ms')})"; }
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then cp tools/oldlib/base/libmi355ppo.so $L; else cp /tmp/lib_new.so $L; fi
    for m in 32768 8192 4096; do
      echo -n "$v "; timeout 120 tools/conv_traffic $m 5 2>&1 | head -1 | sed "s/^{/{\"lib\": \"$v\", /" | tee -a $O/p_ring2_ab.jsonl | show
    done
  done
done
cp /tmp/lib_new.so $L
