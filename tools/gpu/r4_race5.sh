#!/bin/bash
# A/B under contention: the library with the old heads backward (ds_bpermute broadcast) vs the new one (scalar loads).
set -u
export TMPDIR=/tmp DET_HEADS=1 DET_UPDATES=6
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4race5; rm -rf $O; mkdir -p $O
cd $R
L=cleanrl_amd/csrc/libmi355ppo.so
cp $L /tmp/lib_new.so
for v in old new; do
  if [ $v = old ]; then cp tools/oldlib/heads_bperm/libmi355ppo.so $L; else cp /tmp/lib_new.so $L; fi
  export DET_REF=$O/ref_$v.json
  timeout 300 python tools/gpu/determinism.py 128 2>/dev/null | grep -E "MISMATCH" ; ls -la $DET_REF | awk '{print $5}'
  for rep in $(seq 1 8); do
    timeout 300 python tools/gpu/determinism.py 128 2>/dev/null | grep -E "MISMATCH" | sed "s/^/$v rep $rep a: /" & timeout 300 python tools/gpu/determinism.py 128 2>/dev/null | grep -E "MISMATCH" | sed "s/^/$v rep $rep b: /"; wait
  done
done
cp /tmp/lib_new.so $L
