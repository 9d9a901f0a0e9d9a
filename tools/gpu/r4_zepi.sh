#!/bin/bash
# Kernel Z with transposed accumulators / 16-byte epilogue stores against the previous build: output hashes (must be identical:
# the same MFMAs into the same accumulators), timings, then the CNN tests.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4zepi; rm -rf $O; mkdir -p $O; L=$R/cleanrl_amd/csrc/libmi355ppo.so
cd $R
cp $L /tmp/lib_new.so
for m in 32768 8192 2049 600; do
  cp tools/oldlib/base/libmi355ppo.so $L; CONV_TRAFFIC_HASH=1 timeout 120 tools/conv_traffic $m 1 2>&1 | grep "^hash" > $O/h_old_$m.txt
  cp /tmp/lib_new.so $L; CONV_TRAFFIC_HASH=1 timeout 120 tools/conv_traffic $m 1 2>&1 | grep "^hash" > $O/h_new_$m.txt
  d=$(diff $O/h_old_$m.txt $O/h_new_$m.txt | grep "^>" | awk '{print $3}' | tr '\n' ' ')
  echo "images=$m ($(wc -l < $O/h_new_$m.txt) tensors): differs in: ${d:-nothing}"
done
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then cp tools/oldlib/base/libmi355ppo.so $L; else cp /tmp/lib_new.so $L; fi
  for m in 32768 8192; do
    echo -n "$v "; timeout 120 tools/conv_traffic $m 5 2>&1 | grep '^{"images"' | sed "s/^{/{\"build\": \"$v\", /" | tee -a $O/zepi_ab.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], {k:v for k,v in d.items() if k.endswith('_us') or k=='sum_ms'})"
  done
done
done
cp /tmp/lib_new.so $L
timeout 900 python -m pytest tests/test_gpu_cnn.py -q -x 2>&1 | tail -4
