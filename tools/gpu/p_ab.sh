#!/bin/bash
# Same-box A/B of kernel P (conv1 weight gradient): the previous commit's library (tools/oldlib/base: whole-slab staging, 32 spilled
# VGPRs) against the in-tree one (slab staged in three pieces: 2 spilled), at two waves per SIMD and -- MI355PPO_P_OCC=1 -- at one
# wave per SIMD with 372 registers (no spill), grid 512 / 256 (the two switches exist only in the commit that introduced the pieces).  Bit-identity of the dumped results first, then timings, then traffic.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pab; mkdir -p $O; L=$R/cleanrl_amd/csrc/libmi355ppo.so
cd $R
cp $L /tmp/lib_new.so
for m in 32768 4096 2049 3; do
  cp tools/oldlib/base/libmi355ppo.so $L; timeout 120 tools/conv_traffic $m 1 /tmp/d0_$m.bin > /dev/null 2>&1; echo "base rc=$?"
  cp /tmp/lib_new.so $L; timeout 120 tools/conv_traffic $m 1 /tmp/d1_$m.bin > /dev/null 2>&1; echo "new rc=$?"
  cmp /tmp/d0_$m.bin /tmp/d1_$m.bin && echo "images=$m: bit-identical" || python tools/cmp_f32.py /tmp/d0_$m.bin /tmp/d1_$m.bin | tail -2
  MI355PPO_P_OCC=1 MI355PPO_P_GRID=256 timeout 120 tools/conv_traffic $m 1 /tmp/d2_$m.bin > /dev/null 2>&1; echo "occ1 rc=$?"
  cmp /tmp/d0_$m.bin /tmp/d2_$m.bin && echo "images=$m occ1/grid256: bit-identical" || python tools/cmp_f32.py /tmp/d0_$m.bin /tmp/d2_$m.bin | tail -2
done
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], {k:v for k,v in d.items() if k in ('wgrad1_us','sum_ms')})"; }
for rep in 1 2; do
  for cfg in "base 2 512" "new 2 512" "new 1 256" "new 1 512" "new 2 1024"; do
    set -- $cfg
    if [ $1 = base ]; then cp tools/oldlib/base/libmi355ppo.so $L; else cp /tmp/lib_new.so $L; fi
    for m in 32768 8192 4096; do
      echo -n "$1 occ=$2 grid=$3 "; MI355PPO_P_OCC=$2 MI355PPO_P_GRID=$3 timeout 120 tools/conv_traffic $m 5 2>&1 | head -1 | sed "s/^{/{\"lib\": \"$1\", \"occ\": $2, \"grid\": $3, /" | tee -a $O/p_ab.jsonl | show
    done
  done
done
cp /tmp/lib_new.so $L
for cfg in "2 512" "1 256"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_f; MI355PPO_P_OCC=$1 MI355PPO_P_GRID=$2 timeout 90 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_f -o t -- tools/conv_traffic 32768 3 > /dev/null 2>&1
    db=$(ls /tmp/pmc_f/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/${c}_occ$1_grid$2.csv
    echo "== $c occ=$1 grid=$2"; grep conv1p $O/${c}_occ$1_grid$2.csv | cut -c1-40,100-
  done
done
timeout 300 python -m pytest tests/test_gpu_cnn.py -q -x -k "conv1 or layer1 or kernel_p or wgrad" 2>&1 | tail -3
