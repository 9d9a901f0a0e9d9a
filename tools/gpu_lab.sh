#!/bin/bash
# Torch-free A/B loop for conv-kernel work (tools/conv_traffic.cpp; ~12 s of GPU-box time per call):
#   * tuning switches of ONE build against each other (timing JSON + dump comparison with tolerance), and/or
#   * two BUILDS against each other: copy the previous libmi355ppo.so to tools/oldlib/ before rebuilding, then the `old`
#     lines below run it through LD_LIBRARY_PATH (epilogue-only changes must stay bit-identical: `cmp`).
# Example below: kernel T with 4-byte loads (MI355PPO_WGRAD_TAPS=1) against the default paired 8-byte loads (3).
set -u
export TMPDIR=/tmp
B=tools/conv_traffic
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 40 $B ${N:-32768} ${R:-4} /tmp/$tag.bin | tail -2 | head -1; }
N=32768 run default MI355PPO_WGRAD_TAPS=3
N=32768 run pair MI355PPO_WGRAD_TAPS=1
python tools/cmp_f32.py /tmp/default.bin /tmp/pair.bin | grep 'dW2\|db2\|WORST'
for n in 1 7 1001; do
  N=$n R=1 run s_default_$n MI355PPO_WGRAD_TAPS=3 > /dev/null
  N=$n R=1 run s_pair_$n MI355PPO_WGRAD_TAPS=1 > /dev/null
  echo "images=$n: $(python tools/cmp_f32.py /tmp/s_default_$n.bin /tmp/s_pair_$n.bin | grep WORST)"
done
if [ -f tools/oldlib/libmi355ppo.so ]; then
  LD_LIBRARY_PATH=tools/oldlib timeout 40 $B 32768 4 /tmp/old.bin | tail -2 | head -1
  cmp /tmp/old.bin /tmp/default.bin && echo "old build vs new build: dumps bit-identical"
fi
