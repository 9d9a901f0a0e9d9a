#!/bin/bash
# Torch-free A/B of conv kernel variants (tools/conv_traffic.cpp): timing JSON per setting + dump comparison.
set -u
export TMPDIR=/tmp
B=tools/conv_traffic
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 30 $B ${N:-32768} ${R:-4} /tmp/$tag.bin | tail -2 | head -1; }
N=32768 R=4 run base MI355PPO_WGRAD=2
N=32768 R=4 run rows MI355PPO_WGRAD=3
N=32768 R=4 run taps1 MI355PPO_WGRAD=3 MI355PPO_WGRAD_TAPS=1
N=32768 R=4 run taps2 MI355PPO_WGRAD=3 MI355PPO_WGRAD_TAPS=2
python tools/cmp_f32.py /tmp/base.bin /tmp/taps1.bin | grep -v samples
python tools/cmp_f32.py /tmp/base.bin /tmp/taps2.bin | grep WORST
for n in 1 2 7 1001; do
  N=$n R=1 run s_base_$n MI355PPO_WGRAD=2 > /dev/null
  N=$n R=1 run s_new_$n MI355PPO_WGRAD=3 MI355PPO_WGRAD_TAPS=1 > /dev/null
  echo "images=$n: $(python tools/cmp_f32.py /tmp/s_base_$n.bin /tmp/s_new_$n.bin | grep WORST)"
done
