#!/bin/bash
# round 4: kernel P with the zero-extended (subnormal bf16) frame operand (MI355PPO_P_ZEXT=0: byte -> f32 -> bf16 conversions):
# the MFMA subnormal check, bit-identity of dW1 / db1 by hash, per-kernel times, tests, bench A/B
mkdir -p gpurun_out
tools/mfma_denorm > gpurun_out/r04_mfma_denorm.json; cat gpurun_out/r04_mfma_denorm.json
out=gpurun_out/r04_kernel_p_zext_ab.txt; : > $out
for images in 32768 8192 4096; do
  for z in 1 0; do
    echo "images=$images p_zext=$z" >> $out
    CONV_TRAFFIC_HASH=1 MI355PPO_P_ZEXT=$z tools/conv_traffic $images 8 2>&1 | grep -i "hash dW1\|hash db1\|^{" | cut -c1-400 >> $out
  done
done
grep -v "^{" $out; grep "^{" $out | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['images'], 'wgrad1_us', d['wgrad1_us'])"
python -m pytest tests/test_gpu_cnn.py -q -x -k "wgrad or trunk_matches or full_minibatch" 2>&1 | tail -3
out2=gpurun_out/r04_kernel_p_zext_bench.jsonl; : > $out2
for rep in 1 2; do
for cfg in C B; do
  for z in 1 0; do
  MI355PPO_P_ZEXT=$z python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-pcie-inclusive 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'config':'$cfg','p_zext':$z,'value':d['value'],'ms_per_step':d['ms_per_step'],'phases':d.get('phases_ms')}))" >> $out2
  done
done
done
cat $out2
