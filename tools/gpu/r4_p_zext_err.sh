#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r04_kernel_p_zext_error.jsonl; : > $out
MI355PPO_P_ZEXT=1 python tools/gpu/p_zext_err.py /tmp/z1.pt >> $out
MI355PPO_P_ZEXT=0 python tools/gpu/p_zext_err.py /tmp/z0.pt >> $out
python tools/gpu/p_zext_err.py --compare /tmp/z1.pt /tmp/z0.pt >> $out
cat $out
