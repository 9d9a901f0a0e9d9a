#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "fcz or trunk") > $O/pytest_z.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_z.log | cut -c1-300
for m in 1024 512 256 2048 4095; do timeout 60 tools/conv_traffic $m 6 2>&1 | head -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], {k:d[k] for k in ('fwd1_us','fwd2_us','fwd3_us','fc_fwd_us')})"; done
python - <<'PY'
import torch, time
a=torch.randn(1024,3136,device='cuda'); W=torch.randn(512,3136,device='cuda'); b=torch.randn(512,device='cuda')
for M in (1024,512):
    x=a[:M].contiguous()
    for _ in range(5): torch._addmm_activation(b,x,W.t())
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(50): torch._addmm_activation(b,x,W.t())
    e1.record(); torch.cuda.synchronize(); print("library FC fwd M=%d: %.1f us"%(M,e0.elapsed_time(e1)*20))
PY
(timeout 900 python -m pytest tests/test_gpu_learner.py -m gpu -q -x) > $O/pytest_l.log 2>&1; echo "pytest learner rc=$?"; tail -4 $O/pytest_l.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-pcie-inclusive > $O/bench_C_fcsplit.json 2> $O/bench_C_fcsplit.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_C_fcsplit.json').read().strip().splitlines()[-1]); print(d['value'], d['phases_ms'])"
