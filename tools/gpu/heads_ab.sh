#!/bin/bash
# heads forward / backward at 32,768 and 4,096 rows: the in-tree library against tools/oldlib/<V...>, swapped in turn (HIP events around 50 launches)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-heads}; rm -rf $O; mkdir -p $O; cd $R
L=$R/cleanrl_amd/csrc/libmi355ppo.so; cp $L /tmp/lib_tree.so
cat > /tmp/heads_t.py <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
from cleanrl_amd import _lib, cnn
lib = _lib.load(); dev = torch.device("cuda:0"); P = lambda t: t.data_ptr()
g = torch.Generator().manual_seed(1)
for M in (32768, 4096):
    A, H = 4, 512
    h = torch.relu(torch.randn(M, H, generator=g)).to(dev); Wa = torch.randn(A, H, generator=g).to(dev); ba = torch.randn(A, generator=g).to(dev)
    Wc = torch.randn(1, H, generator=g).to(dev); bc = torch.randn(1, generator=g).to(dev)
    logits = torch.empty(M, A, device=dev); value = torch.empty(M, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.mi355ppo_heads_fwd_f32(P(h), P(Wa), P(ba), P(Wc), P(bc), P(logits), P(value), M, A, H, st)
    for _ in range(5): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); e1.synchronize()
    fwd_us = round(e0.elapsed_time(e1) * 20, 2)
    # backward (the ReLU variant with the amax record: what the learner's update runs), dz with a padded row pitch of 516
    dl = torch.randn(M, A, generator=g).to(dev); dv = torch.randn(M, generator=g).to(dev)
    dz = torch.empty(M, 516, device=dev); dWa = torch.empty(A, H, device=dev); dba = torch.empty(A, device=dev); dWc = torch.empty(1, H, device=dev)
    dbc = torch.empty(1, device=dev); dbh = torch.empty(H, device=dev); rec = torch.zeros(256, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.mi355ppo_heads_bwd_workspace_bytes(M, A), dtype=torch.uint8, device=dev)
    b = lambda: lib.mi355ppo_heads_bwd_relu_amax_f32(P(h), P(Wa), P(Wc), P(dl), P(dv), P(dz), 516, P(dWa), P(dba), P(dWc), P(dbc), P(dbh), M, A, H, P(ws), ws.numel(), P(rec), st)
    for _ in range(5): assert b() == 0
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): b()
    e1.record(); e1.synchronize()
    hb = 0
    for t in (dz[:, :512].contiguous(), dWa, dba, dWc, dbc, dbh): hb ^= int(t.view(torch.int32).sum(dtype=torch.int64))
    print(json.dumps(dict(lib=sys.argv[1], M=M, fwd_us=fwd_us, bwd_us=round(e0.elapsed_time(e1) * 20, 2), hash=int(logits.view(torch.int32).sum(dtype=torch.int64)) ^ int(value.view(torch.int32).sum(dtype=torch.int64)), hash_bwd=hb)), flush=True)
PY
for i in 1 2 3; do for v in tree ${V:-prev}; do
  if [ $v = tree ]; then cp /tmp/lib_tree.so $L; else cp tools/oldlib/$v/libmi355ppo.so $L; fi
  python /tmp/heads_t.py $v 2>&1 | grep '^{' | tee -a $O/heads_ab.jsonl
done; done
cp /tmp/lib_tree.so $L
