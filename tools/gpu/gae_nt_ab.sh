#!/bin/bash
# GAE column kernel in its HBM regime: the in-tree library against tools/oldlib/head (the previous commit's), alternating, HIP events
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-gaent}; rm -rf $O; mkdir -p $O; cd $R
L=$R/cleanrl_amd/csrc/libmi355ppo.so; cp $L /tmp/lib_tree.so
cat > /tmp/gae_sweep.py <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
from cleanrl_amd import ops, synthetic
DEV = torch.device("cuda:0"); T = 128
def ev_us(fn, reps):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for N in (1 << 16, 1 << 18, 1 << 20, 1 << 22):
    base = {k: v.to(DEV) for k, v in synthetic.rollout_scalars(T, 4096, 4, seed=1).items()}
    rep = N // 4096
    s = {k: (v.repeat(1, rep) if v.dim() == 2 else v.repeat(rep)).contiguous() for k, v in base.items()}
    adv, ret = torch.empty_like(s["rewards"]), torch.empty_like(s["rewards"])
    f = lambda: ops.gae(s["rewards"], s["dones"], s["values"], s["next_done"], s["next_value"], 0.99, 0.95, adv, ret)
    us = ev_us(f, 20); nbytes = 20 * T * N + 8 * N
    print(json.dumps(dict(lib=sys.argv[1], N=N, us=round(us, 1), GBps=round(nbytes / us / 1e3), frac=round(nbytes / us / 1e3 / 8000, 3), hash=int(adv.view(torch.int32).sum(dtype=torch.int64)))), flush=True)
    del s, adv, ret
PY
for i in 1 2; do for v in tree head; do
  if [ $v = tree ]; then cp /tmp/lib_tree.so $L; else cp tools/oldlib/head/libmi355ppo.so $L; fi
  python /tmp/gae_sweep.py $v 2>&1 | grep '^{' | tee -a $O/gae_nt_ab.jsonl
done; done
cp /tmp/lib_tree.so $L
