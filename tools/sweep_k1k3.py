"""K1 (GAE) N-sweep and K3 (fused loss) M-sweep for rocprofv3 kernel-duration evidence (run under
``rocprofv3 --kernel-trace``; also prints HIP-event timings as JSON lines for cross-checking).
    python tools/sweep_k1k3.py [reps]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cleanrl_amd import ops, synthetic  # noqa: E402

DEV = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def ev_us(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    T = 128
    for N in (1024, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
        base = {k: v.to(DEV) for k, v in synthetic.rollout_scalars(T, min(N, 4096), 4, seed=1).items()}
        rep = max(N // 4096, 1)
        s = {k: (v.repeat(1, rep) if v.dim() == 2 else v.repeat(rep)).contiguous() for k, v in base.items()}
        adv, ret = torch.empty_like(s["rewards"]), torch.empty_like(s["rewards"])
        f = lambda: ops.gae(s["rewards"], s["dones"], s["values"], s["next_done"], s["next_value"], 0.99, 0.95, adv, ret)
        us = ev_us(f, REPS)
        nbytes = 20 * T * N + 8 * N
        print(json.dumps(dict(kernel="gae", T=T, N=N, algorithmic_bytes=nbytes, event_us_per_call=us, GBps=nbytes / us / 1e3)), flush=True)
        del s, adv, ret
    A = 4
    g = torch.Generator(device="cpu").manual_seed(3)
    # K3 modes: "perm" = mb_inds a random permutation of a flat batch as large as the minibatch (every gather a new line once
    # the five arrays outgrow L2); "resident" = random rows of a 131,072-row flat batch (config C's: the five arrays stay in L2);
    # "identity" = mb_inds NULL (pure streaming).
    for mode in ("perm", "resident", "identity"):
        for M in (4096, 32768, 1 << 17, 1 << 20, 1 << 22, 1 << 24):
            B = M if mode != "resident" else 131072
            logits = torch.randn(M, A, generator=g).to(DEV)
            value = torch.randn(M, generator=g).to(DEV)
            if mode == "perm":
                inds = torch.randperm(B, generator=g).to(DEV)
            elif mode == "resident":
                inds = torch.randint(0, B, (M,), generator=g).to(DEV)
            else:
                inds = None
            b_actions = torch.randint(0, A, (B,), generator=g).float().to(DEV)
            b_lp = (torch.randn(B, generator=g) * 0.1 - 1.4).to(DEV)
            b_adv, b_ret, b_val = (torch.randn(B, generator=g).to(DEV) for _ in range(3))
            sc = torch.empty(7, device=DEV)
            dl, dv = torch.empty_like(logits), torch.empty_like(value)
            f = lambda: ops.ppo_loss_categorical(logits, value, inds, b_actions, b_lp, b_adv, b_ret, b_val, 0.1, 0.01, 0.5, True, True,
                                                 scalars_out=sc, dlogits_out=dl, dvalue_out=dv)
            us = ev_us(f, REPS)
            us2 = None
            if True:                # the learner's mode: statistics hoisted to once per epoch, scalar fold deferred to once per update
                md = ops.adv_stats(b_adv, inds, M)[0]
                slots = ops.LossSlots(1, DEV)
                f2 = lambda: ops.ppo_loss_categorical(logits, value, inds, b_actions, b_lp, b_adv, b_ret, b_val, 0.1, 0.01, 0.5, True, True,
                                                      dlogits_out=dl, dvalue_out=dv, adv_mean_den=md, slot=(slots, 0))
                us2 = ev_us(f2, REPS)
                slots.fold(1, sc.view(1, 7))
                # round 3: the five behaviour scalars as one 32-byte packed row per flat index (one gather instead of five)
                pack = ops.batch_pack(b_actions, b_lp, b_adv, b_ret, b_val)
                us_pack = ev_us(lambda: ops.batch_pack(b_actions, b_lp, b_adv, b_ret, b_val, out=pack), REPS)
                f3 = lambda: ops.ppo_loss_categorical_packed(logits, value, inds, pack, 0.1, 0.01, 0.5, True, True, dlogits_out=dl,
                                                             dvalue_out=dv, adv_mean_den=md, slot=(slots, 0))
                us3 = ev_us(f3, REPS)
                slots.fold(1, sc.view(1, 7))
                del pack
            nbytes = (8 * A + 28 + (8 if inds is not None else 0)) * M
            print(json.dumps(dict(kernel="loss_categorical", mode=mode, M=M, B=B, algorithmic_bytes=nbytes, event_us_per_call=us,
                                  GBps=nbytes / us / 1e3, event_us_per_call_learner_mode=us2, GBps_learner_mode=nbytes / us2 / 1e3,
                                  event_us_per_call_packed_rows=us3, GBps_packed_rows=nbytes / us3 / 1e3,
                                  batch_pack_event_us=us_pack)), flush=True)


if __name__ == "__main__":
    main()
