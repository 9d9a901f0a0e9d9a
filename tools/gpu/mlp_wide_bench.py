"""K7 WIDE (csrc/mlp.hip, obs_dim > 32 or > 8 outputs) against the arrangement such agents ran on before round 6 -- the two nn.Sequential MLPs on library
GEMMs with torch autograd (here WITHOUT the distribution / loss kernels that sit behind them: a lower bound of that arrangement).  HIP events, eager launches.
    python tools/gpu/mlp_wide_bench.py            -> one JSON line per (obs_dim, n_out, rows)"""
import json
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleanrl_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def nets(O, nout):
    mk = lambda n: nn.Sequential(nn.Linear(O, 64), nn.Tanh(), nn.Linear(64, 64), nn.Tanh(), nn.Linear(64, n)).to(DEV)      # noqa: E731
    a, c = mk(nout), mk(1)
    for p in list(a.parameters()) + list(c.parameters()):
        p.grad = torch.zeros_like(p)
    return a, c


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    for O, D, M in ((376, 17, 64), (376, 17, 4096), (111, 8, 4096), (17, 6, 4096)):
        a, c = nets(O, D)
        pa, pc = ops.MlpNetPtrs(a), ops.MlpNetPtrs(c)
        Bf = max(M, 2048)
        obs = torch.randn(Bf, O, device=DEV)
        idx = torch.randperm(Bf, device=DEV)[:M]
        acts, lp, adv, ret, val = torch.randn(Bf, D, device=DEV), torch.randn(Bf, device=DEV) - 3, torch.randn(Bf, device=DEV), torch.randn(Bf, device=DEV), torch.randn(Bf, device=DEV)
        logstd, gls = torch.zeros(1, D, device=DEV), torch.zeros(1, D, device=DEV)
        fused = timed(lambda: ops.mlp_ppo_fwd_bwd(obs, idx, pa, pc, acts, lp, adv, ret, val, 0.2, 0.0, 0.5, True, True, logstd=logstd, logstd_grad=gls))
        dmu, dv = torch.randn(M, D, device=DEV), torch.randn(M, 1, device=DEV)

        def lib():
            x = obs[idx]
            mu, v = a(x), c(x)
            torch.autograd.backward([mu, v], [dmu, dv])
        library = timed(lib)
        x1 = torch.randn(64, O, device=DEV)
        act_fused = timed(lambda: ops.mlp_act_normal(x1, pa, pc, logstd, seed=1, offset=0))

        def act_lib():
            with torch.no_grad():
                mu, v = a(x1), c(x1)
                ops.normal_sample(mu, logstd, seed=1, offset=0)
        act_library = timed(act_lib)
        print(json.dumps({"obs_dim": O, "n_out": D, "minibatch_rows": M, "kernel": "WIDE" if (O > 32 or D > 8) else "narrow",
                          "update_fused_us": round(fused, 1), "update_library_gemms_autograd_only_us": round(library, 1),
                          "act_64_rows_fused_us": round(act_fused, 1), "act_64_rows_library_us": round(act_library, 1)}), flush=True)


if __name__ == "__main__":
    main()
