"""K7's minibatch launch pair (mlp_ppo + fold) at config E's shape over rows-per-wave: time per minibatch (HIP events, 200 launches)."""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleanrl_amd import envs as E, ops  # noqa: E402
from cleanrl_amd.agents import ContinuousAgent, fused_mlp_ptrs  # noqa: E402

dev = torch.device("cuda:0")
env = SimpleNamespace(single_observation_space=E.Box(-np.inf, np.inf, (17,)), single_action_space=E.Box(-1.0, 1.0, (6,)))
torch.manual_seed(1)
agent = ContinuousAgent(env).to(dev)
for p in agent.parameters():
    p.grad = torch.zeros_like(p)
pa, pc = fused_mlp_ptrs(agent)
B, M = 131072, 4096
g = torch.Generator(device=dev).manual_seed(3)
obs = torch.randn(B, 17, device=dev, generator=g)
act = torch.randn(B, 6, device=dev, generator=g)
lp, adv, ret, val = (torch.randn(B, device=dev, generator=g) for _ in range(4))
inds = torch.randperm(B, device=dev, generator=g)[:M]
md = torch.tensor([0.0, 1.0], device=dev)
sc = torch.zeros(7, device=dev)
for rpb in [int(x) for x in os.environ.get("K7_RPB", "0,4,8,16,32,64").split(",")]:
    def call():
        ops.mlp_ppo_fwd_bwd(obs, inds, pa, pc, act, lp, adv, ret, val, 0.2, 0.0, 0.5, True, True, adv_mean_den=md, scalars_out=sc,
                            logstd=agent.actor_logstd.detach(), logstd_grad=agent.actor_logstd.grad, rows_per_block=rpb)
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"rows_per_block": rpb, "us_per_minibatch_incl_host_issue": round(e0.elapsed_time(e1) * 1000 / 200, 2)}))
