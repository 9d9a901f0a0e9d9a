"""Where does the captured update part from the eager one at config B's shape?  (tests/test_gpu_learner.py::test_captured_update_slots_...)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleanrl_amd import envs as E, learner_smoke  # noqa: E402
from cleanrl_amd.agents import AtariAgent  # noqa: E402
from cleanrl_amd.learner import PPOLearner  # noqa: E402

DEV = torch.device("cuda:0")
N, T, nmb, epochs = 128, 128, 4, 4


def make(graphs):
    torch.manual_seed(4)
    env = E.DeviceSyntheticAtariVecEnv(N, DEV, seed=6, done_p=0.1)
    agent = AtariAgent(env).to(DEV)
    args = learner_smoke.default_args(num_steps=T, num_minibatches=nmb, update_epochs=epochs)
    L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, DEV, sample_seed=8)
    L.observe(0, env.obs_into(L.stage_obs), L.dones[0])
    if graphs:
        L.capture_update()
    return L, env


(Le, enve), (Lg, envg) = make(False), make(True)
print("split", os.environ.get("MI355PPO_SPLIT", "f16x2"), "init equal", torch.equal(Le.flat.params, Lg.flat.params))
for it in range(3):
    learner_smoke.rollout(Le, enve)
    learner_smoke.rollout(Lg, envg)
    for name in ("obs", "actions", "logprobs", "values", "rewards", "dones", "advantages", "returns"):
        a, b = getattr(Le, name), getattr(Lg, name)
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            print(f"it {it}: rollout {name} differs: max {d.max().item():.3e}, first step {int((d.reshape(T, -1).amax(1) > 0).nonzero()[0])}")
    np.random.seed(100 + it)
    me = Le.update(2.5e-4 * (1 - it / 3))
    np.random.seed(100 + it)
    mg = Lg.update(2.5e-4 * (1 - it / 3))
    Le.start_iteration(); Lg.start_iteration()
    sc_e, sc_g = Le._scalars[:16].cpu(), Lg._scalars[:16].cpu()
    first = [k for k in range(16) if not torch.equal(sc_e[k], sc_g[k])]
    print(f"it {it}: params equal {torch.equal(Le.flat.params, Lg.flat.params)}, max diff {(Le.flat.params - Lg.flat.params).abs().max().item():.3e}, "
          f"first differing scalar row {first[:1]}, adam m equal {torch.equal(Le.flat.exp_avg, Lg.flat.exp_avg)}")
    if first:
        k = first[0]
        print("   eager", sc_e[k].tolist(), "\n   graph", sc_g[k].tolist())
