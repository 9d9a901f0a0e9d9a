"""PCIe-inclusive mode: the same learner fed by a HOST vector env (the reference's arrangement: env on the host, one
D2H of the actions and one H2D of the frames per step).  Frames come from envs.SyntheticAtariVecEnv (numpy, host cores),
go through PPOLearner.observe(): pinned staging + uint8 H2D on a side stream + relayout kernel.

    python tools/host_env_bench.py [--num-envs 1024] [--iters 2]      -> one JSON line
This number is NOT bench.py's `value` (which starts with inputs resident in HBM); it is recorded in DESIGN.md §5.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanrl_amd import learner_smoke  # noqa: E402
from cleanrl_amd.agents import AtariAgent  # noqa: E402
from cleanrl_amd.envs import SyntheticAtariVecEnv  # noqa: E402
from cleanrl_amd.learner import PPOLearner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=1024)
    ap.add_argument("--num-steps", type=int, default=128)
    ap.add_argument("--iters", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, T = a.num_envs, a.num_steps
    env = SyntheticAtariVecEnv(N, seed=1)
    torch.manual_seed(1)
    np.random.seed(1)
    agent = AtariAgent(env).to(dev)
    args = learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=4, clip_coef=0.1)
    L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, dev, sample_seed=1)
    obs, _ = env.reset(seed=1)
    L.observe(0, obs, np.zeros(N, np.float32))
    t_env = t_roll = t_upd = 0.0
    for it in range(a.iters + 1):
        if it == 1:                                   # iteration 0 is warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t_env = t_roll = t_upd = 0.0
        r0 = time.perf_counter()
        for step in range(T):
            act = L.act(step).cpu().numpy()           # D2H + sync, as the reference (:269)
            e0 = time.perf_counter()
            obs, reward, term, trunc, _ = env.step(act)
            t_env += time.perf_counter() - e0
            L.store_reward(step, reward)
            L.observe(step + 1, obs, np.logical_or(term, trunc))
        L.finish_rollout()
        torch.cuda.synchronize()
        t_roll += time.perf_counter() - r0
        u0 = time.perf_counter()
        L.update(args.learning_rate)
        L.start_iteration()
        torch.cuda.synchronize()
        t_upd += time.perf_counter() - u0
    el = time.perf_counter() - t0
    print(json.dumps({"mode": "host env (numpy) -> pinned -> uint8 H2D side stream", "num_envs": N, "num_steps": T,
                      "iters": a.iters, "sps": N * T * a.iters / el, "ms_per_iter": el / a.iters * 1e3,
                      "rollout_ms": t_roll / a.iters * 1e3, "of_which_host_env_ms": t_env / a.iters * 1e3,
                      "update_ms": t_upd / a.iters * 1e3, "h2d_bytes_per_step": N * 28224}))


if __name__ == "__main__":
    main()
