"""Where the host time of the one-thread lane driver goes: cProfile over two rollouts of tools/host_env_bench's arrangement
(worker-process envs, captured lane steps, frames DMA'd from pinned shared memory).  python tools/profile_lanes_host.py [groups]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanrl_amd import learner_smoke  # noqa: E402
from cleanrl_amd.agents import AtariAgent  # noqa: E402
from cleanrl_amd.env_workers import ProcessVecEnv  # noqa: E402
from cleanrl_amd.learner import PPOLearner  # noqa: E402
from cleanrl_amd.pipeline import GroupedRollout, split_env_groups  # noqa: E402

if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    N, T, dev = 1024, 128, torch.device("cuda:0")
    envs = split_env_groups(lambda g, n: ProcessVecEnv(("cleanrl_amd.envs", "SyntheticAtariVecEnv", dict(num_envs=n, seed=1 + g * n, api="gym"))), N, K)
    torch.manual_seed(1)
    L = PPOLearner(AtariAgent(envs[0]).to(dev), learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=4, clip_coef=0.1),
                   envs[0].single_observation_space, envs[0].single_action_space, N, dev, sample_seed=1)
    roll = GroupedRollout(L, K, frame_delta=True)
    for g, e in enumerate(envs):
        roll.first_observation(g, e.reset())
    roll.capture()
    print("pinned:", all([e.pin() for e in envs]))
    roll.run_async(envs)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(2):
        roll.run_async(envs)
    torch.cuda.synchronize()
    pr.disable()
    print("rollout ms:", (time.perf_counter() - t0) * 500)
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    for e in envs:
        e.close()
