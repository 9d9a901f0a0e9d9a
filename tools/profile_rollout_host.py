"""Host-side profile of the device-env rollout loop (cProfile): where the Python time per step goes."""
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
from cleanrl_amd.envs import DeviceSyntheticAtariVecEnv  # noqa: E402
from cleanrl_amd.learner import PPOLearner  # noqa: E402

dev = torch.device("cuda:0")
N, T = 1024, 128
torch.manual_seed(1); np.random.seed(1)
env = DeviceSyntheticAtariVecEnv(N, dev, seed=1)
agent = AtariAgent(env).to(dev)
args = learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=1, clip_coef=0.1)
L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, dev, sample_seed=1)
L.observe(0, env.obs_into(L.stage_obs), L.dones[0])
learner_smoke.rollout(L, env); torch.cuda.synchronize()
t0 = time.perf_counter(); learner_smoke.rollout(L, env); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("rollout: host issue %.1f ms, until GPU done %.1f ms (128 steps)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); learner_smoke.rollout(L, env); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
