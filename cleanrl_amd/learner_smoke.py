"""Tiny end-to-end pass of the learner on a GPU, used by ``__graft_entry__.smoke()``: NatureCNN agent,
device-resident synthetic Atari streams, one rollout + GAE + one update epoch; returns the tensors a
caller needs to check the pass against the oracle."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from .agents import AtariAgent
from .envs import DeviceSyntheticAtariVecEnv
from .learner import PPOLearner


def default_args(**kw):
    d = dict(num_steps=8, num_minibatches=2, update_epochs=1, learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95,
             clip_coef=0.1, ent_coef=0.01, vf_coef=0.5, norm_adv=True, clip_vloss=True, max_grad_norm=0.5, target_kl=None)
    d.update(kw)
    return SimpleNamespace(**d)


def rollout(learner: PPOLearner, env: DeviceSyntheticAtariVecEnv) -> None:
    """T steps with the device env writing straight into the rollout-storage rows."""
    T = learner.T
    if getattr(learner, "_rollout_graphs", None) is not None and learner._graph_env is env:
        learner.replay_rollout()                                                     # T graph launches (PPOLearner.capture_rollout)
        learner.finish_rollout()
        return
    for step in range(T):
        learner.act(step)
        obs_dst, done_dst = learner._slot(step + 1)
        if learner.relayout and hasattr(env, "step_into_rows"):
            env.step_into_rows(obs_dst, learner.rewards[step], done_dst)                 # the rollout row's own layout: no relayout launch
            continue
        frames = env.step_into(learner.stage_obs, learner.rewards[step], done_dst)   # channel-planar uint8 in HBM
        learner.observe(step + 1, frames, done_dst)                                  # relayout into the rollout row
    learner.finish_rollout()


def run(device, num_envs: int = 16, seed: int = 1, verbose: bool = True):
    torch.manual_seed(seed)
    np.random.seed(seed)
    env = DeviceSyntheticAtariVecEnv(num_envs, device, seed=seed)
    agent = AtariAgent(env).to(device)
    args = default_args()
    learner = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, num_envs, device,
                         sample_seed=seed)
    learner.observe(0, env.obs_into(learner.stage_obs), learner.dones[0])
    rollout(learner, env)
    before = learner.flat.params.clone()
    metrics = learner.update(args.learning_rate)
    learner.start_iteration()
    learner.flat.check_views()
    delta = (learner.flat.params - before).abs().max().item()
    assert np.isfinite(metrics["loss"]) and delta > 0, "the update must move the parameters"
    if verbose:
        print("learner smoke:", {k: round(v, 5) if isinstance(v, float) else v for k, v in metrics.items()},
              "max|dparam|=%.3g" % delta)
    return learner, metrics
