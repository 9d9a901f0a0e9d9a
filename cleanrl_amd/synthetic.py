"""Seeded synthetic rollout tensors for parity tests and ``bench.py``.

No real Atari/MuJoCo environment can be installed in this image, so every measured or
parity-checked run uses these generators.  All draws come from a host ``numpy``
``RandomState(seed)`` so that the same seed yields the same tensors on any device.

Distributions follow what the reference's wrappers produce on Atari:
rewards in {-1, 0, +1} (``ClipRewardEnv``, cleanrl_utils/atari_wrappers.py:213-230),
``dones`` Bernoulli(1/200) stored as f32 (ppo_atari_multigpu.py:239,259), values N(0,1),
``logprobs = log_softmax(N(0,1) logits)[a]``, actions uniform stored as **f32**
(ppo_atari_multigpu.py:236,265 -- the reference keeps Discrete actions in a float tensor).
"""
from __future__ import annotations

import numpy as np
import torch


def rollout_scalars(T: int, N: int, n_actions: int = 4, seed: int = 1, done_p: float = 1.0 / 200.0,
                    device="cpu") -> dict:
    """(T,N) f32 ``rewards, dones, values, logprobs, actions`` + (N,) ``next_done, next_value``."""
    rs = np.random.RandomState(seed)
    rewards = rs.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(T, N), p=[0.05, 0.9, 0.05]).astype(np.float32)
    dones = (rs.random_sample((T, N)) < done_p).astype(np.float32)
    values = rs.standard_normal((T, N)).astype(np.float32)
    logits = rs.standard_normal((T, N, n_actions)).astype(np.float32)
    actions = rs.randint(0, n_actions, size=(T, N))
    m = logits.max(-1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(-1, keepdims=True))
    logprobs = np.take_along_axis(logits - lse, actions[..., None], -1)[..., 0].astype(np.float32)
    next_done = np.zeros((N,), np.float32)
    next_value = rs.standard_normal((N,)).astype(np.float32)
    out = dict(rewards=rewards, dones=dones, values=values, logprobs=logprobs,
               actions=actions.astype(np.float32), next_done=next_done, next_value=next_value)
    return {k: torch.from_numpy(v).to(device) for k, v in out.items()}


def atari_frames(n: int, seed: int = 1, frame_stacked: bool = False) -> np.ndarray:
    """``(n, 4, 84, 84)`` uint8 frames, i.i.d. uniform{0..255} (worst case for caches); with
    ``frame_stacked`` channels 0-2 of frame i equal channels 1-3 of frame i-1 (FrameStack(4),
    ppo_atari_multigpu.py:121)."""
    rs = np.random.RandomState(seed)
    if not frame_stacked:
        return rs.randint(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    planes = rs.randint(0, 256, size=(n + 3, 84, 84), dtype=np.uint8)
    idx = np.arange(n)[:, None] + np.arange(4)[None, :]
    return planes[idx]


def continuous_inputs(T: int, N: int, obs_dim: int, seed: int, done_p: float = 1.0 / 1000.0, unit_rewards: bool = False):
    """Vector-observation stand-in streams for the whole-iteration goldens of configs A / E: ``obs_seq`` (T+1, N, obs_dim)
    N(0, 1) f32 (0.5 N(0,1) for the CartPole-shaped case), ``step_done`` (T+1, N) Bernoulli(done_p) as f32 with row 0
    cleared, ``rewards`` (T, N) N(0,1) -- or all ones (CartPole pays 1 per step).  numpy legacy ``RandomState``: the same
    arrays on any box."""
    rs = np.random.RandomState(seed)
    obs_seq = rs.standard_normal((T + 1, N, obs_dim)).astype(np.float32)
    if unit_rewards:
        obs_seq *= np.float32(0.5)
    step_done = (rs.random_sample((T + 1, N)) < done_p).astype(np.float32)
    step_done[0] = 0.0
    rewards = np.ones((T, N), np.float32) if unit_rewards else rs.standard_normal((T, N)).astype(np.float32)
    return obs_seq, step_done, rewards
