"""Atari pre-processing wrappers used by ``ppo_atari.py`` / ``ppo_atari_multigpu.py`` when gymnasium +
ale_py are installed (role of cleanrl_utils/atari_wrappers.py: NoopResetEnv :62, FireResetEnv :94,
EpisodicLifeEnv :117, MaxAndSkipEnv :168, ClipRewardEnv :213).  Host-side Python, outside the GPU hot path.
These follow the standard DQN-Nature preprocessing (Mnih et al. 2015; Machado et al. 2018).
gymnasium is imported lazily: this image does not ship it, so the module must stay importable without it.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    import gymnasium as gym

    _Wrapper, _RewardWrapper = gym.Wrapper, gym.RewardWrapper
except Exception:  # pragma: no cover
    gym = None

    class _Wrapper:  # minimal stand-ins so the classes below can be defined and unit-tested with fake envs
        def __init__(self, env):
            self.env = env

        @property
        def unwrapped(self):
            return getattr(self.env, "unwrapped", self.env)

        def reset(self, **kw):
            return self.env.reset(**kw)

        def step(self, action):
            return self.env.step(action)

    class _RewardWrapper(_Wrapper):
        def step(self, action):
            obs, reward, term, trunc, info = self.env.step(action)
            return obs, self.reward(reward), term, trunc, info


class NoopResetEnv(_Wrapper):
    """Start each episode with a random number (1..noop_max) of no-op actions (action 0)."""

    def __init__(self, env, noop_max: int = 30):
        super().__init__(env)
        self.noop_max = noop_max
        self.noop_action = 0

    def reset(self, **kwargs):
        obs, info = self.env.reset(**kwargs)
        rng = getattr(self.unwrapped, "np_random", np.random)
        n = int(rng.integers(1, self.noop_max + 1)) if hasattr(rng, "integers") else int(rng.randint(1, self.noop_max + 1))
        for _ in range(n):
            obs, _, terminated, truncated, info = self.env.step(self.noop_action)
            if terminated or truncated:
                obs, info = self.env.reset(**kwargs)
        return obs, info


class FireResetEnv(_Wrapper):
    """Press FIRE (then action 2) after reset for games that need it to start."""

    def reset(self, **kwargs):
        self.env.reset(**kwargs)
        obs, _, terminated, truncated, _ = self.env.step(1)
        if terminated or truncated:
            self.env.reset(**kwargs)
        obs, _, terminated, truncated, _ = self.env.step(2)
        if terminated or truncated:
            self.env.reset(**kwargs)
        return obs, {}


class EpisodicLifeEnv(_Wrapper):
    """Treat the loss of a life as the end of an episode; only reset the emulator on true game over."""

    def __init__(self, env):
        super().__init__(env)
        self.lives = 0
        self.was_real_done = True

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        self.was_real_done = terminated or truncated
        lives = self.unwrapped.ale.lives()
        if 0 < lives < self.lives:
            terminated = True
        self.lives = lives
        return obs, reward, terminated, truncated, info

    def reset(self, **kwargs):
        if self.was_real_done:
            obs, info = self.env.reset(**kwargs)
        else:  # continue the same game from the lost-life state with a no-op
            obs, _, terminated, truncated, info = self.env.step(0)
            if terminated or truncated:
                obs, info = self.env.reset(**kwargs)
        self.lives = self.unwrapped.ale.lives()
        return obs, info


class MaxAndSkipEnv(_Wrapper):
    """Repeat the action `skip` times, sum the rewards, return the pixel-wise max of the last two frames."""

    def __init__(self, env, skip: int = 4):
        super().__init__(env)
        self._skip = skip
        self._buf = None

    def step(self, action):
        total, terminated, truncated, info = 0.0, False, False, {}
        for i in range(self._skip):
            obs, reward, terminated, truncated, info = self.env.step(action)
            if self._buf is None:
                self._buf = np.zeros((2,) + np.asarray(obs).shape, dtype=np.asarray(obs).dtype)
            if i == self._skip - 2:
                self._buf[0] = obs
            if i == self._skip - 1:
                self._buf[1] = obs
            total += float(reward)
            if terminated or truncated:
                break
        return self._buf.max(axis=0), total, terminated, truncated, info


class ClipRewardEnv(_RewardWrapper):
    """Rewards -> their sign, {-1, 0, +1}."""

    def reward(self, reward):
        return float(np.sign(float(reward)))
