"""Host-side vector environments.

Environment stepping stays on the host, as in the reference (``gym.vector.SyncVectorEnv``,
ppo_atari_multigpu.py:225-227; ``envpool.make``, ppo_atari_envpool.py:185-196).  This image has no
gymnasium / gym / envpool / ale_py / mujoco (and no network), so besides thin adapters for the real
packages (used when importable) this module provides self-contained vector envs with the same
``reset()/step()`` contracts:

* ``CartPoleVecEnv``            -- CartPole-v1 dynamics in vectorised numpy (a real control task: PPO must
                                   learn it, which is how tests check end-to-end learning without gym).
* ``SyntheticAtariVecEnv``      -- (N,4,84,84) uint8 frame-stacked observations, sign-clipped rewards,
                                   Bernoulli episode ends: the byte streams of a wrapped Atari env
                                   (cleanrl_utils/atari_wrappers.py) without an emulator.
* ``SyntheticContinuousVecEnv`` -- (N,17) observations / (N,6) actions (HalfCheetah-v4 shapes).
* ``DeviceSyntheticAtariVecEnv``-- the Atari byte streams generated directly in HBM (bench.py's
                                   "inputs already resident in HBM" mode; no PCIe in the timed region).

All follow gymnasium 0.29's vector API (5-tuple ``step``, autoreset, ``final_info``); ``api="gym"``
switches to the old 4-tuple + ``info["lives"]/["r"]/["l"]`` layout that ``ppo_atari_envpool.py`` consumes.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np


class Discrete:
    def __init__(self, n: int):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def __repr__(self):
        return f"Discrete({self.n})"


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {np.dtype(self.dtype).name})"


class _EpisodeStats:
    """What ``gym.wrappers.RecordEpisodeStatistics`` (ppo.py:80) / the local wrapper of
    ppo_atari_envpool.py:83-114 record: running return/length per env, reported when an episode ends."""

    def __init__(self, n):
        self.returns = np.zeros(n, np.float32)
        self.lengths = np.zeros(n, np.int32)

    def update(self, reward, done):
        self.returns += reward
        self.lengths += 1
        r, l = self.returns.copy(), self.lengths.copy()
        self.returns *= 1 - done
        self.lengths *= 1 - done
        return r, l


def _final_info(done, r, l):
    if not done.any():
        return {}
    fi = np.empty(len(done), dtype=object)
    for i in np.flatnonzero(done):
        fi[i] = {"episode": {"r": np.array([r[i]], np.float32), "l": np.array([l[i]], np.int32)}}
    return {"final_info": fi, "_final_info": done.copy()}


class CartPoleVecEnv:
    """CartPole-v1 (Barto, Sutton & Anderson 1983 cart-pole; constants and termination of gymnasium's
    ``CartPole-v1``: 12 degree / 2.4 m limits, 500-step truncation, reward 1 per step), vectorised."""

    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    theta_threshold = 12 * 2 * math.pi / 360
    x_threshold = 2.4
    max_episode_steps = 500

    def __init__(self, num_envs: int, seed: int = 0):
        self.num_envs = num_envs
        self.single_observation_space = Box(-np.inf, np.inf, (4,))
        self.single_action_space = Discrete(2)
        self.rng = np.random.RandomState(seed)
        self.state = np.zeros((num_envs, 4), np.float64)
        self.steps = np.zeros(num_envs, np.int32)
        self.stats = _EpisodeStats(num_envs)

    def _reset_rows(self, rows):
        self.state[rows] = self.rng.uniform(-0.05, 0.05, size=(len(rows), 4))
        self.steps[rows] = 0

    def reset(self, seed: Optional[int] = None):
        if seed is not None:
            self.rng = np.random.RandomState(seed)
        self._reset_rows(np.arange(self.num_envs))
        self.stats = _EpisodeStats(self.num_envs)
        return self.state.astype(np.float32), {}

    def step(self, action):
        action = np.asarray(action).reshape(self.num_envs)
        x, x_dot, th, th_dot = self.state.T
        force = np.where(action == 1, self.force_mag, -self.force_mag)
        total_mass = self.masspole + self.masscart
        pml = self.masspole * self.length
        cos, sin = np.cos(th), np.sin(th)
        temp = (force + pml * th_dot**2 * sin) / total_mass
        thacc = (self.gravity * sin - cos * temp) / (self.length * (4.0 / 3.0 - self.masspole * cos**2 / total_mass))
        xacc = temp - pml * thacc * cos / total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        th = th + self.tau * th_dot
        th_dot = th_dot + self.tau * thacc
        self.state = np.stack([x, x_dot, th, th_dot], 1)
        self.steps += 1
        terminated = (np.abs(x) > self.x_threshold) | (np.abs(th) > self.theta_threshold)
        truncated = (self.steps >= self.max_episode_steps) & ~terminated
        done = terminated | truncated
        reward = np.ones(self.num_envs, np.float64)
        r, l = self.stats.update(reward.astype(np.float32), done)
        infos = _final_info(done, r, l)
        if done.any():
            self._reset_rows(np.flatnonzero(done))      # autoreset: the returned obs is the new episode's first
        return self.state.astype(np.float32), reward, terminated, truncated, infos

    def close(self):
        pass


class SyntheticAtariVecEnv:
    """Byte-stream stand-in for a wrapped Atari vector env (NoopReset/MaxAndSkip/EpisodicLife/ClipReward/
    Resize84/GrayScale/FrameStack4, ppo_atari_multigpu.py:105-124): deterministic per seed.

    Observation n at time t is planes[c_n+t .. c_n+t+3] of a fixed random plane pool (so consecutive
    observations share 3 of 4 channels, like FrameStack); reward in {-1,0,+1} with P=(.05,.9,.05);
    episode ends Bernoulli(1/200) (then the cursor jumps).  Actions are accepted and ignored.

    ``autoreset="same_step"`` (gym < 1.0 vector envs): the step that reports done already returns the next episode's first
    stack.  ``"next_step"`` (envpool's gym API, gymnasium >= 1.0): that step returns the episode's last stack, and the
    FOLLOWING call returns the fresh stack with done = False and reward 0.
    """

    def __init__(self, num_envs: int, seed: int = 0, n_actions: int = 4, pool_planes: int = 2048, api: str = "gymnasium",
                 done_p: float = 1.0 / 200.0, frames: int = 4, autoreset: str = "same_step", static_frames: bool = False):
        assert autoreset in ("same_step", "next_step")
        self.autoreset = autoreset
        # static_frames: after the first observation a step leaves the caller's `out` buffer as it is (rewards / dones still drawn) -- an env
        # that costs (almost) nothing, for measuring what the rollout PIPELINE itself sustains (tools/host_env_bench.py); materialising 256
        # four-frame stacks is 0.65 of the stand-in's 0.9 ms per step (np.take of 7.2 MB), which says nothing about a real emulator
        self.static_frames, self._static_filled = bool(static_frames), False
        self._pending = np.zeros(num_envs, bool)       # next_step mode: envs whose next call is their reset
        self.num_envs, self.api, self.done_p = num_envs, api, done_p
        self.single_observation_space = Box(0, 255, (frames, 84, 84), np.uint8)      # frames = 1: FrameStack(1) of ppo_atari_lstm.py:105
        self.single_action_space = Discrete(n_actions)
        self.observation_space, self.action_space = self.single_observation_space, self.single_action_space
        self.rng = np.random.RandomState(seed)
        self.planes = np.random.RandomState(seed + 12345).randint(0, 256, size=(pool_planes, 84, 84), dtype=np.uint8)
        self.cursor = np.zeros(num_envs, np.int64)
        self.stats = _EpisodeStats(num_envs)
        self._win = np.arange(frames)[None, :]

    def _obs(self, out=None):
        if self.static_frames and out is not None and self._static_filled:
            return out
        # (static frames: four copies of ONE plane per env -- a stack that is its own shift, which the frame-delta path of pipeline.py checks for)
        idx = (self.cursor[:, None] + (0 * self._win if self.static_frames else self._win)) % len(self.planes)
        if out is not None:
            np.take(self.planes, idx, axis=0, out=out, mode="clip")     # (idx is in range; mode="raise" would buffer `out`)
            self._static_filled = True
            return out
        return self.planes[idx]

    def reset(self, seed: Optional[int] = None, out=None):
        if seed is not None:
            self.rng = np.random.RandomState(seed)
        self.cursor = self.rng.randint(0, len(self.planes), size=self.num_envs).astype(np.int64)
        self.stats = _EpisodeStats(self.num_envs)
        self._pending[:] = False
        obs = self._obs(out)
        return obs if self.api == "gym" else (obs, {})

    def step(self, action, out=None):
        n = self.num_envs
        reward = self.rng.choice(np.array([-1.0, 0.0, 1.0]), size=n, p=[0.05, 0.9, 0.05])
        done = self.rng.random_sample(n) < self.done_p
        self.cursor += 1
        jump = self.rng.randint(0, len(self.planes), size=n)
        if self.autoreset == "next_step":
            reward = np.where(self._pending, 0.0, reward)
            done = done & ~self._pending                                  # the reset call itself reports done = False
            self.cursor = np.where(self._pending, jump, self.cursor)
            self._pending = done.copy()
        else:
            self.cursor = np.where(done, jump, self.cursor)
        r, l = self.stats.update(reward.astype(np.float32), done)
        obs = self._obs(out)
        if self.api == "gym":       # envpool gym-style 4-tuple (ppo_atari_envpool.py:237-247)
            info = {"lives": np.where(done, 0, 3).astype(np.int32), "r": r, "l": l, "reward": reward,
                    "terminated": done.astype(np.int32)}
            return obs, reward, done, info
        return obs, reward, done, np.zeros(n, bool), _final_info(done, r, l)

    def close(self):
        pass


class SyntheticContinuousVecEnv:
    """HalfCheetah-v4-shaped task (obs 17, act 6): a stable random linear system driven by the action,
    reward = forward velocity proxy - 0.1*|a|^2, 1000-step truncation.  Deterministic per seed."""

    def __init__(self, num_envs: int, seed: int = 0, obs_dim: int = 17, act_dim: int = 6):
        self.num_envs, self.obs_dim, self.act_dim = num_envs, obs_dim, act_dim
        self.single_observation_space = Box(-np.inf, np.inf, (obs_dim,))
        self.single_action_space = Box(-1.0, 1.0, (act_dim,))
        rs = np.random.RandomState(seed + 777)
        a = rs.standard_normal((obs_dim, obs_dim)) / math.sqrt(obs_dim)
        self.A = 0.9 * a / max(1.0, np.abs(np.linalg.eigvals(a)).max())
        self.B = rs.standard_normal((act_dim, obs_dim)) * 0.3
        self.w = rs.standard_normal(obs_dim) / math.sqrt(obs_dim)
        self.rng = np.random.RandomState(seed)
        self.state = np.zeros((num_envs, obs_dim))
        self.steps = np.zeros(num_envs, np.int32)
        self.stats = _EpisodeStats(num_envs)

    def reset(self, seed: Optional[int] = None):
        if seed is not None:
            self.rng = np.random.RandomState(seed)
        self.state = self.rng.standard_normal((self.num_envs, self.obs_dim)) * 0.1
        self.steps[:] = 0
        self.stats = _EpisodeStats(self.num_envs)
        return self.state.astype(np.float32), {}

    def step(self, action):
        a = np.clip(np.asarray(action, np.float64).reshape(self.num_envs, self.act_dim), -1, 1)   # ClipAction (:96)
        self.state = self.state @ self.A.T + a @ self.B + 0.01 * self.rng.standard_normal(self.state.shape)
        reward = self.state @ self.w - 0.1 * (a**2).sum(1)
        self.steps += 1
        truncated = self.steps >= 1000
        terminated = np.zeros(self.num_envs, bool)
        r, l = self.stats.update(reward.astype(np.float32), truncated)
        infos = _final_info(truncated, r, l)
        if truncated.any():
            rows = np.flatnonzero(truncated)
            self.state[rows] = self.rng.standard_normal((len(rows), self.obs_dim)) * 0.1
            self.steps[rows] = 0
        return self.state.astype(np.float32), reward, terminated, truncated, infos

    def close(self):
        pass


class DeviceSyntheticAtariVecEnv:
    """The ``SyntheticAtariVecEnv`` byte streams produced directly in HBM with torch ops on the learner's
    stream: ``step_into(obs_u8_row, reward_row, done_row)`` writes the next (N,4,84,84) uint8 observation,
    rewards and dones straight into rollout-storage rows.  Used by ``bench.py`` so that the timed region
    starts with inputs resident in HBM (no PCIe), and by GPU smoke tests."""

    def __init__(self, num_envs: int, device, seed: int = 0, n_actions: int = 4, pool_planes: int = 4096,
                 done_p: float = 1.0 / 200.0):
        import torch

        self.torch = torch
        self.num_envs, self.device, self.done_p = num_envs, device, done_p
        self.single_observation_space = Box(0, 255, (4, 84, 84), np.uint8)
        self.single_action_space = Discrete(n_actions)
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.planes = torch.randint(0, 256, (pool_planes, 84, 84), dtype=torch.uint8, device=device, generator=self.gen)
        self.cursor = torch.randint(0, pool_planes, (num_envs,), device=device, generator=self.gen)
        self.pool = pool_planes
        self._seed, self._step = int(seed) & (2**64 - 1), 0
        self.step_base = None      # 1-element int64 device tensor once the learner captured its rollout steps (then the step
        self._step_rel = None      # count of a captured launch is step_base + its position in the rollout)

    def _call(self, out, reward, done, advance, hwc=False):
        """Two launches per env step (csrc/synth_env.hip) instead of ~14 torch kernels: the rollout is short enough for
        the stand-in env's own launches to show up in env-steps/sec."""
        from . import _lib

        lib = _lib.load()
        t = self.torch
        assert out.dtype == t.uint8 and out.is_contiguous() and tuple(out.shape) == (self.num_envs,) + ((84, 84, 4) if hwc else (4, 84, 84))
        if advance:
            assert reward.dtype == t.float32 and done.dtype == t.float32 and reward.is_contiguous() and done.is_contiguous()
            self._step += 1
        rel = self._step_rel if (advance and self._step_rel is not None) else None
        fn = lib.mi355ppo_synth_atari_step_hwc_ctr_u8 if hwc else lib.mi355ppo_synth_atari_step_ctr_u8
        st = fn(
            self.planes.data_ptr(), self.pool, self.cursor.data_ptr(), self._seed, self._step if rel is None else rel,
            self.step_base.data_ptr() if rel is not None else None, out.data_ptr(),
            reward.data_ptr() if advance else None, done.data_ptr() if advance else None, self.num_envs, float(self.done_p),
            int(advance), t.cuda.current_stream(self.device).cuda_stream)
        _lib.check(st, "mi355ppo_synth_atari_step_u8")
        return out

    def obs_into(self, out):
        return self._call(out, None, None, False)

    def step_into(self, obs_out, reward_out, done_out):
        return self._call(obs_out, reward_out, done_out, True)

    def step_into_rows(self, rows_out, reward_out, done_out):
        """``step_into`` with the observation written pixel-interleaved, (N, 84, 84, 4) -- a rollout-storage row of the HIP
        learner, no relayout launch behind it."""
        return self._call(rows_out, reward_out, done_out, True, hwc=True)

    def close(self):
        pass


class DeviceSyntheticContinuousVecEnv:
    """``SyntheticContinuousVecEnv`` (HalfCheetah-v4-shaped: obs 17, act 6) stepped in HBM by one kernel launch per step
    (``mi355ppo_synth_continuous_step_f32``) on the learner's stream, for ``bench.py --config E``: observations, rewards and dones
    are device tensors, so the timed region holds no PCIe traffic.  ``step_into`` writes the next observation, the reward and the
    done flag straight into the rollout-storage rows; the position in the noise bank lives in device memory as well
    (``step_base``), so a captured step can be replayed (``PPOLearner.capture_rollout``).  A stand-in env (plumbing), not part of
    the hot path."""

    def __init__(self, num_envs: int, device, seed: int = 0, obs_dim: int = 17, act_dim: int = 6, noise_bank: int = 257,
                 horizon: int = 1000):
        import torch

        from . import ops

        self.torch, self.device, self.ops = torch, device, ops
        self.num_envs, self.obs_dim, self.act_dim, self.horizon = num_envs, obs_dim, act_dim, horizon
        self.single_observation_space = Box(-np.inf, np.inf, (obs_dim,))
        self.single_action_space = Box(-1.0, 1.0, (act_dim,))
        rs = np.random.RandomState(seed + 777)
        a = rs.standard_normal((obs_dim, obs_dim)) / math.sqrt(obs_dim)
        A = 0.9 * a / max(1.0, np.abs(np.linalg.eigvals(a)).max())
        f32 = lambda x: torch.as_tensor(np.asarray(x, np.float32), device=device).contiguous()   # noqa: E731
        self.At, self.B = f32(A.T.copy()), f32(rs.standard_normal((act_dim, obs_dim)) * 0.3)
        self.w = f32(rs.standard_normal(obs_dim) / math.sqrt(obs_dim))
        self.noise = f32(0.01 * rs.standard_normal((noise_bank, num_envs, obs_dim)))
        self.state = f32(0.1 * rs.standard_normal((num_envs, obs_dim)))
        self.reset_state = self.state.clone()
        self.steps = torch.zeros(num_envs, device=device)
        self._reward = torch.zeros(num_envs, device=device)
        self._done = torch.zeros(num_envs, device=device)
        self._step = 0                  # host mirror of the env's step count (position in the noise bank)
        self.step_base = None           # device-resident count: set by PPOLearner.capture_rollout
        self._step_rel = None

    def obs(self):
        return self.state

    def step_into(self, action, obs_out, reward_out, done_out):
        """One step with the results written into caller-provided rows (``obs_out`` may be the env's own ``state``)."""
        if self._step_rel is None:
            k, base = self._step, None
            self._step += 1
        else:                           # inside a capture: position = *step_base + the step's index in the rollout
            k, base = self._step_rel, self.step_base
        self.ops.synth_continuous_step(self.state, self.reset_state, self.At, self.B, self.w, self.noise, k, self.steps, float(self.horizon),
                                       action, obs_out, reward_out, done_out, k_base=base)
        return obs_out

    def step(self, action):
        """-> (next_obs (N,obs) f32, reward (N) f32, done (N) f32), all on the device (views of the env's own buffers)."""
        self.step_into(action.contiguous(), self.state, self._reward, self._done)
        return self.state, self._reward, self._done

    def close(self):
        pass


def have_gymnasium() -> bool:
    try:
        import gymnasium  # noqa: F401

        return True
    except Exception:
        return False


def have_envpool() -> bool:
    try:
        import envpool  # noqa: F401

        return True
    except Exception:
        return False


class _RunningMeanStd:
    """Per-env running mean/variance with the parallel-variance update (Chan et al.), the statistic
    behind gymnasium's NormalizeObservation / NormalizeReward (count starts at 1e-4)."""

    def __init__(self, shape):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = 1e-4

    def update(self, x):
        """``x`` is one new sample per env (the reference wraps each sub-env separately, so every env
        has its own statistics; batch size is 1 -> batch variance 0)."""
        delta = x - self.mean
        tot = self.count + 1.0
        self.mean = self.mean + delta / tot
        self.var = (self.var * self.count + delta**2 * self.count / tot) / tot
        self.count = tot


class NormalizeVecEnv:
    """ClipAction + NormalizeObservation + clip(-10,10) + NormalizeReward(gamma) + clip(-10,10) around a
    vector env: the per-env wrapper stack of ppo_continuous_action.py:94-100, vectorised."""

    def __init__(self, env, gamma: float):
        self.env, self.gamma = env, gamma
        self.num_envs = env.num_envs
        self.single_observation_space, self.single_action_space = env.single_observation_space, env.single_action_space
        n = self.num_envs
        self.obs_rms = _RunningMeanStd((n,) + tuple(env.single_observation_space.shape))
        self.ret_rms = _RunningMeanStd((n,))
        self.ret = np.zeros(n, np.float64)

    def _norm_obs(self, obs):
        self.obs_rms.update(obs.astype(np.float64))
        o = (obs - self.obs_rms.mean) / np.sqrt(self.obs_rms.var + 1e-8)
        return np.clip(o, -10, 10).astype(np.float32)

    def reset(self, seed=None):
        obs, info = self.env.reset(seed=seed)
        self.ret[:] = 0
        return self._norm_obs(obs), info

    def step(self, action):
        lo, hi = self.single_action_space.low, self.single_action_space.high
        obs, reward, term, trunc, info = self.env.step(np.clip(action, lo, hi))
        self.ret = self.ret * self.gamma * (1 - term.astype(np.float64)) + reward
        self.ret_rms.update(self.ret)
        reward = np.clip(reward / np.sqrt(self.ret_rms.var + 1e-8), -10, 10)
        return self._norm_obs(obs), reward, term, trunc, info

    def close(self):
        self.env.close()


class _BatchRunningMeanStd:
    """Scalar running mean/variance updated with a whole batch per call (Chan et al. parallel update) -- the statistic of
    gym's ``NormalizeReward`` around a *vector* env (``is_vector_env``), as ppo_procgen.py:197 uses it."""

    def __init__(self):
        self.mean, self.var, self.count = 0.0, 1.0, 1e-4

    def update(self, x):
        x = np.asarray(x, np.float64)
        bm, bv, bc = x.mean(), x.var(), x.size
        delta = bm - self.mean
        tot = self.count + bc
        m2 = self.var * self.count + bv * bc + delta**2 * self.count * bc / tot
        self.mean, self.var, self.count = self.mean + delta * bc / tot, m2 / tot, tot


class SyntheticProcgenVecEnv:
    """Byte-stream stand-in for ``ProcgenEnv(num_envs, env_name, distribution_mode="easy")`` behind
    ``TransformObservation(obs["rgb"])`` + ``RecordEpisodeStatistics`` + ``NormalizeReward(gamma)`` + clip(-10, 10)
    (ppo_procgen.py:189-198): (N, 64, 64, 3) uint8 pixel-interleaved frames, 15 discrete actions, the old gym API
    (``reset() -> obs``, ``step() -> obs, reward, done, [info per env]``).  Deterministic per seed; actions are ignored."""

    def __init__(self, num_envs: int, seed: int = 0, gamma: float = 0.999, n_actions: int = 15, pool_frames: int = 512,
                 done_p: float = 1.0 / 150.0):
        self.num_envs, self.gamma, self.done_p = num_envs, gamma, done_p
        self.single_observation_space = Box(0, 255, (64, 64, 3), np.uint8)
        self.single_action_space = Discrete(n_actions)
        self.observation_space, self.action_space = self.single_observation_space, self.single_action_space
        self.rng = np.random.RandomState(seed)
        self.frames = np.random.RandomState(seed + 4321).randint(0, 256, size=(pool_frames, 64, 64, 3), dtype=np.uint8)
        self.cursor = np.zeros(num_envs, np.int64)
        self.stats = _EpisodeStats(num_envs)
        self.ret_rms = _BatchRunningMeanStd()
        self.ret = np.zeros(num_envs, np.float64)

    def reset(self):
        self.cursor = self.rng.randint(0, len(self.frames), size=self.num_envs).astype(np.int64)
        self.stats = _EpisodeStats(self.num_envs)
        self.ret[:] = 0
        return self.frames[self.cursor]

    def step(self, action):
        n = self.num_envs
        raw = self.rng.choice(np.array([0.0, 1.0, 10.0]), size=n, p=[0.9, 0.09, 0.01])
        done = self.rng.random_sample(n) < self.done_p
        self.cursor = np.where(done, self.rng.randint(0, len(self.frames), size=n), (self.cursor + 1) % len(self.frames))
        r, l = self.stats.update(raw.astype(np.float32), done)
        info = [{"episode": {"r": r[i], "l": l[i]}} if done[i] else {} for i in range(n)]
        # NormalizeReward (vector form): returns = returns*gamma*(1-done) + reward; reward / sqrt(var + 1e-8); then clip
        self.ret = self.ret * self.gamma * (1.0 - done.astype(np.float64)) + raw
        self.ret_rms.update(self.ret)
        reward = np.clip(raw / np.sqrt(self.ret_rms.var + 1e-8), -10, 10)
        return self.frames[self.cursor], reward, done, info

    def close(self):
        pass


class SyntheticMAAtariVecEnv:
    """Byte-stream stand-in for the supersuit pipeline of ppo_pettingzoo_ma_atari.py:151-165 (two-player PettingZoo Atari:
    max-pool 2 frames, frame-skip 4, clip reward, colour reduction, 84x84, frame-stack 4, agent indicator, ``num_envs // 2``
    games concatenated): (N, 84, 84, 6) uint8 pixel-interleaved observations -- channels 0-3 the stacked frames, 4-5 a one-hot
    player indicator (0 / 255... stored as 0 / 1) --, consecutive envs are the two players of one game (zero-sum rewards,
    shared episode ends), 6 actions, the old gym API with one info dict per env."""

    def __init__(self, num_envs: int, seed: int = 0, n_actions: int = 6, pool_frames: int = 512, done_p: float = 1.0 / 150.0):
        assert num_envs % 2 == 0, "two players per game"
        self.num_envs, self.done_p = num_envs, done_p
        self.single_observation_space = Box(0, 255, (84, 84, 6), np.uint8)
        self.single_action_space = Discrete(n_actions)
        self.observation_space, self.action_space = self.single_observation_space, self.single_action_space
        self.rng = np.random.RandomState(seed)
        self.frames = np.random.RandomState(seed + 777).randint(0, 256, size=(pool_frames, 84, 84), dtype=np.uint8)
        self.cursor = np.zeros(num_envs // 2, np.int64)
        self.stats = _EpisodeStats(num_envs)
        self._win = np.arange(4)[None, :]
        self._indicator = np.zeros((num_envs, 84, 84, 2), np.uint8)
        self._indicator[0::2, :, :, 0] = 1
        self._indicator[1::2, :, :, 1] = 1

    def _obs(self):
        idx = (self.cursor[:, None] + self._win) % len(self.frames)              # (games, 4)
        stack = np.moveaxis(self.frames[idx], 1, -1)                             # (games, 84, 84, 4)
        return np.concatenate([np.repeat(stack, 2, axis=0), self._indicator], axis=-1)

    def reset(self):
        self.cursor = self.rng.randint(0, len(self.frames), size=self.num_envs // 2).astype(np.int64)
        self.stats = _EpisodeStats(self.num_envs)
        return self._obs()

    def step(self, action):
        g = self.num_envs // 2
        r0 = self.rng.choice(np.array([-1.0, 0.0, 1.0]), size=g, p=[0.05, 0.9, 0.05])
        reward = np.stack([r0, -r0], axis=1).reshape(-1)
        gdone = self.rng.random_sample(g) < self.done_p
        done = np.repeat(gdone, 2)
        self.cursor = np.where(gdone, self.rng.randint(0, len(self.frames), size=g), self.cursor + 1)
        r, l = self.stats.update(reward.astype(np.float32), done)
        info = [{"episode": {"r": r[i], "l": l[i]}} if done[i] else {} for i in range(self.num_envs)]
        return self._obs(), reward, done, info

    def close(self):
        pass
