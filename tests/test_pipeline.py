"""Env-group lanes (cleanrl_amd/pipeline.py) on the host path: K overlapped lanes fill the rollout buffers exactly as the
serial loop over the same K vector envs does (the lanes only re-order work in time, never data)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from cleanrl_amd import envs as E
from cleanrl_amd.agents import AtariAgent
from cleanrl_amd.learner import PPOLearner
from cleanrl_amd.learner_smoke import default_args
from cleanrl_amd.pipeline import GroupedRollout, split_env_groups


def _groups(N, K, seed):
    return split_env_groups(lambda g, n: E.SyntheticAtariVecEnv(n, seed=seed + g * n, api="gym", done_p=0.2), N, K)


def _learner(N, T):
    env = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    torch.manual_seed(3)
    agent = AtariAgent(env)
    return PPOLearner(agent, default_args(num_steps=T), env.single_observation_space, env.single_action_space, N, torch.device("cpu"))


@pytest.mark.parametrize("K", [1, 2, 4])
def test_lanes_fill_the_buffers_like_the_serial_loop(K):
    N, T = 8, 6
    ref, L = _learner(N, T), _learner(N, T)
    # serial: one loop over the K vector envs, rows written group by group
    groups = _groups(N, K, seed=11)
    per = N // K
    obs = np.concatenate([g.reset() for g in groups])
    ref.observe(0, obs, np.zeros(N, np.float32))
    for step in range(T):
        a = ref.act(step).numpy()
        res = [g.step(a[i * per:(i + 1) * per]) for i, g in enumerate(groups)]
        ref.store_reward(step, np.concatenate([r[1] for r in res]))
        ref.observe(step + 1, np.concatenate([r[0] for r in res]), np.concatenate([r[2] for r in res]))
    # lanes: K threads
    groups2 = _groups(N, K, seed=11)
    roll = GroupedRollout(L, K)
    for g, ge in enumerate(groups2):
        roll.first_observation(g, ge.reset())
    seen = []

    def step_fn(g, actions, step):
        assert actions.shape == (per,)
        seen.append((g, step))
        o, r, d, _ = groups2[g].step(actions)
        return o, r, d

    roll.run(step_fn)
    assert sorted(seen) == [(g, s) for g in range(K) for s in range(T)]
    for name in ("obs", "boot_obs", "dones", "boot_done", "rewards"):
        assert torch.equal(getattr(L, name), getattr(ref, name)), name
    # the policy is a function of the observation rows alone: same values whichever way the rows were batched
    torch.testing.assert_close(L.values, ref.values, rtol=1e-5, atol=1e-6)
    assert float(L.dones.sum()) > 0                      # the done path was exercised


def test_lane_errors_surface_on_the_callers_thread():
    N, T, K = 4, 3, 2
    L = _learner(N, T)
    roll = GroupedRollout(L, K)
    groups = _groups(N, K, seed=1)
    for g, ge in enumerate(groups):
        roll.first_observation(g, ge.reset())

    def step_fn(g, actions, step):
        if g == 1 and step == 1:
            raise RuntimeError("env group 1 broke")
        o, r, d, _ = groups[g].step(actions)
        return o, r, d

    with pytest.raises(RuntimeError, match="env group 1 broke"):
        roll.run(step_fn)


def test_env_groups_flag_runs_the_script_end_to_end(capsys):
    from cleanrl_amd import ppo_atari_envpool

    L = ppo_atari_envpool.main(["--no-cuda", "--synthetic-env", "--num-envs", "4", "--env-groups", "2", "--num-steps", "8",
                                "--total-timesteps", "64", "--num-minibatches", "2", "--update-epochs", "1"])
    assert np.isfinite(L.last_metrics["loss"]) and "SPS:" in capsys.readouterr().out
    with pytest.raises(AssertionError, match="multiple"):
        ppo_atari_envpool.main(["--no-cuda", "--synthetic-env", "--num-envs", "5", "--env-groups", "2", "--num-steps", "8",
                                "--total-timesteps", "40", "--num-minibatches", "1"])


@pytest.mark.parametrize("autoreset", ["same_step", "next_step"])
def test_newest_frame_rule_rebuilds_every_stack_under_both_autoreset_conventions(autoreset):
    """``pipeline.full_stack_rows`` (which envs must send their whole stack) on the host: rebuilding each observation as
    "previous stack shifted + newest plane" for all other envs reproduces the env's observations exactly -- for gym < 1.0
    vector envs (reset in the step that reports done) and for envpool / gymnasium >= 1.0 (reset in the FOLLOWING call, which
    reports done = False: round-2 advisor finding).  Dropping either half of the rule breaks the convention it covers."""
    from cleanrl_amd.pipeline import full_stack_rows, stack_probe

    env = E.SyntheticAtariVecEnv(12, seed=5, api="gym", done_p=0.2, autoreset=autoreset)
    prev = env.reset().copy()
    prev_done, prev_probe = np.zeros(12, bool), stack_probe(prev)
    stale_without_prev_done = 0
    n_full = 0
    for _ in range(40):
        obs, _, done, _ = env.step(np.zeros(12, np.int64))
        probe = stack_probe(obs)
        rows = full_stack_rows(done, prev_done, probe, prev_probe)
        n_full += len(rows)
        rebuilt = np.concatenate([prev[:, 1:], obs[:, 3:4]], axis=1)       # what obs_shift_append_u8 computes on the device
        rebuilt[rows] = obs[rows]
        assert np.array_equal(rebuilt, obs)
        only_done = np.concatenate([prev[:, 1:], obs[:, 3:4]], axis=1)     # the round-2 rule: done at this step only
        only_done[np.flatnonzero(done)] = obs[np.flatnonzero(done)]
        stale_without_prev_done += int((only_done != obs).any())
        prev, prev_done, prev_probe = obs.copy(), np.asarray(done).astype(bool), probe
    assert 0 < n_full < 12 * 40 // 2                                       # resets happened, and most steps still send one plane
    assert (stale_without_prev_done > 0) == (autoreset == "next_step")
    # the probe alone (no done flags at all) also finds every discontinuity of these byte streams
    env2 = E.SyntheticAtariVecEnv(6, seed=9, api="gym", done_p=0.3, autoreset=autoreset)
    prev = env2.reset().copy()
    for _ in range(20):
        obs, _, done, _ = env2.step(np.zeros(6, np.int64))
        rows = full_stack_rows(np.zeros(6, bool), np.zeros(6, bool), stack_probe(obs), stack_probe(prev))
        rebuilt = np.concatenate([prev[:, 1:], obs[:, 3:4]], axis=1)
        rebuilt[rows] = obs[rows]
        assert np.array_equal(rebuilt, obs)
        prev = obs.copy()


def test_static_frames_env_keeps_the_frame_stack_structure():
    """The zero-cost env of tools/host_env_bench.py's pipeline-ceiling leg (``SyntheticAtariVecEnv(static_frames=True)``): after the first observation a step
    leaves the caller's buffer as it is, and that constant stack must be ITS OWN SHIFT (four copies of one plane) -- otherwise the frame-delta path
    (pipeline.full_stack_rows) takes every env for a discontinuity and re-sends whole stacks one env at a time (13 ms per lane step: seen on the GPU)."""
    import numpy as np

    from cleanrl_amd.envs import SyntheticAtariVecEnv
    from cleanrl_amd.pipeline import full_stack_rows, stack_probe

    env = SyntheticAtariVecEnv(16, seed=3, api="gym", static_frames=True, done_p=0.0)
    out = np.zeros((16, 4, 84, 84), np.uint8)
    env.reset(out=out)
    first = out.copy()
    assert (first[:, 0] == first[:, 3]).all() and first.any()
    prev_probe, prev_done = stack_probe(out), np.zeros(16, bool)
    for _ in range(3):
        obs, reward, done, info = env.step(np.zeros(16, np.int64), out=out)
        assert obs is out and (out == first).all() and not done.any()
        probe = stack_probe(out)
        assert len(full_stack_rows(done, prev_done, probe, prev_probe)) == 0
        prev_probe, prev_done = probe, done
    moving = SyntheticAtariVecEnv(16, seed=3, api="gym", done_p=0.0)      # the ordinary stand-in: a genuine shift every step
    moving.reset(out=out)
    p0 = stack_probe(out)
    moving.step(np.zeros(16, np.int64), out=out)
    assert len(full_stack_rows(np.zeros(16, bool), np.zeros(16, bool), stack_probe(out), p0)) == 0 and (out[:, 3] != first[:, 3]).any()
