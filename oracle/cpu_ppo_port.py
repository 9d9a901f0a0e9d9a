"""CPU port of one full reference PPO iteration (TEST INFRASTRUCTURE -- bench.py's ``cpu_baseline`` leg only).

``cleanrl/ppo_atari_envpool.py`` -- the CPU baseline BASELINE.json names -- cannot run in this image (no
envpool / gym / tyro / tensorboard, no network), so this restates its loop body (ppo_atari_envpool.py:217-341)
in the same stock torch CPU ops, f32 observation storage included, driven by a synthetic vector env with
the Atari byte streams.  kind = "port".  It is timed, never shipped: the product does not import it.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim
from torch.distributions.categorical import Categorical

from . import torch_oracle as TO


def _layer_init(layer, std=np.sqrt(2), bias_const=0.0):
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


class RefAgent(nn.Module):
    """NatureCNN actor-critic as in ppo_atari_envpool.py:123-149."""

    def __init__(self, n_actions):
        super().__init__()
        self.network = nn.Sequential(
            _layer_init(nn.Conv2d(4, 32, 8, stride=4)), nn.ReLU(), _layer_init(nn.Conv2d(32, 64, 4, stride=2)), nn.ReLU(),
            _layer_init(nn.Conv2d(64, 64, 3, stride=1)), nn.ReLU(), nn.Flatten(),
            _layer_init(nn.Linear(64 * 7 * 7, 512)), nn.ReLU())
        self.actor = _layer_init(nn.Linear(512, n_actions), std=0.01)
        self.critic = _layer_init(nn.Linear(512, 1), std=1)

    def get_value(self, x):
        return self.critic(self.network(x / 255.0))

    def get_action_and_value(self, x, action=None):
        hidden = self.network(x / 255.0)
        probs = Categorical(logits=self.actor(hidden))
        if action is None:
            action = probs.sample()
        return action, probs.log_prob(action), probs.entropy(), self.critic(hidden)


class _Env:
    def __init__(self, n, seed, pool=512):
        self.rs = np.random.RandomState(seed)
        self.planes = self.rs.randint(0, 256, size=(pool, 84, 84), dtype=np.uint8)
        self.cur = self.rs.randint(0, pool, size=n)
        self.n, self.pool = n, pool

    def obs(self):
        return self.planes[(self.cur[:, None] + np.arange(4)[None]) % self.pool]

    def step(self, _action):
        reward = self.rs.choice(np.array([-1.0, 0.0, 1.0]), size=self.n, p=[0.05, 0.9, 0.05])
        done = self.rs.random_sample(self.n) < 1 / 200
        self.cur = np.where(done, self.rs.randint(0, self.pool, size=self.n), self.cur + 1)
        return self.obs(), reward, done


def update_from_rollout(agent, optimizer, obs, actions, logprobs, rewards, dones, values, next_obs, next_done, *,
                        num_minibatches=4, update_epochs=4, gamma=0.99, gae_lambda=0.95, clip_coef=0.1, ent_coef=0.01,
                        vf_coef=0.5, max_grad_norm=0.5):
    """GAE + flatten + epochs x minibatches of one iteration (ppo_atari_envpool.py:250-322) on a filled rollout.  The host
    shuffle draws from numpy's global stream as the reference does (seed it before the call).  Returns what the last
    minibatch left behind (advantages, returns, the last loss dict).  ``tests/test_oracle_golden.py`` holds this function to
    the whole-iteration golden minted from the reference's own lines (``atari_iteration.npz``), so the loop ``run`` times
    is pinned as a whole, not only piece by piece."""
    T, N = rewards.shape
    batch = T * N
    mb = batch // num_minibatches
    with torch.no_grad():                                                        # :250-263
        next_value = agent.get_value(next_obs).reshape(-1)
        advantages, returns = TO.gae(rewards, dones, values, next_done, next_value, gamma, gae_lambda)
    b_obs = obs.reshape((-1,) + tuple(obs.shape[2:]))                            # :266-271
    b_logprobs, b_actions = logprobs.reshape(-1), actions.reshape(-1)
    b_advantages, b_returns, b_values = advantages.reshape(-1), returns.reshape(-1), values.reshape(-1)
    b_inds = np.arange(batch)
    out = None
    for epoch in range(update_epochs):                                          # :276-322
        np.random.shuffle(b_inds)
        for start in range(0, batch, mb):
            mb_inds = b_inds[start:start + mb]
            _, newlogprob, entropy, newvalue = agent.get_action_and_value(b_obs[mb_inds], b_actions.long()[mb_inds])
            out = TO.ppo_loss(newlogprob, entropy, newvalue, b_logprobs[mb_inds], b_advantages[mb_inds],
                              b_returns[mb_inds], b_values[mb_inds], clip_coef, ent_coef, vf_coef, True, True)
            optimizer.zero_grad()
            out["loss"].backward()
            nn.utils.clip_grad_norm_(agent.parameters(), max_grad_norm)
            optimizer.step()
    return dict(advantages=advantages, returns=returns, last=out)


def run(num_envs=32, num_steps=128, iterations=1, warmup_iterations=0, seed=1, num_minibatches=4, update_epochs=4,
        gamma=0.99, gae_lambda=0.95, clip_coef=0.1, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, lr=2.5e-4,
        n_actions=4, max_seconds=None, rollout_budget_s=None):
    """Returns dict(sps, seconds, env_steps, cores).  One iteration = rollout + GAE + epochs x minibatches.
    ``rollout_budget_s``: give up (``aborted`` = True, nothing else measured) when the FIRST rollout -- 128 forward steps, about 1/6.5 of an
    iteration on these cores (profiles/r05_cpu_port_vs_reference_lines.json: 22 s of 144 s) -- takes longer than this: bench.py's guard
    around the full-size sample."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    dev = torch.device("cpu")
    envs = _Env(num_envs, seed)
    agent = RefAgent(n_actions)
    optimizer = optim.Adam(agent.parameters(), lr=lr, eps=1e-5)
    T, N = num_steps, num_envs
    obs = torch.zeros((T, N, 4, 84, 84))
    actions, logprobs, rewards = torch.zeros((T, N)), torch.zeros((T, N)), torch.zeros((T, N))
    dones, values = torch.zeros((T, N)), torch.zeros((T, N))
    next_obs = torch.Tensor(envs.obs()).to(dev)
    next_done = torch.zeros(N)
    batch = T * N
    timed_steps, t_start, done_iters = 0, None, 0
    iter_secs = []
    for it in range(warmup_iterations + iterations):
        if it == warmup_iterations:
            t_start = time.perf_counter()
        t_it = time.perf_counter()
        for step in range(T):                                                    # :224-247
            obs[step] = next_obs
            dones[step] = next_done
            with torch.no_grad():
                action, logprob, _, value = agent.get_action_and_value(next_obs)
                values[step] = value.flatten()
            actions[step] = action
            logprobs[step] = logprob
            o, reward, d = envs.step(action.cpu().numpy())
            rewards[step] = torch.tensor(reward, dtype=torch.float32).view(-1)
            next_obs, next_done = torch.Tensor(o).to(dev), torch.Tensor(d).to(dev)
        if rollout_budget_s is not None and it == 0 and time.perf_counter() - t_it > rollout_budget_s:
            return dict(aborted=True, rollout_seconds=time.perf_counter() - t_it, cores=torch.get_num_threads(), num_envs=num_envs, num_steps=num_steps)
        update_from_rollout(agent, optimizer, obs, actions, logprobs, rewards, dones, values, next_obs, next_done,
                            num_minibatches=num_minibatches, update_epochs=update_epochs, gamma=gamma, gae_lambda=gae_lambda,
                            clip_coef=clip_coef, ent_coef=ent_coef, vf_coef=vf_coef, max_grad_norm=max_grad_norm)
        if it >= warmup_iterations:
            timed_steps += batch
            done_iters += 1
            iter_secs.append(time.perf_counter() - t_it)
            if max_seconds is not None and time.perf_counter() - t_start > max_seconds:
                break
    secs = time.perf_counter() - t_start
    return dict(sps=timed_steps / secs, sps_median=batch / float(np.median(iter_secs)), iteration_seconds=iter_secs, seconds=secs,
                env_steps=timed_steps, iterations=done_iters, cores=torch.get_num_threads(), num_envs=num_envs,
                num_steps=num_steps)


def run_metric_sample(num_envs=1024, num_steps=128, rollout_steps_timed=16, minibatches_timed=1, seed=1, num_minibatches=4, update_epochs=4,
                      gamma=0.99, gae_lambda=0.95, clip_coef=0.1, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, lr=2.5e-4, n_actions=4):
    """A BOUNDED sample of one iteration AT THE GIVEN (the metric's) SHAPES, for boxes where a whole iteration takes minutes: every piece of the loop
    body of ``run`` is timed at its full size -- ``rollout_steps_timed`` env steps of ``num_envs`` envs (after one untimed step), the GAE pass over
    the whole (T, N) rollout, ``minibatches_timed`` minibatch updates of T N / num_minibatches rows gathered from the whole f32 observation buffer
    (after one untimed minibatch: oneDNN builds its primitives per shape) -- and the iteration is the pieces times their counts:

        seconds = T x step + gae + (update_epochs x num_minibatches) x minibatch

    The rollout steps that are not timed are filled with copies of the timed ones (untimed), so the gather and the network see full-size tensors.
    Returns dict(sps, seconds (predicted), pieces, cpu_seconds_spent, ...)."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    dev = torch.device("cpu")
    envs = _Env(num_envs, seed)
    agent = RefAgent(n_actions)
    optimizer = optim.Adam(agent.parameters(), lr=lr, eps=1e-5)
    T, N = num_steps, num_envs
    t_all = time.perf_counter()
    obs = torch.zeros((T, N, 4, 84, 84))
    actions, logprobs, rewards = torch.zeros((T, N)), torch.zeros((T, N)), torch.zeros((T, N))
    dones, values = torch.zeros((T, N)), torch.zeros((T, N))
    next_obs = torch.Tensor(envs.obs()).to(dev)
    next_done = torch.zeros(N)
    k = min(rollout_steps_timed, T - 1)
    t0 = None
    for step in range(k + 1):                                                    # :224-247; step 0 untimed
        if step == 1:
            t0 = time.perf_counter()
        obs[step] = next_obs
        dones[step] = next_done
        with torch.no_grad():
            action, logprob, _, value = agent.get_action_and_value(next_obs)
            values[step] = value.flatten()
        actions[step] = action
        logprobs[step] = logprob
        o, reward, d = envs.step(action.cpu().numpy())
        rewards[step] = torch.tensor(reward, dtype=torch.float32).view(-1)
        next_obs, next_done = torch.Tensor(o).to(dev), torch.Tensor(d).to(dev)
    step_s = (time.perf_counter() - t0) / k
    for step in range(k + 1, T):                                                 # untimed: the rest of the rollout = copies of the sampled steps
        src = 1 + (step - 1) % k
        obs[step], dones[step], values[step], actions[step], logprobs[step], rewards[step] = (obs[src], dones[src], values[src], actions[src],
                                                                                              logprobs[src], rewards[src])
    batch, mb = T * N, T * N // num_minibatches
    t0 = time.perf_counter()
    with torch.no_grad():                                                        # :250-263
        next_value = agent.get_value(next_obs).reshape(-1)
        advantages, returns = TO.gae(rewards, dones, values, next_done, next_value, gamma, gae_lambda)
    gae_s = time.perf_counter() - t0
    b_obs = obs.reshape((-1,) + tuple(obs.shape[2:]))
    b_logprobs, b_actions = logprobs.reshape(-1), actions.reshape(-1)
    b_advantages, b_returns, b_values = advantages.reshape(-1), returns.reshape(-1), values.reshape(-1)
    b_inds = np.arange(batch)
    np.random.shuffle(b_inds)
    mb_secs = []
    for i in range(1 + minibatches_timed):                                       # :276-322; minibatch 0 untimed
        t0 = time.perf_counter()
        mb_inds = b_inds[(i % num_minibatches) * mb:(i % num_minibatches + 1) * mb]
        _, newlogprob, entropy, newvalue = agent.get_action_and_value(b_obs[mb_inds], b_actions.long()[mb_inds])
        out = TO.ppo_loss(newlogprob, entropy, newvalue, b_logprobs[mb_inds], b_advantages[mb_inds], b_returns[mb_inds], b_values[mb_inds],
                          clip_coef, ent_coef, vf_coef, True, True)
        optimizer.zero_grad()
        out["loss"].backward()
        nn.utils.clip_grad_norm_(agent.parameters(), max_grad_norm)
        optimizer.step()
        if i > 0:
            mb_secs.append(time.perf_counter() - t0)
    mb_s = float(np.median(mb_secs))
    secs = T * step_s + gae_s + update_epochs * num_minibatches * mb_s
    return dict(sps=batch / secs, seconds=secs, env_steps=batch, cores=torch.get_num_threads(), num_envs=N, num_steps=T,
                pieces=dict(rollout_step_s=step_s, rollout_steps_timed=k, gae_s=gae_s, minibatch_s=mb_s, minibatches_timed=minibatches_timed,
                            minibatch_rows=mb, minibatch_updates_per_iteration=update_epochs * num_minibatches),
                cpu_seconds_spent=time.perf_counter() - t_all, final_loss=float(out["loss"].detach()))


if __name__ == "__main__":
    print(run())
