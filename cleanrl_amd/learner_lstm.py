"""The recurrent PPO learner of ``ppo_atari_lstm.py`` on the same storage, kernels and update machinery as ``PPOLearner``.

What differs from the feed-forward scripts (reference: cleanrl/ppo_atari_lstm.py):

====================================  ==========================================================
reference                              here
====================================  ==========================================================
:225-228  zero LSTM state              ``__init__`` (``next_lstm_state``)
:231      ``initial_lstm_state`` clone ``act(0)`` -- the first action of an iteration snapshots the state
:246-247  action logic with state      ``act``            (K5 convert, network + one LSTM step, K2 sample kernel)
:268-273  bootstrap with state         ``finish_rollout`` (K1 kernel)
:296-312  ENV-WISE minibatches         ``update``: ``mb_inds = flatinds[:, mbenvinds].ravel()`` -- every minibatch is
                                       all T steps of ``num_envs // num_minibatches`` envs, time-major, so that the
                                       LSTM can be unrolled from ``initial_lstm_state[:, mbenvinds]``
:304-309  sequence forward             ``agent.heads_seq`` (T LSTM steps with done resets), then the same fused loss
                                       kernel (K3), autograd through the network, all-reduce, fused clip + Adam
====================================  ==========================================================

The gather kernel (K5) and the loss kernel (K3) take ``mb_inds`` as they come: nothing in them assumes the
feed-forward scripts' random row permutation.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import host_ops
from .learner import PPOLearner


class LSTMPPOLearner(PPOLearner):
    def __init__(self, agent, args, obs_space, act_space, num_envs, device, world_size: int = 1, sample_seed: int = 0):
        super().__init__(agent, args, obs_space, act_space, num_envs, device, world_size=world_size, sample_seed=sample_seed)
        assert self.N % int(args.num_minibatches) == 0, "num_envs must be divisible by num_minibatches"   # :296
        self.next_lstm_state = agent.initial_state(self.N, device)
        self.initial_lstm_state = (self.next_lstm_state[0].clone(), self.next_lstm_state[1].clone())
        if self.hip:
            E = int(args.update_epochs)
            self._env_dev = torch.empty((E, self.N), dtype=torch.int64, device=device)
            self._env_pin = torch.empty((E, self.N), dtype=torch.int64).pin_memory()

    # ------------------------------------------------------------------ rollout
    @torch.no_grad()
    def act(self, step: int):
        if step == 0:                                                      # :231
            self.initial_lstm_state = (self.next_lstm_state[0].clone(), self.next_lstm_state[1].clone())
        if self.hip:
            logits, value, self.next_lstm_state = self.agent.heads_seq(self._features(self.obs[step]), self.next_lstm_state,
                                                                       self.dones[step])
            seed, off = self.agent.rng.next()
            a64, _, _, _ = self.ops.categorical_sample(logits.contiguous(), seed=seed, offset=off,
                                                       action_f32_out=self.actions[step], logprob_out=self.logprobs[step],
                                                       want_entropy=False)
            self.values[step].copy_(value.view(-1))
            return a64
        action, logprob, _, value, self.next_lstm_state = self.agent.get_action_and_value(
            self.obs[step], self.next_lstm_state, self.dones[step])
        self.values[step] = value.flatten()
        self.actions[step] = action
        self.logprobs[step] = logprob
        return action

    @torch.no_grad()
    def finish_rollout(self) -> None:
        a = self.args
        if self.hip:
            _, next_value, _ = self.agent.heads_seq(self._features(self.boot_obs), self.next_lstm_state, self.boot_done)
            self.ops.gae(self.rewards, self.dones, self.values, self.boot_done, next_value.reshape(-1).contiguous(),
                         a.gamma, a.gae_lambda, self.advantages, self.returns)
        else:
            next_value = self.agent.get_value(self.boot_obs, self.next_lstm_state, self.boot_done).reshape(1, -1)
            adv, ret = host_ops.gae(self.rewards, self.dones, self.values, self.boot_done, next_value, a.gamma, a.gae_lambda)
            self.advantages.copy_(adv)
            self.returns.copy_(ret)

    # ------------------------------------------------------------------ update
    def update(self, lr: float) -> dict:
        a = self.args
        T, N, B = self.T, self.N, self.batch_size
        b_obs = self.obs.reshape((-1,) + self.obs_shape)
        b_actions = self.actions.reshape((-1,) + self.act_shape)
        b_logprobs, b_advantages = self.logprobs.reshape(-1), self.advantages.reshape(-1)
        b_returns, b_values, b_dones = self.returns.reshape(-1), self.values.reshape(-1), self.dones.reshape(-1)
        envsperbatch = N // int(a.num_minibatches)                          # :297
        envinds = np.arange(N)
        flatinds = np.arange(B).reshape(T, N)
        if self.hip:
            flat_dev = torch.arange(B, device=self.device).reshape(T, N)
        k = 0
        clipfracs = []
        last = None
        for epoch in range(int(a.update_epochs)):
            np.random.shuffle(envinds)                                     # :302 host MT19937
            if self.hip:
                env_dev = self._env_dev[epoch]                       # one pinned + device row per epoch (no reuse hazard)
                self._env_pin[epoch].copy_(torch.from_numpy(envinds))
                env_dev.copy_(self._env_pin[epoch], non_blocking=True)
            for start in range(0, N, envsperbatch):
                end = start + envsperbatch
                if self.hip:
                    env_idx = env_dev[start:end]
                    idx = flat_dev.index_select(1, env_idx).reshape(-1)    # :306 "be really careful about the index"
                    self._mb_state = (self.initial_lstm_state[0].index_select(1, env_idx),
                                      self.initial_lstm_state[1].index_select(1, env_idx))
                    self._mb_dones = b_dones.index_select(0, idx)
                    self._minibatch_hip(idx, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, lr,
                                        self._scalars[k])
                else:
                    mbenvinds = envinds[start:end]
                    mb_inds = flatinds[:, mbenvinds].ravel()
                    last = self._minibatch_host_lstm(mb_inds, mbenvinds, b_obs, b_actions, b_logprobs, b_advantages,
                                                     b_returns, b_values, b_dones, lr)
                    clipfracs.append(last[6].item())
                k += 1
            if a.target_kl is not None:                                     # :355-356
                approx_kl = (self._scalars[k - 1, 5] if self.hip else last[5]).item()
                if approx_kl > a.target_kl:
                    break
        y_pred, y_true = b_values.cpu().numpy(), b_returns.cpu().numpy()    # :358-360
        var_y = np.var(y_true)
        explained_var = np.nan if var_y == 0 else 1 - np.var(y_true - y_pred) / var_y
        if self.hip:
            sc = self._scalars[:k].cpu().numpy()
            last_np, clipfrac = sc[-1], float(np.mean(sc[:, 6]))
        else:
            last_np, clipfrac = last.numpy(), float(np.mean(clipfracs))
        return dict(loss=float(last_np[0]), policy_loss=float(last_np[1]), value_loss=float(last_np[2]),
                    entropy=float(last_np[3]), old_approx_kl=float(last_np[4]), approx_kl=float(last_np[5]),
                    clipfrac=clipfrac, explained_variance=float(explained_var), num_updates=k)

    def forward_backward_hip(self, idx, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, scalars_out):
        """K5 gather of the minibatch's (T x envs) rows -> conv stack -> T LSTM steps from the minibatch's initial state ->
        K3 fused loss fwd+bwd -> autograd through LSTM and conv stack (:304-352)."""
        a, ops = self.args, self.ops
        if self._x_mb is None or self._x_mb.shape[0] != idx.numel():
            self._x_mb = torch.empty((idx.numel(),) + self.obs_shape, device=self.device)
        x = ops.obs_u8_to_f32(b_obs, idx, self._x_mb).permute(0, 3, 1, 2)           # b_obs[mb_inds] ; x / 255.0
        logits, value, _ = self.agent.heads_seq(x, self._mb_state, self._mb_dones)
        value = value.view(-1)
        _, dlogits, dvalue = ops.ppo_loss_categorical(logits.detach().contiguous(), value.detach().contiguous(), idx,
                                                      b_actions, b_logprobs, b_advantages, b_returns, b_values,
                                                      a.clip_coef, a.ent_coef, a.vf_coef, a.norm_adv, a.clip_vloss,
                                                      scalars_out=scalars_out)
        torch.autograd.backward([logits, value], [dlogits, dvalue])                 # :349

    def _minibatch_host_lstm(self, mb_inds, mbenvinds, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values,
                             b_dones, lr):
        """The reference's minibatch body on CPU tensors (:304-352)."""
        a = self.args
        self.optimizer.param_groups[0]["lr"] = lr
        # the network up to its outputs (:304-309: get_action_and_value's forward), then K3 through the C ABI's host-pointer twin:
        # distribution, loss terms and their gradients down to (logits, value) on the FLAT batch arrays + mb_inds -- the seam the
        # HIP branch crosses (forward_backward_hip)
        logits, newvalue, _ = self.agent.heads_seq(
            self.agent._normalise(b_obs[mb_inds]),
            (self.initial_lstm_state[0][:, mbenvinds], self.initial_lstm_state[1][:, mbenvinds]),
            b_dones[mb_inds])
        loss, scalars = host_ops.ppo_loss_categorical(logits, newvalue, torch.as_tensor(mb_inds, dtype=torch.int64), b_actions,
                                                      b_logprobs, b_advantages, b_returns, b_values, a.clip_coef, a.ent_coef,
                                                      a.vf_coef, a.norm_adv, a.clip_vloss)
        self.optimizer.zero_grad()
        loss.backward()
        self._host_allreduce_grads()
        nn.utils.clip_grad_norm_(self.agent.parameters(), a.max_grad_norm)
        self.optimizer.step()
        return scalars
