"""The Random-Network-Distillation PPO learner of ``ppo_rnd_envpool.py`` on the storage, kernels and update machinery of
``PPOLearner``.

What RND adds to the feed-forward path (reference: cleanrl/ppo_rnd_envpool.py):

====================================  ==========================================================
reference                              here
====================================  ==========================================================
:301-303  reward / observation stats   ``reward_rms``, ``obs_rms`` (``RunningMeanStd``), ``discounted_reward``
:310-314  second value stream          ``int_values``, ``curiosity_rewards`` (+ ``int_advantages`` / ``int_returns``)
:347-355  action logic, two values     ``act``                (K5 convert, network, K2 sample kernel)
:363-371  intrinsic reward             ``curiosity``          (normalised newest frame -> target / predictor nets)
:390-400  intrinsic-reward scaling     ``finish_rollout``     (forward filter per env, running variance; host numpy)
:402-432  TWO GAE streams              ``finish_rollout``     (K1 twice: extrinsic with the dones at gamma, intrinsic
                                                               non-episodic -- all-zero dones -- at int_gamma)
:444      combined advantage           ``update``             ``int_coef * A_int + ext_coef * A_ext``
:446,451  observation statistics       ``update``
:461-521  minibatch update             ``_minibatch_*``       (K5 gather, network forward, K3 fused loss on the combined
                                                               advantage and the clipped EXTRINSIC value; the unclipped
                                                               intrinsic value loss and the masked distillation loss are
                                                               two small torch terms; autograd; fused clip + Adam over
                                                               agent + predictor parameters in one flat buffer)
====================================  ==========================================================
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from . import host_ops
from .learner import PPOLearner


class RunningMeanStd:
    """``gym.wrappers.normalize.RunningMeanStd`` of gym 0.23.1 (the reference's pin, pyproject.toml:17; imported at
    ppo_rnd_envpool.py:16; not installed in this image): running mean / variance with Chan et al.'s parallel update."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, "float64")
        self.var = np.ones(shape, "float64")
        self.count = epsilon

    def update(self, x):
        self.update_from_moments(np.mean(x, axis=0), np.var(x, axis=0), x.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot_count
        self.mean, self.var, self.count = new_mean, m2 / tot_count, tot_count


class RewardForwardFilter:
    """ppo_rnd_envpool.py:236-246."""

    def __init__(self, gamma):
        self.rewems = None
        self.gamma = gamma

    def update(self, rews):
        if self.rewems is None:
            self.rewems = rews
        else:
            self.rewems = self.rewems * self.gamma + rews
        return self.rewems


class _Combined(nn.Module):
    """agent parameters followed by the predictor's: the order of ``combined_parameters`` (:295)."""

    def __init__(self, agent, predictor):
        super().__init__()
        self.agent, self.predictor = agent, predictor


class RNDPPOLearner(PPOLearner):
    def __init__(self, agent, rnd_model, args, obs_space, act_space, num_envs, device, world_size: int = 1,
                 sample_seed: int = 0):
        super().__init__(agent, args, obs_space, act_space, num_envs, device, world_size=world_size, sample_seed=sample_seed)
        assert self.image and tuple(obs_space.shape) == (4, 84, 84), "RND reads the newest of 4 stacked 84x84 frames"
        self.rnd_model = rnd_model
        self.combined_parameters = list(agent.parameters()) + list(rnd_model.predictor.parameters())     # :295
        if self.hip:
            from .flat import FlatParams

            self.flat = FlatParams(_Combined(agent, rnd_model.predictor))    # one flat buffer for the fused clip + Adam
            n_upd = self._scalars.shape[0]
            self._extra = torch.zeros((n_upd, 2), device=device)             # intrinsic value loss, distillation loss
            self._zeros_TN = torch.zeros((self.T, self.N), device=device)
            self._zeros_N = torch.zeros(self.N, device=device)
        else:
            self.optimizer = optim.Adam(self.combined_parameters, lr=args.learning_rate, eps=1e-5)        # :296-300
        T, N = self.T, self.N
        self.int_values = torch.zeros((T, N), device=device)
        self.curiosity_rewards = torch.zeros((T, N), device=device)
        self.int_advantages = torch.zeros((T, N), device=device)
        self.int_returns = torch.zeros((T, N), device=device)
        self.reward_rms = RunningMeanStd()
        self.obs_rms = RunningMeanStd(shape=(1, 1, 84, 84))
        self.discounted_reward = RewardForwardFilter(args.int_gamma)
        self.last_fwd_loss = float("nan")

    # ------------------------------------------------------------------ rollout
    def _newest_frame(self, rows):
        """The last of the 4 stacked frames of observation rows, as (B, 1, 84, 84) f32 (0..255): ``x[:, 3, :, :]``."""
        if self.nhwc:
            return rows[..., 3].reshape(-1, 1, 84, 84).float()
        return rows[:, 3, :, :].reshape(-1, 1, 84, 84).float()

    def _rnd_input(self, rows):
        """:363-368 / :453-458: ((newest frame - obs_rms.mean) / sqrt(obs_rms.var)).clip(-5, 5).float()"""
        mean = torch.from_numpy(self.obs_rms.mean).to(self.device)
        var = torch.from_numpy(self.obs_rms.var).to(self.device)
        return ((self._newest_frame(rows) - mean) / torch.sqrt(var)).clip(-5, 5).float()

    @torch.no_grad()
    def act(self, step: int):
        if self.hip:
            logits, v_ext, v_int = self.agent.heads3(self._features(self.obs[step]))
            seed, off = self.agent.rng.next()
            a64, _, _, _ = self.ops.categorical_sample(logits.contiguous(), seed=seed, offset=off,
                                                       action_f32_out=self.actions[step], logprob_out=self.logprobs[step],
                                                       want_entropy=False)
            self.values[step].copy_(v_ext.view(-1))
            self.int_values[step].copy_(v_int.view(-1))
            return a64
        value_ext, value_int = self.agent.get_value(self.obs[step])                       # :348-352
        self.values[step], self.int_values[step] = value_ext.flatten(), value_int.flatten()
        action, logprob, _, _, _ = self.agent.get_action_and_value(self.obs[step])         # :353
        self.actions[step] = action
        self.logprobs[step] = logprob
        return action

    @torch.no_grad()
    def curiosity(self, step: int):
        """Intrinsic reward of the transition into the observation stored in slot ``step + 1`` (:363-371)."""
        rows, _ = self._slot(step + 1)
        rnd_next_obs = self._rnd_input(rows)
        target_next_feature = self.rnd_model.target(rnd_next_obs)
        predict_next_feature = self.rnd_model.predictor(rnd_next_obs)
        self.curiosity_rewards[step] = (target_next_feature - predict_next_feature).pow(2).sum(1) / 2
        return self.curiosity_rewards[step]

    @torch.no_grad()
    def finish_rollout(self) -> None:
        a = self.args
        # :390-400 intrinsic-reward scaling -- (T, N) floats through the host, as the reference
        curiosity_reward_per_env = np.array(
            [self.discounted_reward.update(reward_per_step) for reward_per_step in self.curiosity_rewards.cpu().data.numpy().T])
        mean, std, count = (np.mean(curiosity_reward_per_env), np.std(curiosity_reward_per_env), len(curiosity_reward_per_env))
        self.reward_rms.update_from_moments(mean, std**2, count)
        self.curiosity_rewards /= np.sqrt(self.reward_rms.var)
        if self.hip:
            _, v_ext, v_int = self.agent.heads3(self._features(self.boot_obs))
            self.ops.gae(self.rewards, self.dones, self.values, self.boot_done, v_ext.reshape(-1).contiguous(),
                         a.gamma, a.gae_lambda, self.advantages, self.returns)
            # the intrinsic stream is non-episodic (int_nextnonterminal = 1.0, :413,418): K1 with all-zero dones
            self.ops.gae(self.curiosity_rewards, self._zeros_TN, self.int_values, self._zeros_N,
                         v_int.reshape(-1).contiguous(), a.int_gamma, a.gae_lambda, self.int_advantages, self.int_returns)
        else:
            next_value_ext, next_value_int = self.agent.get_value(self.boot_obs)
            adv, ret = host_ops.gae(self.rewards, self.dones, self.values, self.boot_done, next_value_ext.reshape(1, -1),
                                    a.gamma, a.gae_lambda)
            self.advantages.copy_(adv)
            self.returns.copy_(ret)
            zeros = torch.zeros_like(self.dones)
            adv, ret = host_ops.gae(self.curiosity_rewards, zeros, self.int_values, torch.zeros_like(self.boot_done),
                                    next_value_int.reshape(1, -1), a.int_gamma, a.gae_lambda)
            self.int_advantages.copy_(adv)
            self.int_returns.copy_(ret)

    # ------------------------------------------------------------------ update
    def update(self, lr: float) -> dict:
        a = self.args
        B, M = self.batch_size, self.minibatch_size
        b_inds = np.arange(B)
        b_obs = self.obs.reshape((-1,) + self.obs_shape)
        b_actions = self.actions.reshape(-1)
        b_logprobs = self.logprobs.reshape(-1)
        b_ext_returns, b_int_returns = self.returns.reshape(-1), self.int_returns.reshape(-1)
        b_ext_values = self.values.reshape(-1)
        b_advantages = self.int_advantages.reshape(-1) * a.int_coef + self.advantages.reshape(-1) * a.ext_coef     # :444
        newest = self._newest_frame(b_obs)
        if self.hip:        # statistics of np.mean / np.var over the batch axis, on the device in f64
            d = newest.double()
            self.obs_rms.update_from_moments(d.mean(0).cpu().numpy(), d.var(0, unbiased=False).cpu().numpy(), d.shape[0])
            del d
        else:
            self.obs_rms.update(newest.cpu().numpy())                                                          # :446
        mean = torch.from_numpy(self.obs_rms.mean).to(self.device)
        var = torch.from_numpy(self.obs_rms.var).to(self.device)
        rnd_next_obs = ((newest - mean) / torch.sqrt(var)).clip(-5, 5).float()                                   # :453-458
        del newest
        k = 0
        clipfracs = []
        last = None
        for epoch in range(int(a.update_epochs)):
            np.random.shuffle(b_inds)
            if self.hip:
                inds_dev = self.upload_permutation(epoch, b_inds)
            for start in range(0, B, M):
                end = start + M
                if self.hip:
                    self._minibatch_rnd_hip(inds_dev[start:end], b_obs, rnd_next_obs, b_actions, b_logprobs, b_advantages,
                                            b_ext_returns, b_int_returns, b_ext_values, lr, k)
                else:
                    last = self._minibatch_rnd_host(b_inds[start:end], b_obs, rnd_next_obs, b_actions, b_logprobs, b_advantages,
                                                    b_ext_returns, b_int_returns, b_ext_values, lr)
                    clipfracs.append(last[6].item())
                k += 1
            if a.target_kl is not None:
                approx_kl = (self._scalars[k - 1, 5] if self.hip else last[5]).item()
                if approx_kl > a.target_kl:
                    break
        if self.hip:
            sc, ex = self._scalars[:k].cpu().numpy(), self._extra[:k].cpu().numpy()
            last_np, clipfrac = sc[-1].copy(), float(np.mean(sc[:, 6]))
            int_v, fwd = float(ex[-1, 0]), float(ex[-1, 1])
            last_np[2] += int_v                                            # v_loss = ext_v_loss + int_v_loss (:513)
            last_np[0] += int_v * a.vf_coef + fwd                          # the K3 scalar holds the terms it computed
        else:
            last_np, clipfrac, fwd = last.numpy(), float(np.mean(clipfracs)), float(last[7])
        self.last_fwd_loss = fwd
        return dict(loss=float(last_np[0]), policy_loss=float(last_np[1]), value_loss=float(last_np[2]),
                    entropy=float(last_np[3]), old_approx_kl=float(last_np[4]), approx_kl=float(last_np[5]),
                    clipfrac=clipfrac, fwd_loss=fwd, num_updates=k)

    def _forward_loss(self, rnd_rows):
        """:467-476: per-sample distillation error, averaged over a random ``update_proportion`` of the minibatch."""
        predict_next_state_feature, target_next_state_feature = self.rnd_model(rnd_rows)
        forward_loss = F.mse_loss(predict_next_state_feature, target_next_state_feature.detach(), reduction="none").mean(-1)
        mask = torch.rand(len(forward_loss), device=self.device)
        mask = (mask < self.args.update_proportion).type(torch.FloatTensor).to(self.device)
        return (forward_loss * mask).sum() / torch.max(mask.sum(), torch.tensor([1], device=self.device, dtype=torch.float32))

    def _minibatch_rnd_hip(self, idx, b_obs, rnd_next_obs, b_actions, b_logprobs, b_advantages, b_ext_returns, b_int_returns,
                           b_ext_values, lr, k):
        a, ops = self.args, self.ops
        forward_loss = self._forward_loss(rnd_next_obs.index_select(0, idx))
        if self._x_mb is None or self._x_mb.shape[0] != idx.numel():
            self._x_mb = torch.empty((idx.numel(),) + self.obs_shape, device=self.device)
        x = ops.obs_u8_to_f32(b_obs, idx, self._x_mb).permute(0, 3, 1, 2)               # K5: b_obs[mb_inds] ; x / 255.0
        logits, v_ext, v_int = self.agent.heads3(x)
        v_ext, v_int = v_ext.view(-1), v_int.view(-1)
        # K3 on the combined advantage and the (clipped) extrinsic value: pg_loss - ent_coef*entropy + vf_coef*ext_v_loss
        _, dlogits, dv_ext = ops.ppo_loss_categorical(logits.detach().contiguous(), v_ext.detach().contiguous(), idx, b_actions,
                                                      b_logprobs, b_advantages, b_ext_returns, b_ext_values, a.clip_coef,
                                                      a.ent_coef, a.vf_coef, a.norm_adv, a.clip_vloss,
                                                      scalars_out=self._scalars[k])
        int_v_loss = 0.5 * ((v_int - b_int_returns.index_select(0, idx)) ** 2).mean()                         # :512
        self._extra[k, 0], self._extra[k, 1] = int_v_loss.detach(), forward_loss.detach().view(())
        rest = int_v_loss * a.vf_coef + forward_loss.view(())
        torch.autograd.backward([logits, v_ext, rest], [dlogits, dv_ext, None])                               # :518
        if self.world_size > 1:
            dist.all_reduce(self.flat.grads, op=dist.ReduceOp.SUM)
        self.optimizer_step_hip(lr)                                                                           # :519-524

    def _minibatch_rnd_host(self, mb_inds, b_obs, rnd_next_obs, b_actions, b_logprobs, b_advantages, b_ext_returns,
                            b_int_returns, b_ext_values, lr):
        """The reference's minibatch body on CPU tensors (:461-524).  Returns the 7 scalars of K3's host-pointer twin with
        ``v_loss = ext + int`` and ``loss`` including the distillation term, followed by the distillation loss."""
        a = self.args
        self.optimizer.param_groups[0]["lr"] = lr
        forward_loss = self._forward_loss(rnd_next_obs[mb_inds])
        # the network's three outputs (get_action_and_value's forward, :478-480), then K3 through the C ABI's host-pointer twin on
        # the combined advantage and the (clipped) extrinsic value -- the seam the HIP branch crosses (_minibatch_rnd_hip); the
        # intrinsic value loss and the distillation term are added outside
        logits, new_ext_values, new_int_values = self.agent.heads3(self.agent._normalise(b_obs[mb_inds]))
        ext_loss, sc = host_ops.ppo_loss_categorical(logits, new_ext_values, torch.as_tensor(mb_inds, dtype=torch.int64), b_actions,
                                                     b_logprobs, b_advantages, b_ext_returns, b_ext_values, a.clip_coef, a.ent_coef,
                                                     a.vf_coef, a.norm_adv, a.clip_vloss)
        int_v_loss = 0.5 * ((new_int_values.view(-1) - b_int_returns[mb_inds]) ** 2).mean()
        loss = ext_loss + int_v_loss * a.vf_coef + forward_loss.view(())
        self.optimizer.zero_grad()
        loss.backward()
        if a.max_grad_norm:
            nn.utils.clip_grad_norm_(self.combined_parameters, a.max_grad_norm)
        self.optimizer.step()
        sc = sc.clone()
        sc[0], sc[2] = loss.detach(), sc[2] + int_v_loss.detach()
        return torch.cat([sc, forward_loss.detach().view(1)])
