"""The Phasic Policy Gradient learner of ``ppg_procgen.py`` on the storage, kernels and update machinery of ``PPOLearner``.

====================================  ==========================================================
reference (ppg_procgen.py)             here
====================================  ==========================================================
:260      Adam(eps=1e-8)               ``adam_eps`` of the fused clip + Adam kernel / the host optimiser
:269-272  aux_obs / aux_returns        ``aux_obs`` (uint8, kept in HBM on a GPU -- 6.4 GB at the default sizes -- instead
                                       of host RAM), ``aux_returns``
:340-342  full-batch advantage norm.   ``update`` (then K3 runs with ``norm_adv`` off)
:344-392  policy phase minibatches     ``PPOLearner.update`` unchanged (K5 gather, K3 fused loss, fused clip + Adam);
                                       the value head reads DETACHED features (``PPGAgent.heads``)
:412-414  store the rollout            ``update`` (after the policy update)
:417-431  old policy on the aux buffer ``aux_phase``: logits of the current policy on every stored rollout
:433-471  auxiliary epochs             ``aux_phase``: whole rollouts (T x num_aux_rollouts envs) per minibatch; joint loss
                                       = aux value + beta_clone * KL(old || new), plus the real value loss; gradient
                                       accumulation, clip, Adam
====================================  ==========================================================
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.distributions as td
import torch.nn as nn
import torch.optim as optim
from torch.distributions.categorical import Categorical

from .learner import PPOLearner


def flatten01(arr):
    return arr.reshape((-1, *arr.shape[2:]))


def unflatten01(arr, targetshape):
    return arr.reshape((*targetshape, *arr.shape[1:]))


class PPGLearner(PPOLearner):
    def __init__(self, agent, args, obs_space, act_space, num_envs, device, world_size: int = 1, sample_seed: int = 0):
        args.norm_adv = False                           # the policy phase normalises over the full batch instead (:340)
        args.update_epochs = int(args.e_policy)         # E_pi
        super().__init__(agent, args, obs_space, act_space, num_envs, device, world_size=world_size, sample_seed=sample_seed)
        self.adam_eps = 1e-8                            # :260
        if not self.hip:
            self.optimizer = optim.Adam(agent.parameters(), lr=args.learning_rate, eps=1e-8)
        R = int(args.aux_batch_rollouts)
        aux_dev = device if self.hip else torch.device("cpu")
        self.aux_obs = torch.zeros((self.T, R) + self.obs_shape, dtype=torch.uint8, device=aux_dev)      # :269-271
        self.aux_returns = torch.zeros((self.T, R), device=aux_dev)
        self._aux_update = 0
        self._capture, self._stale_grads = False, None
        self._aux_tail_step = 0                         # Adam step count of the auxiliary value head (see _aux_step_hip)
        self._last_lr = float(args.learning_rate)
        self.last_aux = {}

    # ------------------------------------------------------------------ policy phase
    def optimizer_step_hip(self, lr: float) -> None:
        """During the last policy update of a phase, remember what ``clip_grad_norm_`` leaves in ``.grad`` (the reference's
        first auxiliary step accumulates onto it): g * grad_scale * min(1, max_norm / (norm + 1e-6))."""
        if self._capture:
            g = self.flat.grads.clone()
        super().optimizer_step_hip(lr)
        if self._capture:
            coef = torch.clamp(self.args.max_grad_norm / (self._total_norm + 1e-6), max=1.0) / self.world_size
            self._stale_grads = g * coef

    def update(self, lr: float) -> dict:
        a = self.args
        self._last_lr = lr
        self._capture = self.hip and (self._aux_update + 1) * self.N == int(a.aux_batch_rollouts)
        if a.adv_norm_fullbatch:                                                           # :340-342
            adv = self.advantages.reshape(-1)
            self.advantages.copy_(((adv - adv.mean()) / (adv.std() + 1e-8)).view_as(self.advantages))
        m = super().update(lr)
        self._capture = False
        sl = slice(self.N * self._aux_update, self.N * (self._aux_update + 1))           # :412-414
        self.aux_obs[:, sl] = self.obs.to(torch.uint8)
        self.aux_returns[:, sl] = self.returns
        self._aux_update += 1
        return m

    # ------------------------------------------------------------------ auxiliary phase
    def _aux_step_hip(self, lr: float) -> None:
        """One optimiser step of the auxiliary phase on the flat buffers with ``torch.optim.Adam``'s bookkeeping: the
        auxiliary value head receives no gradient during the policy phases, so torch never creates its Adam state there
        and its bias-correction step count only advances here, while every other parameter's advances in both phases.  The
        fused kernel takes ONE step count per call, hence two calls on the two contiguous ranges of the flat buffer (the
        head's two tensors are its tail), after one global-norm clip coefficient for both (``clip_grad_norm_`` over all
        parameters, :466): the coefficient goes in as ``grad_scale`` and the kernel's own clip is switched off."""
        f, a = self.flat, self.args
        tail = sum(p.numel() for p in self.agent.aux_critic.parameters())
        n_main = f.numel - tail
        assert f._params_list[-1] is self.agent.aux_critic.bias and n_main % 4 == 0
        scale = 1.0 / self.world_size
        norm = float(torch.linalg.vector_norm(f.grads)) * scale
        coef = min(1.0, a.max_grad_norm / (norm + 1e-6))
        f.step += 1
        self._aux_tail_step += 1
        for lo, hi, step in ((0, n_main, f.step), (n_main, f.numel, self._aux_tail_step)):
            self.ops.clip_adam_(f.params[lo:hi], f.grads[lo:hi], f.exp_avg[lo:hi], f.exp_avg_sq[lo:hi], step, lr, float("inf"),
                                grad_scale=scale * coef, eps=self.adam_eps)

    def _aux_rows(self, cols):
        """Whole rollouts of the stored envs ``cols`` as flat (T * len(cols)) observation rows (:436-438)."""
        if isinstance(cols, np.ndarray):
            cols = torch.from_numpy(cols).to(self.aux_obs.device)
        return flatten01(self.aux_obs.index_select(1, cols))

    def _aux_forward(self, rows):
        """uint8 rows -> (raw logits, value on detached features, auxiliary value)."""
        if self.hip:
            x = self.ops.obs_u8_to_f32(rows.contiguous()).permute(0, 3, 1, 2)                # K5 (no gather): x / 255.0
            return self.agent.heads_aux(x)
        return self.agent.heads_aux(self.agent._normalise(rows))

    def aux_phase(self) -> dict:
        a = self.args
        R, nr, A = int(a.aux_batch_rollouts), int(a.num_aux_rollouts), self.agent.n_actions
        assert self._aux_update * self.N == R, "the auxiliary phase follows n_iteration policy updates"
        aux_inds = np.arange(R)
        # :417-431 the old policy on the aux buffer, before distilling into the network (normalised logits, as
        # Categorical(logits=...).logits stores them)
        aux_pi = torch.zeros((self.T, R, A), device=self.aux_obs.device)
        with torch.no_grad():
            for start in range(0, R, nr):
                cols = aux_inds[start:start + nr]
                logits = self._aux_forward(self._aux_rows(cols))[0]
                logits = logits - logits.logsumexp(dim=-1, keepdim=True)
                aux_pi[:, torch.from_numpy(cols).to(aux_pi.device)] = unflatten01(logits, (self.T, len(cols))).to(aux_pi.device)
        if not self.hip:
            self.optimizer.param_groups[0]["lr"] = self._last_lr
        # Reference quirk, kept: the policy phase leaves the (clipped) gradient of its last minibatch in ``.grad`` and the
        # first auxiliary ``loss.backward()`` accumulates onto it (:393-396 step without a following zero_grad; :462-468).
        # On the host path that simply happens; the fused clip + Adam kernel of the HIP path zeroes the flat gradient buffer
        # after every step, so the clipped gradient of that last minibatch is rebuilt (``optimizer_step_hip`` below) and put
        # back here.
        if self.hip and self._stale_grads is not None:
            self.flat.grads.copy_(self._stale_grads)
            self._stale_grads = None
        kl_loss = aux_value_loss = real_value_loss = None
        for auxiliary_update in range(1, int(a.e_auxiliary) + 1):                           # :433-471
            print(f"aux epoch {auxiliary_update}")
            np.random.shuffle(aux_inds)
            for i, start in enumerate(range(0, R, nr)):
                cols = torch.from_numpy(aux_inds[start:start + nr]).to(self.aux_obs.device)
                m_aux_returns = flatten01(self.aux_returns.index_select(1, cols)).to(torch.float32).to(self.device)
                logits, new_values, new_aux_values = self._aux_forward(self._aux_rows(cols))
                new_values, new_aux_values = new_values.view(-1), new_aux_values.view(-1)
                old_pi = Categorical(logits=flatten01(aux_pi.index_select(1, cols)).to(self.device))
                kl_loss = td.kl_divergence(old_pi, Categorical(logits=logits)).mean()
                real_value_loss = 0.5 * ((new_values - m_aux_returns) ** 2).mean()
                aux_value_loss = 0.5 * ((new_aux_values - m_aux_returns) ** 2).mean()
                joint_loss = aux_value_loss + a.beta_clone * kl_loss
                loss = (joint_loss + real_value_loss) / a.n_aux_grad_accum
                loss.backward()                                   # accumulates: into the flat gradient buffer on a GPU
                if (i + 1) % int(a.n_aux_grad_accum) == 0:
                    if self.hip:
                        if self.world_size > 1:
                            dist.all_reduce(self.flat.grads, op=dist.ReduceOp.SUM)
                        self._aux_step_hip(self._last_lr)        # clip + Adam + zero the gradient buffer
                    else:
                        nn.utils.clip_grad_norm_(self.agent.parameters(), a.max_grad_norm)
                        self.optimizer.step()
                        self.optimizer.zero_grad()
        self._aux_update = 0
        self.last_aux = dict(kl_loss=float(kl_loss.detach()), aux_value_loss=float(aux_value_loss.detach()),
                             real_value_loss=float(real_value_loss.detach()))
        return self.last_aux
