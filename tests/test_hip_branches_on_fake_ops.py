"""The HIP-path BRANCHES of the learners (Python side: index plumbing, buffer shapes, argument order, bookkeeping) run on CPU
against a stand-in for ``cleanrl_amd.ops`` that has the same signatures and the same dtype / shape / contiguity checks as
the real wrappers but computes with the host formulas.  The kernels themselves are checked on the GPU (tests/test_gpu_*);
what this catches without one is a wrong argument, a non-contiguous view, a missing buffer or a mis-indexed minibatch in the
code that calls them -- in particular in the learners added after the round's GPU minutes were spent (LSTM, RND, PPG,
procgen, two-player Atari).  Each learner is run twice on identical inputs, once on its host path and once on its HIP
branch over the stand-in ops, and must agree.

Test-side only: the product never imports this module and has no switch that selects it."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch.distributions.categorical import Categorical

from cleanrl_amd import envs as E, host_ops
from cleanrl_amd.agents import AtariLSTMAgent, MAAtariAgent, PPGAgent, ProcgenAgent, RNDAgent, RNDModel
from cleanrl_amd.flat import FlatParams
from cleanrl_amd.learner import PPOLearner
from cleanrl_amd.learner_lstm import LSTMPPOLearner
from cleanrl_amd.learner_ppg import PPGLearner
from cleanrl_amd.learner_rnd import RNDPPOLearner, _Combined
from cleanrl_amd.learner_smoke import default_args


def _host_ppo_loss(*a):
    """(loss, scalars7) by the oracle's restatement of the reference's loss lines (oracle/torch_oracle.ppo_loss)."""
    from oracle import torch_oracle as TO

    o = TO.ppo_loss(*a)
    return o["loss"], torch.stack([o[k].detach() for k in ("loss", "pg_loss", "v_loss", "entropy", "old_approx_kl", "approx_kl", "clipfrac")])


def _chk(t, dtype, name, shape=None):
    assert isinstance(t, torch.Tensor), name
    assert t.dtype == dtype, f"{name}: dtype {t.dtype}"
    assert t.is_contiguous(), f"{name}: must be contiguous"
    if shape is not None:
        assert tuple(t.shape) == tuple(shape), f"{name}: shape {tuple(t.shape)} != {tuple(shape)}"
    return t


class FakeOps:
    """Signature-compatible CPU stand-in for the functions of cleanrl_amd/ops.py the learners call."""

    LOSS_SCALAR_NAMES = ("loss", "pg_loss", "v_loss", "entropy", "old_approx_kl", "approx_kl", "clipfrac")

    @staticmethod
    def gae(rewards, dones, values, next_done, next_value, gamma, gae_lambda, advantages=None, returns=None, variant=0):
        T, N = rewards.shape
        for t, n in ((rewards, "rewards"), (dones, "dones"), (values, "values"), (advantages, "advantages"), (returns, "returns")):
            _chk(t, torch.float32, n, (T, N))
        _chk(next_done.reshape(-1), torch.float32, "next_done", (N,))
        _chk(next_value.reshape(-1), torch.float32, "next_value", (N,))
        adv, ret = host_ops.gae(rewards, dones, values, next_done.reshape(-1), next_value.reshape(1, -1), gamma, gae_lambda)
        advantages.copy_(adv)
        returns.copy_(ret)
        return advantages, returns

    @staticmethod
    def categorical_sample(logits, noise_exp1=None, seed=0, offset=0, action_f32_out=None, logprob_out=None,
                           want_entropy=True, want_i64=True, offset_base=None):
        B, A = logits.shape
        _chk(logits, torch.float32, "logits", (B, A))
        assert isinstance(seed, int) and isinstance(offset, int) and offset > 0
        assert offset_base is None or (offset_base.dtype == torch.int64 and offset_base.numel() == 1)
        probs = Categorical(logits=logits)
        a = probs.sample()
        if action_f32_out is not None:
            _chk(action_f32_out, torch.float32, "action_f32_out", (B,)).copy_(a.float())
        lp = probs.log_prob(a)
        if logprob_out is not None:
            _chk(logprob_out, torch.float32, "logprob_out", (B,)).copy_(lp)
        return a, action_f32_out, lp, probs.entropy() if want_entropy else None

    class LossSlots:
        """Deferred scalar fold: a slot holds the 7 scalars of the call that used it until fold() hands them out."""

        def __init__(self, n, device):
            self.n, self.rows = int(n), [None] * int(n)

        def fold(self, n, out, first=0):
            _chk(out, torch.float32, "out")
            assert out.dim() == 2 and out.shape[1] == 7 and first + n <= self.n and first + n <= out.shape[0]
            for j in range(first, first + n):
                assert self.rows[j] is not None, f"slot {j} folded before any call filled it"
                out[j].copy_(self.rows[j])
                self.rows[j] = None
            return out

    @staticmethod
    def adv_stats(b_advantages, inds, minibatch_size, out=None):
        flat = _chk(b_advantages.reshape(-1), torch.float32, "b_advantages")
        _chk(inds, torch.int64, "inds")
        rows = [flat[inds[s:s + minibatch_size]] for s in range(0, inds.numel(), minibatch_size)]
        return torch.stack([torch.stack([r.mean(), r.std() + 1e-8]) for r in rows])

    @staticmethod
    def _check_adv_mean_den(adv_mean_den, b_advantages, mb_inds):
        """A hoisted statistics pair must be THIS minibatch's (a wrong row would silently change the normalisation)."""
        if adv_mean_den is None:
            return
        _chk(adv_mean_den, torch.float32, "adv_mean_den", (2,))
        mb = b_advantages.reshape(-1)[mb_inds]
        torch.testing.assert_close(adv_mean_den, torch.stack([mb.mean(), mb.std() + 1e-8]), rtol=1e-5, atol=1e-6)

    @staticmethod
    def ppo_loss_categorical(new_logits, new_value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, clip_coef,
                             ent_coef, vf_coef, norm_adv=True, clip_vloss=True, scalars_out=None, dlogits_out=None, dvalue_out=None,
                             adv_mean_den=None, slot=None):
        M, A = new_logits.shape
        FakeOps._check_adv_mean_den(adv_mean_den, b_advantages, mb_inds)
        _chk(new_logits, torch.float32, "new_logits", (M, A))
        _chk(new_value.reshape(-1), torch.float32, "new_value", (M,))
        _chk(mb_inds, torch.int64, "mb_inds", (M,))
        Bf = b_logprobs.numel()
        for t, n in ((b_logprobs, "b_logprobs"), (b_advantages, "b_advantages"), (b_returns, "b_returns"), (b_values, "b_values"),
                     (b_actions, "b_actions")):
            _chk(t.reshape(-1), torch.float32, n, (Bf,))
        assert not new_logits.requires_grad and not new_value.requires_grad, "the kernel takes detached network outputs"
        logits = new_logits.clone().requires_grad_(True)
        value = new_value.reshape(-1).clone().requires_grad_(True)
        probs = Categorical(logits=logits)
        acts = b_actions.reshape(-1).long()[mb_inds]
        loss, sc = _host_ppo_loss(probs.log_prob(acts), probs.entropy(), value, b_logprobs.reshape(-1)[mb_inds],
                                     b_advantages.reshape(-1)[mb_inds], b_returns.reshape(-1)[mb_inds],
                                     b_values.reshape(-1)[mb_inds], clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss)
        loss.backward()
        if slot is not None:                      # deferred fold: nothing may be written to scalars_out by this call
            slots, k = slot
            assert 0 <= k < slots.n and slots.rows[k] is None, "loss slot reused before it was folded"
            slots.rows[k] = sc.detach().clone()
            return None, logits.grad, value.grad
        if scalars_out is not None:
            _chk(scalars_out, torch.float32, "scalars_out", (7,)).copy_(sc)
        return (scalars_out if scalars_out is not None else sc), logits.grad, value.grad

    PACK_FLOATS = 8

    @staticmethod
    def batch_pack(b_actions, b_logprobs, b_advantages, b_returns, b_values, out=None):
        """include/mi355ppo.h: pack[i] = {action, old log-prob, advantage, return, old value, 0, 0, 0}."""
        Bf = b_logprobs.numel()
        cols = [_chk(t.reshape(-1), torch.float32, n, (Bf,)) for t, n in ((b_actions, "b_actions"), (b_logprobs, "b_logprobs"),
                (b_advantages, "b_advantages"), (b_returns, "b_returns"), (b_values, "b_values"))]
        out = out if out is not None else torch.empty(Bf, 8)
        _chk(out, torch.float32, "pack", (Bf, 8))
        out.zero_()
        for j, c in enumerate(cols):
            out[:, j] = c
        return out

    @staticmethod
    def adv_stats_packed(pack, inds, minibatch_size, out=None):
        _chk(pack, torch.float32, "pack")
        assert pack.dim() == 2 and pack.shape[1] == 8
        return FakeOps.adv_stats(pack[:, 2].contiguous(), inds, minibatch_size, out)

    @staticmethod
    def ppo_loss_categorical_packed(new_logits, new_value, mb_inds, pack, clip_coef, ent_coef, vf_coef, norm_adv=True,
                                    clip_vloss=True, scalars_out=None, dlogits_out=None, dvalue_out=None, adv_mean_den=None,
                                    slot=None):
        _chk(pack, torch.float32, "pack")
        assert pack.dim() == 2 and pack.shape[1] == 8 and float(pack[:, 5:].abs().sum()) == 0.0
        assert not norm_adv or adv_mean_den is not None, "the packed K3 entry point has no statistics launch of its own"
        c = [pack[:, j].contiguous() for j in range(5)]
        return FakeOps.ppo_loss_categorical(new_logits, new_value, mb_inds, c[0], c[1], c[2], c[3], c[4], clip_coef, ent_coef, vf_coef,
                                            norm_adv, clip_vloss, scalars_out, dlogits_out, dvalue_out, adv_mean_den, slot)

    @staticmethod
    def normal_sample(mean, logstd, noise=None, seed=0, offset=0, action_out=None, logprob_out=None):
        B, D = mean.shape
        _chk(mean, torch.float32, "mean", (B, D))
        _chk(logstd.reshape(-1), torch.float32, "logstd", (D,))
        probs = torch.distributions.Normal(mean, torch.exp(logstd.reshape(1, D).expand_as(mean)))
        a = probs.sample()
        lp, ent = probs.log_prob(a).sum(1), probs.entropy().sum(1)
        if action_out is not None:
            _chk(action_out, torch.float32, "action_out", (B, D)).copy_(a)
        if logprob_out is not None:
            _chk(logprob_out, torch.float32, "logprob_out", (B,)).copy_(lp)
        return (action_out if action_out is not None else a), lp, ent

    @staticmethod
    def ppo_loss_normal(new_mean, logstd, new_value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, clip_coef,
                        ent_coef, vf_coef, norm_adv=True, clip_vloss=True, scalars_out=None, adv_mean_den=None):
        M, D = new_mean.shape
        FakeOps._check_adv_mean_den(adv_mean_den, b_advantages, mb_inds)
        _chk(new_mean, torch.float32, "new_mean", (M, D))
        _chk(logstd.reshape(-1), torch.float32, "logstd", (D,))
        _chk(new_value.reshape(-1), torch.float32, "new_value", (M,))
        _chk(mb_inds, torch.int64, "mb_inds", (M,))
        Bf = b_logprobs.numel()
        _chk(b_actions.reshape(Bf, D), torch.float32, "b_actions", (Bf, D))
        assert not new_mean.requires_grad and not new_value.requires_grad and not logstd.requires_grad
        mean = new_mean.clone().requires_grad_(True)
        ls = logstd.reshape(1, D).clone().requires_grad_(True)
        value = new_value.reshape(-1).clone().requires_grad_(True)
        probs = torch.distributions.Normal(mean, torch.exp(ls.expand_as(mean)))
        acts = b_actions.reshape(Bf, D)[mb_inds]
        loss, sc = _host_ppo_loss(probs.log_prob(acts).sum(1), probs.entropy().sum(1), value, b_logprobs.reshape(-1)[mb_inds],
                                     b_advantages.reshape(-1)[mb_inds], b_returns.reshape(-1)[mb_inds],
                                     b_values.reshape(-1)[mb_inds], clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss)
        loss.backward()
        if scalars_out is not None:
            _chk(scalars_out, torch.float32, "scalars_out", (7,)).copy_(sc)
        return (scalars_out if scalars_out is not None else sc), mean.grad, ls.grad.reshape(-1), value.grad

    @staticmethod
    def obs_u8_to_f32(src_u8, inds=None, out=None, scale_255=True):
        _chk(src_u8, torch.uint8, "src_u8")
        if inds is not None:
            _chk(inds, torch.int64, "inds")
        x = (src_u8 if inds is None else src_u8[inds]).float()
        if scale_255:
            x = x / 255.0
        if out is None:
            return x
        return _chk(out, torch.float32, "out", tuple(x.shape)).copy_(x)

    @staticmethod
    def obs_nchw_to_nhwc_u8(src, out=None):
        _chk(src, torch.uint8, "src")
        rows, C, H, W = src.shape
        return _chk(out, torch.uint8, "out", (rows, H, W, C)).copy_(src.permute(0, 2, 3, 1))

    @staticmethod
    def clip_adam_(params, grads, exp_avg, exp_avg_sq, step, lr, max_grad_norm, grad_scale=1.0, beta1=0.9, beta2=0.999,
                   eps=1e-5, total_norm_out=None):
        n = params.numel()
        for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
            _chk(t, torch.float32, nm, (n,))
        assert step >= 1
        with torch.no_grad():
            g = grads * grad_scale
            norm = g.norm()
            g = g * torch.clamp(max_grad_norm / (norm + 1e-6), max=1.0)
            exp_avg.lerp_(g, 1 - beta1)
            exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            bc1, bc2 = 1 - beta1**step, 1 - beta2**step
            params.addcdiv_(exp_avg, (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
            grads.zero_()
            if total_norm_out is not None:
                total_norm_out.fill_(norm)
        return total_norm_out


def _to_fake_hip(L, flat_module=None):
    """Put a learner that was built on the CPU into the state its HIP branch expects (what ``__init__`` allocates when
    ``device.type == 'cuda'``), with the stand-in ops.  Frames are then handed to ``observe`` as uint8 tensors."""
    dev, T, N = L.device, L.T, L.N
    L.hip, L.ops, L.optimizer = True, FakeOps, None
    L.flat = FlatParams(flat_module if flat_module is not None else L.agent)
    L.agent.rng.seed = 1
    if L.image:
        L.nhwc = True
        if not L.hwc_frames:
            c, h, w = L.obs_shape
            L.obs_shape = (h, w, c)
        L.relayout = not L.hwc_frames
        L.obs = torch.zeros((T, N) + L.obs_shape, dtype=torch.uint8)
        L.boot_obs = torch.zeros((N,) + L.obs_shape, dtype=torch.uint8)
        L.stage_obs = torch.zeros((N,) + L.frame_shape, dtype=torch.uint8) if L.relayout else None
        L._x_roll = torch.empty((N,) + L.obs_shape)
    L._x_mb = None
    E_ = int(L.args.update_epochs)
    n_upd = E_ * -(-L.batch_size // L.minibatch_size)
    L._scalars = torch.zeros((n_upd, 7))
    from cleanrl_amd.learner import PPOLearner
    if L.discrete and type(L).forward_backward_hip is PPOLearner.forward_backward_hip:     # as PPOLearner.__init__ on a GPU
        L._loss_slots = FakeOps.LossSlots(n_upd, dev)
    L._inds_dev = torch.empty((E_, L.batch_size), dtype=torch.int64)
    L._inds_pin = torch.empty((E_, L.batch_size), dtype=torch.int64)
    L._total_norm = torch.zeros(1)
    return L


def _frames(rs, T, N, shape):
    return rs.randint(0, 256, size=(T + 1, N) + shape).astype(np.uint8)


def _drive(L, frames, dones, rewards, sample_seed, fake, extra_step=None):
    as_obs = (lambda x: torch.from_numpy(x)) if fake else (lambda x: x)
    L.observe(0, as_obs(frames[0]), torch.from_numpy(dones[0]) if fake else dones[0])
    torch.manual_seed(sample_seed)
    for step in range(L.T):
        L.act(step)
        L.store_reward(step, rewards[step])
        L.observe(step + 1, as_obs(frames[step + 1]), torch.from_numpy(dones[step + 1]) if fake else dones[step + 1])
        if extra_step is not None:
            extra_step(L, step)
    L.finish_rollout()


def _params(mods):
    return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()])


def _pair(make_agent, make_learner, flat_module=None):
    torch.manual_seed(0)
    host = make_learner(make_agent())
    torch.manual_seed(0)
    fake = make_learner(make_agent())
    return host, _to_fake_hip(fake, flat_module(fake) if flat_module else None)


def _episode_streams(rs, T, N):
    dones = (rs.random_sample((T + 1, N)) < 0.2).astype(np.float32)
    dones[0] = 0
    return dones, rs.randint(-1, 2, size=(T, N)).astype(np.float32)


def _compare_rollout_and_update(host, fake, frames, dones, rewards, mods_host, mods_fake, update_seed=3, lr=2.5e-4, extra_step=None,
                                tol=2e-5):
    _drive(host, frames, dones, rewards, 11, False, extra_step)
    _drive(fake, frames, dones, rewards, 11, True, extra_step)
    for name in ("actions", "logprobs", "values", "dones", "rewards", "advantages", "returns"):
        a, b = getattr(host, name), getattr(fake, name)
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6), name
    np.random.seed(update_seed)
    torch.manual_seed(update_seed)
    mh = host.update(lr)
    np.random.seed(update_seed)
    torch.manual_seed(update_seed)
    mf = fake.update(lr)
    assert mh["num_updates"] == mf["num_updates"]
    for k in ("loss", "policy_loss", "value_loss", "entropy", "approx_kl", "clipfrac"):
        assert abs(mh[k] - mf[k]) <= 1e-5 * max(1.0, abs(mh[k])), (k, mh[k], mf[k])
    d = (_params(mods_host) - _params(mods_fake)).abs()
    # Adam's first steps move a parameter by ~lr * sign(g): the handful whose gradient is rounding noise may differ by up to
    # 2 lr between two correct implementations; everything else must agree closely
    assert (d <= tol).float().mean().item() >= 1.0 - 1e-5 and d.max().item() <= 2.5 * lr, \
        f"parameters diverge between the host path and the HIP branch: max {d.max().item()}, {(d > tol).sum().item()} above {tol}"
    fake.flat.check_views()
    return mh, mf


def test_lstm_hip_branch():
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (1, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    T, N = 6, 4
    args = lambda: default_args(num_steps=T, num_minibatches=2, update_epochs=2)
    host, fake = _pair(lambda: AtariLSTMAgent(envs),
                       lambda ag: LSTMPPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    fake._env_dev, fake._env_pin = torch.empty((2, N), dtype=torch.int64), torch.empty((2, N), dtype=torch.int64)
    rs = np.random.RandomState(1)
    dones, rewards = _episode_streams(rs, T, N)
    _compare_rollout_and_update(host, fake, _frames(rs, T, N, (1, 84, 84)), dones, rewards, [host.agent], [fake.agent])
    assert torch.allclose(host.next_lstm_state[0], fake.next_lstm_state[0], atol=1e-6)
    assert fake.obs.dtype == torch.uint8 and tuple(fake.obs.shape[2:]) == (84, 84, 1)


def test_rnd_hip_branch():
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(5))
    T, N = 4, 4
    args = lambda: default_args(num_steps=T, num_minibatches=2, update_epochs=1, gamma=0.999, int_gamma=0.99, ent_coef=0.001,
                                update_proportion=0.25, int_coef=1.0, ext_coef=2.0, learning_rate=1e-4)

    def make(ag):
        return RNDPPOLearner(ag, RNDModel(4, 5), args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu"))

    host, fake = _pair(lambda: RNDAgent(envs), make, flat_module=lambda L: _Combined(L.agent, L.rnd_model.predictor))
    fake._extra = torch.zeros((2, 2))
    fake._zeros_TN, fake._zeros_N = torch.zeros((T, N)), torch.zeros(N)
    rs = np.random.RandomState(2)
    warm = rs.randint(0, 256, size=(32, 1, 84, 84)).astype(np.float64)
    for L in (host, fake):
        L.obs_rms.update(warm)
    dones, rewards = _episode_streams(rs, T, N)
    mh, mf = _compare_rollout_and_update(host, fake, _frames(rs, T, N, (4, 84, 84)), dones, rewards,
                                         [host.agent, host.rnd_model.predictor], [fake.agent, fake.rnd_model.predictor], lr=1e-4,
                                         extra_step=lambda L, step: L.curiosity(step))
    assert torch.allclose(host.int_returns, fake.int_returns, rtol=1e-5, atol=1e-6) and host.int_returns.abs().sum() > 0
    assert torch.allclose(host.curiosity_rewards, fake.curiosity_rewards, rtol=1e-5, atol=1e-7)
    assert abs(mh["fwd_loss"] - mf["fwd_loss"]) <= 1e-5 * max(1.0, abs(mh["fwd_loss"]))
    assert np.allclose(host.obs_rms.mean, fake.obs_rms.mean, atol=1e-4) and np.allclose(host.obs_rms.var, fake.obs_rms.var, rtol=1e-5)
    assert fake.flat.numel == sum(p.numel() for p in fake.combined_parameters)


def test_procgen_and_ppg_hip_branches(capsys):
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (64, 64, 3), np.uint8), single_action_space=E.Discrete(15))
    T, N = 4, 4
    rs = np.random.RandomState(3)
    frames = _frames(rs, T, N, (64, 64, 3))
    dones, rewards = _episode_streams(rs, T, N)
    args = lambda: default_args(num_steps=T, num_minibatches=2, update_epochs=2, gamma=0.999, clip_coef=0.2)
    host, fake = _pair(lambda: ProcgenAgent(envs),
                       lambda ag: PPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    _compare_rollout_and_update(host, fake, frames, dones, rewards, [host.agent], [fake.agent], lr=5e-4)
    assert fake.hwc_frames and not fake.relayout and fake.stage_obs is None and tuple(fake.obs.shape[2:]) == (64, 64, 3)

    pargs = lambda: default_args(num_steps=T, num_minibatches=2, gamma=0.999, clip_coef=0.2, adv_norm_fullbatch=True, e_policy=1,
                                 e_auxiliary=2, beta_clone=1.0, num_aux_rollouts=2, n_aux_grad_accum=1, aux_batch_rollouts=N,
                                 n_iteration=1, learning_rate=5e-4)
    host, fake = _pair(lambda: PPGAgent(envs),
                       lambda ag: PPGLearner(ag, pargs(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    _compare_rollout_and_update(host, fake, frames, dones, rewards, [host.agent], [fake.agent], lr=5e-4)
    assert fake.adam_eps == 1e-8 and fake._stale_grads is not None            # the last policy update of the phase was captured
    np.random.seed(4)
    ah = host.aux_phase()
    np.random.seed(4)
    af = fake.aux_phase()
    capsys.readouterr()
    for k in ah:
        assert abs(ah[k] - af[k]) <= 2e-5 * max(1.0, abs(ah[k])), (k, ah[k], af[k])
    # the rebuilt stale gradient reproduces the reference quirk: both paths end at the same parameters
    assert (_params([host.agent]) - _params([fake.agent])).abs().max().item() <= 5e-5
    assert torch.equal(host.aux_obs, fake.aux_obs)


def test_two_player_atari_hip_branch():
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (84, 84, 6), np.uint8), single_action_space=E.Discrete(6))
    T, N = 4, 4
    rs = np.random.RandomState(5)
    frames = _frames(rs, T, N, (84, 84, 6))
    frames[..., 4:] = rs.randint(0, 2, size=frames[..., 4:].shape)
    dones, rewards = _episode_streams(rs, T, N)
    args = lambda: default_args(num_steps=T, num_minibatches=2, update_epochs=1)
    host, fake = _pair(lambda: MAAtariAgent(envs),
                       lambda ag: PPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    _compare_rollout_and_update(host, fake, frames, dones, rewards, [host.agent], [fake.agent])
    assert fake.partial_scale and fake.hwc_frames


def test_core_learner_hip_branch_mlp_and_unfused_atari(monkeypatch):
    """The same harness on the GPU-validated core: ppo.py's MLP agent (vector observations) and the NatureCNN agent on the
    library-convolution branch (``MI355PPO_CNN=miopen``: K5 gather + torch Conv2d) -- a CPU regression net for the HIP branch
    of ``PPOLearner`` itself."""
    from cleanrl_amd.agents import AtariAgent, MlpAgent

    T, N = 8, 4
    rs = np.random.RandomState(7)
    dones, rewards = _episode_streams(rs, T, N)
    envs = SimpleNamespace(single_observation_space=E.Box(-1, 1, (4,)), single_action_space=E.Discrete(2))
    args = lambda: default_args(num_steps=T, num_minibatches=2, update_epochs=2, clip_coef=0.2)
    host, fake = _pair(lambda: MlpAgent(envs),
                       lambda ag: PPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    obs = rs.standard_normal((T + 1, N, 4)).astype(np.float32)
    _compare_rollout_and_update(host, fake, obs, dones, rewards, [host.agent], [fake.agent])

    monkeypatch.setenv("MI355PPO_CNN", "miopen")
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    host, fake = _pair(lambda: AtariAgent(envs),
                       lambda ag: PPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    fake.fused_cnn = False
    _compare_rollout_and_update(host, fake, _frames(rs, T, N, (4, 84, 84)), dones, rewards, [host.agent], [fake.agent])
    assert fake.relayout and tuple(fake.obs.shape[2:]) == (84, 84, 4)


def test_core_learner_hip_branch_continuous_actions():
    """ppo_continuous_action.py's agent (Normal policy with a state-independent log-std): K2-normal sampling, K3-normal loss,
    the log-std gradient added onto the flat buffer."""
    from cleanrl_amd.agents import ContinuousAgent

    T, N = 8, 4
    rs = np.random.RandomState(9)
    dones, rewards = _episode_streams(rs, T, N)
    envs = SimpleNamespace(single_observation_space=E.Box(-10, 10, (17,)), single_action_space=E.Box(-1, 1, (6,)))
    args = lambda: default_args(num_steps=T, num_minibatches=2, update_epochs=2, clip_coef=0.2, ent_coef=0.0, learning_rate=3e-4)
    host, fake = _pair(lambda: ContinuousAgent(envs),
                       lambda ag: PPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    obs = rs.standard_normal((T + 1, N, 17)).astype(np.float32)
    _compare_rollout_and_update(host, fake, obs, dones, rewards, [host.agent], [fake.agent], lr=3e-4)
    assert fake.agent.actor_logstd.abs().sum().item() > 0                    # the shared log-std moved


class FakeCnn:
    """CPU stand-ins for the conv entry points of cleanrl_amd/cnn.py (same signatures, layouts and layer / mode pairing).  A
    repacked matrix is represented by its buffer's address -> a snapshot of the weights AT REPACK TIME, so a stale cached
    matrix (a missed ``weights_version`` bump) computes with stale weights and the comparison with the host path fails."""

    def __init__(self):
        self.reg = {}
        self.repacks = 0

    def repack_weights(self, W, layer, mode=0, out=None):
        from cleanrl_amd import cnn

        cin, cout, k, _, _, _ = cnn.LAYERS[layer]
        _chk(W, torch.float32, f"W{layer}", (cout, cin, k, k))
        assert (layer, mode) in ((1, 0), (1, 4), (2, 0), (3, 0), (3, 1), (2, 2), (3, 3), (2, 5))
        numel = (cnn.BT_CLASSES_NUMEL if mode == 3 else cnn.QPACK_NUMEL if mode == 4 else cnn.BT2_CLASSES_NUMEL if mode == 5
                 else W.numel())
        if out is None:
            out = torch.empty(numel)
        _chk(out, torch.float32, "Bt", (numel,))
        self.reg[out.data_ptr()] = (W.detach().clone(), layer, mode)
        self.repacks += 1
        return out

    def _w(self, Bt, layer, modes):
        W, l, m = self.reg[Bt.data_ptr()]
        assert l == layer and m in modes, (l, m, layer, modes)
        return W

    def conv_fwd(self, src, Bt, bias, layer, inds=None, out=None, variant=0):
        from cleanrl_amd import cnn

        cin, cout, k, s, hin, hout = cnn.LAYERS[layer]
        assert variant in (0, cnn.VARIANT_Q) and (variant == 0 or layer == 1)
        W = self._w(Bt, layer, (4,) if variant == cnn.VARIANT_Q else (0,))      # the pack and the kernel must belong together
        x = src if inds is None else src[inds]
        if layer == 1:
            _chk(src, torch.uint8, "src")
            x = x.float() / 255.0
        else:
            assert inds is None
            _chk(src, torch.float32, "src", (src.shape[0], hin, hin, cin))
        y = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), W, bias, stride=s)).permute(0, 2, 3, 1)
        if out is None:
            return y.contiguous()
        return _chk(out, torch.float32, "out", tuple(y.shape)).copy_(y)

    def conv_dgrad(self, dz, Bt, act_in, layer, out=None, variant=0):
        from cleanrl_amd import cnn

        cin, cout, k, s, hin, hout = cnn.LAYERS[layer]
        W = self._w(Bt, layer, (3,) if variant == 5 else (5,) if variant == 6 else ((1,) if layer == 3 else (2,)))
        _chk(dz, torch.float32, "dz", (dz.shape[0], hout, hout, cout))
        _chk(act_in, torch.float32, "act_in", (dz.shape[0], hin, hin, cin))
        gi = torch.nn.functional.conv_transpose2d(dz.permute(0, 3, 1, 2), W, stride=s).permute(0, 2, 3, 1) * (act_in > 0)
        if out is None:
            return gi.contiguous()
        return _chk(out, torch.float32, "out", tuple(gi.shape)).copy_(gi)

    def conv_wgrad(self, src, dz, layer, inds=None, out=None, amax=None):
        from cleanrl_amd import cnn

        if amax is not None:                       # (src record, dz record); layer 1: (None, dz record) -- the uint8 frames are exact
            if amax[0] is not None:
                self._holds(amax[0], src, f"conv{layer} weight gradient's input")
            self._holds(amax[1], dz, f"conv{layer} weight gradient's dz")

        cin, cout, k, s, hin, hout = cnn.LAYERS[layer]
        _chk(dz, torch.float32, "dz", (dz.shape[0], hout, hout, cout))
        x = src if inds is None else src[inds]
        if layer == 1:
            _chk(src, torch.uint8, "src")
            x = x.float() / 255.0
        dW = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (cout, cin, k, k), dz.permute(0, 3, 1, 2), stride=s)
        if out is not None:                        # direct_grads: the kernel writes where the optimizer reads
            _chk(out[0], torch.float32, "dW", (cout, cin, k, k)).copy_(dW)
            _chk(out[1], torch.float32, "db", (cout,)).copy_(dz.sum((0, 1, 2)))
            return out
        return dW, dz.sum((0, 1, 2))

    # ---- kernel Z (round 3): a pack is derived from a repacked (N, K) matrix and remembers which (weights, layer, mode) that was
    def fc_pack(self, B, out=None):
        assert B.dim() == 2 and B.stride(1) == 1 and B.dtype == torch.float32
        key = B.data_ptr()
        src = self.reg.get(key)                      # a view of a repacked conv matrix, or an FC weight (snapshot it)
        if src is None:
            src = (B.detach().clone(), "fc", tuple(B.shape))
        if out is None:
            out = torch.empty(16, dtype=torch.uint8)
        self.reg[out.data_ptr()] = ("zpack",) + tuple(src)
        self.packs = getattr(self, "packs", 0) + 1
        return out

    def _zw(self, pack, layer, modes):
        tag, W, l, m = self.reg[pack.data_ptr()]
        assert tag == "zpack" and l == layer and m in modes, (tag, l, m, layer, modes)
        return W

    # ReLU masks as bits (include/mi355ppo.h): word w, bit b <-> flat element 32 w + b, set where the activation is > 0
    @staticmethod
    def _write_bits(bits, y):
        w = (y.reshape(-1, 32) > 0).to(torch.int64)
        words = (w << torch.arange(32)).sum(1)
        _chk(bits, torch.int32, "bits", (y.numel() // 32,)).copy_(torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32))
        self_bits = getattr(FakeCnn, "bits_written", 0)
        FakeCnn.bits_written = self_bits + 1

    def conv1q_fwd_bits(self, obs_u8, pack, bias, inds, out, bits):
        from cleanrl_amd import cnn

        y = self.conv_fwd(obs_u8, pack, bias, 1, inds, out, variant=cnn.VARIANT_Q)
        self._write_bits(bits, y)
        return y

    def conv_fwd_packed(self, src, pack, bias, layer, out=None, bits=None, amax=None):
        from cleanrl_amd import cnn

        if amax is not None:
            return self.conv_fwd_packed_h(src, pack, bias, layer, out, bits, amax)

        cin, cout, k, s, hin, hout = cnn.LAYERS[layer]
        W = self._zw(pack, layer, (0,))
        _chk(src, torch.float32, "src", (src.shape[0], hin, hin, cin))
        y = torch.relu(torch.nn.functional.conv2d(src.permute(0, 3, 1, 2), W, bias, stride=s)).permute(0, 2, 3, 1)
        y = y.contiguous() if out is None else _chk(out, torch.float32, "out", tuple(y.shape)).copy_(y)
        if bits is not None:
            self._write_bits(bits, y)
        return y

    def conv_dgrad_packed(self, dz, pack, act_in, layer, out=None, bits=None, amax=None):
        from cleanrl_amd import cnn

        if amax is not None:
            return self.conv_dgrad_packed_h(dz, pack, act_in, layer, out, bits, amax)

        cin, cout, k, s, hin, hout = cnn.LAYERS[layer]
        W = self._zw(pack, layer, (1,) if layer == 3 else (2,))
        _chk(dz, torch.float32, "dz", (dz.shape[0], hout, hout, cout))
        if bits is not None:                              # the kernel reads ONLY the bits: they must be the activation's sign
            mask = cnn.unpack_mask_bits(_chk(bits, torch.int32, "bits", (dz.shape[0] * hin * hin * cin // 32,)), (dz.shape[0], hin, hin, cin))
            assert act_in is None or torch.equal(mask, act_in > 0)
            FakeCnn.bits_read = getattr(FakeCnn, "bits_read", 0) + 1
        else:
            _chk(act_in, torch.float32, "act_in", (dz.shape[0], hin, hin, cin))
            mask = act_in > 0
        gi = torch.nn.functional.conv_transpose2d(dz.permute(0, 3, 1, 2), W, stride=s).permute(0, 2, 3, 1) * mask
        return gi.contiguous() if out is None else _chk(out, torch.float32, "out", tuple(gi.shape)).copy_(gi)

    # ---- round 5: the amax-aware entry points of the two-term f16 split.  A stand-in PRODUCER writes the exact maximum of what it stored
    # into the record it was handed; a stand-in CONSUMER asserts that the record it was handed holds the exact maximum of the tensor it is
    # about to split -- so a stale, missing or mis-assigned record (cnn._Buffers.begin_pass / owns / rec_of) fails here, on the CPU.
    @staticmethod
    def _put(rec, t):
        from cleanrl_amd import cnn

        _chk(rec, torch.int32, "amax record", (cnn.AMAX_WORDS,))
        rec.zero_()
        rec[0] = torch.tensor(float(t.abs().max()), dtype=torch.float32).view(torch.int32)
        FakeCnn.recs_written = getattr(FakeCnn, "recs_written", 0) + 1

    @staticmethod
    def _holds(rec, t, what):
        from cleanrl_amd import cnn

        got, want = cnn.amax_value(rec), float(t.abs().max())
        assert got == want, f"{what}: the record says {got!r}, the tensor's maximum is {want!r}"
        FakeCnn.recs_checked = getattr(FakeCnn, "recs_checked", 0) + 1

    def absmax(self, x, rec):
        cur = torch.tensor(float(x.abs().max()), dtype=torch.float32)
        old = rec[:1].view(torch.float32)
        rec[0] = torch.maximum(cur, old[0]).view(torch.int32)
        FakeCnn.absmax_calls = getattr(FakeCnn, "absmax_calls", 0) + 1
        return rec

    def fc_pack_f16x2(self, B, b_amax=None, out=None):
        out = self.fc_pack(B, out)
        tag = self.reg[out.data_ptr()]
        self.reg[out.data_ptr()] = ("zpack_h",) + tuple(tag[1:])
        return out

    def _zwh(self, pack, layer, modes):
        tag, W, l, m = self.reg[pack.data_ptr()]
        assert tag == "zpack_h" and l == layer and m in modes, (tag, l, m, layer, modes)      # an f16x2 entry point needs an f16x2 pack
        return W

    def conv1q_fwd_amax(self, obs_u8, pack, bias, inds, out, bits, dst_amax):
        from cleanrl_amd import cnn

        y = self.conv_fwd(obs_u8, pack, bias, 1, inds, out, variant=cnn.VARIANT_Q)
        if bits is not None:
            self._write_bits(bits, y)
        self._put(dst_amax, y)
        return y

    def conv_fwd_packed_h(self, src, pack, bias, layer, out, bits, amax):
        from cleanrl_amd import cnn

        cin, cout, k, s, hin, hout = cnn.LAYERS[layer]
        W = self._zwh(pack, layer, (0,))
        self._holds(amax[0], src, f"conv{layer} forward's input")
        y = torch.relu(torch.nn.functional.conv2d(src.permute(0, 3, 1, 2), W, bias, stride=s)).permute(0, 2, 3, 1)
        y = y.contiguous() if out is None else _chk(out, torch.float32, "out", tuple(y.shape)).copy_(y)
        if bits is not None:
            self._write_bits(bits, y)
        if amax[1] is not None:
            self._put(amax[1], y)
        return y

    def conv_dgrad_packed_h(self, dz, pack, act_in, layer, out, bits, amax):
        from cleanrl_amd import cnn

        cin, cout, k, s, hin, hout = cnn.LAYERS[layer]
        W = self._zwh(pack, layer, (1,) if layer == 3 else (2,))
        self._holds(amax[0], dz, f"conv{layer} data gradient's dz")
        if bits is not None:
            mask = cnn.unpack_mask_bits(bits, (dz.shape[0], hin, hin, cin))
            FakeCnn.bits_read = getattr(FakeCnn, "bits_read", 0) + 1
        else:
            mask = act_in > 0
        gi = torch.nn.functional.conv_transpose2d(dz.permute(0, 3, 1, 2), W, stride=s).permute(0, 2, 3, 1) * mask
        gi = gi.contiguous() if out is None else _chk(out, torch.float32, "out", tuple(gi.shape)).copy_(gi)
        if amax[1] is not None:
            self._put(amax[1], gi)
        return gi

    def trunk_fwd(self, obs_u8, inds, bt1, b1, bt2, b2, bt3, b3, a1, a2, a3, conv1_variant=0):
        self.conv_fwd(obs_u8, bt1, b1, 1, inds, a1, variant=conv1_variant)
        self.conv_fwd(a1, bt2, b2, 2, None, a2)
        return self.conv_fwd(a2, bt3, b3, 3, None, a3)


def test_fused_cnn_branch_and_weight_matrix_cache(monkeypatch):
    """The main path's Python (``AtariAgent.heads_u8`` -> ``cnn.NatureTrunkFn`` / ``LinearReLUHwcFn``, the learner's repacked-
    matrix cache with its version bumps) over CPU stand-ins for the conv entry points: rollout, GAE and 2 x 2 minibatch
    updates must agree with the host path, which they only can if every forward after an optimiser step re-derives the
    matrices."""
    from cleanrl_amd import cnn
    from cleanrl_amd.agents import AtariAgent

    fk = FakeCnn()
    for name in ("repack_weights", "conv_fwd", "conv_dgrad", "conv_wgrad", "trunk_fwd", "fc_pack", "conv_fwd_packed", "conv_dgrad_packed",
                 "conv1q_fwd_bits"):
        monkeypatch.setattr(cnn, name, getattr(fk, name))
    FakeCnn.bits_written = FakeCnn.bits_read = 0
    monkeypatch.setattr(cnn, "heads_supported", lambda actor, critic: False)      # HeadsFn binds the library directly
    T, N = 6, 4
    rs = np.random.RandomState(13)
    dones, rewards = _episode_streams(rs, T, N)
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    args = lambda: default_args(num_steps=T, num_minibatches=2, update_epochs=2)
    host, fake = _pair(lambda: AtariAgent(envs),
                       lambda ag: PPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    fake.fused_cnn = True
    fake._x_roll = None
    _compare_rollout_and_update(host, fake, _frames(rs, T, N, (4, 84, 84)), dones, rewards, [host.agent], [fake.agent])
    bufs = fake.agent._trunk.bufs
    assert bufs.cache_weights and bufs.weights_version == 4                          # one bump per optimiser step
    # rollout: 3 forward matrices derived once for its T + 1 forwards and still valid for the first minibatch's forward;
    # then 2 data-gradient matrices, and 3 + 2 after each of the following three optimiser steps: never a stale one (the
    # comparison above), never a redundant one.  Layers 2 / 3 run on kernel Z: each of their four matrices is followed by its
    # pack, derived exactly as often (2 forward packs per forward-matrix set, 2 data-gradient packs per backward set).
    assert fk.repacks == 3 + 2 + 3 * 5
    assert fk.packs == 2 + 2 + 3 * 4
    # ReLU masks as bits: written by the three forwards of each of the 4 minibatch updates (never by a rollout forward, which no
    # backward follows), read by the two conv data gradients of each
    assert FakeCnn.bits_written == 3 * 4 and FakeCnn.bits_read == 2 * 4
    # a second rollout sees the updated weights (the cache was invalidated by the last step)
    logits_a, _ = fake._heads_rollout(fake.obs[0])
    with torch.no_grad():
        logits_b, _ = fake.agent.heads(fake.obs[0].float().permute(0, 3, 1, 2) / 255.0)
    assert torch.allclose(logits_a, logits_b, atol=1e-5)


def test_fused_cnn_branch_on_the_f16_split_record_protocol(monkeypatch):
    """Round 5: the same rollout + 2 x 2 minibatch updates with the trunk on the two-term f16 split (``cnn._Buffers.f16``): every amax record a
    consumer is handed must hold the exact maximum of the tensor it describes (the stand-ins assert it), whichever kernel -- or the
    ``rec_of`` fallback, here for dz3, which comes from torch's threshold_backward -- produced it; results equal the host path's."""
    from cleanrl_amd import cnn
    from cleanrl_amd.agents import AtariAgent

    fk = FakeCnn()
    for name in ("repack_weights", "conv_fwd", "conv_dgrad", "conv_wgrad", "trunk_fwd", "fc_pack", "fc_pack_f16x2", "conv_fwd_packed", "conv_dgrad_packed",
                 "conv1q_fwd_bits", "conv1q_fwd_amax", "absmax"):
        monkeypatch.setattr(cnn, name, getattr(fk, name))
    FakeCnn.bits_written = FakeCnn.bits_read = FakeCnn.recs_written = FakeCnn.recs_checked = FakeCnn.absmax_calls = 0
    monkeypatch.setattr(cnn, "heads_supported", lambda actor, critic: False)
    T, N = 4, 4
    rs = np.random.RandomState(17)
    dones, rewards = _episode_streams(rs, T, N)
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    args = lambda: default_args(num_steps=T, num_minibatches=2, update_epochs=2)
    host, fake = _pair(lambda: AtariAgent(envs),
                       lambda ag: PPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    fake.fused_cnn = True
    fake._x_roll = None
    fake.agent._trunk = cnn.NatureTrunk()
    fake.agent._trunk.bufs.split = "f16x2"
    fake.agent._trunk.bufs.f16_on_cpu = True
    _compare_rollout_and_update(host, fake, _frames(rs, T, N, (4, 84, 84)), dones, rewards, [host.agent], [fake.agent])
    # per forward 3 records written; per update pass the two data gradients write 2 more; consumers: 2 per forward (conv2, conv3), and per
    # backward: wgrad3 (2), dgrad3 (1), wgrad2 (2), dgrad2 (1), wgrad1 (1) = 7
    n_fwd, n_upd = (T + 1) + 4, 4
    assert FakeCnn.recs_written == 3 * n_fwd + 2 * n_upd, FakeCnn.recs_written
    assert FakeCnn.recs_checked == 2 * n_fwd + 7 * n_upd, FakeCnn.recs_checked
    assert FakeCnn.absmax_calls == n_upd                # dz3 of every update: the one tensor no kernel of this path produced
    assert FakeCnn.bits_written == 3 * 4 and FakeCnn.bits_read == 2 * 4


def test_ragged_tail_minibatch_on_the_hip_branch():
    """B % num_minibatches != 0: ``range(0, B, M)`` yields one more (short) minibatch (ppo.py:246-248); the HIP branch's
    per-minibatch scalar rows must cover it."""
    from cleanrl_amd.agents import MlpAgent

    T, N = 5, 3                                          # B = 15, 4 minibatches -> M = 3 -> 5 minibatches per epoch
    rs = np.random.RandomState(5)
    dones, rewards = _episode_streams(rs, T, N)
    envs = SimpleNamespace(single_observation_space=E.Box(-1, 1, (4,)), single_action_space=E.Discrete(2))
    args = lambda: default_args(num_steps=T, num_minibatches=4, update_epochs=2, clip_coef=0.2)
    host, fake = _pair(lambda: MlpAgent(envs),
                       lambda ag: PPOLearner(ag, args(), envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    obs = rs.standard_normal((T + 1, N, 4)).astype(np.float32)
    mh, mf = _compare_rollout_and_update(host, fake, obs, dones, rewards, [host.agent], [fake.agent])
    assert mh["num_updates"] == mf["num_updates"] == 10


def test_every_epoch_has_its_own_permutation_rows():
    """The host runs ahead of the GPU: epoch e+1's permutation must not overwrite the pinned row epoch e's asynchronous
    H2D copy still reads.  Every epoch therefore owns a pinned row and a device row."""
    from cleanrl_amd.agents import MlpAgent

    T, N = 4, 4
    envs = SimpleNamespace(single_observation_space=E.Box(-1, 1, (4,)), single_action_space=E.Discrete(2))
    torch.manual_seed(0)
    L = _to_fake_hip(PPOLearner(MlpAgent(envs), default_args(num_steps=T, num_minibatches=2, update_epochs=3),
                                envs.single_observation_space, envs.single_action_space, N, torch.device("cpu")))
    assert tuple(L._inds_pin.shape) == tuple(L._inds_dev.shape) == (3, T * N)
    perms = [np.random.RandomState(e).permutation(T * N) for e in range(3)]
    rows = [L.upload_permutation(e, p) for e, p in enumerate(perms)]
    assert len({r.data_ptr() for r in rows}) == 3 and len({L._inds_pin[e].data_ptr() for e in range(3)}) == 3
    for e in range(3):                                   # later uploads left the earlier rows alone
        assert np.array_equal(L._inds_pin[e].numpy(), perms[e]) and np.array_equal(rows[e].numpy(), perms[e])


def test_ppg_hip_branch_world_size_two_allreduce(capsys, monkeypatch):
    """The auxiliary phase's ``dist.all_reduce`` + ``/world_size`` on the HIP branch (round-1 advisor finding: ``dist`` was
    never imported there).  Two ranks that hold identical data: SUM doubles the flat gradient, ``/world_size`` halves it
    (both exact in f32), so a world_size=2 learner must end exactly where a world_size=1 learner ends -- through the
    policy phase (learner.py) and the auxiliary phase (learner_ppg.py), the stale-gradient quirk included."""
    from cleanrl_amd import learner as learner_mod, learner_ppg as ppg_mod

    calls = []

    def identical_peer_sum(t, op=None):
        assert op == ppg_mod.dist.ReduceOp.SUM
        calls.append(t.data_ptr())
        t.mul_(2.0)

    monkeypatch.setattr(learner_mod.dist, "all_reduce", identical_peer_sum)
    assert ppg_mod.dist is learner_mod.dist
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (64, 64, 3), np.uint8), single_action_space=E.Discrete(15))
    T, N = 4, 4
    rs = np.random.RandomState(3)
    frames = _frames(rs, T, N, (64, 64, 3))
    dones, rewards = _episode_streams(rs, T, N)
    pargs = lambda: default_args(num_steps=T, num_minibatches=2, gamma=0.999, clip_coef=0.2, adv_norm_fullbatch=True, e_policy=1,
                                 e_auxiliary=2, beta_clone=1.0, num_aux_rollouts=2, n_aux_grad_accum=1, aux_batch_rollouts=N,
                                 n_iteration=1, learning_rate=5e-4)
    mk = lambda w: _to_fake_hip(PPGLearner(PPGAgent(envs), pargs(), envs.single_observation_space, envs.single_action_space, N,
                                           torch.device("cpu"), world_size=w))
    torch.manual_seed(0)
    two = mk(2)
    torch.manual_seed(0)
    one = mk(1)
    for L in (two, one):
        _drive(L, frames, dones, rewards, 11, True)
        np.random.seed(3)
        torch.manual_seed(3)
        L.update(5e-4)
        np.random.seed(4)
        L.aux_phase()
    capsys.readouterr()
    assert len(calls) == 2 + 4 and set(calls) == {two.flat.grads.data_ptr()}     # 2 policy minibatches + 2 epochs x 2 aux steps
    d = (_params([two.agent]) - _params([one.agent])).abs().max().item()
    assert d <= 1e-7, d
    for k in one.last_aux:
        assert abs(two.last_aux[k] - one.last_aux[k]) <= 1e-6 * max(1.0, abs(one.last_aux[k]))
