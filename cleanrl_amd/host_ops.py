"""Host (CPU) path: the ``*_cpu`` host-pointer twins of ``libmi355ppo.so`` behind the same seams as the GPU path.

Selected ONLY when the user asks for a CPU device (``--no-cuda``: BASELINE config A "CartPole on CPU,
plumbing", and the world_size-2 ``gloo`` tests of the data-parallel logic).  It is not a fallback: a
CUDA device always runs the HIP kernels and raises if ``libmi355ppo.so`` is missing, and these functions
refuse CUDA tensors.  The twins (csrc/host_twins.hip, declared in include/mi355ppo.h) are the device
kernels' own row / element functions compiled for the host, so the CPU loop crosses the SAME C ABI seams
as the GPU loop: GAE (ppo.py:218-231), the fused loss forward + backward (ppo.py:250-285 and its autograd)
and, for tests, sampling and clip + Adam.  The learners whose loss has extra terms (LSTM state, RND's second value
head and distillation loss) cross the same twin on their logits / value and add their terms outside.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

LOSS_SCALARS = 7


def _p(t):
    if t is None:
        return None
    if t.device.type != "cpu":
        raise TypeError("cleanrl_amd.host_ops works on CPU tensors only (CUDA tensors run the HIP kernels: cleanrl_amd.ops)")
    assert t.is_contiguous(), "host twins take contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def gae(rewards, dones, values, next_done, next_value, gamma, gae_lambda):
    """ppo.py:218-231 through ``mi355ppo_gae_f32_cpu`` -> (advantages, returns); bit-equal to the reference's loop."""
    lib = _lib.load()
    T, N = rewards.shape
    r, d, v = _f32(rewards), _f32(dones), _f32(values)
    nd, nv = _f32(next_done).reshape(-1), _f32(next_value).reshape(-1)
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    _lib.check(lib.mi355ppo_gae_f32_cpu(_p(r), _p(d), _p(v), _p(nd), _p(nv), _p(adv), _p(ret), T, N, float(gamma),
                                        float(gae_lambda)), "mi355ppo_gae_f32_cpu")
    return adv, ret


class _CategoricalLossTwin(torch.autograd.Function):
    """K3 on the host: ``mi355ppo_loss_categorical_fwd_bwd_f32_cpu`` computes the loss scalars AND d loss / d (logits, value)
    in one pass (ppo.py:250-285 + its autograd down to the network outputs); backward hands the stored gradients on."""

    @staticmethod
    def forward(ctx, logits, value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, hp):
        lib = _lib.load()
        M, A = logits.shape
        lg, vl = _f32(logits), _f32(value).reshape(-1)
        inds = mb_inds.to(torch.int64).contiguous()
        scalars = torch.empty(LOSS_SCALARS)
        dlogits, dvalue = torch.empty_like(lg), torch.empty_like(vl)
        _lib.check(lib.mi355ppo_loss_categorical_fwd_bwd_f32_cpu(
            _p(lg), _p(vl), _p(inds), _p(_f32(b_actions)), _p(_f32(b_logprobs)), _p(_f32(b_advantages)), _p(_f32(b_returns)),
            _p(_f32(b_values)), M, A, hp["clip_coef"], hp["ent_coef"], hp["vf_coef"], int(hp["norm_adv"]), int(hp["clip_vloss"]),
            None, _p(scalars), _p(dlogits), _p(dvalue)), "mi355ppo_loss_categorical_fwd_bwd_f32_cpu")
        ctx.save_for_backward(dlogits, dvalue.reshape(value.shape))
        ctx.mark_non_differentiable(scalars)
        return scalars[0].clone(), scalars

    @staticmethod
    def backward(ctx, g_loss, _g_scalars):
        dlogits, dvalue = ctx.saved_tensors
        return dlogits * g_loss, dvalue * g_loss, None, None, None, None, None, None, None


class _NormalLossTwin(torch.autograd.Function):
    """K3' on the host (``mi355ppo_loss_normal_fwd_bwd_f32_cpu``, ppo_continuous_action.py:265-300 + autograd)."""

    @staticmethod
    def forward(ctx, mean, logstd, value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, hp):
        lib = _lib.load()
        M, D = mean.shape
        mu, ls, vl = _f32(mean), _f32(logstd).reshape(-1), _f32(value).reshape(-1)
        inds = mb_inds.to(torch.int64).contiguous()
        scalars = torch.empty(LOSS_SCALARS)
        dmean, dlogstd, dvalue = torch.empty_like(mu), torch.empty_like(ls), torch.empty_like(vl)
        _lib.check(lib.mi355ppo_loss_normal_fwd_bwd_f32_cpu(
            _p(mu), _p(ls), _p(vl), _p(inds), _p(_f32(b_actions)), _p(_f32(b_logprobs)), _p(_f32(b_advantages)), _p(_f32(b_returns)),
            _p(_f32(b_values)), M, D, hp["clip_coef"], hp["ent_coef"], hp["vf_coef"], int(hp["norm_adv"]), int(hp["clip_vloss"]),
            None, _p(scalars), _p(dmean), _p(dlogstd), _p(dvalue)), "mi355ppo_loss_normal_fwd_bwd_f32_cpu")
        ctx.save_for_backward(dmean, dlogstd.reshape(logstd.shape), dvalue.reshape(value.shape))
        ctx.mark_non_differentiable(scalars)
        return scalars[0].clone(), scalars

    @staticmethod
    def backward(ctx, g_loss, _g_scalars):
        dmean, dlogstd, dvalue = ctx.saved_tensors
        return dmean * g_loss, dlogstd * g_loss, dvalue * g_loss, None, None, None, None, None, None, None


def _hp(clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss):
    return dict(clip_coef=float(clip_coef), ent_coef=float(ent_coef), vf_coef=float(vf_coef), norm_adv=bool(norm_adv),
                clip_vloss=bool(clip_vloss))


def ppo_loss_categorical(logits, newvalue, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, clip_coef,
                         ent_coef, vf_coef, norm_adv, clip_vloss):
    """(loss, scalars7 in ops.LOSS_SCALAR_NAMES order) of one minibatch from the network outputs and the FLAT batch arrays
    (the twin gathers rows ``mb_inds`` itself, as the device kernel does); ``loss.backward()`` reaches logits and value."""
    return _CategoricalLossTwin.apply(logits, newvalue, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values,
                                      _hp(clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss))


def ppo_loss_normal(mean, logstd, newvalue, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, clip_coef,
                    ent_coef, vf_coef, norm_adv, clip_vloss):
    """The continuous-action twin: mean (M, D), logstd (D) or (1, D); ``loss.backward()`` reaches mean, logstd and value."""
    return _NormalLossTwin.apply(mean, logstd, newvalue, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values,
                                 _hp(clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss))


def categorical_sample(logits, noise_exp1=None, seed=0, offset=0):
    """``mi355ppo_categorical_sample_f32_cpu`` -> (action int64, logprob, entropy).  With ``noise_exp1`` None the draws come from
    the device kernel's Philox stream (same words for the same (seed, offset, row))."""
    lib = _lib.load()
    lg = _f32(logits)
    B, A = lg.shape
    act, lp, ent = torch.empty(B, dtype=torch.int64), torch.empty(B), torch.empty(B)
    nz = _f32(noise_exp1) if noise_exp1 is not None else None
    _lib.check(lib.mi355ppo_categorical_sample_f32_cpu(_p(lg), _p(nz), int(seed), int(offset), _p(act), None, _p(lp), _p(ent), B, A),
               "mi355ppo_categorical_sample_f32_cpu")
    return act, lp, ent


def categorical_logprob_entropy(logits, action):
    lib = _lib.load()
    lg = _f32(logits)
    B, A = lg.shape
    act = action.to(torch.int64).contiguous()
    lp, ent = torch.empty(B), torch.empty(B)
    _lib.check(lib.mi355ppo_categorical_logprob_entropy_f32_cpu(_p(lg), _p(act), None, _p(lp), _p(ent), B, A),
               "mi355ppo_categorical_logprob_entropy_f32_cpu")
    return lp, ent


def categorical_logprob_entropy_bwd(logits, action, g_logprob, g_entropy):
    lib = _lib.load()
    lg = _f32(logits)
    B, A = lg.shape
    act = action.to(torch.int64).contiguous()
    out = torch.empty_like(lg)
    _lib.check(lib.mi355ppo_categorical_logprob_entropy_bwd_f32_cpu(
        _p(lg), _p(act), None, _p(_f32(g_logprob)) if g_logprob is not None else None,
        _p(_f32(g_entropy)) if g_entropy is not None else None, _p(out), B, A), "mi355ppo_categorical_logprob_entropy_bwd_f32_cpu")
    return out


def normal_sample(mean, logstd, noise=None, seed=0, offset=0):
    lib = _lib.load()
    mu, ls = _f32(mean), _f32(logstd).reshape(-1)
    B, D = mu.shape
    act, lp, ent = torch.empty_like(mu), torch.empty(B), torch.empty(B)
    nz = _f32(noise) if noise is not None else None
    _lib.check(lib.mi355ppo_normal_sample_f32_cpu(_p(mu), _p(ls), _p(nz), int(seed), int(offset), _p(act), _p(lp), _p(ent), B, D),
               "mi355ppo_normal_sample_f32_cpu")
    return act, lp, ent


def normal_logprob_entropy(mean, logstd, action):
    lib = _lib.load()
    mu, ls, ac = _f32(mean), _f32(logstd).reshape(-1), _f32(action)
    B, D = mu.shape
    lp, ent = torch.empty(B), torch.empty(B)
    _lib.check(lib.mi355ppo_normal_logprob_entropy_f32_cpu(_p(mu), _p(ls), _p(ac), _p(lp), _p(ent), B, D),
               "mi355ppo_normal_logprob_entropy_f32_cpu")
    return lp, ent


def normal_logprob_entropy_bwd(mean, logstd, action, g_logprob, g_entropy):
    lib = _lib.load()
    mu, ls, ac = _f32(mean), _f32(logstd).reshape(-1), _f32(action)
    B, D = mu.shape
    dmean, dls = torch.empty_like(mu), torch.empty_like(mu)
    _lib.check(lib.mi355ppo_normal_logprob_entropy_bwd_f32_cpu(
        _p(mu), _p(ls), _p(ac), _p(_f32(g_logprob)) if g_logprob is not None else None,
        _p(_f32(g_entropy)) if g_entropy is not None else None, _p(dmean), _p(dls), B, D), "mi355ppo_normal_logprob_entropy_bwd_f32_cpu")
    return dmean, dls


def clip_adam_(params, grads, exp_avg, exp_avg_sq, step, lr, max_grad_norm, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-5):
    """In place on flat f32 CPU buffers (``mi355ppo_clip_adam_f32_cpu``): grads * grad_scale -> global-norm clip -> Adam; the
    gradient buffer is zeroed.  Returns the pre-clip norm."""
    lib = _lib.load()
    for t in (params, grads, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.dim() == 1
    norm = torch.empty(1)
    _lib.check(lib.mi355ppo_clip_adam_f32_cpu(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), params.numel(), float(grad_scale),
                                              float(max_grad_norm), float(lr), beta1, beta2, eps, int(step), _p(norm)),
               "mi355ppo_clip_adam_f32_cpu")
    return norm


def obs_u8_to_f32(src_u8, inds=None, scale_255=True):
    lib = _lib.load()
    src = src_u8.contiguous()
    rows_total = src.shape[0]
    row_bytes = src.numel() // max(rows_total, 1)
    idx = inds.to(torch.int64).contiguous() if inds is not None else None
    rows = idx.numel() if idx is not None else rows_total
    out = torch.empty((rows,) + tuple(src.shape[1:]), dtype=torch.float32)
    _lib.check(lib.mi355ppo_obs_u8_to_f32_cpu(_p(src), _p(idx), _p(out), rows, row_bytes, int(bool(scale_255))),
               "mi355ppo_obs_u8_to_f32_cpu")
    return out
