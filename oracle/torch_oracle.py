"""CPU / fp32 restatement of the PPO hot path in plain torch ops.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) -- never imported by the
product.  The reference computes this path with stock torch ops, so a torch
restatement that keeps the reference's operation order reproduces its
floating-point results on CPU; each function cites the reference lines it
follows (``cleanrl/ppo_atari_multigpu.py`` unless another file is named; the
same blocks appear verbatim in ``ppo.py``, ``ppo_atari.py``,
``ppo_atari_envpool.py`` and ``ppo_continuous_action.py``).

Third-party code on the path (not vendored under /root/reference):
``torch==2.4.1`` is the reference's pin (``pyproject.toml``); the formulas of
``torch.distributions.Categorical`` / ``Normal``, ``torch.multinomial``,
``nn.utils.clip_grad_norm_`` and ``optim.Adam`` were read from the installed
torch 2.10.0 and restated here; goldens are minted with 2.10.0.

Pinned by ``tests/test_oracle_golden.py`` against ``tests/golden/*.npz`` (outputs
of the reference's own lines, minted by ``oracle/mint_goldens.py``).
"""
from __future__ import annotations

import math

import torch


# --------------------------------------------------------------------------------------
# K1  GAE   (ppo_atari_multigpu.py:288-301 == ppo.py:218-231 == ppo_atari_envpool.py:250-263)
# --------------------------------------------------------------------------------------
def gae(rewards, dones, values, next_done, next_value, gamma: float, gae_lambda: float):
    """rewards/dones/values: (T,N) f32; next_done/next_value: (N,) f32 -> advantages, returns (T,N).

    Operation order kept exactly: ``((gamma*nv)*nnt)``, ``((r + .) - v)``,
    ``(((gamma*gae_lambda)*nnt)*last)`` with ``gamma*gae_lambda`` formed in Python
    double and only then rounded to f32 by torch's scalar multiply; ``last`` starts
    as the Python int 0 (``:291``).
    """
    T = rewards.shape[0]
    next_value = next_value.reshape(1, -1)
    advantages = torch.zeros_like(rewards)
    lastgaelam = 0
    for t in reversed(range(T)):
        if t == T - 1:
            nextnonterminal = 1.0 - next_done
            nextvalues = next_value
        else:
            nextnonterminal = 1.0 - dones[t + 1]
            nextvalues = values[t + 1]
        delta = rewards[t] + gamma * nextvalues * nextnonterminal - values[t]
        lastgaelam = delta + gamma * gae_lambda * nextnonterminal * lastgaelam
        advantages[t] = lastgaelam
    returns = advantages + values
    return advantages, returns


# --------------------------------------------------------------------------------------
# K2  Categorical(logits)   (Agent.get_action_and_value :153-159 -> torch categorical.py)
# --------------------------------------------------------------------------------------
def categorical_normalize(logits):
    """``Categorical.__init__``: ``logits - logits.logsumexp(-1, keepdim=True)``; ``probs = softmax(.)``."""
    norm = logits - logits.logsumexp(dim=-1, keepdim=True)
    probs = torch.softmax(norm, dim=-1)
    return norm, probs


def categorical_sample_from_noise(logits, noise_exp1):
    """``Categorical.sample`` -> ``torch.multinomial(probs, 1, True)`` whose one-draw fast path is
    ``argmax(probs / q)``, ``q ~ Exponential(1)`` (ATen ``multinomial_out``).  ``noise_exp1`` is ``q``."""
    _, probs = categorical_normalize(logits)
    return torch.argmax(probs / noise_exp1, dim=-1)


def categorical_logprob_entropy(logits, action):
    """``log_prob`` = gather of the normalised logits; ``entropy`` = ``-(clamp(logits, min=finfo.min) * probs).sum(-1)``."""
    norm, probs = categorical_normalize(logits)
    logprob = norm.gather(-1, action.long().unsqueeze(-1)).squeeze(-1)
    min_real = torch.finfo(norm.dtype).min
    entropy = -(torch.clamp(norm, min=min_real) * probs).sum(-1)
    return logprob, entropy


# --------------------------------------------------------------------------------------
# K2' Normal(mean, exp(logstd))   (ppo_continuous_action.py:134-141 -> torch normal.py)
# --------------------------------------------------------------------------------------
def normal_sample_from_noise(mean, logstd, noise_std_normal):
    """``Normal.sample`` = ``torch.normal(loc, scale)`` = ``z*scale + loc`` (mul then add)."""
    std = torch.exp(logstd.expand_as(mean))
    return noise_std_normal * std + mean


def normal_logprob_entropy(mean, logstd, action):
    """``log_prob(a).sum(1)``, ``entropy().sum(1)`` with ``scale = exp(logstd)`` and
    ``log_scale = scale.log()`` (NOT ``logstd`` itself), as torch's ``Normal`` does."""
    std = torch.exp(logstd.expand_as(mean))
    var = std**2
    log_scale = std.log()
    logprob = -((action - mean) ** 2) / (2 * var) - log_scale - math.log(math.sqrt(2 * math.pi))
    entropy = 0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)
    return logprob.sum(1), entropy.sum(1)


# --------------------------------------------------------------------------------------
# K3  minibatch loss   (ppo_atari_multigpu.py:321-355)
# --------------------------------------------------------------------------------------
def ppo_loss(newlogprob, entropy, newvalue, mb_logprobs, mb_advantages, mb_returns, mb_values,
             clip_coef: float, ent_coef: float, vf_coef: float, norm_adv: bool, clip_vloss: bool):
    logratio = newlogprob - mb_logprobs
    ratio = logratio.exp()
    with torch.no_grad():
        old_approx_kl = (-logratio).mean()
        approx_kl = ((ratio - 1) - logratio).mean()
        clipfrac = ((ratio - 1.0).abs() > clip_coef).float().mean()
    if norm_adv:
        mb_advantages = (mb_advantages - mb_advantages.mean()) / (mb_advantages.std() + 1e-8)
    pg_loss1 = -mb_advantages * ratio
    pg_loss2 = -mb_advantages * torch.clamp(ratio, 1 - clip_coef, 1 + clip_coef)
    pg_loss = torch.max(pg_loss1, pg_loss2).mean()
    newvalue = newvalue.view(-1)
    if clip_vloss:
        v_loss_unclipped = (newvalue - mb_returns) ** 2
        v_clipped = mb_values + torch.clamp(newvalue - mb_values, -clip_coef, clip_coef)
        v_loss_clipped = (v_clipped - mb_returns) ** 2
        v_loss = 0.5 * torch.max(v_loss_unclipped, v_loss_clipped).mean()
    else:
        v_loss = 0.5 * ((newvalue - mb_returns) ** 2).mean()
    entropy_loss = entropy.mean()
    loss = pg_loss - ent_coef * entropy_loss + v_loss * vf_coef
    return dict(loss=loss, pg_loss=pg_loss, v_loss=v_loss, entropy=entropy_loss,
                old_approx_kl=old_approx_kl, approx_kl=approx_kl, clipfrac=clipfrac)


def loss_categorical_seam(new_logits, new_value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns,
                          b_values, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss):
    """The loss at the C-ABI seam: logits/value in, 7 scalars + dlogits/dvalue out (autograd)."""
    logits = new_logits.detach().clone().requires_grad_(True)
    value = new_value.detach().clone().requires_grad_(True)
    idx = torch.arange(logits.shape[0]) if mb_inds is None else torch.as_tensor(mb_inds).long()
    act = b_actions.long()[idx]                                   # :320  b_actions.long()[mb_inds]
    lp, ent = categorical_logprob_entropy(logits, act)
    out = ppo_loss(lp, ent, value, b_logprobs[idx], b_advantages[idx], b_returns[idx], b_values[idx],
                   clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss)
    out["loss"].backward()
    res = {k: v.detach() for k, v in out.items()}
    res["dlogits"] = logits.grad
    res["dvalue"] = value.grad
    return res


def loss_normal_seam(new_mean, logstd, new_value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns,
                     b_values, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss):
    """Continuous-action seam (ppo_continuous_action.py:265-300): mean/logstd/value in, grads out."""
    mean = new_mean.detach().clone().requires_grad_(True)
    ls = logstd.detach().clone().requires_grad_(True)
    value = new_value.detach().clone().requires_grad_(True)
    idx = torch.arange(mean.shape[0]) if mb_inds is None else torch.as_tensor(mb_inds).long()
    lp, ent = normal_logprob_entropy(mean, ls.reshape(1, -1), b_actions[idx])
    out = ppo_loss(lp, ent, value, b_logprobs[idx], b_advantages[idx], b_returns[idx], b_values[idx],
                   clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss)
    out["loss"].backward()
    res = {k: v.detach() for k, v in out.items()}
    res["dmean"] = mean.grad
    res["dlogstd"] = ls.grad.reshape(-1)
    res["dvalue"] = value.grad
    return res


# --------------------------------------------------------------------------------------
# a8  clip_grad_norm_ + Adam on a flat vector   (:376-377 -> torch clip_grad.py / adam.py)
# --------------------------------------------------------------------------------------
def clip_adam_flat(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, max_grad_norm: float,
                   segments, grad_scale: float = 1.0, beta1=0.9, beta2=0.999, eps=1e-5):
    """Restates ``clip_grad_norm_(params, max_norm)`` followed by one ``Adam.step`` on flat f32 vectors.

    ``segments`` = list of (offset, numel) per parameter tensor: the reference's total norm is the
    2-norm of the per-tensor 2-norms.  ``grad_scale`` is the ``/ world_size`` of ``:372``.
    ``step`` is the 1-based Adam step count.  Returns new (params, exp_avg, exp_avg_sq, total_norm).
    """
    g = grads * grad_scale if grad_scale != 1.0 else grads.clone()
    norms = torch.stack([torch.linalg.vector_norm(g[o:o + n], 2) for o, n in segments])
    total_norm = torch.linalg.vector_norm(norms, 2)
    clip_coef = torch.clamp(max_grad_norm / (total_norm + 1e-6), max=1.0)
    g = g * clip_coef
    m = torch.lerp(exp_avg, g, 1 - beta1)
    v = exp_avg_sq * beta2 + (1 - beta2) * g * g
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = params - step_size * (m / denom)
    return p, m, v, total_norm


def explained_variance(b_values, b_returns):
    """:382-384 (host numpy, float32 var)."""
    import numpy as np

    y_pred, y_true = b_values.cpu().numpy(), b_returns.cpu().numpy()
    var_y = np.var(y_true)
    return np.nan if var_y == 0 else 1 - np.var(y_true - y_pred) / var_y
