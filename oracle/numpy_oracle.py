"""Independent float64 numpy derivation of the hot path (TEST INFRASTRUCTURE -- see ``oracle/__init__.py``).

``torch_oracle.py`` restates the reference with the same torch ops and lets autograd produce the gradients; this file
derives everything a second way -- plain numpy in float64, gradients in CLOSED FORM -- so that
a shared mistake in the restatement and in the kernels cannot hide.  It is compared with the goldens minted from the
reference's own lines at float32 round-off tolerances (``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import numpy as np


def gae(rewards, dones, values, next_done, next_value, gamma, gae_lambda):
    """cleanrl/ppo.py:218-231 in float64 (not bit-comparable with the f32 reference; tolerance check only)."""
    r, d, v = (np.asarray(x, np.float64) for x in (rewards, dones, values))
    T = r.shape[0]
    adv = np.zeros_like(r)
    last = np.zeros(r.shape[1])
    for t in reversed(range(T)):
        nnt = 1.0 - (np.asarray(next_done, np.float64) if t == T - 1 else d[t + 1])
        nv = np.asarray(next_value, np.float64).reshape(-1) if t == T - 1 else v[t + 1]
        delta = r[t] + gamma * nv * nnt - v[t]
        adv[t] = last = delta + gamma * gae_lambda * nnt * last
    return adv, adv + v


def loss_categorical(new_logits, new_value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, clip_coef,
                     ent_coef, vf_coef, norm_adv=True, clip_vloss=True):
    """cleanrl/ppo.py:250-285 and its gradient w.r.t. (new_logits, new_value), closed form, float64.
    Returns ``(scalars7, dlogits, dvalue)`` in the library's order (loss pg v ent old_kl kl clipfrac)."""
    x = np.asarray(new_logits, np.float64)
    v = np.asarray(new_value, np.float64).reshape(-1)
    idx = np.asarray(mb_inds)
    a = np.asarray(b_actions)[idx].astype(np.int64)
    old_lp, adv, ret, old_v = (np.asarray(t, np.float64).reshape(-1)[idx] for t in (b_logprobs, b_advantages, b_returns, b_values))
    M = x.shape[0]
    lse = np.log(np.exp(x - x.max(1, keepdims=True)).sum(1, keepdims=True)) + x.max(1, keepdims=True)
    lp = x - lse
    p = np.exp(lp)
    H = -(p * lp).sum(1)
    newlp = lp[np.arange(M), a]
    logratio = newlp - old_lp
    ratio = np.exp(logratio)
    A = (adv - adv.mean()) / (adv.std(ddof=1) + 1e-8) if norm_adv else adv
    pg1, pg2 = -A * ratio, -A * np.clip(ratio, 1 - clip_coef, 1 + clip_coef)
    pg_loss = np.maximum(pg1, pg2).mean()
    u = (v - ret) ** 2
    if clip_vloss:
        vc = old_v + np.clip(v - old_v, -clip_coef, clip_coef)
        c = (vc - ret) ** 2
        v_loss = 0.5 * np.maximum(u, c).mean()
    else:
        v_loss = 0.5 * u.mean()
    ent = H.mean()
    loss = pg_loss - ent_coef * ent + vf_coef * v_loss
    scalars = np.array([loss, pg_loss, v_loss, ent, (-logratio).mean(), ((ratio - 1) - logratio).mean(),
                        (np.abs(ratio - 1) > clip_coef).mean()])
    # gradients (torch.max splits ties 1/2 + 1/2; clamp passes gradient on the closed interval)
    inr = ((ratio >= 1 - clip_coef) & (ratio <= 1 + clip_coef)).astype(np.float64)
    w = np.where(pg1 > pg2, 1.0, np.where(pg2 > pg1, inr, 0.5 + 0.5 * inr))
    g_lp = (-A * w) * ratio / M
    onehot = np.zeros_like(x)
    onehot[np.arange(M), a] = 1.0
    dlogits = g_lp[:, None] * (onehot - p) + (ent_coef / M) * p * (lp + H[:, None])
    if clip_vloss:
        inv = (np.abs(v - old_v) <= clip_coef).astype(np.float64)
        gv = np.where(u > c, 2 * (v - ret), np.where(c > u, 2 * (vc - ret) * inv, (v - ret) + (vc - ret) * inv))
    else:
        gv = 2 * (v - ret)
    dvalue = vf_coef * 0.5 * gv / M
    return scalars, dlogits, dvalue
