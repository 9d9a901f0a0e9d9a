"""Host (CPU) path with the reference's exact torch semantics.

Selected ONLY when the user asks for a CPU device (``--no-cuda``: BASELINE config A "CartPole on CPU,
plumbing", and the world_size-2 ``gloo`` tests of the data-parallel logic).  It is not a fallback: a
CUDA device always runs the HIP kernels and raises if ``libmi355ppo.so`` is missing.  Each function
follows the reference lines it cites (cleanrl/ppo.py).
"""
from __future__ import annotations

import torch


def gae(rewards, dones, values, next_done, next_value, gamma, gae_lambda):
    """ppo.py:218-231."""
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
        advantages[t] = lastgaelam = delta + gamma * gae_lambda * nextnonterminal * lastgaelam
    return advantages, advantages + values


def ppo_loss(newlogprob, entropy, newvalue, mb_logprobs, mb_advantages, mb_returns, mb_values, clip_coef, ent_coef,
             vf_coef, norm_adv, clip_vloss):
    """ppo.py:251-285 -> (loss, scalars7 in ops.LOSS_SCALAR_NAMES order)."""
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
    scalars = torch.stack([loss.detach(), pg_loss.detach(), v_loss.detach(), entropy_loss.detach(), old_approx_kl,
                           approx_kl, clipfrac])
    return loss, scalars
