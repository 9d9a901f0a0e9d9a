"""Tensor-level operators over ``libmi355ppo.so`` -- the Python host side of the C ABI.

Each function mirrors one seam of the reference's inline PPO code (cited per function) and takes
CUDA (=HIP) ``torch`` tensors only: torch is plumbing here (device memory + the current stream), the
computation is the HIP kernels.  Passing a CPU tensor raises; there is no CPU fallback in this module.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib

__all__ = [
    "gae", "categorical_sample", "categorical_logprob_entropy", "normal_sample", "normal_logprob_entropy",
    "ppo_loss_categorical", "ppo_loss_normal", "obs_u8_to_f32", "obs_nchw_to_nhwc_u8", "clip_adam_", "PPOLossCategorical", "PPOLossNormal",
    "CategoricalLogProbEntropy", "NormalLogProbEntropy", "LOSS_SCALAR_NAMES",
]

LOSS_SCALAR_NAMES = ("loss", "pg_loss", "v_loss", "entropy", "old_approx_kl", "approx_kl", "clipfrac")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(dev: torch.device):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _NoSwitch:
    """Context manager that does nothing (the tensor's device is already the current one)."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(dev: torch.device):
    """``with _on(dev):`` == ``with _on(dev):`` without the get/set-device round trips when ``dev`` is
    already current -- the normal case (one process per GPU); the wrappers sit on the rollout's per-step critical path,
    which is host-bound."""
    return _NO_SWITCH if dev.index is None or torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def _chk(t: torch.Tensor, dtype, name: str, shape=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA/HIP tensor (libmi355ppo has no CPU path), got "
                        f"{type(t).__name__}{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


_workspaces: dict = {}
_retired: list = []


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    """Per-(device, stream) scratch; calls ordered on one stream may share it (see mi355ppo.h)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _retired.append(ws)      # a captured hipGraph may have baked this pointer in: an outgrown workspace is kept, never freed
        ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws


# ------------------------------------------------------------------------------------------- K1
def gae(rewards, dones, values, next_done, next_value, gamma: float, gae_lambda: float, advantages=None, returns=None,
        variant: int = 0):
    """Fused GAE (reference: ppo_atari_multigpu.py:290-301).  Returns ``(advantages, returns)`` (T,N)."""
    lib = _lib.load()
    T, N = rewards.shape
    _chk(rewards, torch.float32, "rewards", (T, N))
    _chk(dones, torch.float32, "dones", (T, N))
    _chk(values, torch.float32, "values", (T, N))
    next_done = _chk(next_done.reshape(-1), torch.float32, "next_done", (N,))
    next_value = _chk(next_value.reshape(-1), torch.float32, "next_value", (N,))
    if advantages is None:
        advantages = torch.empty_like(rewards)
    if returns is None:
        returns = torch.empty_like(rewards)
    _chk(advantages, torch.float32, "advantages", (T, N))
    _chk(returns, torch.float32, "returns", (T, N))
    with _on(rewards.device):
        st = lib.mi355ppo_gae_f32_variant(_ptr(rewards), _ptr(dones), _ptr(values), _ptr(next_done), _ptr(next_value),
                                          _ptr(advantages), _ptr(returns), T, N, float(gamma), float(gae_lambda),
                                          int(variant), _stream(rewards.device))
    _lib.check(st, "mi355ppo_gae_f32")
    return advantages, returns


# ------------------------------------------------------------------------------------------- K2
def categorical_sample(logits, noise_exp1=None, seed: int = 0, offset: int = 0, action_f32_out=None,
                       logprob_out=None, want_entropy: bool = True, want_i64: bool = True, offset_base=None):
    """``Categorical(logits=logits)``: sample, log_prob, entropy (ppo_atari_multigpu.py:156-159).

    ``noise_exp1`` (B,A) Exponential(1) draws reproduces torch's multinomial draw for that noise
    (parity mode); otherwise a Philox stream keyed by ``(seed, offset)`` is used; ``offset_base`` (1-element int64 device
    tensor) is added to ``offset`` on the device, so a captured launch can be replayed at a new stream position.
    Returns ``(action_i64 | None, action_f32 | None, logprob, entropy | None)``.
    """
    lib = _lib.load()
    B, A = logits.shape
    _chk(logits, torch.float32, "logits", (B, A))
    if noise_exp1 is not None:
        _chk(noise_exp1, torch.float32, "noise_exp1", (B, A))
    dev = logits.device
    a64 = torch.empty(B, dtype=torch.int64, device=dev) if want_i64 else None
    af = action_f32_out
    if af is not None:
        _chk(af, torch.float32, "action_f32_out", (B,))
    elif not want_i64:
        af = torch.empty(B, dtype=torch.float32, device=dev)
    lp = logprob_out if logprob_out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    _chk(lp, torch.float32, "logprob_out", (B,))
    ent = torch.empty(B, dtype=torch.float32, device=dev) if want_entropy else None
    if offset_base is not None:
        _chk(offset_base, torch.int64, "offset_base", (1,))
    with _on(dev):
        st = lib.mi355ppo_categorical_sample_ctr_f32(_ptr(logits), _ptr(noise_exp1), int(seed) & (2**64 - 1),
                                                     int(offset) & (2**64 - 1), _ptr(offset_base), _ptr(a64), _ptr(af), _ptr(lp),
                                                     _ptr(ent), B, A, _stream(dev))
    _lib.check(st, "mi355ppo_categorical_sample_f32")
    return a64, af, lp, ent


def categorical_logprob_entropy(logits, action):
    """log_prob / entropy of given actions (int64 or the reference's f32 storage)."""
    lib = _lib.load()
    B, A = logits.shape
    _chk(logits, torch.float32, "logits", (B, A))
    dev = logits.device
    if action.dtype == torch.int64:
        a64, af = _chk(action, torch.int64, "action", (B,)), None
    else:
        a64, af = None, _chk(action, torch.float32, "action", (B,))
    lp = torch.empty(B, dtype=torch.float32, device=dev)
    ent = torch.empty(B, dtype=torch.float32, device=dev)
    with _on(dev):
        st = lib.mi355ppo_categorical_logprob_entropy_f32(_ptr(logits), _ptr(a64), _ptr(af), _ptr(lp), _ptr(ent), B, A,
                                                          _stream(dev))
    _lib.check(st, "mi355ppo_categorical_logprob_entropy_f32")
    return lp, ent


def normal_sample(mean, logstd, noise=None, seed: int = 0, offset: int = 0, action_out=None, logprob_out=None):
    """``Normal(mean, exp(logstd))`` sample + summed log_prob/entropy (ppo_continuous_action.py:134-141)."""
    lib = _lib.load()
    B, D = mean.shape
    _chk(mean, torch.float32, "mean", (B, D))
    logstd = _chk(logstd.reshape(-1), torch.float32, "logstd", (D,))
    if noise is not None:
        _chk(noise, torch.float32, "noise", (B, D))
    dev = mean.device
    act = action_out if action_out is not None else torch.empty_like(mean)
    _chk(act, torch.float32, "action_out", (B, D))
    lp = logprob_out if logprob_out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    ent = torch.empty(B, dtype=torch.float32, device=dev)
    with _on(dev):
        st = lib.mi355ppo_normal_sample_f32(_ptr(mean), _ptr(logstd), _ptr(noise), int(seed) & (2**64 - 1),
                                            int(offset) & (2**64 - 1), _ptr(act), _ptr(lp), _ptr(ent), B, D,
                                            _stream(dev))
    _lib.check(st, "mi355ppo_normal_sample_f32")
    return act, lp, ent


def normal_logprob_entropy(mean, logstd, action):
    lib = _lib.load()
    B, D = mean.shape
    _chk(mean, torch.float32, "mean", (B, D))
    logstd = _chk(logstd.reshape(-1), torch.float32, "logstd", (D,))
    _chk(action, torch.float32, "action", (B, D))
    dev = mean.device
    lp = torch.empty(B, dtype=torch.float32, device=dev)
    ent = torch.empty(B, dtype=torch.float32, device=dev)
    with _on(dev):
        st = lib.mi355ppo_normal_logprob_entropy_f32(_ptr(mean), _ptr(logstd), _ptr(action), _ptr(lp), _ptr(ent), B, D,
                                                     _stream(dev))
    _lib.check(st, "mi355ppo_normal_logprob_entropy_f32")
    return lp, ent


class CategoricalLogProbEntropy(torch.autograd.Function):
    """Differentiable ``(log_prob, entropy)`` of given actions: forward and backward are HIP kernels."""

    @staticmethod
    def forward(ctx, logits, action):
        logits_c = logits.detach().contiguous()
        action = action.contiguous()
        lp, ent = categorical_logprob_entropy(logits_c, action)
        ctx.save_for_backward(logits_c, action)
        return lp, ent

    @staticmethod
    def backward(ctx, g_lp, g_ent):
        logits, action = ctx.saved_tensors
        lib = _lib.load()
        B, A = logits.shape
        dlogits = torch.empty_like(logits)
        a64, af = (action, None) if action.dtype == torch.int64 else (None, action)
        g_lp = None if g_lp is None else _chk(g_lp.contiguous(), torch.float32, "g_logprob", (B,))
        g_ent = None if g_ent is None else _chk(g_ent.contiguous(), torch.float32, "g_entropy", (B,))
        with _on(logits.device):
            st = lib.mi355ppo_categorical_logprob_entropy_bwd_f32(_ptr(logits), _ptr(a64), _ptr(af), _ptr(g_lp),
                                                                  _ptr(g_ent), _ptr(dlogits), B, A,
                                                                  _stream(logits.device))
        _lib.check(st, "mi355ppo_categorical_logprob_entropy_bwd_f32")
        return dlogits, None


class NormalLogProbEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, logstd, action):
        mean_c, action = mean.detach().contiguous(), action.contiguous()
        ls = logstd.detach().reshape(-1).contiguous()
        lp, ent = normal_logprob_entropy(mean_c, ls, action)
        ctx.save_for_backward(mean_c, ls, action)
        ctx.logstd_shape = logstd.shape
        return lp, ent

    @staticmethod
    def backward(ctx, g_lp, g_ent):
        mean, ls, action = ctx.saved_tensors
        lib = _lib.load()
        B, D = mean.shape
        dmean, drows = torch.empty_like(mean), torch.empty_like(mean)
        g_lp = None if g_lp is None else _chk(g_lp.contiguous(), torch.float32, "g_logprob", (B,))
        g_ent = None if g_ent is None else _chk(g_ent.contiguous(), torch.float32, "g_entropy", (B,))
        with _on(mean.device):
            st = lib.mi355ppo_normal_logprob_entropy_bwd_f32(_ptr(mean), _ptr(ls), _ptr(action), _ptr(g_lp), _ptr(g_ent),
                                                             _ptr(dmean), _ptr(drows), B, D, _stream(mean.device))
        _lib.check(st, "mi355ppo_normal_logprob_entropy_bwd_f32")
        return dmean, drows.sum(0).reshape(ctx.logstd_shape), None


# ------------------------------------------------------------------------------------------- K3
def _flat_batch(b_logprobs, b_advantages, b_returns, b_values):
    Bf = b_logprobs.numel()
    return (Bf, _chk(b_logprobs.reshape(-1), torch.float32, "b_logprobs"),
            _chk(b_advantages.reshape(-1), torch.float32, "b_advantages", (Bf,)),
            _chk(b_returns.reshape(-1), torch.float32, "b_returns", (Bf,)),
            _chk(b_values.reshape(-1), torch.float32, "b_values", (Bf,)))


class LossSlots:
    """Workspace slots for the deferred scalar fold of K3: minibatch k of an update runs ``ppo_loss_categorical(..., slot=(slots,
    k))`` (no fold launch) and ``slots.fold(n, out)`` turns the first n slots into rows of the (n, 7) scalar table in one
    launch."""

    def __init__(self, n: int, device: torch.device):
        lib = _lib.load()
        self.n, self.device = int(n), device
        self.stride = (lib.mi355ppo_loss_workspace_bytes(1, 0) + 255) // 256 * 256
        self.buf = torch.empty(self.n * self.stride, dtype=torch.uint8, device=device)

    def ptr(self, k: int):
        if not 0 <= k < self.n:
            raise IndexError(f"loss slot {k} out of range 0..{self.n - 1}")
        return ctypes.c_void_p(self.buf.data_ptr() + k * self.stride)

    def fold(self, n: int, out: torch.Tensor, first: int = 0) -> torch.Tensor:
        """rows first..first+n-1 of ``out`` (>= first+n, 7) <- slots first..first+n-1."""
        lib = _lib.load()
        _chk(out, torch.float32, "out")
        if out.dim() != 2 or out.shape[1] != 7 or out.shape[0] < first + n or first + n > self.n:
            raise ValueError(f"fold: out {tuple(out.shape)} / slots {self.n} cannot hold rows {first}..{first + n - 1}")
        with _on(self.device):
            st = lib.mi355ppo_loss_scalars_f32(self.ptr(first), self.stride, int(n), ctypes.c_void_p(out.data_ptr() + 28 * first),
                                               _stream(self.device))
        _lib.check(st, "mi355ppo_loss_scalars_f32")
        return out


def adv_stats(b_advantages, inds, minibatch_size: int, out=None):
    """``(mean, unbiased std + 1e-8)`` of ``b_advantages[inds[j*M:(j+1)*M]]`` for every minibatch j of one epoch's permutation,
    in one launch (ppo_atari_multigpu.py:337-338).  Returns a ``(ceil(len/M), 2)`` f32 tensor whose rows are the
    ``adv_mean_den`` argument of the loss entry points."""
    lib = _lib.load()
    flat = _chk(b_advantages.reshape(-1), torch.float32, "b_advantages")
    total = flat.numel() if inds is None else inds.numel()
    if inds is not None:
        _chk(inds, torch.int64, "inds", (total,))
    nseg = (total + minibatch_size - 1) // minibatch_size
    out = out if out is not None else torch.empty(nseg, 2, dtype=torch.float32, device=flat.device)
    _chk(out, torch.float32, "out", (nseg, 2))
    ws = _workspace(flat.device, lib.mi355ppo_adv_stats_workspace_bytes(total, int(minibatch_size)))
    with _on(flat.device):
        st = lib.mi355ppo_adv_stats_f32(_ptr(flat), _ptr(inds), total, int(minibatch_size), _ptr(out), _ptr(ws), ws.numel(),
                                        _stream(flat.device))
    _lib.check(st, "mi355ppo_adv_stats_f32")
    return out


PACK_FLOATS = 8       # floats per packed behaviour row: {action, old log-prob, advantage, return, old value, 0, 0, 0}


def batch_pack(b_actions, b_logprobs, b_advantages, b_returns, b_values, out=None):
    """The five per-row behaviour arrays of the flat batch -> ``(B, 8)`` packed rows (one 32-byte gather per minibatch row in
    K3 instead of five 4-byte gathers; ppo_atari_multigpu.py:320-352).  Once per iteration, after GAE."""
    lib = _lib.load()
    Bf, b_logprobs, b_advantages, b_returns, b_values = _flat_batch(b_logprobs, b_advantages, b_returns, b_values)
    b_actions = _chk(b_actions.reshape(-1), torch.float32, "b_actions (f32 storage, as the reference)", (Bf,))
    dev = b_logprobs.device
    out = out if out is not None else torch.empty((Bf, PACK_FLOATS), dtype=torch.float32, device=dev)
    _chk(out, torch.float32, "pack", (Bf, PACK_FLOATS))
    with _on(dev):
        st = lib.mi355ppo_batch_pack_f32(_ptr(b_actions), _ptr(b_logprobs), _ptr(b_advantages), _ptr(b_returns), _ptr(b_values),
                                         _ptr(out), Bf, _stream(dev))
    _lib.check(st, "mi355ppo_batch_pack_f32")
    return out


def adv_stats_packed(pack, inds, minibatch_size: int, out=None):
    """:func:`adv_stats` reading the advantages out of the packed rows of :func:`batch_pack`."""
    lib = _lib.load()
    _chk(pack, torch.float32, "pack")
    if pack.dim() != 2 or pack.shape[1] != PACK_FLOATS:
        raise ValueError(f"pack: expected (B, {PACK_FLOATS}), got {tuple(pack.shape)}")
    total = pack.shape[0] if inds is None else inds.numel()
    if inds is not None:
        _chk(inds, torch.int64, "inds", (total,))
    nseg = (total + minibatch_size - 1) // minibatch_size
    out = out if out is not None else torch.empty(nseg, 2, dtype=torch.float32, device=pack.device)
    _chk(out, torch.float32, "out", (nseg, 2))
    ws = _workspace(pack.device, lib.mi355ppo_adv_stats_workspace_bytes(total, int(minibatch_size)))
    with _on(pack.device):
        st = lib.mi355ppo_adv_stats_packed_f32(_ptr(pack), _ptr(inds), total, int(minibatch_size), _ptr(out), _ptr(ws), ws.numel(),
                                               _stream(pack.device))
    _lib.check(st, "mi355ppo_adv_stats_packed_f32")
    return out


def ppo_loss_categorical_packed(new_logits, new_value, mb_inds, pack, clip_coef: float, ent_coef: float, vf_coef: float,
                                norm_adv: bool = True, clip_vloss: bool = True, scalars_out=None, dlogits_out=None,
                                dvalue_out=None, adv_mean_den=None, slot=None):
    """:func:`ppo_loss_categorical` on the packed rows of :func:`batch_pack` (bit-identical results).  ``adv_mean_den`` (a row
    of :func:`adv_stats_packed`) is required when ``norm_adv`` is set."""
    lib = _lib.load()
    M, A = new_logits.shape
    _chk(new_logits, torch.float32, "new_logits", (M, A))
    new_value = _chk(new_value.reshape(-1), torch.float32, "new_value", (M,))
    dev = new_logits.device
    if mb_inds is not None:
        _chk(mb_inds, torch.int64, "mb_inds", (M,))
    _chk(pack, torch.float32, "pack")
    if pack.dim() != 2 or pack.shape[1] != PACK_FLOATS:
        raise ValueError(f"pack: expected (B, {PACK_FLOATS}), got {tuple(pack.shape)}")
    if norm_adv and adv_mean_den is None:
        adv_mean_den = adv_stats_packed(pack, mb_inds, M)[0] if mb_inds is not None else adv_stats_packed(pack[:M], None, M)[0]
    if adv_mean_den is not None:
        _chk(adv_mean_den, torch.float32, "adv_mean_den", (2,))
    dlogits = dlogits_out if dlogits_out is not None else torch.empty_like(new_logits)
    dvalue = dvalue_out if dvalue_out is not None else torch.empty(M, dtype=torch.float32, device=dev)
    if slot is not None:
        slots, k = slot
        ws_ptr, ws_bytes, scalars = slots.ptr(k), slots.stride, None
    else:
        scalars = scalars_out if scalars_out is not None else torch.empty(7, dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.mi355ppo_loss_workspace_bytes(M, 0))
        ws_ptr, ws_bytes = _ptr(ws), ws.numel()
    with _on(dev):
        st = lib.mi355ppo_loss_categorical_packed_fwd_bwd_f32(
            _ptr(new_logits), _ptr(new_value), _ptr(mb_inds), _ptr(pack), M, A, float(clip_coef), float(ent_coef), float(vf_coef),
            int(bool(norm_adv)), int(bool(clip_vloss)), _ptr(adv_mean_den), _ptr(scalars), _ptr(dlogits), _ptr(dvalue), ws_ptr,
            ws_bytes, _stream(dev))
    _lib.check(st, "mi355ppo_loss_categorical_packed_fwd_bwd_f32")
    return scalars, dlogits, dvalue


def ppo_loss_categorical(new_logits, new_value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values,
                         clip_coef: float, ent_coef: float, vf_coef: float, norm_adv: bool = True,
                         clip_vloss: bool = True, scalars_out=None, dlogits_out=None, dvalue_out=None, adv_mean_den=None,
                         slot=None):
    """Fused minibatch loss fwd+bwd (ppo_atari_multigpu.py:320-355 + autograd backward).

    Returns ``(scalars7, dlogits, dvalue)``; ``scalars7`` = LOSS_SCALAR_NAMES order, on device.  ``adv_mean_den``: the
    minibatch's row of :func:`adv_stats` (skips the statistics launch).  ``slot=(LossSlots, k)``: defer the scalar fold
    (``scalars7`` is then None; ``LossSlots.fold`` produces it later).
    """
    lib = _lib.load()
    M, A = new_logits.shape
    _chk(new_logits, torch.float32, "new_logits", (M, A))
    new_value = _chk(new_value.reshape(-1), torch.float32, "new_value", (M,))
    dev = new_logits.device
    if mb_inds is not None:
        _chk(mb_inds, torch.int64, "mb_inds", (M,))
    Bf, b_logprobs, b_advantages, b_returns, b_values = _flat_batch(b_logprobs, b_advantages, b_returns, b_values)
    b_actions = _chk(b_actions.reshape(-1), torch.float32, "b_actions (f32 storage, as the reference)", (Bf,))
    if adv_mean_den is not None:
        _chk(adv_mean_den, torch.float32, "adv_mean_den", (2,))
    dlogits = dlogits_out if dlogits_out is not None else torch.empty_like(new_logits)
    dvalue = dvalue_out if dvalue_out is not None else torch.empty(M, dtype=torch.float32, device=dev)
    if slot is not None:
        slots, k = slot
        ws_ptr, ws_bytes, scalars = slots.ptr(k), slots.stride, None
    else:
        scalars = scalars_out if scalars_out is not None else torch.empty(7, dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.mi355ppo_loss_workspace_bytes(M, 0))
        ws_ptr, ws_bytes = _ptr(ws), ws.numel()
    with _on(dev):
        st = lib.mi355ppo_loss_categorical_fwd_bwd_f32(
            _ptr(new_logits), _ptr(new_value), _ptr(mb_inds), _ptr(b_actions), _ptr(b_logprobs), _ptr(b_advantages),
            _ptr(b_returns), _ptr(b_values), M, A, float(clip_coef), float(ent_coef), float(vf_coef), int(bool(norm_adv)),
            int(bool(clip_vloss)), _ptr(adv_mean_den), _ptr(scalars), _ptr(dlogits), _ptr(dvalue), ws_ptr, ws_bytes,
            _stream(dev))
    _lib.check(st, "mi355ppo_loss_categorical_fwd_bwd_f32")
    return scalars, dlogits, dvalue


def ppo_loss_normal(new_mean, logstd, new_value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values,
                    clip_coef: float, ent_coef: float, vf_coef: float, norm_adv: bool = True, clip_vloss: bool = True,
                    scalars_out=None, adv_mean_den=None):
    """Continuous-action loss fwd+bwd (ppo_continuous_action.py:265-300).
    Returns ``(scalars7, dmean, dlogstd, dvalue)``."""
    lib = _lib.load()
    M, D = new_mean.shape
    _chk(new_mean, torch.float32, "new_mean", (M, D))
    logstd_flat = _chk(logstd.reshape(-1), torch.float32, "logstd", (D,))
    new_value = _chk(new_value.reshape(-1), torch.float32, "new_value", (M,))
    dev = new_mean.device
    if mb_inds is not None:
        _chk(mb_inds, torch.int64, "mb_inds", (M,))
    Bf, b_logprobs, b_advantages, b_returns, b_values = _flat_batch(b_logprobs, b_advantages, b_returns, b_values)
    b_actions = _chk(b_actions.reshape(Bf, D), torch.float32, "b_actions", (Bf, D))
    if adv_mean_den is not None:
        _chk(adv_mean_den, torch.float32, "adv_mean_den", (2,))
    scalars = scalars_out if scalars_out is not None else torch.empty(7, dtype=torch.float32, device=dev)
    dmean = torch.empty_like(new_mean)
    dlogstd = torch.empty(D, dtype=torch.float32, device=dev)
    dvalue = torch.empty(M, dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.mi355ppo_loss_workspace_bytes(M, D))
    with _on(dev):
        st = lib.mi355ppo_loss_normal_fwd_bwd_f32(
            _ptr(new_mean), _ptr(logstd_flat), _ptr(new_value), _ptr(mb_inds), _ptr(b_actions), _ptr(b_logprobs),
            _ptr(b_advantages), _ptr(b_returns), _ptr(b_values), M, D, float(clip_coef), float(ent_coef), float(vf_coef),
            int(bool(norm_adv)), int(bool(clip_vloss)), _ptr(adv_mean_den), _ptr(scalars), _ptr(dmean), _ptr(dlogstd),
            _ptr(dvalue), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(st, "mi355ppo_loss_normal_fwd_bwd_f32")
    return scalars, dmean, dlogstd, dvalue


class PPOLossCategorical(torch.autograd.Function):
    """``loss = PPOLossCategorical.apply(logits, value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns,
    b_values, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss)`` -> (loss, scalars7).  ``loss.backward()``
    feeds the precomputed dlogits / dvalue into the network's autograd graph."""

    @staticmethod
    def forward(ctx, logits, value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, clip_coef,
                ent_coef, vf_coef, norm_adv, clip_vloss):
        scalars, dlogits, dvalue = ppo_loss_categorical(
            logits.detach().contiguous(), value.detach().contiguous(), mb_inds, b_actions, b_logprobs, b_advantages,
            b_returns, b_values, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss)
        ctx.save_for_backward(dlogits, dvalue)
        ctx.value_shape = value.shape
        ctx.mark_non_differentiable(scalars)
        return scalars[0].clone(), scalars

    @staticmethod
    def backward(ctx, grad_loss, _grad_scalars):
        dlogits, dvalue = ctx.saved_tensors
        return (grad_loss * dlogits, (grad_loss * dvalue).reshape(ctx.value_shape)) + (None,) * 11


class PPOLossNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, logstd, value, mb_inds, b_actions, b_logprobs, b_advantages, b_returns, b_values, clip_coef,
                ent_coef, vf_coef, norm_adv, clip_vloss):
        scalars, dmean, dlogstd, dvalue = ppo_loss_normal(
            mean.detach().contiguous(), logstd.detach().contiguous(), value.detach().contiguous(), mb_inds, b_actions,
            b_logprobs, b_advantages, b_returns, b_values, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss)
        ctx.save_for_backward(dmean, dlogstd, dvalue)
        ctx.value_shape, ctx.logstd_shape = value.shape, logstd.shape
        ctx.mark_non_differentiable(scalars)
        return scalars[0].clone(), scalars

    @staticmethod
    def backward(ctx, grad_loss, _grad_scalars):
        dmean, dlogstd, dvalue = ctx.saved_tensors
        return (grad_loss * dmean, (grad_loss * dlogstd).reshape(ctx.logstd_shape),
                (grad_loss * dvalue).reshape(ctx.value_shape)) + (None,) * 11


# ------------------------------------------------------------------------------------------- K5
def obs_u8_to_f32(src_u8, inds=None, out=None, scale_255: bool = True):
    """Gather rows of a uint8 observation buffer and convert to f32 (``b_obs[mb_inds]`` then ``x / 255.0``,
    ppo_atari_multigpu.py:320,154).  ``src_u8``: (R, ...) uint8; ``inds``: (rows,) int64 or None."""
    lib = _lib.load()
    _chk(src_u8, torch.uint8, "src_u8")
    dev = src_u8.device
    row_shape = tuple(src_u8.shape[1:])
    row_bytes = 1
    for s in row_shape:
        row_bytes *= s
    if inds is not None:
        _chk(inds, torch.int64, "inds")
        rows = inds.numel()
    else:
        rows = src_u8.shape[0]
    if out is None:
        out = torch.empty((rows,) + row_shape, dtype=torch.float32, device=dev)
    _chk(out, torch.float32, "out", (rows,) + row_shape)
    with _on(dev):
        st = lib.mi355ppo_obs_u8_to_f32(_ptr(src_u8), _ptr(inds), _ptr(out), rows, row_bytes, int(bool(scale_255)),
                                        _stream(dev))
    _lib.check(st, "mi355ppo_obs_u8_to_f32")
    return out


def obs_nchw_to_nhwc_u8(src, out=None):
    """(rows, C, H, W) uint8 -> (rows, H, W, C) uint8, the rollout buffer's pixel-interleaved layout."""
    lib = _lib.load()
    _chk(src, torch.uint8, "src")
    rows, C, H, W = src.shape
    if out is None:
        out = torch.empty((rows, H, W, C), dtype=torch.uint8, device=src.device)
    _chk(out, torch.uint8, "out", (rows, H, W, C))
    with _on(src.device):
        st = lib.mi355ppo_obs_nchw_to_nhwc_u8(_ptr(src), _ptr(out), rows, C, H * W, _stream(src.device))
    _lib.check(st, "mi355ppo_obs_nchw_to_nhwc_u8")
    return out


def obs_shift_append_u8(prev_rows, newest, out):
    """FrameStack(4) delta store: ``out[r] = concat(prev_rows[r][..., 1:4], newest[r][..., None])`` on (rows, H, W, 4) uint8
    rows and (rows, H, W) uint8 planes -- the next observation of envs that were not reset."""
    lib = _lib.load()
    rows, H, W, C = prev_rows.shape
    assert C == 4, "the delta store is defined for 4-frame stacks"
    _chk(prev_rows, torch.uint8, "prev_rows", (rows, H, W, 4))
    _chk(newest, torch.uint8, "newest", (rows, H, W))
    _chk(out, torch.uint8, "out", (rows, H, W, 4))
    with _on(out.device):
        st = lib.mi355ppo_obs_shift_append_u8_c4(_ptr(prev_rows), _ptr(newest), _ptr(out), rows, H * W, _stream(out.device))
    _lib.check(st, "mi355ppo_obs_shift_append_u8_c4")
    return out


# ---------------------------------------------------------------------------------------- a8/a9
def clip_adam_(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, max_grad_norm: float, grad_scale: float = 1.0,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-5, total_norm_out=None):
    """In-place fused ``grads*grad_scale -> clip_grad_norm_ -> Adam.step`` on flat f32 buffers
    (ppo_atari_multigpu.py:368-377).  Zeroes ``grads`` for the next backward.  ``step`` is 1-based."""
    lib = _lib.load()
    n = params.numel()
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, torch.float32, nm, (n,))
    dev = params.device
    if total_norm_out is None:
        total_norm_out = torch.empty(1, dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.mi355ppo_clip_adam_workspace_bytes(n))
    with _on(dev):
        st = lib.mi355ppo_clip_adam_f32(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), n, float(grad_scale),
                                        float(max_grad_norm), float(lr), float(beta1), float(beta2), float(eps),
                                        int(step), _ptr(total_norm_out), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(st, "mi355ppo_clip_adam_f32")
    return total_norm_out


def adam_schedule(lr: float, step: int, beta1: float = 0.9, beta2: float = 0.999):
    """The two schedule-dependent constants of Adam step ``step`` (1-based) as the kernels consume them (host floats):
    ``(-(lr / (1 - beta1^step)), sqrt(1 - beta2^step))`` -- computed by the library, so that eager and captured steps agree bit
    for bit."""
    out = (ctypes.c_float * 2)()
    _lib.check(_lib.load().mi355ppo_adam_schedule_f32(float(lr), float(beta1), float(beta2), int(step), out), "mi355ppo_adam_schedule_f32")
    return float(out[0]), float(out[1])


def clip_adam_sched_(params, grads, exp_avg, exp_avg_sq, sched2, max_grad_norm: float, grad_scale: float = 1.0, beta1: float = 0.9,
                     beta2: float = 0.999, eps: float = 1e-5, total_norm_out=None):
    """``clip_adam_`` with the step's (step_size, bias_correction2_sqrt) read from the 2-float DEVICE tensor ``sched2``
    (``adam_schedule`` values copied there by the caller): capturable, replayable with the next step's schedule."""
    lib = _lib.load()
    n = params.numel()
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, torch.float32, nm, (n,))
    _chk(sched2, torch.float32, "sched2", (2,))
    dev = params.device
    if total_norm_out is None:
        total_norm_out = torch.empty(1, dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.mi355ppo_clip_adam_workspace_bytes(n))
    with _on(dev):
        st = lib.mi355ppo_clip_adam_sched_f32(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), n, float(grad_scale),
                                              float(max_grad_norm), float(beta1), float(beta2), float(eps), _ptr(sched2),
                                              _ptr(total_norm_out), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(st, "mi355ppo_clip_adam_sched_f32")
    return total_norm_out


# ------------------------------------------------------------------------------------------- K7: the fused MLP agents
MLP_MAX_OBS, MLP_MAX_OUT, MLP_HIDDEN = 512, 20, 64      # (round 6: the WIDE kernels of csrc/mlp.hip -- Humanoid's 376 / 17; up to 32 / 8: the narrow ones)


class MlpNetPtrs:
    """The six parameter tensors of one 64-64 tanh MLP (``nn.Sequential(Linear, Tanh, Linear, Tanh, Linear)``, ppo.py:100-117)
    as the C ABI wants them: a host array of six device pointers (weights in torch's (out, in) layout), and the same for their
    ``.grad`` views.  Built once per network: parameters and gradients of a flat-buffer agent never move."""

    def __init__(self, seq):
        lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
        assert len(lin) == 3 and lin[0].out_features == MLP_HIDDEN and lin[1].in_features == MLP_HIDDEN and \
            lin[1].out_features == MLP_HIDDEN and lin[2].in_features == MLP_HIDDEN, "the fused MLP kernels are the reference's 64-64 networks"
        self.obs_dim, self.n_out = lin[0].in_features, lin[2].out_features
        self.tensors = [t for m in lin for t in (m.weight, m.bias)]
        for t in self.tensors:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self.params = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in self.tensors])
        self._grad_ptrs = None

    def grads(self):
        g = [t.grad for t in self.tensors]
        assert all(x is not None and x.is_contiguous() for x in g), "fused MLP backward needs allocated .grad tensors (flat buffers)"
        key = tuple(x.data_ptr() for x in g)
        if self._grad_ptrs is None or self._grad_ptrs[0] != key:
            self._grad_ptrs = (key, (ctypes.c_void_p * 6)(*key))
        return self._grad_ptrs[1]

    def refresh(self):
        """Re-read the parameter pointers (after ``module.to(...)`` / a flat-buffer rebind)."""
        self.params = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in self.tensors])


def mlp_supported(obs_dim: int, n_out: int) -> bool:
    return 0 < obs_dim <= MLP_MAX_OBS and 0 < n_out <= MLP_MAX_OUT


def mlp_forward(obs, actor: MlpNetPtrs, critic: MlpNetPtrs, actor_out=None, value_out=None):
    """Both networks' forward in one launch -> ``(actor_out (B, n_out), value (B))``."""
    lib = _lib.load()
    B, O = obs.shape
    _chk(obs, torch.float32, "obs", (B, actor.obs_dim))
    dev = obs.device
    out = actor_out if actor_out is not None else torch.empty((B, actor.n_out), dtype=torch.float32, device=dev)
    val = value_out if value_out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    _chk(out, torch.float32, "actor_out", (B, actor.n_out))
    _chk(val, torch.float32, "value_out", (B,))
    with _on(dev):
        st = lib.mi355ppo_mlp_fwd_f32(_ptr(obs), B, O, actor.params, critic.params, actor.n_out, _ptr(out), _ptr(val), _stream(dev))
    _lib.check(st, "mi355ppo_mlp_fwd_f32")
    return out, val


def mlp_act_categorical(obs, actor: MlpNetPtrs, critic: MlpNetPtrs, noise_exp1=None, seed: int = 0, offset: int = 0, offset_base=None,
                        action_f32_out=None, logprob_out=None, value_out=None, want_i64: bool = True, want_entropy: bool = False,
                        want_logits: bool = False):
    """One rollout step of the Categorical MLP agent (ppo.py:205-210): both forwards + sample + log_prob (+ entropy) in one launch.
    -> ``(action_i64 | None, action_f32 | None, logprob, entropy | None, value, logits | None)``."""
    lib = _lib.load()
    B, O = obs.shape
    _chk(obs, torch.float32, "obs", (B, actor.obs_dim))
    dev, A = obs.device, actor.n_out
    if noise_exp1 is not None:
        _chk(noise_exp1, torch.float32, "noise_exp1", (B, A))
    a64 = torch.empty(B, dtype=torch.int64, device=dev) if want_i64 else None
    af = action_f32_out if action_f32_out is not None else (None if want_i64 else torch.empty(B, dtype=torch.float32, device=dev))
    lp = logprob_out if logprob_out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    val = value_out if value_out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    for t, nm in ((af, "action_f32_out"), (lp, "logprob_out"), (val, "value_out")):
        if t is not None:
            _chk(t, torch.float32, nm, (B,))
    ent = torch.empty(B, dtype=torch.float32, device=dev) if want_entropy else None
    logits = torch.empty((B, A), dtype=torch.float32, device=dev) if want_logits else None
    if offset_base is not None:
        _chk(offset_base, torch.int64, "offset_base", (1,))
    with _on(dev):
        st = lib.mi355ppo_mlp_act_categorical_f32(_ptr(obs), B, O, actor.params, critic.params, A, _ptr(noise_exp1), int(seed) & (2**64 - 1),
                                                  int(offset) & (2**64 - 1), _ptr(offset_base), _ptr(a64), _ptr(af), _ptr(lp), _ptr(ent),
                                                  _ptr(val), _ptr(logits), _stream(dev))
    _lib.check(st, "mi355ppo_mlp_act_categorical_f32")
    return a64, af, lp, ent, val, logits


def mlp_act_normal(obs, actor: MlpNetPtrs, critic: MlpNetPtrs, logstd, noise=None, seed: int = 0, offset: int = 0, offset_base=None,
                   action_out=None, logprob_out=None, value_out=None, want_entropy: bool = False, want_mean: bool = False):
    """One rollout step of the continuous-action agent (ppo_continuous_action.py:221-226) in one launch.
    -> ``(action (B, D), logprob, entropy | None, value, mean | None)``."""
    lib = _lib.load()
    B, O = obs.shape
    _chk(obs, torch.float32, "obs", (B, actor.obs_dim))
    dev, D = obs.device, actor.n_out
    logstd = _chk(logstd.reshape(-1), torch.float32, "logstd", (D,))
    if noise is not None:
        _chk(noise, torch.float32, "noise", (B, D))
    act = action_out if action_out is not None else torch.empty((B, D), dtype=torch.float32, device=dev)
    _chk(act, torch.float32, "action_out", (B, D))
    lp = logprob_out if logprob_out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    val = value_out if value_out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    _chk(lp, torch.float32, "logprob_out", (B,))
    _chk(val, torch.float32, "value_out", (B,))
    ent = torch.empty(B, dtype=torch.float32, device=dev) if want_entropy else None
    mean = torch.empty((B, D), dtype=torch.float32, device=dev) if want_mean else None
    if offset_base is not None:
        _chk(offset_base, torch.int64, "offset_base", (1,))
    with _on(dev):
        st = lib.mi355ppo_mlp_act_normal_f32(_ptr(obs), B, O, actor.params, critic.params, _ptr(logstd), D, _ptr(noise),
                                             int(seed) & (2**64 - 1), int(offset) & (2**64 - 1), _ptr(offset_base), _ptr(act), _ptr(lp),
                                             _ptr(ent), _ptr(val), _ptr(mean), _stream(dev))
    _lib.check(st, "mi355ppo_mlp_act_normal_f32")
    return act, lp, ent, val, mean


def mlp_ppo_fwd_bwd(b_obs, mb_inds, actor: MlpNetPtrs, critic: MlpNetPtrs, b_actions, b_logprobs, b_advantages, b_returns, b_values,
                    clip_coef, ent_coef, vf_coef, norm_adv=True, clip_vloss=True, adv_mean_den=None, scalars_out=None, logstd=None,
                    logstd_grad=None, mean_shift=None, rows_per_block: int = 0):
    """One minibatch of the update of an MLP agent in two launches (ppo.py:250-287 / ppo_continuous_action.py:265-302 up to and
    including ``loss.backward()``): gather, both forwards, distribution, PPO loss, both backward passes.  Gradients are ADDED to
    the networks' ``.grad`` tensors (and ``logstd_grad``); -> the seven scalars of K3.  ``logstd`` given = Normal head."""
    lib = _lib.load()
    Bf, O = b_obs.shape
    _chk(b_obs, torch.float32, "b_obs", (Bf, actor.obs_dim))
    dev, nout = b_obs.device, actor.n_out
    if mb_inds is not None:
        _chk(mb_inds, torch.int64, "mb_inds")
        M = mb_inds.numel()
    else:
        M = Bf
    _, lpv, advv, retv, valv = _flat_batch(b_logprobs, b_advantages, b_returns, b_values)
    normal = logstd is not None
    _chk(b_actions, torch.float32, "b_actions", (Bf, nout) if normal else (Bf,))
    if norm_adv:
        if adv_mean_den is None:
            adv_mean_den = adv_stats(advv, mb_inds, M)[0]
        _chk(adv_mean_den, torch.float32, "adv_mean_den", (2,))
    sc = scalars_out if scalars_out is not None else torch.empty(7, dtype=torch.float32, device=dev)
    _chk(sc, torch.float32, "scalars_out", (7,))
    nbytes = lib.mi355ppo_mlp_ppo_workspace_bytes(M, O, nout, int(rows_per_block))
    if nbytes == 0:
        raise ValueError(f"mlp_ppo_fwd_bwd: unsupported shape (obs_dim={O} <= {MLP_MAX_OBS}, n_out={nout} <= {MLP_MAX_OUT})")
    ws = _workspace(dev, nbytes)
    with _on(dev):
        if normal:
            ls = _chk(logstd.reshape(-1), torch.float32, "logstd", (nout,))
            lg = _chk(logstd_grad.reshape(-1), torch.float32, "logstd_grad", (nout,))
            if mean_shift is not None:
                _chk(mean_shift, torch.float32, "mean_shift", (M, nout))
            st = lib.mi355ppo_mlp_ppo_normal_fwd_bwd_f32(
                _ptr(b_obs), _ptr(mb_inds), M, O, actor.params, critic.params, _ptr(ls), nout, _ptr(mean_shift), _ptr(b_actions), _ptr(lpv),
                _ptr(advv), _ptr(retv), _ptr(valv), float(clip_coef), float(ent_coef), float(vf_coef), int(bool(norm_adv)),
                int(bool(clip_vloss)), _ptr(adv_mean_den), actor.grads(), critic.grads(), _ptr(lg), _ptr(sc), int(rows_per_block), _ptr(ws),
                ws.numel(), _stream(dev))
        else:
            st = lib.mi355ppo_mlp_ppo_categorical_fwd_bwd_f32(
                _ptr(b_obs), _ptr(mb_inds), M, O, actor.params, critic.params, nout, _ptr(b_actions), _ptr(lpv), _ptr(advv), _ptr(retv),
                _ptr(valv), float(clip_coef), float(ent_coef), float(vf_coef), int(bool(norm_adv)), int(bool(clip_vloss)),
                _ptr(adv_mean_den), actor.grads(), critic.grads(), _ptr(sc), int(rows_per_block), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(st, "mi355ppo_mlp_ppo_fwd_bwd_f32")
    return sc


def synth_continuous_step(state, reset_state, At, Bm, w, noise, k: int, steps, horizon: float, action, obs_out, reward, done, k_base=None):
    """One step of the device-resident continuous-control stand-in env (test / bench support, not the reference path)."""
    lib = _lib.load()
    N, O = state.shape
    D = action.shape[1]
    bank = noise.shape[0]
    for t, nm, shp in ((state, "state", (N, O)), (reset_state, "reset_state", (N, O)), (At, "At", (O, O)), (Bm, "Bm", (D, O)), (w, "w", (O,)),
                       (noise, "noise", (bank, N, O)), (steps, "steps", (N,)), (action, "action", (N, D)), (obs_out, "obs_out", (N, O)),
                       (reward, "reward", (N,)), (done, "done", (N,))):
        _chk(t, torch.float32, nm, shp)
    if k_base is not None:
        _chk(k_base, torch.int64, "k_base", (1,))
    dev = state.device
    with _on(dev):
        st = lib.mi355ppo_synth_continuous_step_f32(_ptr(state), _ptr(reset_state), _ptr(At), _ptr(Bm), _ptr(w), _ptr(noise), bank,
                                                    int(k) & (2**64 - 1), _ptr(k_base), _ptr(steps), float(horizon), _ptr(action),
                                                    _ptr(obs_out), _ptr(reward), _ptr(done), N, O, D, _stream(dev))
    _lib.check(st, "mi355ppo_synth_continuous_step_f32")
