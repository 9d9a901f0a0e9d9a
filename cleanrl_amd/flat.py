"""Persistent flat parameter / gradient / Adam-state buffers.

The reference packs ``torch.cat([p.grad.view(-1) ...])`` for every all-reduce and unpacks it with one
``copy_`` + divide per parameter (ppo_atari_multigpu.py:360-374), then runs per-tensor clip and Adam.
Here every ``nn.Parameter`` and its ``.grad`` are *views* into one contiguous f32 buffer each, laid out
in ``agent.parameters()`` order (the reference's cat order), so the collective and the fused
clip+Adam kernel operate on the buffers directly: no pack, no unpack.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class FlatParams:
    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters()]
        assert all(p.dtype == torch.float32 for p in params), "flat buffers are f32"
        dev = params[0].device
        self.numel = sum(p.numel() for p in params)
        self.segments = []
        self.params = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.params[off:off + n].copy_(p.reshape(-1))
                p.data = self.params[off:off + n].view(p.shape)
                p.grad = self.grads[off:off + n].view(p.shape)
                self.segments.append((off, n))
                off += n
        self._params_list = params
        self.step = 0

    def check_views(self) -> None:
        """Autograd accumulates in place into ``p.grad``; fail loudly if something replaced the views."""
        for p, (off, n) in zip(self._params_list, self.segments):
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * off:
                raise RuntimeError("a parameter's .grad no longer aliases the flat gradient buffer "
                                   "(do not call zero_grad(set_to_none=True) on a flat-buffer agent)")
