"""NatureCNN convolution stack on the libmi355ppo f32-MFMA kernels (``csrc/conv.hip``).

Host side of ``mi355ppo_cnn_*``: tensor-level wrappers, and ``NatureTrunk`` -- the autograd node that replaces
``Agent.network[0:6]`` (cleanrl/ppo_atari_multigpu.py:136-142: three Conv2d + ReLU) for uint8 channels-last rollout
rows.  Everything the reference runs as separate passes around the convolutions is fused into them:
``b_obs[mb_inds]`` and ``x / 255.0`` (:320,154) into conv1's loader, bias + ReLU into every forward epilogue, ReLU
backward into the data-gradient epilogues, bias gradients into the weight-gradient kernels.

Activations are (images, H, W, C) f32; parameters keep torch's (Cout, Cin, KH, KW) layout (they are views of the
flat parameter buffer) and are repacked into the kernels' matrix layouts on use (<= 147 KB each).
"""
from __future__ import annotations

import os

import weakref

import torch

from . import _lib
from .ops import _chk, _on, _ptr, _stream, _workspace

# layer -> (Cin, Cout, K, stride, Hin, Hout)
LAYERS = {1: (4, 32, 8, 4, 84, 20), 2: (32, 64, 4, 2, 20, 9), 3: (64, 64, 3, 1, 9, 7)}
MODE_FWD, MODE_DGRAD_S1, MODE_DGRAD_S2, MODE_DGRAD_S1_CLASSES, MODE_FWD_Q, MODE_DGRAD_S2_CLASSES = 0, 1, 2, 3, 4, 5
BT_CLASSES_NUMEL = 81 * 4096
BT2_CLASSES_NUMEL = 16 * 128 * 64   # layer-2 data gradient, one matrix per border class of the 10x10 class grid (mode 5)
VARIANT_DGRAD2_CLASSES = 6
QPACK_NUMEL = 33408 // 4            # mi355ppo_cnn_conv1q_pack_bytes() as f32 storage elements (kernel Q's integer-digit pack)
VARIANT_Q = 6                       # layer-1 forward on the integer matrix pipe (csrc/conv1q.hip); Bt = the mode-4 pack
_CONV_Z = os.environ.get("MI355PPO_CONV", "z") != "f"    # layers 2 / 3 forward + data gradients: kernel Z, or the f32-pipe kernel F
_MASK_BITS = True    # ReLU masks travel to the data gradients as bits (the A/B switch MI355PPO_MASK_BITS is gone since round 6: profiles/r03_mask_bits_ab.jsonl)
_FUSED_PACKS = True   # all weight packs of the NatureCNN agent in one launch (the A/B switch MI355PPO_FUSED_PACKS is gone since round 6)
BUF_LIMIT = (1 << 32) - 8192     # kernels Z / F address a tensor with 32-bit buffer offsets: larger tensors take kernel S (64-bit pointers)
# Operand split of kernels Z / V / W (round 5): "f16x2" = two f16 terms per f32 under per-tensor power-of-two scales, three matrix
# instructions per product (csrc/f16split.h); "bf16x3" = round 3's three bf16 terms, six instructions.  Read once per process.
_SPLIT = os.environ.get("MI355PPO_SPLIT", "f16x2")
if _SPLIT not in ("f16x2", "bf16x3"):
    raise ValueError(f"MI355PPO_SPLIT={_SPLIT!r}: expected f16x2 or bf16x3")
AMAX_WORDS = 256                 # MI355PPO_AMAX_WORDS: uint32 words of an amax record (16 slots, 64 bytes apart)
REC_A1, REC_A2, REC_A3, REC_DH, REC_DZ3, REC_DZ2, REC_DZ1, N_REC = 0, 1, 2, 3, 4, 5, 6, 8     # the records of one forward / backward pass of the trunk


def new_amax(n: int, device) -> torch.Tensor:
    """``n`` zeroed amax records (csrc/f16split.h): row r = the record of one tensor."""
    return torch.zeros((n, AMAX_WORDS), dtype=torch.int32, device=device)


def absmax(x: torch.Tensor, rec: torch.Tensor) -> torch.Tensor:
    """Fold ``max |x|`` into the amax record ``rec`` (one row of ``new_amax``; zero it first unless it already holds part of x)."""
    lib = _lib.load()
    _chk(rec, torch.int32, "amax record", (AMAX_WORDS,))
    if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous():
        raise ValueError(f"absmax: expected a contiguous f32 device tensor, got {tuple(x.shape)} {x.dtype} on {x.device}")
    with _on(x.device):
        st = lib.mi355ppo_absmax_f32(_ptr(x), x.numel(), _ptr(rec), _stream(x.device))
    _lib.check(st, "mi355ppo_absmax_f32")
    return rec


def amax_value(rec: torch.Tensor) -> float:
    """The record's value as a float (tests, debugging): the slots hold bit patterns of non-negative f32."""
    return float(rec.view(-1)[::16].contiguous().view(torch.float32).max().item())


def _rec(t, name):
    return None if t is None else _ptr(_chk(t, torch.int32, name, (AMAX_WORDS,)))



def warm_forward_packs(bufs, net) -> None:
    """Kernel Z's forward packs of layers 2 / 3 and of the FC layer (re-derived into the same buffers: captured rollout steps
    keep reading one address)."""
    if _CONV_Z:
        bufs.conv_zpack(net[2].weight, 2, MODE_FWD)
        bufs.conv_zpack(net[4].weight, 3, MODE_FWD)
    bufs.fc_pack_fwd(net[7].weight)        # Linear(3136, 512): kernel Z at every batch size (K split over the grid for a rollout step)


def repack_weights(W: torch.Tensor, layer: int, mode: int = MODE_FWD, out: torch.Tensor | None = None) -> torch.Tensor:
    lib = _lib.load()
    cin, cout, k, _, _, _ = LAYERS[layer]
    _chk(W, torch.float32, f"W{layer}", (cout, cin, k, k))
    numel = (BT_CLASSES_NUMEL if mode == MODE_DGRAD_S1_CLASSES else QPACK_NUMEL if mode == MODE_FWD_Q
             else BT2_CLASSES_NUMEL if mode == MODE_DGRAD_S2_CLASSES else W.numel())
    if out is None:
        out = torch.empty(numel, dtype=torch.float32, device=W.device)
    _chk(out, torch.float32, "Bt", (numel,))
    with _on(W.device):
        st = lib.mi355ppo_cnn_repack_weights_f32(_ptr(W), _ptr(out), layer, mode, _stream(W.device))
    _lib.check(st, "mi355ppo_cnn_repack_weights_f32")
    return out


def conv_fwd(src: torch.Tensor, Bt: torch.Tensor, bias: torch.Tensor, layer: int, inds: torch.Tensor | None = None,
             out: torch.Tensor | None = None, variant: int = 0) -> torch.Tensor:
    """``relu(conv(src) + bias)``; layer 1 takes the uint8 rollout rows (+ optional int64 row gather)."""
    lib = _lib.load()
    cin, cout, k, _, hin, hout = LAYERS[layer]
    if layer == 1:
        _chk(src, torch.uint8, "src")
        assert tuple(src.shape[1:]) == (hin, hin, cin), f"layer 1 source rows must be (84,84,4) uint8, got {tuple(src.shape)}"
        images = src.shape[0] if inds is None else inds.numel()
        if inds is not None:
            _chk(inds, torch.int64, "inds")
    else:
        assert inds is None
        images = src.shape[0]
        _chk(src, torch.float32, "src", (images, hin, hin, cin))
    _chk(Bt, torch.float32, "Bt", (QPACK_NUMEL if variant == VARIANT_Q else cout * cin * k * k,))
    _chk(bias, torch.float32, "bias", (cout,))
    if out is None:
        out = torch.empty((images, hout, hout, cout), dtype=torch.float32, device=src.device)
    _chk(out, torch.float32, "out", (images, hout, hout, cout))
    with _on(src.device):
        st = lib.mi355ppo_cnn_conv_fwd_f32_variant(_ptr(src), _ptr(inds), _ptr(Bt), _ptr(bias), _ptr(out), images, layer,
                                                   int(variant), _stream(src.device))
    _lib.check(st, "mi355ppo_cnn_conv_fwd_f32")
    return out


def conv_dgrad(dz: torch.Tensor, Bt: torch.Tensor, act_in: torch.Tensor, layer: int,
               out: torch.Tensor | None = None, variant: int = 0) -> torch.Tensor:
    """Gradient w.r.t. the layer's input activation, masked by ``act_in > 0`` (ReLU backward fused)."""
    lib = _lib.load()
    cin, cout, k, _, hin, hout = LAYERS[layer]
    images = dz.shape[0]
    _chk(dz, torch.float32, "dz", (images, hout, hout, cout))
    _chk(act_in, torch.float32, "act_in", (images, hin, hin, cin))
    _chk(Bt, torch.float32, "Bt", (BT_CLASSES_NUMEL if variant == 5 else BT2_CLASSES_NUMEL if variant == VARIANT_DGRAD2_CLASSES
                                   else cout * cin * k * k,))
    if out is None:
        out = torch.empty_like(act_in)
    _chk(out, torch.float32, "out", (images, hin, hin, cin))
    with _on(dz.device):
        st = lib.mi355ppo_cnn_conv_dgrad_f32_variant(_ptr(dz), _ptr(Bt), _ptr(act_in), _ptr(out), images, layer, int(variant),
                                                     _stream(dz.device))
    _lib.check(st, "mi355ppo_cnn_conv_dgrad_f32")
    return out


def conv_wgrad(src: torch.Tensor, dz: torch.Tensor, layer: int, inds: torch.Tensor | None = None, out=None, amax=None):
    """``(dW (Cout,Cin,K,K), db (Cout))`` from the layer input and the pre-activation gradient; ``out=(dW, db)`` writes into the
    caller's tensors (e.g. the parameters' ``.grad`` views of a flat gradient buffer).  ``amax = (src_rec, dz_rec)`` (layers 2 / 3):
    kernel V on the f16 split."""
    lib = _lib.load()
    cin, cout, k, _, hin, hout = LAYERS[layer]
    images = dz.shape[0]
    _chk(dz, torch.float32, "dz", (images, hout, hout, cout))
    if layer == 1:
        _chk(src, torch.uint8, "src")
        if inds is not None:
            _chk(inds, torch.int64, "inds", (images,))
        else:
            assert src.shape[0] == images
    else:
        assert inds is None
        _chk(src, torch.float32, "src", (images, hin, hin, cin))
    dev = dz.device
    if out is not None:
        dW, db = _chk(out[0], torch.float32, "dW", (cout, cin, k, k)), _chk(out[1], torch.float32, "db", (cout,))
    else:
        dW = torch.empty((cout, cin, k, k), dtype=torch.float32, device=dev)
        db = torch.empty(cout, dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.mi355ppo_cnn_conv_wgrad_workspace_bytes(images, layer))
    if amax is not None and layer == 1:          # kernel P with dz in two f16 terms: ``amax = (None, dz_rec)`` (the uint8 frames are exact)
        with _on(dev):
            st = lib.mi355ppo_cnn_conv1_wgrad_f16x2(_ptr(src), _ptr(inds), _ptr(dz), _ptr(dW), _ptr(db), images, _ptr(ws), ws.numel(),
                                                    _rec(amax[1], "dz_amax"), _stream(dev))
        _lib.check(st, "mi355ppo_cnn_conv1_wgrad_f16x2")
        return dW, db
    if amax is not None:
        with _on(dev):
            st = lib.mi355ppo_cnn_conv_wgrad_f16x2_f32(_ptr(src), _ptr(dz), _ptr(dW), _ptr(db), images, layer, _ptr(ws), ws.numel(),
                                                       _rec(amax[0], "src_amax"), _rec(amax[1], "dz_amax"), _stream(dev))
        _lib.check(st, "mi355ppo_cnn_conv_wgrad_f16x2_f32")
        return dW, db
    with _on(dev):
        st = lib.mi355ppo_cnn_conv_wgrad_f32(_ptr(src), _ptr(inds), _ptr(dz), _ptr(dW), _ptr(db), images, layer, _ptr(ws),
                                             ws.numel(), _stream(dev))
    _lib.check(st, "mi355ppo_cnn_conv_wgrad_f32")
    return dW, db


def trunk_fwd(obs_u8, inds, bt1, b1, bt2, b2, bt3, b3, a1, a2, a3, conv1_variant: int = 0):
    """conv1 -> conv2 -> conv3 (each with bias + ReLU) in one library call; buffers as produced by ``_Buffers``.
    ``conv1_variant=VARIANT_Q``: ``bt1`` is the mode-4 pack."""
    lib = _lib.load()
    _chk(obs_u8, torch.uint8, "obs_u8")
    images = a1.shape[0]
    dev = obs_u8.device
    with _on(dev):
        st = lib.mi355ppo_cnn_trunk_fwd_f32(_ptr(obs_u8), _ptr(inds), _ptr(bt1), _ptr(b1), _ptr(bt2), _ptr(b2), _ptr(bt3), _ptr(b3),
                                            _ptr(a1), _ptr(a2), _ptr(a3), images, int(conv1_variant), _stream(dev))
    _lib.check(st, "mi355ppo_cnn_trunk_fwd_f32")
    return a3


def fc_pack(B: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """(N, K) f32 matrix (unit column stride, any row pitch >= K) -> kernel Z's pre-split fragment-order pack (csrc/gemmz.hip)."""
    lib = _lib.load()
    if B.dtype != torch.float32 or not B.is_cuda or B.dim() != 2 or B.stride(1) != 1 or B.stride(0) < B.shape[1]:
        raise ValueError(f"fc_pack: expected a row-major f32 device matrix, got {tuple(B.shape)} strides {B.stride()} {B.dtype} on {B.device}")
    N, K = B.shape
    nbytes = lib.mi355ppo_fc_pack_bytes(N, K)
    if nbytes == 0:
        raise ValueError(f"fc_pack: K={K} must be a positive multiple of 16")
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=B.device)
    _chk(out, torch.uint8, "pack", (nbytes,))
    with _on(B.device):
        st = lib.mi355ppo_fc_pack_f32(_ptr(B), B.stride(0), N, K, _ptr(out), _stream(B.device))
    _lib.check(st, "mi355ppo_fc_pack_f32")
    return out


def fc_pack_f16x2(B: torch.Tensor, b_amax: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """``fc_pack`` in the f16x2 format (header + two f16 planes of s B; csrc/f16split.h).  ``b_amax``: B's amax record (computed here
    when absent)."""
    lib = _lib.load()
    if B.dtype != torch.float32 or not B.is_cuda or B.dim() != 2 or B.stride(1) != 1 or B.stride(0) < B.shape[1]:
        raise ValueError(f"fc_pack_f16x2: expected a row-major f32 device matrix, got {tuple(B.shape)} strides {B.stride()} {B.dtype} on {B.device}")
    N, K = B.shape
    nbytes = lib.mi355ppo_fc_pack_f16x2_bytes(N, K)
    if nbytes == 0:
        raise ValueError(f"fc_pack_f16x2: K={K} must be a positive multiple of 16")
    if b_amax is None:
        b_amax = new_amax(1, B.device)[0]
        if B.is_contiguous():
            absmax(B, b_amax)
        else:                                   # (a padded row pitch: the pad columns are not B's)
            absmax(B.contiguous(), b_amax)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=B.device)
    _chk(out, torch.uint8, "pack", (nbytes,))
    with _on(B.device):
        st = lib.mi355ppo_fc_pack_f16x2_f32(_ptr(B), B.stride(0), N, K, _rec(b_amax, "b_amax"), _ptr(out), _stream(B.device))
    _lib.check(st, "mi355ppo_fc_pack_f16x2_f32")
    return out


def fc_fwd_relu_packed(a: torch.Tensor, pack: torch.Tensor, bias: torch.Tensor, N: int, out: torch.Tensor | None = None,
                       amax=None) -> torch.Tensor:
    """``relu(a @ B.T + bias)`` with ``pack = fc_pack(B)``, B (N, K): kernel Z.  ``amax = (a_rec, h_rec | None)``: the f16 split --
    ``pack = fc_pack_f16x2(B)``, ``a_rec`` = a's amax record, ``h_rec`` receives the result's (whole-K route only)."""
    lib = _lib.load()
    M, K = a.shape
    lda = _row_major(a, "a")
    _chk(bias, torch.float32, "bias", (N,))
    _chk(pack, torch.uint8, "pack", ((lib.mi355ppo_fc_pack_f16x2_bytes if amax is not None else lib.mi355ppo_fc_pack_bytes)(N, K),))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _chk(out, torch.float32, "out", (M, N))
    nws = lib.mi355ppo_fc_fwd_workspace_bytes(M, N, K)       # rollout-sized batches: K split over the grid, partials in a workspace
    if amax is not None:
        ws = _workspace(a.device, nws) if nws else None
        with _on(a.device):
            st = lib.mi355ppo_fc_fwd_relu_packed_f16x2_f32(_ptr(a), lda, _ptr(pack), _ptr(bias), _ptr(out), M, N, K, _ptr(ws), ws.numel() if nws else 0,
                                                           _rec(amax[0], "a_amax"), None if nws else _rec(amax[1], "h_amax"), _stream(a.device))
        _lib.check(st, "mi355ppo_fc_fwd_relu_packed_f16x2_f32")
        return out
    with _on(a.device):
        if nws:
            ws = _workspace(a.device, nws)
            st = lib.mi355ppo_fc_fwd_relu_packed_ws_f32(_ptr(a), lda, _ptr(pack), _ptr(bias), _ptr(out), M, N, K, _ptr(ws), ws.numel(), _stream(a.device))
        else:
            st = lib.mi355ppo_fc_fwd_relu_packed_f32(_ptr(a), lda, _ptr(pack), _ptr(bias), _ptr(out), M, N, K, _stream(a.device))
    _lib.check(st, "mi355ppo_fc_fwd_relu_packed_f32")
    return out


def fc_heads_act_supported(a: torch.Tensor, N: int = 512) -> bool:
    """The fused rollout step behind the trunk applies wherever the FC forward splits K (below 8,192 rows)."""
    return bool(a.is_cuda and a.dim() == 2 and a.shape[1] % 16 == 0 and a.stride(1) == 1 and
                _lib.load().mi355ppo_fc_fwd_workspace_bytes(a.shape[0], N, a.shape[1]) > 0)


def fc_heads_act_supported_rows(rows: int, K: int = 3136, N: int = 512) -> bool:
    """The same question from the row count alone -- asked BEFORE the trunk runs, so that a batch the fused step does not take
    (8,192 rows and more) is not pushed through the trunk twice (once here, once by the fallback)."""
    return rows > 0 and _lib.load().mi355ppo_fc_fwd_workspace_bytes(int(rows), N, K) > 0


def fc_heads_act_categorical(a: torch.Tensor, pack: torch.Tensor, fc_bias: torch.Tensor, Wa, ba, Wc, bc, seed: int, offset: int,
                             offset_base=None, action_f32_out=None, logprob_out=None, value_out=None, noise_exp1=None, want_i64: bool = True,
                             amax=None):
    """``a`` = the trunk's (M, 3136) features -> Linear(3136,512) + ReLU (K split over the grid), actor / critic heads, Categorical
    sample + log-prob: two launches (``mi355ppo_fc_heads_act_categorical_f32``).  -> ``(action_i64 | None, action_f32, logprob, value)``."""
    lib = _lib.load()
    M, K = a.shape
    lda = _row_major(a, "a")
    A, H = Wa.shape
    dev = a.device
    _chk(fc_bias, torch.float32, "fc_bias", (H,))
    _chk(pack, torch.uint8, "pack", ((lib.mi355ppo_fc_pack_f16x2_bytes if amax is not None else lib.mi355ppo_fc_pack_bytes)(H, K),))
    a64 = torch.empty(M, dtype=torch.int64, device=dev) if want_i64 else None
    af = action_f32_out if action_f32_out is not None else torch.empty(M, dtype=torch.float32, device=dev)
    lp = logprob_out if logprob_out is not None else torch.empty(M, dtype=torch.float32, device=dev)
    val = value_out if value_out is not None else torch.empty(M, dtype=torch.float32, device=dev)
    for t, nm in ((af, "action_f32_out"), (lp, "logprob_out"), (val, "value_out")):
        _chk(t, torch.float32, nm, (M,))
    if offset_base is not None:
        _chk(offset_base, torch.int64, "offset_base", (1,))
    if noise_exp1 is not None:
        _chk(noise_exp1, torch.float32, "noise_exp1", (M, A))
    ws = _workspace(dev, lib.mi355ppo_fc_fwd_workspace_bytes(M, H, K))
    with _on(dev):
        if amax is not None:        # ``amax`` = a's amax record, ``pack`` an f16x2 pack
            st = lib.mi355ppo_fc_heads_act_categorical_f16x2_f32(_ptr(a), lda, _ptr(pack), _ptr(fc_bias), _ptr(Wa), _ptr(ba), _ptr(Wc), _ptr(bc), M, A,
                                                                 H, K, _ptr(noise_exp1), int(seed) & (2**64 - 1), int(offset) & (2**64 - 1),
                                                                 _ptr(offset_base), _ptr(a64), _ptr(af), _ptr(lp), _ptr(val), None, _ptr(ws),
                                                                 ws.numel(), _rec(amax, "a_amax"), _stream(dev))
        else:
            st = lib.mi355ppo_fc_heads_act_categorical_f32(_ptr(a), lda, _ptr(pack), _ptr(fc_bias), _ptr(Wa), _ptr(ba), _ptr(Wc), _ptr(bc), M, A, H, K,
                                                           _ptr(noise_exp1), int(seed) & (2**64 - 1), int(offset) & (2**64 - 1), _ptr(offset_base),
                                                           _ptr(a64), _ptr(af), _ptr(lp), _ptr(val), None, _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(st, "mi355ppo_fc_heads_act_categorical_f32")
    return a64, af, lp, val


def fc_dgrad_mask_packed(dz: torch.Tensor, pack: torch.Tensor, act_in: torch.Tensor, out: torch.Tensor | None = None,
                         bits: torch.Tensor | None = None, amax=None) -> torch.Tensor:
    """``(dz @ B.T) * (act_in > 0)`` with ``pack = fc_pack(B)``, B (N, K) = the transposed weight: kernel Z.  ``bits``: the mask
    ``act_in > 0`` as written by ``conv_fwd_packed(..., bits=)`` (N % 32 == 0); ``act_in`` then only gives the shape."""
    lib = _lib.load()
    M, K = dz.shape
    N = act_in.shape[1]
    lddz = _row_major(dz, "dz")
    _chk(act_in, torch.float32, "act_in", (M, N))
    _chk(pack, torch.uint8, "pack", ((lib.mi355ppo_fc_pack_f16x2_bytes if amax is not None else lib.mi355ppo_fc_pack_bytes)(N, K),))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=dz.device)
    _chk(out, torch.float32, "out", (M, N))
    if amax is not None:            # the f16 split: ``amax = (dz_rec, out_rec | None)``, ``pack = fc_pack_f16x2(B)``
        if bits is not None:
            _chk(bits, torch.int32, "bits", (mask_words(M * N),))
        with _on(dz.device):
            st = lib.mi355ppo_fc_dgrad_packed_f16x2_f32(_ptr(dz), lddz, _ptr(pack), _ptr(act_in), _ptr(bits), _ptr(out), M, N, K,
                                                        _rec(amax[0], "dz_amax"), _rec(amax[1], "da_amax"), _stream(dz.device))
        _lib.check(st, "mi355ppo_fc_dgrad_packed_f16x2_f32")
        return out
    with _on(dz.device):
        if bits is not None:
            _chk(bits, torch.int32, "bits", (mask_words(M * N),))
            st = lib.mi355ppo_fc_dgrad_maskbits_packed_f32(_ptr(dz), lddz, _ptr(pack), _ptr(bits), _ptr(out), M, N, K, _stream(dz.device))
        else:
            st = lib.mi355ppo_fc_dgrad_mask_packed_f32(_ptr(dz), lddz, _ptr(pack), _ptr(act_in), _ptr(out), M, N, K, _stream(dz.device))
    _lib.check(st, "mi355ppo_fc_dgrad_mask_packed_f32")
    return out


ZPACK_SHAPE = {(2, MODE_FWD): (64, 512), (3, MODE_FWD): (64, 576), (3, MODE_DGRAD_S1): (64, 576), (2, MODE_DGRAD_S2): (128, 256)}


def conv_zpack(W: torch.Tensor, layer: int, mode: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """Kernel Z's pack of a conv layer's (N, K) matrix: ``fc_pack(repack_weights(W, layer, mode))`` -- mode 0 for the forward,
    MODE_DGRAD_S1 (layer 3) / MODE_DGRAD_S2 (layer 2) for the data gradients."""
    n, k = ZPACK_SHAPE[(layer, mode)]
    return fc_pack(repack_weights(W, layer, mode).view(n, k), out)


def conv_zpack_f16x2(W: torch.Tensor, layer: int, mode: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """``conv_zpack`` in the f16x2 format (the scale from max |W|)."""
    n, k = ZPACK_SHAPE[(layer, mode)]
    return fc_pack_f16x2(repack_weights(W, layer, mode).view(n, k), None, out)


def mask_words(numel: int) -> int:
    """uint32 words of a ReLU bit mask over ``numel`` activation elements (bit b of word w <-> flat element 32 w + b)."""
    assert numel % 32 == 0
    return numel // 32


def unpack_mask_bits(bits: torch.Tensor, shape) -> torch.Tensor:
    """The bit mask written by the ``*_bits`` forwards as a bool tensor of the activation's shape (tests, debugging)."""
    w = bits.view(-1).to(torch.int64) & 0xFFFFFFFF
    return ((w[:, None] >> torch.arange(32, device=bits.device)) & 1).bool().view(*shape)


def conv1q_fwd_amax(obs_u8: torch.Tensor, pack: torch.Tensor, bias: torch.Tensor, inds: torch.Tensor | None, out: torch.Tensor,
                    bits: torch.Tensor | None, dst_amax: torch.Tensor) -> torch.Tensor:
    """Layer-1 forward on kernel Q that also folds ``max(out)`` into the amax record ``dst_amax`` (and writes the mask bits when given)."""
    lib = _lib.load()
    _chk(obs_u8, torch.uint8, "src")
    images = obs_u8.shape[0] if inds is None else inds.numel()
    if inds is not None:
        _chk(inds, torch.int64, "inds")
    _chk(pack, torch.float32, "Bt", (QPACK_NUMEL,))
    _chk(bias, torch.float32, "bias", (32,))
    _chk(out, torch.float32, "out", (images, 20, 20, 32))
    if bits is not None:
        _chk(bits, torch.int32, "bits", (mask_words(out.numel()),))
    with _on(obs_u8.device):
        st = lib.mi355ppo_cnn_conv1q_fwd_amax(_ptr(obs_u8), _ptr(inds), _ptr(pack), _ptr(bias), _ptr(out), _ptr(bits), images,
                                              _rec(dst_amax, "dst_amax"), _stream(obs_u8.device))
    _lib.check(st, "mi355ppo_cnn_conv1q_fwd_amax")
    return out


def conv1q_fwd_bits(obs_u8: torch.Tensor, pack: torch.Tensor, bias: torch.Tensor, inds: torch.Tensor | None, out: torch.Tensor,
                    bits: torch.Tensor) -> torch.Tensor:
    """Layer-1 forward on kernel Q that also writes ``(out > 0)`` as bits (``bits``: int32, ``out.numel() // 32`` words)."""
    lib = _lib.load()
    _chk(obs_u8, torch.uint8, "src")
    images = obs_u8.shape[0] if inds is None else inds.numel()
    if inds is not None:
        _chk(inds, torch.int64, "inds")
    _chk(pack, torch.float32, "Bt", (QPACK_NUMEL,))
    _chk(bias, torch.float32, "bias", (32,))
    _chk(out, torch.float32, "out", (images, 20, 20, 32))
    _chk(bits, torch.int32, "bits", (mask_words(out.numel()),))
    with _on(obs_u8.device):
        st = lib.mi355ppo_cnn_conv1q_fwd_bits(_ptr(obs_u8), _ptr(inds), _ptr(pack), _ptr(bias), _ptr(out), _ptr(bits), images, _stream(obs_u8.device))
    _lib.check(st, "mi355ppo_cnn_conv1q_fwd_bits")
    return out


def conv_fwd_packed(src: torch.Tensor, pack: torch.Tensor, bias: torch.Tensor, layer: int, out: torch.Tensor | None = None,
                    bits: torch.Tensor | None = None, amax=None) -> torch.Tensor:
    """``relu(conv(src) + bias)`` of layer 2 / 3 on kernel Z (csrc/gemmz.hip); ``pack = conv_zpack(W, layer, MODE_FWD)``.
    ``bits`` (int32, ``out.numel() // 32`` words): also receives ``(out > 0)`` as a bit mask."""
    lib = _lib.load()
    cin, cout, k, _, hin, hout = LAYERS[layer]
    images = src.shape[0]
    _chk(src, torch.float32, "src", (images, hin, hin, cin))
    _chk(pack, torch.uint8, "pack", ((lib.mi355ppo_fc_pack_f16x2_bytes if amax is not None else lib.mi355ppo_fc_pack_bytes)(cout, cin * k * k),))
    _chk(bias, torch.float32, "bias", (cout,))
    if out is None:
        out = torch.empty((images, hout, hout, cout), dtype=torch.float32, device=src.device)
    _chk(out, torch.float32, "out", (images, hout, hout, cout))
    if amax is not None:            # the f16 split: ``amax = (src_rec, out_rec | None)``, ``pack = conv_zpack_f16x2(W, layer, MODE_FWD)``
        if bits is not None:
            _chk(bits, torch.int32, "bits", (mask_words(out.numel()),))
        with _on(src.device):
            st = lib.mi355ppo_cnn_conv_fwd_packed_f16x2_f32(_ptr(src), _ptr(pack), _ptr(bias), _ptr(out), _ptr(bits), images, layer,
                                                            _rec(amax[0], "src_amax"), _rec(amax[1], "dst_amax"), _stream(src.device))
        _lib.check(st, "mi355ppo_cnn_conv_fwd_packed_f16x2_f32")
        return out
    with _on(src.device):
        if bits is not None:
            _chk(bits, torch.int32, "bits", (mask_words(out.numel()),))
            st = lib.mi355ppo_cnn_conv_fwd_packed_bits_f32(_ptr(src), _ptr(pack), _ptr(bias), _ptr(out), _ptr(bits), images, layer, _stream(src.device))
        else:
            st = lib.mi355ppo_cnn_conv_fwd_packed_f32(_ptr(src), _ptr(pack), _ptr(bias), _ptr(out), images, layer, _stream(src.device))
    _lib.check(st, "mi355ppo_cnn_conv_fwd_packed_f32")
    return out


def conv_dgrad_packed(dz: torch.Tensor, pack: torch.Tensor, act_in: torch.Tensor | None, layer: int, out: torch.Tensor | None = None,
                      bits: torch.Tensor | None = None, amax=None) -> torch.Tensor:
    """Data gradient of layer 2 / 3 masked by ``act_in > 0`` on kernel Z; ``pack = conv_zpack(W, layer, MODE_DGRAD_S2 / _S1)``.
    ``bits``: the mask as written by the layer below's ``*_bits`` forward instead of ``act_in`` (which may then be None)."""
    lib = _lib.load()
    cin, cout, k, _, hin, hout = LAYERS[layer]
    images = dz.shape[0]
    _chk(dz, torch.float32, "dz", (images, hout, hout, cout))
    if bits is None:
        _chk(act_in, torch.float32, "act_in", (images, hin, hin, cin))
    n, kk = ZPACK_SHAPE[(layer, MODE_DGRAD_S2 if layer == 2 else MODE_DGRAD_S1)]
    _chk(pack, torch.uint8, "pack", ((lib.mi355ppo_fc_pack_f16x2_bytes if amax is not None else lib.mi355ppo_fc_pack_bytes)(n, kk),))
    if out is None:
        out = torch.empty((images, hin, hin, cin), dtype=torch.float32, device=dz.device)
    _chk(out, torch.float32, "out", (images, hin, hin, cin))
    if amax is not None:            # the f16 split: ``amax = (dz_rec, out_rec | None)``
        if bits is not None:
            _chk(bits, torch.int32, "bits", (mask_words(out.numel()),))
        with _on(dz.device):
            st = lib.mi355ppo_cnn_conv_dgrad_packed_f16x2_f32(_ptr(dz), _ptr(pack), _ptr(act_in), _ptr(bits), _ptr(out), images, layer,
                                                              _rec(amax[0], "dz_amax"), _rec(amax[1], "dsrc_amax"), _stream(dz.device))
        _lib.check(st, "mi355ppo_cnn_conv_dgrad_packed_f16x2_f32")
        return out
    with _on(dz.device):
        if bits is not None:
            _chk(bits, torch.int32, "bits", (mask_words(out.numel()),))
            st = lib.mi355ppo_cnn_conv_dgrad_packed_bits_f32(_ptr(dz), _ptr(pack), _ptr(bits), _ptr(out), images, layer, _stream(dz.device))
        else:
            st = lib.mi355ppo_cnn_conv_dgrad_packed_f32(_ptr(dz), _ptr(pack), _ptr(act_in), _ptr(out), images, layer, _stream(dz.device))
    _lib.check(st, "mi355ppo_cnn_conv_dgrad_packed_f32")
    return out


FC_PAD = 4          # extra floats per row of the K = 512 operands of the FC data gradient: a dense 2 KiB pitch puts the 32 rows of
                    # a fragment load on one cache channel (measured: 1.47 ms against 0.70 ms for the forward of the same size)


def _row_major(t: torch.Tensor, name: str):
    """(rows, cols) f32 device tensor with unit column stride and a row pitch that is a multiple of 4 floats -> its pitch."""
    if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 4 or t.stride(0) < t.shape[1]:
        raise ValueError(f"{name}: expected a row-major f32 device matrix with a 16-byte-multiple row pitch, got shape "
                         f"{tuple(t.shape)} strides {t.stride()} {t.dtype} on {t.device}")
    return t.stride(0)


def padded_rows(rows: int, cols: int, device, pad: int = FC_PAD) -> torch.Tensor:
    """A (rows, cols) view with row pitch cols + pad."""
    return torch.empty((rows, cols + pad), dtype=torch.float32, device=device)[:, :cols]


def fc_wgrad(dz: torch.Tensor, a: torch.Tensor, hwc_channels: int = 0, out: torch.Tensor | None = None, amax=None) -> torch.Tensor:
    """``dz.T @ a`` (N, K) (csrc/fcw.hip: kernel W on the bf16 pipe at minibatch sizes, kernel Y on the f32 pipe otherwise), the
    batch slabs added in a fixed order.  With
    ``hwc_channels = C`` the columns of ``a`` are (h, w, c)-ordered features and the result comes out in the reference's
    (c, h, w) order -- the gradient of ``Linear.weight`` itself."""
    lib = _lib.load()
    M, N = dz.shape
    K = a.shape[1]
    lddz = _row_major(dz, "dz")
    _chk(a, torch.float32, "a", (M, K))
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=dz.device)
    _chk(out, torch.float32, "out", (N, K))
    ws = _workspace(dz.device, lib.mi355ppo_fc_wgrad_workspace_bytes(M, N, K))
    if amax is not None:            # kernel W on the f16 split: ``amax = (dz_rec, a_rec)``
        with _on(dz.device):
            st = lib.mi355ppo_fc_wgrad_f16x2_f32(_ptr(dz), lddz, _ptr(a), _ptr(out), M, N, K, int(hwc_channels), _ptr(ws), ws.numel(),
                                                 _rec(amax[0], "dz_amax"), _rec(amax[1], "a_amax"), _stream(dz.device))
        _lib.check(st, "mi355ppo_fc_wgrad_f16x2_f32")
        return out
    with _on(dz.device):
        st = lib.mi355ppo_fc_wgrad_f32(_ptr(dz), lddz, _ptr(a), _ptr(out), M, N, K, int(hwc_channels), _ptr(ws), ws.numel(), _stream(dz.device))
    _lib.check(st, "mi355ppo_fc_wgrad_f32")
    return out


FCZ_MIN_ROWS = 1             # kernel Z at every batch size: from 8,192 rows whole-K wave tiles, below (a rollout step, config B's minibatch) K split over the grid


class _Buffers:
    """Activation / gradient buffers reused across calls of one batch size (no allocator traffic in the loop), and the
    repacked weight matrices.  The matrices are re-derived from the parameters on every use unless the owner opts in to
    caching (``cache_weights = True``) and promises to bump ``weights_version`` whenever it changes the parameters
    behind torch's back -- the learner does, after each fused clip+Adam step, which writes the flat buffer through raw
    pointers; in-place torch updates are caught through the tensors' own version counters."""

    def __init__(self):
        self.by_m = {}
        self.cache_weights = False
        self.weights_version = 0
        self._bt = {}
        self.last_a3_ptr = None            # data pointer of the a3 the trunk produced last (LinearReLUHwcFn checks its input is it)
        self.last_a3_bits = None           # ... and its ReLU mask as bits, when the trunk's forward wrote one (kernel Z, gradients enabled)
        self.a3_grad_is_masked = False     # set by LinearReLUHwcFn.backward when conv3's ReLU backward rode in the FC data gradient
        self.fc_dz_from_heads = None       # (data pointer of dz, FC bias gradient) when the FC layer's ReLU backward rode in HeadsFn.backward
        # direct_grads: the backward nodes WRITE their parameter gradients into the parameters' .grad tensors (views of a flat gradient
        # buffer that the optimizer kernel leaves zeroed) and hand autograd None -- no temporary, no AccumulateGrad add per parameter
        # (12 small launches per minibatch).  Only an owner that guarantees zeroed .grad before every backward sets it (PPOLearner).
        self.direct_grads = False
        self.after_fc_wgrad = None         # callback(param) once Linear(3136,512).weight's gradient is final (world > 1: early all-reduce)
        # (W1, W2, W3, Wfc) of the owning NatureCNN agent: with cached weights, a stale pack then means ALL of kernel Q's / kernel Z's
        # packs are rebuilt together from the parameters in one launch (mi355ppo_nature_packs_f32) instead of 13 small ones
        self.pack_params = None
        # round 5, the two-term f16 split (csrc/f16split.h): the amax records of the pass in flight (one set per batch size and stream,
        # like the activations they describe), which tensor each record currently belongs to, and the weights' records (f16x2 packs)
        self.split = _SPLIT
        self._owners = {}
        self.w_amax = None

    def f16(self, t: torch.Tensor) -> bool:
        """Kernels Z / V / W take the f16 split for this tensor's pass (device tensors; the CPU stand-ins of the tests keep the bf16 names
        unless they set ``f16_on_cpu`` and stand in for the amax-aware entry points too)."""
        return self.split == "f16x2" and _CONV_Z and (t.is_cuda or getattr(self, "f16_on_cpu", False))

    @staticmethod
    def _amax_key(m: int, dev):
        return ("amax", m, dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)

    def has_pass(self, m: int, dev) -> bool:
        return self._amax_key(m, dev) in self.by_m

    def begin_pass(self, m: int, dev) -> torch.Tensor:
        """The amax records of a forward (+ backward) pass over ``m`` images on the current stream (one set per batch size and stream,
        like the activations they describe -- the env-group lanes of a rollout run side by side), zeroed: ONE fill for all of them."""
        key = self._amax_key(m, dev)
        rec = self.by_m.get(key)
        if rec is None:
            rec = new_amax(N_REC, dev)
            self.by_m[key] = rec
        else:
            rec.zero_()
        self._owners[key] = {}
        return rec

    def owns(self, idx: int, t: torch.Tensor) -> torch.Tensor:
        """Record ``idx`` of t's pass, for a producer about to write ``t`` (its epilogue fills the record)."""
        key = self._amax_key(t.shape[0], t.device)
        self._owners[key][idx] = (t.data_ptr(), weakref.ref(t))
        return self.by_m[key][idx]

    def rec_of(self, idx: int, t: torch.Tensor) -> torch.Tensor:
        """Record ``idx`` holding ``max |t|``: as filled by t's producer, or -- a tensor that came from somewhere else (torch's
        threshold_backward, a head wider than the fused kernel takes) -- computed here with one pass over it."""
        key = self._amax_key(t.shape[0], t.device)
        rec = self.by_m.get(key)
        if rec is None:
            rec = self.begin_pass(t.shape[0], t.device)
        # A record is trusted only for the tensor its PRODUCER filled it for: same address AND the producer's tensor object still alive -- once it
        # is gone the allocator may hand the address to another tensor (a second backward over one forward, an auxiliary pass), and a record that
        # understates a tensor overflows the f16 split.  A tensor from elsewhere is measured on every use; it never becomes an owner.
        o = self._owners[key].get(idx)
        if o is None or o[0] != t.data_ptr() or o[1]() is None:
            self._owners[key].pop(idx, None)
            rec[idx].zero_()
            absmax(t if t.is_contiguous() else t.contiguous(), rec[idx])      # (module attribute at call time: the tests' stand-in)
        return rec[idx]

    def _pack_keys(self):
        W1, W2, W3, Wfc = self.pack_params
        return [((1, MODE_FWD_Q), W1, None), (("zpack", 2, MODE_FWD), W2, (64, 512)), (("zpack", 3, MODE_FWD), W3, (64, 576)),
                (("zpack", 3, MODE_DGRAD_S1), W3, (64, 576)), (("zpack", 2, MODE_DGRAD_S2), W2, (128, 256)),
                ("fc_pack_fwd", Wfc, (512, 3136)), ("fc_pack_dgrad", Wfc, (3136, 512))]

    def _repack_all(self, key, W) -> bool:
        """Rebuild every cached pack in one launch when ``key`` (stale) is one of them and ``W`` its parameter.  -> done?"""
        if self.pack_params is None or not self.cache_weights or not _FUSED_PACKS:
            return False
        keys = self._pack_keys()
        if not any(k == key and w.data_ptr() == W.data_ptr() for k, w, _ in keys) or not all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for _, w, _ in keys):
            return False
        lib = _lib.load()
        dev = W.device
        f16 = self.f16(W)
        nbytes = lib.mi355ppo_fc_pack_f16x2_bytes if f16 else lib.mi355ppo_fc_pack_bytes
        outs = []
        for k, w, shape in keys:
            hit = self._bt.get(k)
            if hit is None:
                buf = (torch.empty(QPACK_NUMEL, dtype=torch.float32, device=dev) if shape is None
                       else torch.empty(nbytes(*shape), dtype=torch.uint8, device=dev))
            else:
                buf = hit[1]
            outs.append(buf)
        W1, W2, W3, Wfc = (w.detach() for w in self.pack_params)
        with _on(dev):
            if f16:
                if self.w_amax is None:
                    self.w_amax = new_amax(3, dev)
                st = lib.mi355ppo_nature_packs_f16x2_f32(_ptr(W1), _ptr(W2), _ptr(W3), _ptr(Wfc), *[_ptr(o) for o in outs], _ptr(self.w_amax), _stream(dev))
            else:
                st = lib.mi355ppo_nature_packs_f32(_ptr(W1), _ptr(W2), _ptr(W3), _ptr(Wfc), *[_ptr(o) for o in outs], _stream(dev))
        _lib.check(st, "mi355ppo_nature_packs_f16x2_f32" if f16 else "mi355ppo_nature_packs_f32")
        for (k, w, _), buf in zip(keys, outs):
            self._bt[k] = ((self.weights_version, w._version, w.data_ptr()), buf)
        return True

    def fc_weight(self, W: torch.Tensor) -> torch.Tensor:
        """Linear(3136,512) weight with (h,w,c)-ordered input features, cached like the conv matrices (re-derived INTO the
        same buffer: captured rollout steps keep reading one address)."""
        if not self.cache_weights:
            return fc_weight_hwc(W.detach()).contiguous()
        tag = (self.weights_version, W._version, W.data_ptr())
        hit = self._bt.get("fc")
        if hit is None or hit[0] != tag:
            if hit is None:
                hit = (tag, fc_weight_hwc(W.detach()).contiguous())
            else:
                hit[1].copy_(fc_weight_hwc(W.detach()))
                hit = (tag, hit[1])
            self._bt["fc"] = hit
        return hit[1]

    def fc_pack_fwd(self, W: torch.Tensor) -> torch.Tensor:
        """Kernel Z's pack of the (h,w,c)-ordered Linear(3136,512) weight (the forward's B operand), cached like the matrices."""
        return self._pack_cached("fc_pack_fwd", W, lambda: self.fc_weight(W))

    def fc_pack_dgrad(self, W: torch.Tensor) -> torch.Tensor:
        """Kernel Z's pack of its transpose (3136, 512): the data gradient's B operand."""
        return self._pack_cached("fc_pack_dgrad", W, lambda: self.fc_weight_t(W))

    def conv_zpack(self, W: torch.Tensor, layer: int, mode: int) -> torch.Tensor:
        """Kernel Z's pack of a conv layer's matrix (forward: MODE_FWD; data gradients: MODE_DGRAD_S1 / _S2), cached likewise."""
        return self._pack_cached(("zpack", layer, mode), W, lambda: self.weights(W, layer, mode).view(*ZPACK_SHAPE[(layer, mode)]))

    def _pack_cached(self, key, W, source):
        pack = (lambda B, out=None: fc_pack_f16x2(B, None, out)) if self.f16(W) else fc_pack      # (looked up per call: the tests patch cnn.fc_pack)
        if not self.cache_weights:
            return pack(source())
        tag = (self.weights_version, W._version, W.data_ptr())
        hit = self._bt.get(key)
        if hit is None or hit[0] != tag:
            if self._repack_all(key, W):
                return self._bt[key][1]
            hit = (tag, pack(source(), hit[1] if hit is not None else None))
            self._bt[key] = hit
        return hit[1]

    def fc_dz(self, m: int, n: int, dev) -> torch.Tensor:
        """Reusable (m, n) buffer with a padded row pitch for the FC layer's pre-activation gradient."""
        key = ("fc_dz", m, n, dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        b = self.by_m.get(key)
        if b is None:
            b = padded_rows(m, n, dev)
            self.by_m[key] = b
        return b

    def fc_weight_t(self, W: torch.Tensor) -> torch.Tensor:
        """``fc_weight(W).T`` as a dense (3136, 512) matrix (the B operand of the FC data gradient), cached likewise."""
        if not self.cache_weights:
            wt = padded_rows(W.shape[1], W.shape[0], W.device)
            wt.copy_(fc_weight_hwc(W.detach()).t())
            return wt
        tag = (self.weights_version, W._version, W.data_ptr())
        hit = self._bt.get("fc_t")
        if hit is None or hit[0] != tag:
            wt = hit[1] if hit is not None else padded_rows(W.shape[1], W.shape[0], W.device)     # (3136, 512), pitch 516
            wt.copy_(self.fc_weight(W).t())
            hit = (tag, wt)
            self._bt["fc_t"] = hit
        return hit[1]

    def weights(self, W: torch.Tensor, layer: int, mode: int) -> torch.Tensor:
        if not self.cache_weights:
            return repack_weights(W.detach(), layer, mode)
        key = (layer, mode)
        tag = (self.weights_version, W._version, W.data_ptr())
        hit = self._bt.get(key)
        if hit is None or hit[0] != tag:
            if mode == MODE_FWD_Q and self._repack_all(key, W):
                return self._bt[key][1]
            hit = (tag, repack_weights(W.detach(), layer, mode, hit[1] if hit is not None else None))
            self._bt[key] = hit
        return hit[1]

    def get_bits(self, m: int, dev):
        """ReLU masks of a1, a2, a3 as bits (int32 words; one set per batch size and stream, like the activations)."""
        key = ("bits", m, dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        b = self.by_m.get(key)
        if b is None:
            b = [torch.empty(mask_words(m * n), dtype=torch.int32, device=dev) for n in (20 * 20 * 32, 9 * 9 * 64, 7 * 7 * 64)]
            self.by_m[key] = b
        return b

    def get(self, m: int, dev, grads: bool):
        # one set per (batch size, stream): the env-group lanes of a rollout (pipeline.py) run the same batch size concurrently
        key = (m, dev, grads, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        b = self.by_m.get(key)
        if b is None:
            shapes = [(m, 20, 20, 32), (m, 9, 9, 64), (m, 7, 7, 64)]
            b = [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]
            self.by_m[key] = b
        return b


class NatureTrunkFn(torch.autograd.Function):
    """a3 = relu(conv3(relu(conv2(relu(conv1(obs[inds] / 255)))))) as one autograd node; a3 is (M, 7, 7, 64)."""

    @staticmethod
    def forward(ctx, obs_u8, inds, W1, b1, W2, b2, W3, b3, bufs, want_bits=False):
        m = obs_u8.shape[0] if inds is None else inds.numel()
        a1, a2, a3 = bufs.get(m, obs_u8.device, False)
        if a1.numel() * 4 >= BUF_LIMIT:     # beyond 32-bit buffer offsets: the f32-pipe kernels with 64-bit pointers (kernel S) throughout
            conv_fwd(obs_u8, bufs.weights(W1, 1, MODE_FWD), b1.detach(), 1, inds, a1)
            conv_fwd(a1, bufs.weights(W2, 2, MODE_FWD), b2.detach(), 2, None, a2)
            conv_fwd(a2, bufs.weights(W3, 3, MODE_FWD), b3.detach(), 3, None, a3)
            ctx.obs, ctx.inds, ctx.acts, ctx.bufs, ctx.bits = obs_u8, inds, (a1, a2, a3), bufs, None
            ctx.params = (W1, b1, W2, b2, W3, b3)
            bufs.last_a3_ptr, bufs.last_a3_bits = a3.data_ptr(), None
            ctx.save_for_backward(W2, W3)
            return a3
        # layer 1 runs on the integer matrix pipe (kernel Q): uint8 taps are exact int8 operands, weights four int8 digits
        bt1 = bufs.weights(W1, 1, MODE_FWD_Q)
        ctx.bits = None
        bufs.last_a3_bits = None
        ctx.f16 = f16 = bufs.f16(obs_u8)
        if f16:
            # the two-term f16 split (csrc/f16split.h): every forward folds its output's maximum into the tensor's amax record, the
            # next layer scales its A operand by it; one fill zeroes the records of this forward and of the backward that may follow
            rec = bufs.begin_pass(m, obs_u8.device)
            mb1 = mb2 = mb3 = None
            if _MASK_BITS and want_bits and any(ctx.needs_input_grad):
                mb1, mb2, mb3 = ctx.bits = bufs.get_bits(m, obs_u8.device)
                bufs.last_a3_bits = mb3
            conv1q_fwd_amax(obs_u8, bt1, b1.detach(), inds, a1, mb1, bufs.owns(REC_A1, a1))
            conv_fwd_packed(a1, bufs.conv_zpack(W2, 2, MODE_FWD), b2.detach(), 2, a2, bits=mb2, amax=(rec[REC_A1], bufs.owns(REC_A2, a2)))
            conv_fwd_packed(a2, bufs.conv_zpack(W3, 3, MODE_FWD), b3.detach(), 3, a3, bits=mb3, amax=(rec[REC_A2], bufs.owns(REC_A3, a3)))
        elif _CONV_Z and _MASK_BITS and want_bits and any(ctx.needs_input_grad):
            # a backward will follow (the caller saw gradients enabled; inside forward() they never are): every forward also writes its ReLU mask as bits -- the data gradients then read 1/32 of the
            # mask bytes (the f32 activations stay: the weight gradients read them)
            mb1, mb2, mb3 = bufs.get_bits(m, obs_u8.device)
            conv1q_fwd_bits(obs_u8, bt1, b1.detach(), inds, a1, mb1)
            conv_fwd_packed(a1, bufs.conv_zpack(W2, 2, MODE_FWD), b2.detach(), 2, a2, bits=mb2)
            conv_fwd_packed(a2, bufs.conv_zpack(W3, 3, MODE_FWD), b3.detach(), 3, a3, bits=mb3)
            ctx.bits = (mb1, mb2, mb3)
            bufs.last_a3_bits = mb3
        elif _CONV_Z:               # layers 2 and 3 on kernel Z (bf16 pipe, pre-split weights, coalesced window loads)
            conv_fwd(obs_u8, bt1, b1.detach(), 1, inds, a1, variant=VARIANT_Q)
            conv_fwd_packed(a1, bufs.conv_zpack(W2, 2, MODE_FWD), b2.detach(), 2, a2)
            conv_fwd_packed(a2, bufs.conv_zpack(W3, 3, MODE_FWD), b3.detach(), 3, a3)
        elif m <= 4096 and obs_u8.is_contiguous() and tuple(obs_u8.shape[1:]) == (84, 84, 4):     # inference-sized: one call
            bt2, bt3 = bufs.weights(W2, 2, MODE_FWD), bufs.weights(W3, 3, MODE_FWD)
            trunk_fwd(obs_u8, inds, bt1, b1.detach(), bt2, b2.detach(), bt3, b3.detach(), a1, a2, a3, conv1_variant=VARIANT_Q)
        else:
            conv_fwd(obs_u8, bt1, b1.detach(), 1, inds, a1, variant=VARIANT_Q)
            conv_fwd(a1, bufs.weights(W2, 2, MODE_FWD), b2.detach(), 2, None, a2)
            conv_fwd(a2, bufs.weights(W3, 3, MODE_FWD), b3.detach(), 3, None, a3)
        ctx.obs, ctx.inds, ctx.acts, ctx.bufs = obs_u8, inds, (a1, a2, a3), bufs
        ctx.params = (W1, b1, W2, b2, W3, b3)
        bufs.last_a3_ptr = a3.data_ptr()
        ctx.save_for_backward(W2, W3)
        return a3

    @staticmethod
    def backward(ctx, da3):
        W2, W3 = ctx.saved_tensors
        a1, a2, a3 = ctx.acts
        m = a3.shape[0]
        dz1, dz2, _ = ctx.bufs.get(m, a3.device, True)
        if ctx.bufs.a3_grad_is_masked:          # the FC data-gradient kernel already applied (a3 > 0)
            ctx.bufs.a3_grad_is_masked = False
            dz3 = da3.contiguous().view(a3.shape)
        else:
            dz3 = torch.ops.aten.threshold_backward(da3.contiguous(), a3, 0.0)    # ReLU backward of the last conv
        direct = ctx.bufs.direct_grads and all(p.grad is not None for p in ctx.params)
        gout = (lambda l: (ctx.params[2 * l - 2].grad, ctx.params[2 * l - 1].grad)) if direct else (lambda l: None)
        if ctx.bufs.f16(dz3) and a1.numel() * 4 < BUF_LIMIT:      # the two-term f16 split (records: from the producers, else computed here)
            bufs, bits = ctx.bufs, ctx.bits
            r3 = bufs.rec_of(REC_DZ3, dz3)          # filled by the FC data gradient's epilogue when it produced dz3
            dW3, db3 = conv_wgrad(a2, dz3, 3, out=gout(3), amax=(bufs.rec_of(REC_A2, a2), r3))
            conv_dgrad_packed(dz3, bufs.conv_zpack(W3, 3, MODE_DGRAD_S1), a2, 3, dz2, bits=bits[1] if bits else None,
                              amax=(r3, bufs.owns(REC_DZ2, dz2)))
            r2 = bufs.rec_of(REC_DZ2, dz2)
            dW2, db2 = conv_wgrad(a1, dz2, 2, out=gout(2), amax=(bufs.rec_of(REC_A1, a1), r2))
            conv_dgrad_packed(dz2, bufs.conv_zpack(W2, 2, MODE_DGRAD_S2), a1, 2, dz1, bits=bits[0] if bits else None,
                              amax=(r2, bufs.owns(REC_DZ1, dz1)))
            dW1, db1 = conv_wgrad(ctx.obs, dz1, 1, ctx.inds, out=gout(1), amax=(None, bufs.rec_of(REC_DZ1, dz1)))      # kernel P, dz in two f16 terms
            if direct:
                return (None,) * 10
            return None, None, dW1, db1, dW2, db2, dW3, db3, None, None
        dW3, db3 = conv_wgrad(a2, dz3, 3, out=gout(3))
        # (layer-3 data gradient: kernel Z multiplies the padding taps, 1.65 x the MFMAs, and still beats kernel F's nine
        # border-class launches at every size measured: profiles/r03_conv_traffic_ab_same_box.jsonl)
        bits = ctx.bits
        if _CONV_Z and a2.numel() * 4 < BUF_LIMIT and ctx.bufs.f16(dz3):      # (a1 beyond 4 GiB, a2 within: kernel Z for layer 3, on its f16 pack)
            conv_dgrad_packed(dz3, ctx.bufs.conv_zpack(W3, 3, MODE_DGRAD_S1), a2, 3, dz2, amax=(ctx.bufs.rec_of(REC_DZ3, dz3), None))
        elif _CONV_Z and a2.numel() * 4 < BUF_LIMIT:
            conv_dgrad_packed(dz3, ctx.bufs.conv_zpack(W3, 3, MODE_DGRAD_S1), a2, 3, dz2, bits=bits[1] if bits else None)
        elif a2.numel() * 4 < BUF_LIMIT:            # the border-class kernels address tensors with 32-bit buffer offsets
            conv_dgrad(dz3, ctx.bufs.weights(W3, 3, MODE_DGRAD_S1_CLASSES), a2, 3, dz2, variant=5)   # no padding zeros
        else:
            conv_dgrad(dz3, ctx.bufs.weights(W3, 3, MODE_DGRAD_S1), a2, 3, dz2)
        dW2, db2 = conv_wgrad(a1, dz2, 2, out=gout(2))
        if _CONV_Z and a1.numel() * 4 < BUF_LIMIT:
            conv_dgrad_packed(dz2, ctx.bufs.conv_zpack(W2, 2, MODE_DGRAD_S2), a1, 2, dz1, bits=bits[0] if bits else None)
        elif a1.numel() * 4 < BUF_LIMIT:
            conv_dgrad(dz2, ctx.bufs.weights(W2, 2, MODE_DGRAD_S2_CLASSES), a1, 2, dz1, variant=VARIANT_DGRAD2_CLASSES)   # no padding zeros
        else:
            conv_dgrad(dz2, ctx.bufs.weights(W2, 2, MODE_DGRAD_S2), a1, 2, dz1)
        dW1, db1 = conv_wgrad(ctx.obs, dz1, 1, ctx.inds, out=gout(1))
        if direct:
            return (None,) * 10
        return None, None, dW1, db1, dW2, db2, dW3, db3, None, None


class NatureTrunk:
    """Callable wrapper owning the reusable buffers: ``trunk(obs_u8, inds, conv1, conv2, conv3) -> (M, 3136)`` with
    features in (h, w, c) order (use ``fc_weight_hwc`` for the Linear that follows)."""

    def __init__(self):
        self.bufs = _Buffers()

    def __call__(self, obs_u8, inds, conv1, conv2, conv3):
        a3 = NatureTrunkFn.apply(obs_u8, inds, conv1.weight, conv1.bias, conv2.weight, conv2.bias, conv3.weight,
                                 conv3.bias, self.bufs, torch.is_grad_enabled())
        return a3.reshape(a3.shape[0], -1)


def fc_weight_hwc(weight: torch.Tensor) -> torch.Tensor:
    """Linear(64*7*7, 512) weight with its input features re-ordered from the reference's flatten order (c, h, w)
    to the trunk's (h, w, c); differentiable (the gradient flows back into the original layout)."""
    return weight.view(weight.shape[0], 64, 7, 7).permute(0, 2, 3, 1).reshape(weight.shape[0], 64 * 7 * 7)


class LinearReLUHwcFn(torch.autograd.Function):
    """``relu(a @ fc_weight_hwc(W).T + b)`` for the trunk's (h, w, c)-ordered features -- Agent.network[7:9]
    (Linear(3136, 512) + ReLU, cleanrl/ppo_atari_multigpu.py:144-145).  Minibatch-sized batches: forward and data gradient
    on kernel Z (bf16 pipe, pre-split weights; bias + ReLU / the ReLU backward of conv3 in the epilogues, csrc/gemmz.hip), the
    weight gradient on kernel W (bf16 pipe, the batch cut into slabs, csrc/fcw.hip); this layer's own ReLU backward and bias
    gradient ride in ``HeadsFn.backward``.  Rollout-sized calls (no backward) use the library GEMM with the fused epilogue."""

    @staticmethod
    def forward(ctx, a, W, b, bufs=None):
        # on the GPU: kernel Z (csrc/gemmz.hip), bf16 matrix pipe, bias + ReLU in its epilogue -- minibatch-sized batches with
        # whole-K wave tiles, rollout-sized ones with K split over the grid.  The host path (and operands kernel Z cannot take):
        # the library GEMM with the same fused epilogue.
        ctx.fcz = bool(a.is_cuda and bufs is not None and a.shape[0] >= FCZ_MIN_ROWS and a.shape[1] % 16 == 0 and a.is_contiguous()
                       and a.numel() * 4 < BUF_LIMIT)
        ctx.f16 = bool(ctx.fcz and bufs.f16(a))
        if ctx.f16:
            h = fc_fwd_relu_packed(a, bufs.fc_pack_fwd(W), b.detach().contiguous(), W.shape[0], amax=(bufs.rec_of(REC_A3, a), None))
        elif ctx.fcz:
            h = fc_fwd_relu_packed(a, bufs.fc_pack_fwd(W), b.detach().contiguous(), W.shape[0])
        else:
            Wp = bufs.fc_weight(W) if bufs is not None else fc_weight_hwc(W.detach()).contiguous()
            h = torch._addmm_activation(b.detach(), a, Wp.t())                 # bias + ReLU fused into the GEMM epilogue
        ctx.bufs = bufs
        ctx.bias = b
        ctx.save_for_backward(a, h, W)
        return h

    @staticmethod
    def backward(ctx, dh):
        a, h, W = ctx.saved_tensors
        bufs = ctx.bufs
        pre = bufs.fc_dz_from_heads if bufs is not None else None
        db = None
        if pre is not None and pre[0] == dh.data_ptr() and dh.stride(1) == 1 and dh.stride(0) % 4 == 0:
            dz, db = dh, pre[1]                     # HeadsFn.backward already applied this layer's ReLU mask and summed the bias gradient
        else:
            dz = torch.ops.aten.threshold_backward(dh.contiguous(), h, 0.0)
        if bufs is not None:
            bufs.fc_dz_from_heads = None
        fused = bool(ctx.fcz and ctx.needs_input_grad[0] and bufs.last_a3_ptr == a.data_ptr())
        da = None
        if ctx.needs_input_grad[0]:
            if fused:
                # `a` is the trunk's ReLU output a3: its ReLU backward rides in this GEMM's epilogue (kernel Z, Z_MASK) and
                # NatureTrunkFn.backward is told not to mask again -- the separate pass over the 411 MB tensor is gone
                if ctx.f16:
                    da = torch.empty((dz.shape[0], a.shape[1]), dtype=torch.float32, device=dz.device)
                    fc_dgrad_mask_packed(dz, bufs.fc_pack_dgrad(W), a, out=da, bits=bufs.last_a3_bits,
                                         amax=(bufs.rec_of(REC_DH, dz), bufs.owns(REC_DZ3, da)))
                else:
                    da = fc_dgrad_mask_packed(dz, bufs.fc_pack_dgrad(W), a, bits=bufs.last_a3_bits)
                bufs.a3_grad_is_masked = True
            else:
                da = dz @ (bufs.fc_weight(W) if bufs is not None else fc_weight_hwc(W.detach()))
        m, n = dz.shape
        wk = ctx.fcz and n % 64 == 0 and a.shape[1] % 224 == 0 and a.shape[1] % 64 == 0
        direct = bool(wk and bufs is not None and bufs.direct_grads and W.grad is not None and ctx.bias.grad is not None
                      and W.grad.is_contiguous())
        if wk and ctx.f16:
            dW = fc_wgrad(dz, a, 64, out=W.grad if direct else None, amax=(bufs.rec_of(REC_DH, dz), bufs.rec_of(REC_A3, a)))
        elif wk:
            dW = fc_wgrad(dz, a, 64, out=W.grad if direct else None)   # kernel W (bf16 pipe), written in the (c, h, w) feature order of W itself
        else:
            dW = (dz.t() @ a).view(n, 7, 7, 64).permute(0, 3, 1, 2).reshape(n, 64 * 7 * 7)     # back to the (c, h, w) feature order
        if direct:
            ctx.bias.grad.copy_(db if db is not None else dz.sum(0))      # (512 floats)
            if bufs.after_fc_wgrad is not None:
                bufs.after_fc_wgrad(W)
            return da, None, None, None
        return da, dW, (db if db is not None else dz.sum(0)), None


def heads_supported(actor: torch.nn.Linear, critic: torch.nn.Linear) -> bool:
    return actor.in_features == 512 and critic.in_features == 512 and critic.out_features == 1 and 1 <= actor.out_features <= 18


class HeadsFn(torch.autograd.Function):
    """``(actor(h), critic(h))`` -- Agent's two output Linear layers (ppo_atari_multigpu.py:148-149) -- as one
    bandwidth-bound pass over ``h`` forward and one backward (``csrc/heads.hip``) instead of six degenerate GEMMs."""

    @staticmethod
    def forward(ctx, h, Wa, ba, Wc, bc, relu_bufs=None):
        """``relu_bufs``: the trunk's ``_Buffers`` when ``h`` is the output of ``LinearReLUHwcFn`` -- backward then returns the
        gradient with respect to that layer's PRE-activation (its ReLU mask applied here, where h and dh are in registers) and
        leaves its bias gradient in ``relu_bufs.fc_dz_from_heads``."""
        lib = _lib.load()
        ctx.relu_bufs = relu_bufs
        M, H = h.shape
        A = Wa.shape[0]
        h = _chk(h.contiguous(), torch.float32, "h", (M, H))
        _chk(Wa, torch.float32, "actor.weight", (A, H))
        _chk(Wc, torch.float32, "critic.weight", (1, H))
        logits = torch.empty((M, A), dtype=torch.float32, device=h.device)
        value = torch.empty((M, 1), dtype=torch.float32, device=h.device)
        with _on(h.device):
            st = lib.mi355ppo_heads_fwd_f32(_ptr(h), _ptr(Wa), _ptr(ba), _ptr(Wc), _ptr(bc), _ptr(logits), _ptr(value), M, A, H,
                                            _stream(h.device))
        _lib.check(st, "mi355ppo_heads_fwd_f32")
        ctx.save_for_backward(h, Wa, Wc)
        ctx.biases = (ba, bc)
        return logits, value

    @staticmethod
    def backward(ctx, dlogits, dvalue):
        lib = _lib.load()
        h, Wa, Wc = ctx.saved_tensors
        M, H = h.shape
        A = Wa.shape[0]
        dev = h.device
        dlogits = dlogits.contiguous() if dlogits is not None else torch.zeros((M, A), device=dev)
        dvalue = dvalue.contiguous() if dvalue is not None else torch.zeros((M, 1), device=dev)
        bufs = ctx.relu_bufs
        # (ReLU variant: dz with a padded row pitch -- kernel Z's A operand at K = 512, see FC_PAD)
        dh = bufs.fc_dz(M, H, dev) if bufs is not None else torch.empty_like(h)
        ba, bc = ctx.biases
        direct = bool(bufs is not None and bufs.direct_grads and all(p.grad is not None and p.grad.is_contiguous() for p in (Wa, ba, Wc, bc)))
        if direct:                                        # the kernel's reduction writes the gradients where the optimizer reads them
            dWa, dba, dWc, dbc = Wa.grad, ba.grad, Wc.grad, bc.grad
        else:
            dWa, dba = torch.empty_like(Wa), torch.empty(A, dtype=torch.float32, device=dev)
            dWc, dbc = torch.empty_like(Wc), torch.empty(1, dtype=torch.float32, device=dev)
        ws = _workspace(dev, lib.mi355ppo_heads_bwd_workspace_bytes(M, A))
        with _on(dev):
            if bufs is not None and bufs.f16(h) and bufs.has_pass(M, dev):      # the FC layer's gradients will split dz: record max |dz| here
                dbh = torch.empty(H, dtype=torch.float32, device=dev)
                st = lib.mi355ppo_heads_bwd_relu_amax_f32(_ptr(h), _ptr(Wa), _ptr(Wc), _ptr(dlogits), _ptr(dvalue), _ptr(dh), dh.stride(0),
                                                          _ptr(dWa), _ptr(dba), _ptr(dWc), _ptr(dbc), _ptr(dbh), M, A, H, _ptr(ws), ws.numel(),
                                                          _ptr(bufs.owns(REC_DH, dh)), _stream(dev))
                bufs.fc_dz_from_heads = (dh.data_ptr(), dbh)
            elif bufs is not None:
                dbh = torch.empty(H, dtype=torch.float32, device=dev)
                st = lib.mi355ppo_heads_bwd_relu_f32(_ptr(h), _ptr(Wa), _ptr(Wc), _ptr(dlogits), _ptr(dvalue), _ptr(dh), dh.stride(0),
                                                     _ptr(dWa), _ptr(dba), _ptr(dWc), _ptr(dbc), _ptr(dbh), M, A, H, _ptr(ws), ws.numel(),
                                                     _stream(dev))
                bufs.fc_dz_from_heads = (dh.data_ptr(), dbh)
            else:
                st = lib.mi355ppo_heads_bwd_f32(_ptr(h), _ptr(Wa), _ptr(Wc), _ptr(dlogits), _ptr(dvalue), _ptr(dh), _ptr(dWa),
                                                _ptr(dba), _ptr(dWc), _ptr(dbc), M, A, H, _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(st, "mi355ppo_heads_bwd_relu_f32" if bufs is not None else "mi355ppo_heads_bwd_f32")
        if direct:
            return dh, None, None, None, None, None
        return dh, dWa, dba, dWc, dbc, None
