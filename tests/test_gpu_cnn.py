"""Parity of the f32-MFMA NatureCNN kernels (csrc/conv.hip) with a float64 CPU convolution of the reference's
conv stack (cleanrl/ppo_atari_multigpu.py:136-142).  Tolerance: the kernels are f32-in / f32-accumulate, so results
differ from the float64 truth by f32 summation round-off only: |err| <= 2e-5 * max|ref| (K <= 576 terms; measured
~1e-6) -- the same class as torch's own f32 convolution, which is checked against the same bound for calibration."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from cleanrl_amd import cnn, synthetic

DEV = torch.device("cuda:0")
SPEC = {1: (4, 32, 8, 4, 84, 20), 2: (32, 64, 4, 2, 20, 9), 3: (64, 64, 3, 1, 9, 7)}


def _close(got, ref, what, tol=2e-5):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= tol * scale + 1e-30, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / max(scale, 1e-30):.2e})"


def _params(layer, seed):
    cin, cout, k, _, _, _ = SPEC[layer]
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g) * 0.1
    return W, b


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("layer,mode", [(1, 0), (2, 0), (3, 0), (3, 1), (2, 2)])
def test_repack_layouts(layer, mode):
    cin, cout, k, _, _, _ = SPEC[layer]
    W = torch.arange(cout * cin * k * k, dtype=torch.float32).reshape(cout, cin, k, k)   # asymmetric on purpose
    got = cnn.repack_weights(W.to(DEV), layer, mode).cpu()
    if mode == 0:
        ref = W.permute(0, 2, 3, 1).reshape(-1)                                # [cout][(kh,kw,cin)]
        if layer == 1:
            ref = ref * np.float32(1.0 / 255.0)                                # layer 1 consumes raw uint8 taps
    elif mode == 1:
        ref = W.flip(2, 3).permute(1, 2, 3, 0).reshape(-1)                     # [cin][(r,c,cout)], taps flipped
    else:
        parts = []
        for ph in (0, 1):
            for pw in (0, 1):
                sub = W[:, :, [ph + 2, ph], :][:, :, :, [pw + 2, pw]]           # r=0 -> kh=ph+2, r=1 -> kh=ph
                parts.append(sub.permute(1, 2, 3, 0).reshape(-1))
        ref = torch.cat(parts)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("variant", [2, 4])
@pytest.mark.parametrize("images", [1, 37, 256, 1100])
def test_conv1_fwd_u8_gather(images, variant):
    frames = torch.from_numpy(synthetic.atari_frames(images + 5, seed=3))            # (R,4,84,84) uint8
    rows = _nhwc(frames).to(DEV)
    inds = torch.from_numpy(np.random.RandomState(0).randint(0, images + 5, size=images)).to(DEV)
    W, b = _params(1, 1)
    ref = F.relu(F.conv2d(frames[inds.cpu()].double() / 255.0, W.double(), b.double(), stride=4))
    got = cnn.conv_fwd(rows, cnn.repack_weights(W.to(DEV), 1), b.to(DEV), 1, inds, variant=variant)
    _close(got, _nhwc(ref), "conv1 fwd (gather)")
    got2 = cnn.conv_fwd(rows[:images].contiguous(), cnn.repack_weights(W.to(DEV), 1), b.to(DEV), 1, None, variant=variant)
    ref2 = F.relu(F.conv2d(frames[:images].double() / 255.0, W.double(), b.double(), stride=4))
    _close(got2, _nhwc(ref2), "conv1 fwd (identity rows)")
    # calibration: torch's own f32 GPU convolution meets the same bound
    t32 = F.relu(F.conv2d((frames[:images].float() / 255.0).to(DEV), W.to(DEV), b.to(DEV), stride=4))
    _close(_nhwc(t32), _nhwc(ref2), "torch f32 conv1 (calibration)")


def _unpack_q(pack_f32):
    """Decode kernel Q's mode-4 pack back into (q (32,4,8,8) int64, acc0 (4,32), scale (32,)): the inverse of conv1q_pack_kernel."""
    raw = pack_f32.cpu().numpy().view(np.uint8)
    dig = raw[:8 * 4 * 64 * 16].view(np.int8).reshape(8, 4, 2, 32, 16).astype(np.int64)     # [row][digit][lh][n][e]
    q = np.zeros((32, 4, 8, 8), np.int64)
    for r in range(8):
        for lh in range(2):
            for e in range(16):
                e32 = 16 * lh + e
                kw, c = e32 >> 2, e32 & 3
                q[:, c, r, kw] = sum(dig[r, d, lh, :, e] << (8 * (3 - d)) for d in range(4))
    acc0 = raw[32768:32768 + 512].view(np.int32).reshape(4, 32)
    scale = raw[32768 + 512:].view(np.float32)
    return q, dig, acc0, scale


@pytest.mark.parametrize("kind", ["init", "wide", "zero_channel", "tiny"])
def test_conv1q_pack_is_the_weights_to_31_bits(kind):
    """The integer-digit pack of kernel Q decodes to rint(w * 2^(30-E_n)) exactly: every weight is represented to 2^-30 of
    its channel's largest weight (1/64 of that weight's f32 ulp), digits stay in int8, the accumulator start values and the
    scales are what the kernel's algebra needs."""
    W, _ = _params(1, 4)
    if kind == "wide":
        W = W * torch.logspace(-6, 0, 256).reshape(1, 4, 8, 8)            # seven decades inside every channel
    elif kind == "zero_channel":
        W[5] = 0.0
        W[9, :, :, :] = 0.0
        W[9, 0, 0, 0] = -0.75
    elif kind == "tiny":
        W = W * 1e-30
    q, dig, acc0, scale = _unpack_q(cnn.repack_weights(W.to(DEV), 1, cnn.MODE_FWD_Q))
    Wd = W.double().numpy()
    m = np.abs(Wd).reshape(32, -1).max(1)
    E = np.where(m > 0, np.frexp(m)[1], 0)
    want = np.rint(Wd * np.exp2(30.0 - E).reshape(32, 1, 1, 1)).astype(np.int64)
    assert np.array_equal(q, want)
    assert dig.min() >= -128 and dig.max() <= 127 and np.abs(q).max() <= 2 ** 30
    err = np.abs(q * np.exp2(E - 30.0).reshape(32, 1, 1, 1) - Wd).reshape(32, -1).max(1)
    assert np.all(err <= np.exp2(E - 31.0))                   # half a unit of 2^(E-30): <= 2^-30 of the channel's largest weight
    assert np.array_equal(acc0, 128 * dig.sum(axis=(0, 2, 4)).astype(np.int32))           # [digit][channel]
    np.testing.assert_array_equal(scale, (np.exp2(E - 6.0) / 255.0).astype(np.float32))


@pytest.mark.parametrize("images", [1, 37, 256, 1100, 5000])
def test_conv1q_fwd_matches_float64_and_beats_the_f32_pipe(images):
    """Kernel Q (integer matrix pipe) against the float64 convolution of the reference's layer: same bound as the f32-MFMA
    kernels, and its error is not larger than theirs (exact int32 accumulation + 5 roundings vs a 256-term f32 chain)."""
    frames = torch.from_numpy(synthetic.atari_frames(images + 5, seed=3))
    rows = _nhwc(frames).to(DEV)
    inds = torch.from_numpy(np.random.RandomState(0).randint(0, images + 5, size=images)).to(DEV)
    W, b = _params(1, 1)
    ref = _nhwc(F.relu(F.conv2d(frames[inds.cpu()].double() / 255.0, W.double(), b.double(), stride=4)))
    pack = cnn.repack_weights(W.to(DEV), 1, cnn.MODE_FWD_Q)
    got = cnn.conv_fwd(rows, pack, b.to(DEV), 1, inds, variant=cnn.VARIANT_Q)
    _close(got, ref, "conv1 fwd, kernel Q (gather)")
    f32 = cnn.conv_fwd(rows, cnn.repack_weights(W.to(DEV), 1), b.to(DEV), 1, inds, variant=2)
    e_q = (got.cpu().double() - ref).abs()
    e_f = (f32.cpu().double() - ref).abs()
    assert e_q.max() <= e_f.max() * 1.05 + 1e-12 and e_q.mean() <= e_f.mean() * 1.05 + 1e-12, \
        f"kernel Q err max {e_q.max():.3e} mean {e_q.mean():.3e} vs f32-MFMA max {e_f.max():.3e} mean {e_f.mean():.3e}"
    # identity rows, bit-identical run to run, inference path of the trunk call
    got2 = cnn.conv_fwd(rows[:images].contiguous(), pack, b.to(DEV), 1, None, variant=cnn.VARIANT_Q)
    ref2 = _nhwc(F.relu(F.conv2d(frames[:images].double() / 255.0, W.double(), b.double(), stride=4)))
    _close(got2, ref2, "conv1 fwd, kernel Q (identity rows)")
    assert torch.equal(got, cnn.conv_fwd(rows, pack, b.to(DEV), 1, inds, variant=cnn.VARIANT_Q))


def test_conv1q_extreme_inputs_and_weights():
    """All-0 / all-255 frames (the v - 128 offset at both ends), weights spanning seven decades, a zero channel, a large
    bias: still within the f32 bound of the float64 result."""
    frames = torch.zeros(6, 4, 84, 84, dtype=torch.uint8)
    frames[1] = 255
    frames[2, :, ::2] = 255
    frames[3:] = torch.from_numpy(synthetic.atari_frames(3, seed=5))
    W, b = _params(1, 7)
    W = W * torch.logspace(-6, 0, 256).reshape(1, 4, 8, 8)[:, :, torch.randperm(8)]
    W[3] = 0.0
    b[4] = 50.0
    ref = _nhwc(F.relu(F.conv2d(frames.double() / 255.0, W.double(), b.double(), stride=4)))
    got = cnn.conv_fwd(_nhwc(frames).to(DEV), cnn.repack_weights(W.to(DEV), 1, cnn.MODE_FWD_Q), b.to(DEV), 1, None, variant=cnn.VARIANT_Q)
    _close(got, ref, "conv1 fwd, kernel Q (extremes)", tol=1e-6)
    assert torch.equal(got[..., 3].cpu(), torch.relu(b[3]).expand(6, 20, 20))              # the zero channel is exactly its bias


@pytest.mark.parametrize("variant", [2, 4])
@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 19, 128, 700, 7000])
def test_conv_fwd_f32(layer, images, variant):
    if images == 7000 and variant != 2:
        pytest.skip("the 7000-image case (64-pixel-tile launch path) is checked on the default kernel only: its float64 "
                    "CPU reference dominates the suite's run time")
    cin, cout, k, s, hin, hout = SPEC[layer]
    g = torch.Generator().manual_seed(10 + layer)
    x = torch.relu(torch.randn(images, cin, hin, hin, generator=g))
    W, b = _params(layer, 2)
    ref = F.relu(F.conv2d(x.double(), W.double(), b.double(), stride=s))
    got = cnn.conv_fwd(_nhwc(x).to(DEV), cnn.repack_weights(W.to(DEV), layer), b.to(DEV), layer, variant=variant)
    assert got.shape == (images, hout, hout, cout)
    _close(got, _nhwc(ref), f"conv{layer} fwd")


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 19, 700, 2049, 7000])
def test_conv_fwd_kernel_z(layer, images):
    """Kernel Z (csrc/gemmz.hip) on the forward of layers 2 / 3: output pixels as GEMM rows, coalesced window loads through the
    wave-private LDS transposition, weights pre-split into fragment order -- the float64 bound of every forward kernel (2e-5 of
    the result's scale) on activations with a wide dynamic range, next to kernel F on the same inputs; deterministic; tails
    (1 and 19 images: partial 64-pixel row blocks).  From 2,048 images on the workgroup shares B through its LDS ring: 2,049
    images leave the last workgroup with waves past the batch that still take part in the ring."""
    cin, cout, k, s, hin, hout = SPEC[layer]
    g = torch.Generator().manual_seed(400 + layer + images)
    x = torch.relu(torch.randn(images, cin, hin, hin, generator=g)) * torch.exp(torch.randn(images, cin, hin, hin, generator=g))
    W, b = _params(layer, 4)
    ref = F.relu(F.conv2d(x.double(), W.double(), b.double(), stride=s))
    pack = cnn.conv_zpack(W.to(DEV), layer, cnn.MODE_FWD)
    xd = _nhwc(x).to(DEV)
    got = cnn.conv_fwd_packed(xd, pack, b.to(DEV), layer)
    assert got.shape == (images, hout, hout, cout)
    _close(got, _nhwc(ref), f"conv{layer} fwd (kernel Z)")
    f32 = cnn.conv_fwd(xd, cnn.repack_weights(W.to(DEV), layer), b.to(DEV), layer)
    _close(got, f32, f"conv{layer} fwd: kernel Z vs kernel F", tol=4e-6)
    assert torch.equal(got, cnn.conv_fwd_packed(xd, pack, b.to(DEV), layer))


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 19, 128, 700, 2049, 2310, 7000])
def test_conv_dgrad_kernel_z_with_relu_mask(layer, images):
    """Kernel Z on the data gradients: layer 3 as the zero-padded full correlation of dz3 with the flipped taps (padding = the
    buffer range check), layer 2 as ONE 128-column GEMM over the 10 x 10 grid whose four column tiles are the stride-parity
    classes; ReLU-backward mask fused.  float64 bound of every data-gradient kernel; exact zeros where the activation is 0.
    From 2,048 images on (B ring) a workgroup's four waves take one class tile of four consecutive 64-image groups: 2,049 and
    2,310 images leave the last workgroups with one and one-and-a-bit live groups."""
    cin, cout, k, s, hin, hout = SPEC[layer]
    g = torch.Generator().manual_seed(420 + layer)
    pre = torch.randn(images, cin, hin, hin, generator=g).double().requires_grad_(True)
    act = torch.relu(pre)
    W, _ = _params(layer, 3)
    dz = torch.randn(images, cout, hout, hout, generator=g) * torch.exp(torch.randn(images, cout, hout, hout, generator=g))
    out = F.conv2d(act, W.double(), None, stride=s)
    (ref,) = torch.autograd.grad(out, pre, dz.double())
    mode = cnn.MODE_DGRAD_S1 if layer == 3 else cnn.MODE_DGRAD_S2
    actd = _nhwc(act.detach().float()).to(DEV)
    got = cnn.conv_dgrad_packed(_nhwc(dz).to(DEV), cnn.conv_zpack(W.to(DEV), layer, mode), actd, layer)
    _close(got, _nhwc(ref), f"conv{layer} dgrad (kernel Z)")
    assert ((actd > 0) | (got == 0)).all()
    assert torch.equal(got, cnn.conv_dgrad_packed(_nhwc(dz).to(DEV), cnn.conv_zpack(W.to(DEV), layer, mode), actd, layer))


@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 19, 128, 700, 7000])
def test_conv_dgrad_with_relu_mask(layer, images, variant):
    if variant == 3 and layer != 2:
        pytest.skip("variant 3 (one launch per stride-parity class) is the layer-2 data gradient")
    if variant == 5 and layer != 3:
        pytest.skip("variant 5 (border classes) is the layer-3 data gradient")
    if variant == 6 and layer != 2:
        pytest.skip("variant 6 (border classes) is the layer-2 data gradient")
    if images == 7000 and variant not in (2, 5, 6):
        pytest.skip("the 7000-image case is checked on the default kernels only (CPU reference time)")
    cin, cout, k, s, hin, hout = SPEC[layer]
    g = torch.Generator().manual_seed(20 + layer)
    pre = torch.randn(images, cin, hin, hin, generator=g).double().requires_grad_(True)
    act = torch.relu(pre)                                                            # the layer's input activation
    W, _ = _params(layer, 3)
    dz = torch.randn(images, cout, hout, hout, generator=g)
    out = F.conv2d(act, W.double(), None, stride=s)
    (ref,) = torch.autograd.grad(out, pre, dz.double())                              # conv_transpose * (act > 0)
    mode = ((cnn.MODE_DGRAD_S1_CLASSES if variant == 5 else cnn.MODE_DGRAD_S1) if layer == 3
            else (cnn.MODE_DGRAD_S2_CLASSES if variant == 6 else cnn.MODE_DGRAD_S2))
    got = cnn.conv_dgrad(_nhwc(dz).to(DEV), cnn.repack_weights(W.to(DEV), layer, mode),
                         _nhwc(act.detach().float()).to(DEV), layer, variant=variant)
    _close(got, _nhwc(ref), f"conv{layer} dgrad")


@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("images", [1, 7, 600, 608, 2064])
def test_conv_wgrad(layer, images):
    # 1 / 7 / 600 images: kernels P (layer 1) and T (layers 2, 3).  608 / 2,064 images (multiples of 16, enough 16-pixel blocks for
    # every slab): layers 2 / 3 run on kernel V (bf16 pipe, convw.hip) -- slabs with uneven step counts, image boundaries inside
    # the 16-pixel blocks, all four / six wave groups
    cin, cout, k, s, hin, hout = SPEC[layer]
    g = torch.Generator().manual_seed(30 + layer)
    W, b = _params(layer, 4)
    Wd, bd = W.double().requires_grad_(True), b.double().requires_grad_(True)
    dz = torch.randn(images, cout, hout, hout, generator=g)
    if layer == 1:
        frames = torch.from_numpy(synthetic.atari_frames(images + 3, seed=5))
        inds = torch.from_numpy(np.random.RandomState(1).randint(0, images + 3, size=images))
        x = frames[inds].double() / 255.0
        src, idx = _nhwc(frames).to(DEV), inds.to(DEV)
    else:
        x = torch.relu(torch.randn(images, cin, hin, hin, generator=g)).double()
        src, idx = _nhwc(x.float()).to(DEV), None
    out = F.conv2d(x, Wd, bd, stride=s)
    refW, refb = torch.autograd.grad(out, (Wd, bd), dz.double())
    dW, db = cnn.conv_wgrad(src, _nhwc(dz).to(DEV), layer, idx)
    _close(dW, refW, f"conv{layer} wgrad dW")
    _close(db, refb, f"conv{layer} wgrad db")
    dW2, db2 = cnn.conv_wgrad(src, _nhwc(dz).to(DEV), layer, idx)
    assert torch.equal(dW, dW2) and torch.equal(db, db2), "weight gradient must be deterministic"


@pytest.mark.parametrize("layer", [1, 2, 3])
def test_conv_wgrad_full_minibatch_against_float64_on_a_strided_slab(layer):
    """Kernels R (layer 1) and T (layers 2, 3) at the config-C minibatch size, where every workgroup / partial / image-pair
    combination of the 512-partial workspace is in play: dz is non-zero on a slab of 512 images strided through the whole
    batch (every 64th image), so the float64 weight gradient of those 512 images IS the exact answer for the 32,768-image
    launch; the other 32,256 images still stream through the kernel (their products are exact zeros)."""
    M, step = 32768, 64
    cin, cout, k, s, hin, hout = SPEC[layer]
    g = torch.Generator().manual_seed(70 + layer)
    slab = torch.arange(step // 2 - 1, M, step)                                   # 512 images, odd and even image parities
    dz_slab = torch.randn(len(slab), cout, hout, hout, generator=g)
    dz = torch.zeros(M, hout, hout, cout, device=DEV)
    dz[slab.to(DEV)] = _nhwc(dz_slab).to(DEV)
    gd = torch.Generator(device=DEV).manual_seed(71)
    if layer == 1:
        obs = torch.randint(0, 256, (M + 100, 84, 84, 4), dtype=torch.uint8, device=DEV, generator=gd)
        inds = torch.randperm(M + 100, device=DEV, generator=gd)[:M]
        x_slab = obs[inds[slab.to(DEV)]].cpu().permute(0, 3, 1, 2).double() / 255.0
        dW, db = cnn.conv_wgrad(obs, dz, 1, inds)
    else:
        src = torch.relu(torch.randn(M, hin, hin, cin, device=DEV, generator=gd))
        x_slab = src[slab.to(DEV)].cpu().permute(0, 3, 1, 2).double()
        dW, db = cnn.conv_wgrad(src, dz, layer)
    Wd = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    bd = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    refW, refb = torch.autograd.grad(F.conv2d(x_slab, Wd, bd, stride=s), (Wd, bd), dz_slab.double())
    _close(dW, refW, f"conv{layer} wgrad dW at M=32768 (512-image slab)")
    _close(db, refb, f"conv{layer} wgrad db at M=32768 (512-image slab)")


@pytest.mark.parametrize("scale", [1e-15, 1.0, 1e10])
def test_conv1_wgrad_zero_extended_frame_operand_over_the_gradient_range(scale):
    """Kernel P feeds the uint8 frames to the bf16 pipe as SUBNORMAL bf16 (0x00vv = v * 2^-133) and pre-scales dz by 2^80 (csrc/conv1p.hip):
    a dense gradient of magnitude 1e-15 / 1 / 1e+10 against float64 -- the accumulators (2^-53 x the sums) must neither underflow nor
    overflow; same bar as every f32 kernel (2e-5 of the result's scale; measured 4.5e-7)."""
    M = 512
    g = torch.Generator(device=DEV).manual_seed(9)
    obs = torch.randint(0, 256, (M, 84, 84, 4), dtype=torch.uint8, device=DEV, generator=g)
    obs[:4] = 255
    obs[4:8] = 0
    obs[8:12] = 127                                                # the largest subnormal pattern ...
    obs[12:16] = 128                                               # ... and the smallest normal one
    dz = torch.randn(M, 20, 20, 32, device=DEV, generator=g) * scale
    dW, db = cnn.conv_wgrad(obs, dz, 1, None)
    Wd = torch.zeros(32, 4, 8, 8, dtype=torch.float64, device=DEV, requires_grad=True)
    bd = torch.zeros(32, dtype=torch.float64, device=DEV, requires_grad=True)
    refW, refb = torch.autograd.grad(F.conv2d(obs.permute(0, 3, 1, 2).double() / 255.0, Wd, bd, stride=4), (Wd, bd), dz.permute(0, 3, 1, 2).double())
    assert torch.isfinite(dW).all()
    _close(dW, refW, f"conv1 wgrad dW at gradient scale {scale:g}")
    _close(db, refb, f"conv1 wgrad db at gradient scale {scale:g}")


@pytest.mark.parametrize("images", [48, 4096])
def test_trunk_matches_reference_network_forward_backward(images):
    """The whole conv stack + Linear(3136,512) against the reference's nn.Sequential in float64 (same weights)."""
    torch.manual_seed(0)
    frames = torch.from_numpy(synthetic.atari_frames(images, seed=11))
    net = torch.nn.Sequential(torch.nn.Conv2d(4, 32, 8, stride=4), torch.nn.ReLU(), torch.nn.Conv2d(32, 64, 4, stride=2),
                              torch.nn.ReLU(), torch.nn.Conv2d(64, 64, 3, stride=1), torch.nn.ReLU(), torch.nn.Flatten(),
                              torch.nn.Linear(64 * 7 * 7, 512), torch.nn.ReLU())
    import copy

    ref_net = copy.deepcopy(net).double()
    gout = torch.randn(images, 512)
    ref_h = ref_net(frames.double() / 255.0)
    ref_h.backward(gout.double())

    gnet = copy.deepcopy(net).to(DEV)
    trunk = cnn.NatureTrunk()
    feats = trunk(_nhwc(frames).to(DEV), None, gnet[0], gnet[2], gnet[4])
    h = cnn.LinearReLUHwcFn.apply(feats, gnet[7].weight, gnet[7].bias)
    _close(h, ref_h, "trunk+fc forward", tol=5e-5)
    h.backward(gout.to(DEV))
    # calibration: torch's own f32 network on the same device.  Gradients of parameters are sums over images x pixels
    # (1.6 M f32 terms for conv1 at 4096 images), so the bound is the larger of 5e-5 of the gradient's scale and
    # 4x the error torch's f32 backward shows against the same float64 reference.
    tnet = copy.deepcopy(net).to(DEV)
    th = tnet((frames.float() / 255.0).to(DEV))
    th.backward(gout.to(DEV))
    for (name, p), (_, pr), (_, pt) in zip(gnet.named_parameters(), ref_net.named_parameters(), tnet.named_parameters()):
        scale = pr.grad.abs().max().item()
        err = (p.grad.cpu().double() - pr.grad).abs().max().item()
        err_torch = (pt.grad.cpu().double() - pr.grad).abs().max().item()
        assert err <= max(5e-5 * scale, 4.0 * err_torch), f"grad of {name}: err {err:.3e}, torch f32 err {err_torch:.3e}, scale {scale:.3e}"


@pytest.mark.parametrize("M,A", [(1, 4), (37, 4), (1024, 6), (4096, 1), (32768, 4), (300, 7), (5, 8), (1024, 9), (300, 13), (4100, 18), (32768, 18)])      # (A > 7, round 6: the weight rows in LDS, one instantiation for 8 .. 18 actions)
def test_heads_forward_backward(M, A):
    """actor + critic heads (ppo_atari_multigpu.py:148-149) as one pass each way vs float64 Linear layers; the weight
    gradients are sums over M rows, so their bound scales like the f32 error of torch's own GEMM (calibrated)."""
    g = torch.Generator().manual_seed(M + A)
    h = torch.relu(torch.randn(M, 512, generator=g))
    actor, critic = torch.nn.Linear(512, A), torch.nn.Linear(512, 1)
    gl, gv = torch.randn(M, A, generator=g), torch.randn(M, 1, generator=g)
    import copy
    a64, c64 = copy.deepcopy(actor).double(), copy.deepcopy(critic).double()
    h64 = h.double().requires_grad_(True)
    l_ref, v_ref = a64(h64), c64(h64)
    torch.autograd.backward([l_ref, v_ref], [gl.double(), gv.double()])
    ad, cd = copy.deepcopy(actor).to(DEV), copy.deepcopy(critic).to(DEV)
    hd = h.to(DEV).requires_grad_(True)
    assert cnn.heads_supported(ad, cd)
    logits, value = cnn.HeadsFn.apply(hd, ad.weight, ad.bias, cd.weight, cd.bias)
    _close(logits, l_ref, "logits", tol=1e-5)
    _close(value, v_ref, "value", tol=1e-5)
    torch.autograd.backward([logits, value], [gl.to(DEV), gv.to(DEV)])
    _close(hd.grad, h64.grad, "dh", tol=1e-5)
    at, ct = copy.deepcopy(actor).to(DEV), copy.deepcopy(critic).to(DEV)          # torch f32 on the device, for calibration
    ht = h.to(DEV)
    torch.autograd.backward([at(ht), ct(ht)], [gl.to(DEV), gv.to(DEV)])
    for name, got, ref, cal in (("dWa", ad.weight.grad, a64.weight.grad, at.weight.grad), ("dba", ad.bias.grad, a64.bias.grad, at.bias.grad),
                                ("dWc", cd.weight.grad, c64.weight.grad, ct.weight.grad), ("dbc", cd.bias.grad, c64.bias.grad, ct.bias.grad)):
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        err_t = (cal.cpu().double() - ref).abs().max().item()
        assert err <= max(2e-5 * scale, 4.0 * err_t), f"{name}: err {err:.3e}, torch f32 err {err_t:.3e}, scale {scale:.3e}"


def test_all_weight_packs_in_one_launch_are_bit_identical_to_the_per_matrix_route():
    """``mi355ppo_nature_packs_f32`` (through ``_Buffers._repack_all``) against repack_weights + fc_pack per matrix, the route it replaces;
    then the cache follows ``weights_version`` and rebuilds into the SAME buffers (captured launches keep their addresses)."""
    g = torch.Generator().manual_seed(5)
    W1, W2, W3 = (torch.randn(*s, generator=g).to(DEV) * 0.05 for s in ((32, 4, 8, 8), (64, 32, 4, 4), (64, 64, 3, 3)))
    Wfc = (torch.randn(512, 3136, generator=g) * 0.02).to(DEV)
    W1[3] = 0.0                                                   # a dead channel (kernel Q's per-channel exponent)
    ref = {(1, cnn.MODE_FWD_Q): cnn.repack_weights(W1, 1, cnn.MODE_FWD_Q),
           ("zpack", 2, cnn.MODE_FWD): cnn.conv_zpack(W2, 2, cnn.MODE_FWD), ("zpack", 3, cnn.MODE_FWD): cnn.conv_zpack(W3, 3, cnn.MODE_FWD),
           ("zpack", 3, cnn.MODE_DGRAD_S1): cnn.conv_zpack(W3, 3, cnn.MODE_DGRAD_S1),
           ("zpack", 2, cnn.MODE_DGRAD_S2): cnn.conv_zpack(W2, 2, cnn.MODE_DGRAD_S2),
           "fc_pack_fwd": cnn.fc_pack(cnn.fc_weight_hwc(Wfc).contiguous()), "fc_pack_dgrad": cnn.fc_pack(cnn.fc_weight_hwc(Wfc).t().contiguous())}
    bufs = cnn._Buffers()
    bufs.split = "bf16x3"                                         # (the f16x2 twin of this test: tests/test_gpu_f16x2.py)
    bufs.cache_weights = True
    bufs.pack_params = (W1, W2, W3, Wfc)
    got = bufs.conv_zpack(W2, 2, cnn.MODE_FWD)                    # one stale pack -> all of them
    assert set(bufs._bt) >= set(ref)
    for k, want in ref.items():
        assert torch.equal(bufs._bt[k][1].view(torch.uint8), want.view(torch.uint8)), f"pack {k} differs from the per-matrix route"
    assert got.data_ptr() == bufs._bt[("zpack", 2, cnn.MODE_FWD)][1].data_ptr()
    ptrs = {k: bufs._bt[k][1].data_ptr() for k in ref}
    assert bufs.fc_pack_dgrad(Wfc).data_ptr() == ptrs["fc_pack_dgrad"] and bufs.weights(W1, 1, cnn.MODE_FWD_Q).data_ptr() == ptrs[(1, cnn.MODE_FWD_Q)]
    with torch.no_grad():                                         # the optimizer kernel writes through raw pointers, then bumps the version
        for w in (W1, W2, W3, Wfc):
            w.view(-1)[::7] *= 1.5
    torch.cuda.synchronize()
    bufs.weights_version += 1
    q2 = bufs.weights(W1, 1, cnn.MODE_FWD_Q)
    assert {k: bufs._bt[k][1].data_ptr() for k in ref} == ptrs
    assert torch.equal(q2.view(torch.uint8), cnn.repack_weights(W1, 1, cnn.MODE_FWD_Q).view(torch.uint8))
    assert torch.equal(bufs._bt["fc_pack_fwd"][1], cnn.fc_pack(cnn.fc_weight_hwc(Wfc).contiguous()))
    assert torch.equal(bufs._bt[("zpack", 2, cnn.MODE_DGRAD_S2)][1], cnn.conv_zpack(W2, 2, cnn.MODE_DGRAD_S2))


@pytest.mark.parametrize("M,A,philox", [(1024, 4, True), (1024, 4, False), (256, 6, True), (100, 7, True), (8, 1, True), (3000, 4, True)])
def test_fused_rollout_step_behind_the_trunk_is_bit_identical_to_its_four_launches(M, A, philox):
    """``fc_heads_act_categorical`` (K-split FC partials, then ONE kernel for fold + bias + ReLU + heads + Categorical draw) against
    the launches it replaces: ``fc_fwd_relu_packed`` -> ``HeadsFn`` -> ``ops.categorical_sample`` (+ the value copy)."""
    from cleanrl_amd import ops

    g = torch.Generator().manual_seed(7 * M + A)
    a = torch.relu(torch.randn(M, 3136, generator=g)).to(DEV)
    fc, actor, critic = torch.nn.Linear(3136, 512).to(DEV), torch.nn.Linear(512, A).to(DEV), torch.nn.Linear(512, 1).to(DEV)
    with torch.no_grad():
        actor.weight.mul_(30.0)                                     # spread the logits so that the draw is not near-uniform
    pack = cnn.fc_pack(fc.weight.detach())
    assert cnn.fc_heads_act_supported(a)
    noise = None if philox else torch.empty(M, A, device=DEV).exponential_(1.0, generator=torch.Generator(DEV).manual_seed(3))
    base = torch.full((1,), 1000, dtype=torch.int64, device=DEV) if philox else None
    with torch.no_grad():
        h = cnn.fc_fwd_relu_packed(a, pack, fc.bias.detach(), 512)
        logits, value = cnn.HeadsFn.apply(h, actor.weight, actor.bias, critic.weight, critic.bias)
        a64, af, lp, _ = ops.categorical_sample(logits.contiguous(), noise_exp1=noise, seed=11, offset=5, offset_base=base, want_entropy=False)
        out_a, out_lp, out_v = (torch.full((M,), -7.0, device=DEV) for _ in range(3))
        b64, bf, blp, bv = cnn.fc_heads_act_categorical(a, pack, fc.bias.detach(), actor.weight.detach(), actor.bias.detach(), critic.weight.detach(),
                                                        critic.bias.detach(), 11, 5, base, out_a, out_lp, out_v, noise_exp1=noise)
    assert bf is out_a and blp is out_lp and bv is out_v
    assert torch.equal(b64, a64) and torch.equal(bf, a64.float())
    assert torch.equal(blp, lp), f"log-probs differ: {(blp - lp).abs().max().item():.3e}"
    assert torch.equal(bv, value.view(-1)), f"values differ: {(bv - value.view(-1)).abs().max().item():.3e}"
    if A > 1:
        assert len(torch.unique(a64)) > 1


@pytest.mark.parametrize("M", [1, 700, 4099, 8192])
def test_fc_weight_gradient_kernel_y_against_float64(M):
    """Kernel Y (csrc/fcw.hip): dW = dz^T a of Linear(3136, 512) on the f32 pipe, the batch cut into slabs (1 at M <= 1023,
    8 at 4099 with an odd last slab, 9 from 4608) added in a fixed order, against float64; the library's f32 GEMM on the
    same inputs calibrates the bound (both accumulate M exact products in f32).  With the channel count given the columns
    come out in the reference's (c, h, w) order."""
    g = torch.Generator().manual_seed(170 + M)
    a = torch.relu(torch.randn(M, 3136, generator=g)) * torch.exp(torch.randn(M, 3136, generator=g))
    dz = torch.randn(M, 512, generator=g) * (torch.rand(M, 512, generator=g) > 0.4)
    ref = dz.double().t() @ a.double()
    cal = (dz.to(DEV).t() @ a.to(DEV)).cpu().double()
    got = cnn.fc_wgrad(dz.to(DEV), a.to(DEV))
    scale = ref.abs().max().item()
    err, err_t = (got.cpu().double() - ref).abs().max().item(), (cal - ref).abs().max().item()
    assert err <= max(2e-5 * scale, 4.0 * err_t), f"dW: err {err:.3e}, library f32 err {err_t:.3e}, scale {scale:.3e}"
    dz_p = torch.empty((M, 516), device=DEV)[:, :512]                       # the learner's padded row pitch: same bits
    dz_p.copy_(dz)
    assert torch.equal(cnn.fc_wgrad(dz_p, a.to(DEV)), got)
    chw = cnn.fc_wgrad(dz.to(DEV), a.to(DEV), 64)                          # (hw, c) -> (c, hw) columns
    assert torch.equal(chw, got.view(512, 49, 64).permute(0, 2, 1).reshape(512, 3136))
    assert torch.equal(got, cnn.fc_wgrad(dz.to(DEV), a.to(DEV)))            # deterministic


@pytest.mark.parametrize("M,A", [(1, 4), (4100, 6), (32768, 4), (37, 9), (4100, 18)])
def test_heads_backward_relu_variant(M, A):
    """The ReLU variant of the heads' backward: the gradient it writes is the plain one times (h > 0), bit for bit, with a
    padded row pitch; the extra output is the column sum of that (the FC layer's bias gradient)."""
    lib = cnn._lib.load()
    g = torch.Generator().manual_seed(7 * M + A)
    h = torch.relu(torch.randn(M, 512, generator=g)).to(DEV)
    Wa, Wc = (torch.randn(A, 512, generator=g) * 0.05).to(DEV), (torch.randn(1, 512, generator=g) * 0.05).to(DEV)
    gl, gv = torch.randn(M, A, generator=g).to(DEV), torch.randn(M, 1, generator=g).to(DEV)
    ws = torch.empty(lib.mi355ppo_heads_bwd_workspace_bytes(M, A), dtype=torch.uint8, device=DEV)
    P = cnn._ptr
    outs = {}
    for relu in (False, True):
        dh = torch.full((M, 516), 7.0, device=DEV)[:, :512] if relu else torch.empty((M, 512), device=DEV)
        dWa, dba, dWc, dbc, dbh = (torch.empty_like(Wa), torch.empty(A, device=DEV), torch.empty_like(Wc), torch.empty(1, device=DEV),
                                   torch.empty(512, device=DEV))
        with torch.cuda.device(DEV):
            st = (lib.mi355ppo_heads_bwd_relu_f32(P(h), P(Wa), P(Wc), P(gl), P(gv), P(dh), 516, P(dWa), P(dba), P(dWc), P(dbc), P(dbh), M, A,
                                                  512, P(ws), ws.numel(), None) if relu else
                  lib.mi355ppo_heads_bwd_f32(P(h), P(Wa), P(Wc), P(gl), P(gv), P(dh), P(dWa), P(dba), P(dWc), P(dbc), M, A, 512, P(ws),
                                             ws.numel(), None))
        assert st == 0, lib.mi355ppo_last_error()
        outs[relu] = (dh, dWa, dba, dWc, dbc, dbh)
    plain, fused = outs[False], outs[True]
    assert torch.equal(fused[0], plain[0] * (h > 0))
    for k in range(1, 5):
        assert torch.equal(fused[k], plain[k])                               # the head gradients do not change
    ref = fused[0].double().sum(0)
    assert (fused[5].double() - ref).abs().max().item() <= 2e-6 * max(ref.abs().max().item(), fused[0].abs().sum(0).max().item())
    assert torch.all(fused[0].as_strided((M, 4), (516, 1), fused[0].storage_offset() + 512) == 7.0)       # the padding is not written


def test_fc_kernels_at_the_full_minibatch_size_against_float64_on_the_device():
    """Config-C minibatch (32,768 rows): kernels Z (forward, data gradient + mask) and W (weight gradient, batch slabs) against
    torch's float64 GEMMs run on the GPU -- the CPU cannot produce a 105-GFLOP float64 reference in test time, the device can.
    Same bounds as at the small sizes; the library's f32 GEMM calibrates the weight-gradient bound (32,768 f32 accumulations)."""
    M = 32768
    g = torch.Generator(device=DEV).manual_seed(77)
    a = torch.relu(torch.randn(M, 3136, device=DEV, generator=g)) * torch.exp(torch.randn(M, 3136, device=DEV, generator=g))
    W = torch.randn(512, 3136, device=DEV, generator=g) / 56.0
    b = torch.randn(512, device=DEV, generator=g) * 0.1
    dz = torch.randn(M, 512, device=DEV, generator=g) * (torch.rand(M, 512, device=DEV, generator=g) > 0.4)
    a64, W64, dz64 = a.double(), W.double(), dz.double()
    _close(cnn.fc_fwd_relu_packed(a, cnn.fc_pack(W), b, 512), torch.relu(a64 @ W64.t() + b.double()), "fc fwd (kernel Z) at 32768")
    Wt = torch.empty((3136, 516), device=DEV)[:, :512]
    Wt.copy_(W.t())
    _close(cnn.fc_dgrad_mask_packed(dz, cnn.fc_pack(Wt), a), (dz64 @ W64) * (a > 0), "fc dgrad + mask (kernel Z) at 32768")
    ref = dz64.t() @ a64
    got = cnn.fc_wgrad(dz, a)
    scale = ref.abs().max().item()
    err, err_t = (got.double() - ref).abs().max().item(), ((dz.t() @ a).double() - ref).abs().max().item()
    assert err <= max(2e-5 * scale, 4.0 * err_t), f"dW at 32768: err {err:.3e}, library f32 err {err_t:.3e}, scale {scale:.3e}"


def _conv64_nhwc(x_nhwc, W, b, stride):
    """float64 convolution ON THE DEVICE as im2col (F.unfold) + one float64 GEMM: (B,H,W,C) -> (B,Ho,Wo,Cout), no ReLU."""
    B, H, _, C = x_nhwc.shape
    cout, cin, k, _ = W.shape
    ho = (H - k) // stride + 1
    cols = F.unfold(x_nhwc.permute(0, 3, 1, 2), kernel_size=k, stride=stride)            # (B, C*k*k, Ho*Wo), (c, kh, kw) order
    y = torch.einsum("nk,bkl->bln", W.reshape(cout, -1), cols)
    if b is not None:
        y = y + b
    return y.reshape(B, ho, ho, cout)


def _slab(M, n=2048):
    """Image indices of a float64 slab strided through a batch of M images, including both ends of the batch."""
    idx = torch.arange(0, M, max(1, M // n))[:n]
    return torch.unique(torch.cat([idx, torch.arange(0, min(64, M)), torch.arange(max(0, M - 64), M)])).to(DEV)


def test_conv_forward_and_data_gradient_kernels_at_the_full_minibatch_size_against_float64_on_the_device():
    """Config-C minibatch (32,768 images), every DEFAULT convolution kernel of the forward and data-gradient path at its bench
    launch size, against a float64 convolution computed on the device (im2col + float64 GEMM; autograd of it for the data
    gradients) on a slab of ~2,100 images strided through the batch with both ends included:
      kernel Q (layer-1 forward through the row gather), kernel F (layer-2 / layer-3 forward), the layer-3 data gradient per
      border class (variant 5, nine launches) and the layer-2 data gradient per border class (variant 6, four launches), each
      with its fused ReLU-backward mask.  Bound: 2e-5 of the reference's scale (the bound of the small-size float64 tests).
    Reference layers: ppo_atari_multigpu.py:136-142."""
    M = 32768
    g = torch.Generator(device=DEV).manual_seed(11)
    sl = _slab(M)
    params = {l: tuple(t.to(DEV) for t in _params(l, 70 + l)) for l in (1, 2, 3)}
    # ---- layer 1 (kernel Q): uint8 rows through a permutation
    obs = torch.randint(0, 256, (M, 84, 84, 4), dtype=torch.uint8, device=DEV, generator=g)
    inds = torch.randperm(M, device=DEV, generator=g)
    W1, b1 = params[1]
    a1 = cnn.conv_fwd(obs, cnn.repack_weights(W1, 1, cnn.MODE_FWD_Q), b1, 1, inds, variant=cnn.VARIANT_Q)
    ref = torch.relu(_conv64_nhwc(obs[inds[sl]].double() / 255.0, W1.double(), b1.double(), 4))
    _close(a1[sl], ref, "conv1 fwd (kernel Q) at 32768 images")
    del obs, ref
    # ---- layers 2 and 3 forward (kernel F) on the activations the previous kernel produced
    W2, b2 = params[2]
    a2 = cnn.conv_fwd(a1, cnn.repack_weights(W2, 2), b2, 2)
    _close(a2[sl], torch.relu(_conv64_nhwc(a1[sl].double(), W2.double(), b2.double(), 2)), "conv2 fwd (kernel F) at 32768 images")
    z2 = cnn.conv_fwd_packed(a1, cnn.conv_zpack(W2, 2, cnn.MODE_FWD), b2, 2)
    _close(z2[sl], torch.relu(_conv64_nhwc(a1[sl].double(), W2.double(), b2.double(), 2)), "conv2 fwd (kernel Z) at 32768 images")
    del z2
    W3, b3 = params[3]
    a3 = cnn.conv_fwd(a2, cnn.repack_weights(W3, 3), b3, 3)
    _close(a3[sl], torch.relu(_conv64_nhwc(a2[sl].double(), W3.double(), b3.double(), 1)), "conv3 fwd (kernel F) at 32768 images")
    z3 = cnn.conv_fwd_packed(a2, cnn.conv_zpack(W3, 3, cnn.MODE_FWD), b3, 3)
    _close(z3[sl], torch.relu(_conv64_nhwc(a2[sl].double(), W3.double(), b3.double(), 1)), "conv3 fwd (kernel Z) at 32768 images")
    del z3
    # ---- data gradients, default variants, masks fused
    dz3 = torch.randn(a3.shape, device=DEV, generator=g) * (a3 > 0)
    dz2 = cnn.conv_dgrad(dz3, cnn.repack_weights(W3, 3, cnn.MODE_DGRAD_S1_CLASSES), a2, 3, variant=5)
    x = a2[sl].double().requires_grad_(True)
    _conv64_nhwc(x, W3.double(), None, 1).backward(dz3[sl].double())
    _close(dz2[sl], x.grad * (a2[sl] > 0), "conv3 dgrad (border classes, variant 5) at 32768 images")
    zd = cnn.conv_dgrad_packed(dz3, cnn.conv_zpack(W3, 3, cnn.MODE_DGRAD_S1), a2, 3)
    _close(zd[sl], x.grad * (a2[sl] > 0), "conv3 dgrad (kernel Z) at 32768 images")
    del zd
    assert ((a2 > 0) | (dz2 == 0)).all()
    del x, dz3, a3
    dz1 = cnn.conv_dgrad(dz2, cnn.repack_weights(W2, 2, cnn.MODE_DGRAD_S2_CLASSES), a1, 2, variant=cnn.VARIANT_DGRAD2_CLASSES)
    x = a1[sl].double().requires_grad_(True)
    _conv64_nhwc(x, W2.double(), None, 2).backward(dz2[sl].double())
    _close(dz1[sl], x.grad * (a1[sl] > 0), "conv2 dgrad (border classes, variant 6) at 32768 images")
    zd = cnn.conv_dgrad_packed(dz2, cnn.conv_zpack(W2, 2, cnn.MODE_DGRAD_S2), a1, 2)
    _close(zd[sl], x.grad * (a1[sl] > 0), "conv2 dgrad (kernel Z) at 32768 images")
    assert ((a1 > 0) | (zd == 0)).all()
    assert ((a1 > 0) | (dz1 == 0)).all()


def test_full_minibatch_size_properties():
    """Config-C minibatch (32,768 images, where a float64 CPU convolution is out of reach): size-independent properties.
    The data and weight gradients are LINEAR in dz (the ReLU mask depends on the activation only); the forward of a
    gathered batch equals the forward of the rows gathered beforehand; a zero dz gives exactly zero."""
    M = 32768
    g = torch.Generator(device=DEV).manual_seed(5)
    obs = torch.randint(0, 256, (M, 84, 84, 4), dtype=torch.uint8, device=DEV, generator=g)
    inds = torch.randperm(M, device=DEV, generator=g)
    params = {l: tuple(t.to(DEV) for t in _params(l, 40 + l)) for l in (1, 2, 3)}
    bt = {l: cnn.repack_weights(params[l][0], l) for l in (1, 2, 3)}
    a1 = cnn.conv_fwd(obs, bt[1], params[1][1], 1, inds)
    assert torch.equal(a1[:4096], cnn.conv_fwd(obs[inds[:4096]].contiguous(), bt[1], params[1][1], 1))    # gather == pre-gathered rows
    # kernel Q (integer matrix pipe) at full size: gather == pre-gathered rows bit for bit, and the f32-MFMA kernel's result
    # within the f32 bound (both are within it of the float64 truth at the sizes the float64 reference can reach)
    pack = cnn.repack_weights(params[1][0], 1, cnn.MODE_FWD_Q)
    a1q = cnn.conv_fwd(obs, pack, params[1][1], 1, inds, variant=cnn.VARIANT_Q)
    assert torch.equal(a1q[-4096:], cnn.conv_fwd(obs[inds[-4096:]].contiguous(), pack, params[1][1], 1, variant=cnn.VARIANT_Q))
    assert (a1q - a1).abs().max().item() <= 2e-5 * a1.abs().max().item()
    del a1q
    a2 = cnn.conv_fwd(a1, bt[2], params[2][1], 2)
    assert a1.min().item() >= 0.0 and a2.min().item() >= 0.0 and torch.isfinite(a2).all()
    dz_a, dz_b = torch.randn(a2.shape, device=DEV, generator=g), torch.randn(a2.shape, device=DEV, generator=g)
    btd = cnn.repack_weights(params[2][0], 2, cnn.MODE_DGRAD_S2)
    da, db_ = cnn.conv_dgrad(dz_a, btd, a1, 2), cnn.conv_dgrad(dz_b, btd, a1, 2)
    dab = cnn.conv_dgrad(2.0 * dz_a - 0.5 * dz_b, btd, a1, 2)
    lin = 2.0 * da - 0.5 * db_
    assert (dab - lin).abs().max().item() <= 2e-5 * lin.abs().max().item()
    assert ((a1 > 0) | (da == 0)).all()                                                         # masked where the activation is 0
    assert ((a1 > 0) | (dab == 0)).all()
    Wa, ba = cnn.conv_wgrad(a1, dz_a, 2)
    Wb, bb = cnn.conv_wgrad(a1, dz_b, 2)
    Wab, bab = cnn.conv_wgrad(a1, 2.0 * dz_a - 0.5 * dz_b, 2)
    # sums of 2.65 M f32 terms: linear within f32 accumulation error of the sum's scale
    scale = (Wa.abs().max() + Wb.abs().max()).item()
    assert (Wab - (2.0 * Wa - 0.5 * Wb)).abs().max().item() <= 2e-4 * scale
    assert (bab - (2.0 * ba - 0.5 * bb)).abs().max().item() <= 2e-4 * (ba.abs().max() + bb.abs().max()).item()
    W0, b0 = cnn.conv_wgrad(a1, torch.zeros_like(dz_a), 2)
    assert W0.abs().max().item() == 0.0 and b0.abs().max().item() == 0.0
    # layer 1 weight gradient on the uint8 rows, through the gather: linear as well
    d1a, d1b = torch.randn(a1.shape, device=DEV, generator=g), torch.randn(a1.shape, device=DEV, generator=g)
    W1a, _ = cnn.conv_wgrad(obs, d1a, 1, inds)
    W1b, _ = cnn.conv_wgrad(obs, d1b, 1, inds)
    W1ab, _ = cnn.conv_wgrad(obs, d1a + d1b, 1, inds)
    assert (W1ab - (W1a + W1b)).abs().max().item() <= 2e-4 * (W1a.abs().max() + W1b.abs().max()).item()


def _unpack_z(pack, N, K):
    """Decode kernel Z's pack back into three (N, K) float64 planes: the inverse of zpack_kernel (csrc/gemmz.hip)."""
    ntiles = (N + 31) // 32
    raw = pack.cpu().numpy().view(np.uint16).reshape(K // 16, ntiles, 3, 64, 8)          # [s][j][t][lane][e]
    f32 = (raw.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    planes = np.zeros((3, ntiles * 32, K))
    for lane in range(64):
        n_in, kb = lane & 31, lane >> 5
        for t in range(3):
            # rows 32 j + n_in, columns 16 s + 8 kb + e
            planes[t, n_in::32, :].reshape(ntiles, K // 16, 16)[:, :, 8 * kb:8 * kb + 8] = f32[:, :, t, lane, :].transpose(1, 0, 2)
    return planes[:, :N]


@pytest.mark.parametrize("N,K", [(512, 3136), (3136, 512), (40, 32)])
def test_fc_pack_planes_sum_back_to_the_matrix_exactly(N, K):
    """Kernel Z's pre-split weight pack: the three bf16 planes sum back to every weight bit for bit (the split loses nothing),
    each plane is bf16-representable by construction, rows past N in the last 32-row tile are zero; padded row pitch allowed."""
    g = torch.Generator().manual_seed(N + K)
    W = torch.randn(N, K, generator=g) * torch.exp(2.0 * torch.randn(N, K, generator=g))      # wide dynamic range, all 24 bits used
    Wp = cnn.padded_rows(N, K, DEV)
    Wp.copy_(W)
    for src in (W.to(DEV), Wp):
        planes = _unpack_z(cnn.fc_pack(src), N, K)
        assert np.array_equal(planes.sum(0), W.double().numpy())
        assert np.all(np.abs(planes[1]) <= np.abs(planes[0]) * 2.0 ** -7 + 1e-300) and np.all(np.abs(planes[2]) <= np.abs(planes[0]) * 2.0 ** -15 + 1e-300)


@pytest.mark.parametrize("M", [1, 130, 1024, 4095, 4100])
def test_fcz_forward_and_masked_data_gradient_against_float64(M):
    """Kernel Z (csrc/gemmz.hip): the FC layer's forward and data gradient with the weight pre-split into fragment order and the
    activations loaded coalesced through LDS, against float64 on operands with low-order bits set everywhere and a wide dynamic
    range (the three-term split must lose nothing): 2e-5 of the result's scale; the library's f32 GEMM is held to the same bound
    for calibration, and kernel Z's mean error to 2.5 x the library's (both accumulate 3,136 products in f32 -- the library as a
    tree, kernel Z in order; the six term pairs kernel Z multiplies leave out less than the rounding of one f32 product).
    Below 4,096 rows the forward splits K over the grid (raw partials in a workspace, added in order, then bias + ReLU)."""
    g = torch.Generator().manual_seed(190 + M)
    a = torch.relu(torch.randn(M, 3136, generator=g)) * torch.exp(torch.randn(M, 3136, generator=g))
    W = torch.randn(512, 3136, generator=g) / 56.0
    b = torch.randn(512, generator=g) * 0.1
    ref = torch.relu(a.double() @ W.double().t() + b.double())
    got = cnn.fc_fwd_relu_packed(a.to(DEV), cnn.fc_pack(W.to(DEV)), b.to(DEV), 512)
    _close(got, ref, "fc fwd (kernel Z)")
    lib32 = torch.relu(a.to(DEV) @ W.to(DEV).t() + b.to(DEV))
    _close(lib32, ref, "fc fwd (library f32 GEMM, calibration)")
    e_z, e_l = (got.cpu().double() - ref).abs().mean().item(), (lib32.cpu().double() - ref).abs().mean().item()
    # (the library splits K across workgroups / lanes and adds the parts as a tree, which halves the error of ANY sequential f32
    # accumulation; kernel Z accumulates its 196 k-steps in order, as kernel X and the f32-MFMA kernels did)
    assert e_z <= 2.5 * e_l + 1e-12, f"kernel Z mean error {e_z:.3e} vs the library f32 GEMM's {e_l:.3e}"
    dz = torch.randn(M, 512, generator=g) * torch.exp(torch.randn(M, 512, generator=g))
    Wt = W.t().contiguous()                                    # (3136, 512)
    ref_da = (dz.double() @ W.double()) * (a > 0).double()
    pk = cnn.fc_pack(Wt.to(DEV))
    got_da = cnn.fc_dgrad_mask_packed(dz.to(DEV), pk, a.to(DEV))
    _close(got_da, ref_da, "fc dgrad + mask (kernel Z)")
    dz_p = cnn.padded_rows(M, 512, DEV)                        # the learner's padded row pitch: same bits
    dz_p.copy_(dz)
    assert torch.equal(cnn.fc_dgrad_mask_packed(dz_p, pk, a.to(DEV)), got_da)
    assert torch.equal(got_da.cpu() == 0, (ref_da == 0).to(torch.bool) | (got_da.cpu() == 0))       # masked entries are exact zeros
    assert torch.equal(got, cnn.fc_fwd_relu_packed(a.to(DEV), cnn.fc_pack(W.to(DEV)), b.to(DEV), 512))     # deterministic



# ---- ReLU masks as bits (include/mi355ppo.h "ReLU masks as bits"): the *_bits forwards write (activation > 0) as one bit per
# element beside the f32 tensor, the *_bits data gradients select by those bits -- everything must equal the f32-mask kernels
# bit for bit, at sizes with partial tiles and (from 2,048 images) under the B ring.
@pytest.mark.parametrize("images", [1, 37, 1100])
def test_conv1q_forward_also_writes_the_relu_mask_as_bits(images):
    g = torch.Generator().manual_seed(930 + images)
    rows = images + 5
    obs = torch.randint(0, 256, (rows, 84, 84, 4), dtype=torch.uint8, generator=g).to(DEV)
    inds = torch.randperm(rows, generator=g)[:images].to(DEV)
    W, b = _params(1, 12)
    b = b - 0.3                                                # a good share of exact zeros after the ReLU
    pack = cnn.repack_weights(W.to(DEV), 1, cnn.MODE_FWD_Q)
    ref = cnn.conv_fwd(obs, pack, b.to(DEV), 1, inds, variant=cnn.VARIANT_Q)
    out = torch.full((images, 20, 20, 32), float("nan"), device=DEV)
    bits = torch.full((cnn.mask_words(out.numel()),), -1, dtype=torch.int32, device=DEV)
    cnn.conv1q_fwd_bits(obs, pack, b.to(DEV), inds, out, bits)
    assert torch.equal(out, ref)
    assert torch.equal(cnn.unpack_mask_bits(bits, out.shape), out > 0)
    assert 0.05 < (out > 0).float().mean().item() < 0.95


@pytest.mark.parametrize("images", [1, 37, 1027, 1100])
def test_conv1q_forward_writes_nothing_past_its_tensors(images):
    """Kernel Q's tiles are 32 pixels; images x 400 pixels end inside a tile for most batch sizes (1: 400 = 12.5 tiles), and the kernel leaves the pixels past the
    batch to the buffer's range check (round 6: the tile's base must therefore travel in the store's VECTOR offset -- a scalar offset is not range-checked).
    Activations and mask words carved out of sentinel-filled buffers: every word behind them keeps its sentinel."""
    g = torch.Generator().manual_seed(960 + images)
    obs = torch.randint(0, 256, (images, 84, 84, 4), dtype=torch.uint8, generator=g).to(DEV)
    W, b = _params(1, 13)
    pack = cnn.repack_weights(W.to(DEV), 1, cnn.MODE_FWD_Q)
    n, guard = images * 12800, 1 << 16
    abuf = torch.full((n + guard,), 0x7FC0DEAD, dtype=torch.int32, device=DEV)
    bbuf = torch.full((n // 32 + guard,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    out, bits = abuf[:n].view(torch.float32).view(images, 20, 20, 32), bbuf[: n // 32]
    cnn.conv1q_fwd_bits(obs, pack, b.to(DEV), None, out, bits)
    torch.cuda.synchronize()
    assert bool((abuf[n:] == 0x7FC0DEAD).all()) and bool((bbuf[n // 32:] == 0x5A5A5A5A).all())
    ref = cnn.conv_fwd(obs, pack, b.to(DEV), 1, None, variant=cnn.VARIANT_Q)
    assert torch.equal(out, ref) and torch.equal(cnn.unpack_mask_bits(bits, out.shape), out > 0)


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 19, 2049])
def test_conv_fwd_kernel_z_also_writes_the_relu_mask_as_bits(layer, images):
    cin, cout, k, s, hin, hout = SPEC[layer]
    g = torch.Generator().manual_seed(940 + layer + images)
    x = _nhwc(torch.relu(torch.randn(images, cin, hin, hin, generator=g))).to(DEV)
    W, b = _params(layer, 4)
    pack = cnn.conv_zpack(W.to(DEV), layer, cnn.MODE_FWD)
    ref = cnn.conv_fwd_packed(x, pack, b.to(DEV), layer)
    bits = torch.full((cnn.mask_words(ref.numel()),), -1, dtype=torch.int32, device=DEV)
    out = cnn.conv_fwd_packed(x, pack, b.to(DEV), layer, bits=bits)
    assert torch.equal(out, ref)
    assert torch.equal(cnn.unpack_mask_bits(bits, out.shape), out > 0)
    assert 0.05 < (out > 0).float().mean().item() < 0.95


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 19, 700, 2049, 2310])
def test_conv_dgrad_kernel_z_with_the_relu_mask_as_bits(layer, images):
    """The mask words come from the layer below's forward (layer 2's data gradient: kernel Q's a1 mask; layer 3's: kernel Z's a2
    mask), the result must equal the data gradient masked by the f32 activation itself."""
    cin, cout, k, s, hin, hout = SPEC[layer]
    g = torch.Generator().manual_seed(950 + layer + images)
    W, _ = _params(layer, 3)
    mode = cnn.MODE_DGRAD_S1 if layer == 3 else cnn.MODE_DGRAD_S2
    pack = cnn.conv_zpack(W.to(DEV), layer, mode)
    dz = _nhwc(torch.randn(images, cout, hout, hout, generator=g)).to(DEV)
    if layer == 2:                                             # a1 and its mask from kernel Q
        obs = torch.randint(0, 256, (images, 84, 84, 4), dtype=torch.uint8, generator=g).to(DEV)
        W1, b1 = _params(1, 12)
        act = torch.empty((images, 20, 20, 32), device=DEV)
        bits = torch.empty(cnn.mask_words(act.numel()), dtype=torch.int32, device=DEV)
        cnn.conv1q_fwd_bits(obs, cnn.repack_weights(W1.to(DEV), 1, cnn.MODE_FWD_Q), (b1 - 0.3).to(DEV), None, act, bits)
    else:                                                      # a2 and its mask from kernel Z's layer-2 forward
        x = _nhwc(torch.relu(torch.randn(images, 32, 20, 20, generator=g))).to(DEV)
        W2, b2 = _params(2, 4)
        bits = torch.empty(cnn.mask_words(images * 81 * 64), dtype=torch.int32, device=DEV)
        act = cnn.conv_fwd_packed(x, cnn.conv_zpack(W2.to(DEV), 2, cnn.MODE_FWD), b2.to(DEV), 2, bits=bits)
    ref = cnn.conv_dgrad_packed(dz, pack, act, layer)
    got = cnn.conv_dgrad_packed(dz, pack, None, layer, bits=bits)
    assert torch.equal(got, ref)
    assert ((act > 0) | (got == 0)).all() and (got != 0).any()


@pytest.mark.parametrize("M", [1, 130, 4100])
def test_fcz_masked_data_gradient_with_the_relu_mask_as_bits(M):
    g = torch.Generator().manual_seed(960 + M)
    a2 = _nhwc(torch.relu(torch.randn(M, 64, 9, 9, generator=g))).to(DEV)
    W3, b3 = _params(3, 4)
    bits = torch.empty(cnn.mask_words(M * 3136), dtype=torch.int32, device=DEV)
    a3 = cnn.conv_fwd_packed(a2, cnn.conv_zpack(W3.to(DEV), 3, cnn.MODE_FWD), b3.to(DEV), 3, bits=bits).view(M, 3136)
    Wt = (torch.randn(3136, 512, generator=g) / 56.0).to(DEV)
    pk = cnn.fc_pack(Wt)
    dz = torch.randn(M, 512, generator=g).to(DEV)
    ref = cnn.fc_dgrad_mask_packed(dz, pk, a3)
    got = cnn.fc_dgrad_mask_packed(dz, pk, a3, bits=bits)
    assert torch.equal(got, ref)
    assert ((a3 > 0) | (got == 0)).all() and (got != 0).any()


def test_tensors_beyond_4GiB_take_the_64bit_pointer_kernel():
    """Kernels Z, F and V address a tensor with 32-bit buffer offsets.  84,000 images put layer 2's input (and layer 1's output
    and gradient) at 4.30 GB: the entry points and the trunk then route to kernel S (64-bit pointers) and kernel T.  Held to
    the results of the default kernels on the two halves of the batch (4e-6 of the scale: summation order only)."""
    M, H = 84000, 42000
    g = torch.Generator(device=DEV).manual_seed(91)
    W2, b2 = (t.to(DEV) for t in _params(2, 71))
    a1 = torch.relu(torch.randn(M, 20, 20, 32, device=DEV, generator=g))
    assert a1.numel() * 4 > (1 << 32)
    a2 = cnn.conv_fwd(a1, cnn.repack_weights(W2, 2), b2, 2)                          # variant 0 -> kernel S at this size
    pk = cnn.conv_zpack(W2, 2, cnn.MODE_FWD)
    for lo in (0, H):
        _close(a2[lo:lo + H], cnn.conv_fwd_packed(a1[lo:lo + H], pk, b2, 2), f"conv2 fwd beyond 4 GiB, images {lo}..", tol=4e-6)
    with pytest.raises(RuntimeError, match="4 GiB"):
        cnn.conv_fwd_packed(a1, pk, b2, 2)                                           # kernel Z refuses: loudly, not silently wrong
    dz2 = torch.randn(M, 9, 9, 64, device=DEV, generator=g)
    d1 = cnn.conv_dgrad(dz2, cnn.repack_weights(W2, 2, cnn.MODE_DGRAD_S2), a1, 2)    # destination beyond 4 GiB -> kernel S
    pkd = cnn.conv_zpack(W2, 2, cnn.MODE_DGRAD_S2)
    for lo in (0, H):
        _close(d1[lo:lo + H], cnn.conv_dgrad_packed(dz2[lo:lo + H], pkd, a1[lo:lo + H], 2), f"conv2 dgrad beyond 4 GiB, images {lo}..", tol=4e-6)
    dW, db = cnn.conv_wgrad(a1, dz2, 2)                                              # kernel V refuses the size -> kernel T
    parts = [cnn.conv_wgrad(a1[lo:lo + H], dz2[lo:lo + H], 2) for lo in (0, H)]
    _close(dW, parts[0][0].double() + parts[1][0].double(), "conv2 wgrad beyond 4 GiB", tol=2e-5)
    _close(db, parts[0][1].double() + parts[1][1].double(), "conv2 bias gradient beyond 4 GiB", tol=2e-5)
    del a2, d1, dz2
    # the trunk itself at this size: forward + backward run (no kernel refuses) and agree with the two halves
    net = [torch.nn.Conv2d(4, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1)]
    for i, c in enumerate(net):
        Wl, bl = _params(i + 1, 80 + i)
        c.weight.data, c.bias.data = Wl.to(DEV), bl.to(DEV)
    obs = torch.randint(0, 256, (M, 84, 84, 4), dtype=torch.uint8, device=DEV, generator=g)
    da3 = torch.randn(M, 3136, device=DEV, generator=g)
    trunk = cnn.NatureTrunk()

    def run(lo, hi):
        for c in net:
            c.weight.grad = c.bias.grad = None
        f = trunk(obs[lo:hi], None, *net)
        f.backward(da3[lo:hi])
        return f.detach().clone(), [c.weight.grad.double().clone() for c in net]

    f_all, g_all = run(0, M)
    f0, g0 = run(0, H)
    f1, g1 = run(H, M)
    _close(f_all[:H], f0, "trunk forward beyond 4 GiB (first half)", tol=4e-6)
    _close(f_all[H:], f1, "trunk forward beyond 4 GiB (second half)", tol=4e-6)
    # (weight gradients: 84,000 x 400 / 81 / 49 f32 accumulations per element, cut into partial sums differently at the two batch
    # sizes -- a routing check, not an accuracy bar: a dropped or doubled image shows as an O(1e-2) difference)
    for i in range(3):
        _close(g_all[i], g0[i] + g1[i], f"trunk dW{i + 1} beyond 4 GiB", tol=2e-3)
