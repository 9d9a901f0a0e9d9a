"""Round 5: kernels Z / V / W on the two-term f16 split (csrc/f16split.h, the ``*_f16x2`` / ``*_amax`` entry points), against float64.

Reference arithmetic: the f32 Conv2d / Linear layers of the NatureCNN, forward and backward (cleanrl/ppo_atari_multigpu.py:136-148,358).
The bars are the ones of the three-term bf16 kernels in tests/test_gpu_cnn.py, unchanged: |err| <= 2e-5 of the float64 result's scale
(K <= 3,136-term f32 accumulations; the weight gradients, whose reduction runs over the batch, are calibrated by the library's f32 GEMM
exactly as there).  Also held: the amax records are the EXACT maxima of the tensors they describe (every producer), the packs' two
planes sum back to the scaled matrix within 2^-22, results are deterministic, and a record that understates a tensor's maximum
produces infinities (loud), never a silently wrong finite result."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from cleanrl_amd import cnn

DEV = torch.device("cuda:0")
SPEC = {1: (4, 32, 8, 4, 84, 20), 2: (32, 64, 4, 2, 20, 9), 3: (64, 64, 3, 1, 9, 7)}


def _close(got, ref, what, tol=2e-5):
    got, ref = got.detach().double(), ref.detach().double()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert np.isfinite(err) and err <= tol * scale + 1e-30, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / max(scale, 1e-30):.2e})"
    return err / max(scale, 1e-30)


def _params(layer, seed, wscale=1.0):
    cin, cout, k, _, _, _ = SPEC[layer]
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(cout, cin, k, k, generator=g) * (wscale / np.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g) * 0.1
    return W.to(DEV), b.to(DEV)


def _rec_of(x):
    r = cnn.new_amax(1, x.device)[0]
    cnn.absmax(x.contiguous(), r)
    return r


def _conv64(x_nhwc, W, b, stride):
    B, H, _, C = x_nhwc.shape
    cout, cin, k, _ = W.shape
    ho = (H - k) // stride + 1
    cols = F.unfold(x_nhwc.permute(0, 3, 1, 2), kernel_size=k, stride=stride)
    y = torch.einsum("nk,bkl->bln", W.reshape(cout, -1), cols)
    if b is not None:
        y = y + b
    return y.reshape(B, ho, ho, cout)


def _scale_exp(amax: float) -> int:
    bits = int(np.float32(amax).view(np.uint32))
    e = 141 - ((bits >> 23) & 0xFF)
    return max(-100, min(60, e))


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 4096, 1 << 20, (1 << 22) + 7])
def test_absmax_record_is_the_exact_maximum(n):
    g = torch.Generator(device=DEV).manual_seed(n)
    x = torch.randn(n + 1, device=DEV, generator=g) * torch.exp(4 * torch.randn(n + 1, device=DEV, generator=g))
    for view in (x[:n], x[1:]):                     # (the second view is 4-byte aligned only: the scalar path)
        v = view if view.is_contiguous() else view.contiguous()
        r = cnn.new_amax(1, DEV)[0]
        lib = cnn._lib.load()
        assert lib.mi355ppo_absmax_f32(cnn._ptr(view), n, cnn._ptr(r), None) == 0, lib.mi355ppo_last_error()
        assert cnn.amax_value(r) == v.abs().max().item()
    # accumulation: a second tensor folds in
    r = _rec_of(x[:n])
    y = torch.full((7,), 1e30, device=DEV)
    cnn.absmax(y, r)
    assert cnn.amax_value(r) == max(x[:n].abs().max().item(), float(np.float32(1e30)))
    z = cnn.new_amax(1, DEV)[0]
    cnn.absmax(torch.zeros(9, device=DEV), z)
    assert cnn.amax_value(z) == 0.0


def _unpack_h(pack, N, K):
    """f16x2 pack -> (amax bits, hi (N, K), lo (N, K)) as float64."""
    hdr = pack[:64].view(torch.int32)
    body = pack[64:].view(torch.float16)
    nt = (N + 31) // 32
    t = body.view(K // 16, nt, 2, 64, 8).double()               # [s][j][term][lane][e]
    out = []
    for term in range(2):
        p = t[:, :, term]                                       # [s][j][lane][e]: n = 32 j + (lane & 31), k = 16 s + 8 (lane >> 5) + e
        p = p.view(K // 16, nt, 2, 32, 8)                        # lane = 32 lh + li
        m = p.permute(1, 3, 0, 2, 4).reshape(nt * 32, K)         # [j][li] x [s][lh][e]
        out.append(m[:N])
    return int(hdr[0].item()) & 0xFFFFFFFF, out[0], out[1], hdr[1:]


@pytest.mark.parametrize("N,K,scale", [(64, 512, 0.05), (128, 256, 3e-4), (512, 3136, 40.0), (3136, 512, 1e-9), (70, 48, 1.0)])
def test_f16x2_pack_planes_sum_back_to_the_scaled_matrix(N, K, scale):
    g = torch.Generator(device=DEV).manual_seed(N + K)
    B = torch.randn(N, K + 4, device=DEV, generator=g)[:, :K] * scale            # (a padded row pitch, like the FC transpose)
    B = B * torch.exp(2 * torch.randn(N, K, device=DEV, generator=g))
    pack = cnn.fc_pack_f16x2(B)
    bits, hi, lo, rest = _unpack_h(pack, N, K)
    amax = B.abs().max().item()
    assert bits == int(np.float32(amax).view(np.uint32)) and (rest == 0).all()
    s = 2.0 ** _scale_exp(amax)
    sb = B.double() * s
    assert 2 ** 14 <= hi.abs().max().item() <= 2 ** 15
    resid = (sb - hi - lo).abs()
    # hi = f16(s B) to nearest, lo = f16(s B - hi): the residue is below 2^-22 of the element -- or below half an f16 subnormal step for tiny ones
    assert (resid <= sb.abs() * 2.0 ** -22 + 2.0 ** -25).all(), f"worst residue {(resid / sb.abs().clamp_min(1e-300)).max().item():.3e}"
    assert torch.equal(pack, cnn.fc_pack_f16x2(B))


@pytest.mark.parametrize("M", [5, 77, 300, 1024, 4096, 9000, 17000])
def test_fc_forward_and_data_gradient_f16x2_against_float64(M):
    g = torch.Generator(device=DEV).manual_seed(M)
    a = torch.relu(torch.randn(M, 3136, device=DEV, generator=g)) * torch.exp(torch.randn(M, 3136, device=DEV, generator=g))
    W = torch.randn(512, 3136, device=DEV, generator=g) / 56.0
    b = torch.randn(512, device=DEV, generator=g) * 0.1
    ra = _rec_of(a)
    rh = cnn.new_amax(1, DEV)[0]
    lib = cnn._lib.load()
    split_k = lib.mi355ppo_fc_fwd_workspace_bytes(M, 512, 3136) > 0
    pack = cnn.fc_pack_f16x2(W)
    h = cnn.fc_fwd_relu_packed(a, pack, b, 512, amax=(ra, rh))
    _close(h, torch.relu(a.double() @ W.double().t() + b.double()), f"fc fwd f16x2 M={M}")
    if not split_k:
        assert cnn.amax_value(rh) == h.max().item()             # the epilogue's record = the stored tensor's maximum, bit for bit
    assert torch.equal(h, cnn.fc_fwd_relu_packed(a, pack, b, 512, amax=(ra, None)))
    # data gradient: dz (M, 512) with a wide dynamic range and a padded pitch, mask from a (f32) -- tiny magnitudes, as PPO's are
    dz = (torch.randn(M, 516, device=DEV, generator=g)[:, :512] * torch.exp2(-16 * torch.rand(M, 1, device=DEV, generator=g)) * 1e-4
          * (torch.rand(M, 512, device=DEV, generator=g) > 0.4))
    Wt = torch.empty((3136, 516), device=DEV)[:, :512]
    Wt.copy_(W.t())
    rz, rd = _rec_of(dz), cnn.new_amax(1, DEV)[0]
    da = cnn.fc_dgrad_mask_packed(dz, cnn.fc_pack_f16x2(Wt), a, amax=(rz, rd))
    _close(da, (dz.double() @ W.double()) * (a > 0), f"fc dgrad f16x2 M={M}")
    assert cnn.amax_value(rd) == da.abs().max().item()


def _bits_of(mask):
    """bool tensor -> the int32 words of its bit mask (word w, bit b <-> flat element 32 w + b)."""
    m = mask.reshape(-1, 32).to(torch.int64)
    w = (m << torch.arange(32, device=mask.device)).sum(1)
    return ((w + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32)


@pytest.mark.parametrize("M", [1, 77, 128, 300, 1027, 8192, 9000])
def test_kernel_g_fc_forward_and_data_gradient_are_kernel_z_bit_for_bit(monkeypatch, M):
    """Kernel G (csrc/gemmg.hip, round 6: both operands of the FC layer through workgroup-wide LDS rings, A split once per workgroup; the
    default from 16,384 / 8,192 rows on) against kernel Z's whole-K route on Linear(3136, 512) forward (cleanrl/ppo_atari_multigpu.py:144)
    and its data gradient under conv3's ReLU mask bits, at sizes with partial row blocks and with the data gradient's partial last column
    block (3,136 = 12 x 256 + 64 columns): outputs and amax records equal bit for bit -- the same products in the same order --, each within
    the float64 bar."""
    g = torch.Generator(device=DEV).manual_seed(1000 + M)
    a = torch.relu(torch.randn(M, 3136, device=DEV, generator=g)) * torch.exp(torch.randn(M, 3136, device=DEV, generator=g))
    W = torch.randn(512, 3136, device=DEV, generator=g) / 56.0
    b = torch.randn(512, device=DEV, generator=g) * 0.1
    dz = (torch.randn(M, 516, device=DEV, generator=g)[:, :512] * torch.exp2(-16 * torch.rand(M, 1, device=DEV, generator=g)) * 1e-4
          * (torch.rand(M, 512, device=DEV, generator=g) > 0.4))
    Wt = torch.empty((3136, 516), device=DEV)[:, :512]
    Wt.copy_(W.t())
    pack, packt = cnn.fc_pack_f16x2(W), cnn.fc_pack_f16x2(Wt)
    bits = _bits_of(a > 0)
    ra, rz = _rec_of(a), _rec_of(dz)
    lib = cnn._lib.load()
    out = {}
    for route, env in (("Z", {"MI355PPO_FC_G": "0"}), ("G", {"MI355PPO_FC_G": "min:1"})):
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        rh, rd = cnn.new_amax(2, DEV)
        h = torch.full((M, 512), float("nan"), device=DEV)
        # (no workspace: the whole-K route at every size -- below 8,192 rows the wrapper would split K over the grid on both routes)
        st = lib.mi355ppo_fc_fwd_relu_packed_f16x2_f32(cnn._ptr(a), 3136, cnn._ptr(pack), cnn._ptr(b), cnn._ptr(h), M, 512, 3136, None, 0,
                                                       cnn._ptr(ra), cnn._ptr(rh), None)
        assert st == 0, lib.mi355ppo_last_error()
        da = cnn.fc_dgrad_mask_packed(dz, packt, a, out=torch.full((M, 3136), float("nan"), device=DEV), bits=bits, amax=(rz, rd))
        torch.cuda.synchronize()
        out[route] = (h, da, cnn.amax_value(rh), cnn.amax_value(rd))
    (hz, dz_, ahz, adz), (hg, dg, ahg, adg) = out["Z"], out["G"]
    assert torch.equal(hg.view(torch.int32), hz.view(torch.int32)) and ahg == ahz == hg.max().item()
    assert torch.equal(dg.view(torch.int32), dz_.view(torch.int32)) and adg == adz == dg.abs().max().item()
    _close(hg, torch.relu(a.double() @ W.double().t() + b.double()), f"kernel G fc fwd M={M}")
    _close(dg, (dz.double() @ W.double()) * (a > 0), f"kernel G fc dgrad M={M}")


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 3, 37, 256, 700, 2100])
def test_conv_forward_f16x2_against_float64(layer, images):
    """32-row tiles below 768 images, 64-row tiles above, the B ring from 2,048 images on -- each against float64, with and without the
    mask bits; the output's amax record exact."""
    cin, cout, k, st, hin, hout = SPEC[layer]
    W, b = _params(layer, 10 * layer + images)
    g = torch.Generator(device=DEV).manual_seed(images + layer)
    x = torch.relu(torch.randn(images, hin, hin, cin, device=DEV, generator=g)) * torch.exp(torch.randn(images, hin, hin, cin, device=DEV, generator=g))
    ref = torch.relu(_conv64(x.double(), W.double(), b.double(), st))
    pack = cnn.conv_zpack_f16x2(W, layer, cnn.MODE_FWD)
    rx, ry = _rec_of(x), cnn.new_amax(1, DEV)[0]
    y = cnn.conv_fwd_packed(x, pack, b, layer, amax=(rx, ry))
    _close(y, ref, f"conv{layer} fwd f16x2, {images} images")
    assert cnn.amax_value(ry) == y.max().item()
    bits = torch.empty(cnn.mask_words(y.numel()), dtype=torch.int32, device=DEV)
    ry2 = cnn.new_amax(1, DEV)[0]
    y2 = cnn.conv_fwd_packed(x, pack, b, layer, bits=bits, amax=(rx, ry2))
    if images >= 768:
        assert torch.equal(y2, y)                       # the same 64-row tiles with and without the bit epilogue
    else:
        _close(y2, ref, f"conv{layer} fwd f16x2 + bits, {images} images")
    assert torch.equal(cnn.unpack_mask_bits(bits, y2.shape), y2 > 0) and cnn.amax_value(ry2) == y2.max().item()


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 4, 5, 61, 1027])
def test_kernel_r_forwards_are_kernel_z_bit_for_bit(monkeypatch, layer, images):
    """Kernel R (csrc/convr.hip: the source of an image group resident in LDS, split once) against kernel Z on the layer-2 (stride 2, 32
    source channels, 2 images per group) and layer-3 (5 images per group) forwards, with and without the mask bits, at sizes with partial
    last groups: output, mask words and the output's amax record equal bit for bit -- the same products in the same order --, and the
    output within the float64 bar."""
    cin, cout, k, st, hin, hout = SPEC[layer]
    W, b = _params(layer, 77 + images)
    g = torch.Generator(device=DEV).manual_seed(images)
    x = torch.relu(torch.randn(images, hin, hin, cin, device=DEV, generator=g)) * torch.exp(torch.randn(images, hin, hin, cin, device=DEV, generator=g))
    pack = cnn.conv_zpack_f16x2(W, layer, cnn.MODE_FWD)
    rx = _rec_of(x)
    out = {}
    for route, env in (("Z", {"MI355PPO_CONV_R": "0"}), ("R", {"MI355PPO_CONV_R": "min:1"})):
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        ry, ry2 = cnn.new_amax(2, DEV)
        y = cnn.conv_fwd_packed(x, pack, b, layer, amax=(rx, ry))
        bits = torch.zeros(cnn.mask_words(y.numel()), dtype=torch.int32, device=DEV)
        y2 = cnn.conv_fwd_packed(x, pack, b, layer, bits=bits, amax=(rx, ry2))
        out[route] = (y, y2, bits, cnn.amax_value(ry), cnn.amax_value(ry2))
    (yz, yz2, bz, az, az2), (yr, yr2, br, ar, ar2) = out["Z"], out["R"]
    assert torch.equal(yr.view(torch.int32), yr2.view(torch.int32)) and torch.equal(br, bz) and ar == az == ar2 == yr.max().item()
    if images >= 768:                                    # (below, kernel Z runs its 32-row tiles without the bits and 64-row tiles with them)
        assert torch.equal(yr.view(torch.int32), yz.view(torch.int32))
    assert torch.equal(yr2.view(torch.int32), yz2.view(torch.int32))
    _close(yr, torch.relu(_conv64(x.double(), W.double(), b.double(), st)), f"kernel R conv{layer} fwd, {images} images")


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 2, 3, 61, 1027])
def test_kernel_r_data_gradients_are_kernel_z_bit_for_bit(monkeypatch, layer, images):
    """The layer-2 (default from 512 images) and layer-3 data gradients on kernel R against kernel Z: gradient and its
    amax record bit-equal (the zero border adds exact zeros where kernel Z's border classes skip the taps), within the float64 bar."""
    cin, cout, k, st, hin, hout = SPEC[layer]
    W, _ = _params(layer, 20 * layer + images)
    g = torch.Generator(device=DEV).manual_seed(3 * images + layer)
    act = torch.randn(images, hin, hin, cin, device=DEV, generator=g)
    dz = (torch.randn(images, hout, hout, cout, device=DEV, generator=g) * torch.exp2(-14 * torch.rand(images, 1, 1, 1, device=DEV, generator=g)) * 1e-3
          * (torch.rand(images, hout, hout, cout, device=DEV, generator=g) > 0.5))
    m = (act > 0).reshape(-1, 32).to(torch.int64)                     # word w, bit b <-> element 32 w + b of the flat activation
    w = (m << torch.arange(32, device=DEV)).sum(1)
    bits = ((w + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32)
    assert torch.equal(cnn.unpack_mask_bits(bits, act.shape), act > 0)
    pack = cnn.conv_zpack_f16x2(W, layer, cnn.MODE_DGRAD_S2 if layer == 2 else cnn.MODE_DGRAD_S1)
    rz = _rec_of(dz)
    out = {}
    for route, env in (("Z", {"MI355PPO_CONV_R": "0"}), ("R", {"MI355PPO_CONV_R": "min:1"})):
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        rd = cnn.new_amax(1, DEV)[0]
        got = cnn.conv_dgrad_packed(dz, pack, None, layer, bits=bits, amax=(rz, rd))
        out[route] = (got, cnn.amax_value(rd))
    assert torch.equal(out["R"][0].view(torch.int32), out["Z"][0].view(torch.int32)) and out["R"][1] == out["Z"][1] == out["R"][0].abs().max().item()
    x = act.double().requires_grad_(True)
    _conv64(x, W.double(), None, st).backward(dz.double())
    _close(out["R"][0], x.grad * (act > 0), f"kernel R conv{layer} dgrad, {images} images")


@pytest.mark.parametrize("images", [3072, 3073, 3074, 4100])
def test_kernel_rb_is_kernel_z_bit_for_bit_and_writes_nothing_past_its_tensor(monkeypatch, images):
    """Kernel RB (csrc/convrb.hip: the layer-2 data gradient from 3,072 images on -- three images per group, the rows dealt to tiles by border class, the products
    with a zero-border operand not issued) against kernel Z through the C ABI: gradient and amax record bit-equal, at batch sizes whose last group holds three,
    one and two images; the gradient carved out of a sentinel-filled buffer -- the rows of the images a last group does not have are left to the buffer
    descriptor's range check, every word behind the tensor must keep its sentinel; a slice against float64."""
    layer = 2
    cin, cout, k, st, hin, hout = SPEC[layer]
    lib = cnn._lib.load()
    monkeypatch.delenv("MI355PPO_CONV_R", raising=False)
    assert chr(lib.mi355ppo_cnn_conv_packed_kernel_f16x2(images, 2, 1)) == "B"
    W, _ = _params(layer, 40 + images % 7)
    g = torch.Generator(device=DEV).manual_seed(5 * images + 1)
    act = torch.randn(images, hin, hin, cin, device=DEV, generator=g)
    dz = (torch.randn(images, hout, hout, cout, device=DEV, generator=g) * torch.exp2(-14 * torch.rand(images, 1, 1, 1, device=DEV, generator=g)) * 1e-3
          * (torch.rand(images, hout, hout, cout, device=DEV, generator=g) > 0.5))
    m = (act > 0).reshape(-1, 32).to(torch.int64)
    w = (m << torch.arange(32, device=DEV)).sum(1)
    bits = ((w + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32)
    pack = cnn.conv_zpack_f16x2(W, layer, cnn.MODE_DGRAD_S2)
    rz = _rec_of(dz)
    n, guard = act.numel(), 1 << 18
    out = {}
    for route, env in (("Z", {"MI355PPO_CONV_R": "0"}), ("B", {})):
        monkeypatch.delenv("MI355PPO_CONV_R", raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        buf = torch.full((n + guard,), 0x7FC0DEAD, dtype=torch.int32, device=DEV)
        rd = cnn.new_amax(1, DEV)[0]
        got = cnn.conv_dgrad_packed(dz, pack, None, layer, buf[:n].view(torch.float32).view(act.shape), bits=bits, amax=(rz, rd))
        torch.cuda.synchronize()
        assert bool((buf[n:] == 0x7FC0DEAD).all()), f"route {route}: words behind the gradient were written"
        out[route] = (got, cnn.amax_value(rd))
    assert torch.equal(out["B"][0].view(torch.int32), out["Z"][0].view(torch.int32)) and out["B"][1] == out["Z"][1] == out["B"][0].abs().max().item()
    sl = slice(images - 5, images)                                     # the last group(s)
    x = act[sl].double().requires_grad_(True)
    _conv64(x, W.double(), None, st).backward(dz[sl].double())
    _close(out["B"][0][sl], x.grad * (act[sl] > 0), f"kernel RB conv2 dgrad, {images} images")


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("images", [1, 37, 64, 700, 2100])
def test_conv_data_gradient_f16x2_against_float64(layer, images):
    cin, cout, k, st, hin, hout = SPEC[layer]
    W, _ = _params(layer, 20 * layer + images)
    g = torch.Generator(device=DEV).manual_seed(3 * images + layer)
    act = torch.randn(images, hin, hin, cin, device=DEV, generator=g)
    dz = (torch.randn(images, hout, hout, cout, device=DEV, generator=g) * torch.exp2(-14 * torch.rand(images, 1, 1, 1, device=DEV, generator=g)) * 1e-3
          * (torch.rand(images, hout, hout, cout, device=DEV, generator=g) > 0.5))
    x = act.double().requires_grad_(True)
    _conv64(x, W.double(), None, st).backward(dz.double())
    ref = x.grad * (act > 0)
    mode = cnn.MODE_DGRAD_S2 if layer == 2 else cnn.MODE_DGRAD_S1
    pack = cnn.conv_zpack_f16x2(W, layer, mode)
    rz, rd = _rec_of(dz), cnn.new_amax(1, DEV)[0]
    got = cnn.conv_dgrad_packed(dz, pack, act, layer, amax=(rz, rd))
    _close(got, ref, f"conv{layer} dgrad f16x2, {images} images")
    assert cnn.amax_value(rd) == got.abs().max().item()
    # the mask as bits
    bits = torch.empty(cnn.mask_words(act.numel()), dtype=torch.int32, device=DEV)
    w = (act.reshape(-1, 32) > 0).to(torch.int64)
    words = (w << torch.arange(32, device=DEV)).sum(1)
    bits.copy_(torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32))
    rd2 = cnn.new_amax(1, DEV)[0]
    got2 = cnn.conv_dgrad_packed(dz, pack, None, layer, bits=bits, amax=(rz, rd2))
    assert torch.equal(got2, got) and cnn.amax_value(rd2) == cnn.amax_value(rd)


@pytest.mark.parametrize("layer,images", [(2, 208), (2, 256), (2, 2048), (3, 304), (3, 1024), (3, 2064), (3, 1), (3, 5), (3, 1027), (2, 1), (2, 3), (2, 1027)])      # (208: fewer images than kernel V's 256 slabs)
def test_conv_weight_gradient_f16x2_against_float64(layer, images):
    """Layer 2: kernel V; layer 3: kernel U (csrc/convu.hip: both operands of an image group resident in LDS, fragments by LDS transpose
    reads) at every size, partial last groups included -- against float64, against the three-term bf16 kernel and run to run."""
    cin, cout, k, st, hin, hout = SPEC[layer]
    lib = cnn._lib.load()
    assert chr(lib.mi355ppo_cnn_conv_wgrad_kernel(images, layer)) == ("V" if images % 16 == 0 else "T")
    g = torch.Generator(device=DEV).manual_seed(images + 7 * layer)
    src = torch.relu(torch.randn(images, hin, hin, cin, device=DEV, generator=g)) * torch.exp(torch.randn(images, hin, hin, cin, device=DEV, generator=g))
    dz = (torch.randn(images, hout, hout, cout, device=DEV, generator=g) * torch.exp2(-10 * torch.rand(images, 1, 1, 1, device=DEV, generator=g)) * 1e-3)
    ref = torch.nn.grad.conv2d_weight(src.double().permute(0, 3, 1, 2), (cout, cin, k, k), dz.double().permute(0, 3, 1, 2), stride=st)
    dW, db = cnn.conv_wgrad(src, dz, layer, amax=(_rec_of(src), _rec_of(dz)))
    dWb, dbb = cnn.conv_wgrad(src, dz, layer)                   # the three-term bf16 kernel on the same operands
    _close(db, dz.double().sum((0, 1, 2)), f"conv{layer} bias gradient f16x2, {images} images", tol=2e-5)
    e_h = _close(dW, ref, f"conv{layer} wgrad f16x2, {images} images")
    e_b = _close(dWb, ref, f"conv{layer} wgrad bf16x3, {images} images")
    assert e_h <= max(4.0 * e_b, 2e-6), f"f16x2 {e_h:.2e} vs bf16x3 {e_b:.2e}"
    # the bias gradient sums the f32 fragments: independent of the split -- bit-equal where both kernels cut the batch into the same slabs
    assert torch.equal(db, dbb) or (db.double() - dbb.double()).abs().max().item() <= 2e-6 * dbb.abs().max().item()
    assert torch.equal(dW, cnn.conv_wgrad(src, dz, layer, amax=(_rec_of(src), _rec_of(dz)))[0])


@pytest.mark.parametrize("images,scale", [(8, 1.0), (300, 1e-6), (4096, 3e-4), (8192, 1e-4)])      # (8,192: config D's minibatch -- the direct float64 bar behind the relaxed trajectory bar of tests/test_gpu_multirank.py)
def test_conv1_weight_gradient_f16x2_against_float64(images, scale):
    """The layer-1 weight gradient of the f16 split (kernel U's layer-1 variant, csrc/convu.hip: one image per pass resident in LDS, the uint8
    frame as zero-extended 16-bit = exact f16 subnormals; MI355PPO_CONV_U=23: kernel P) against float64 and against kernel P's three-term bf16
    variant, through a row gather; the bias gradient sums the f32 values as loaded (another order than kernel P's)."""
    g = torch.Generator(device=DEV).manual_seed(images)
    obs = torch.randint(0, 256, (images + 5, 84, 84, 4), dtype=torch.uint8, device=DEV, generator=g)
    inds = torch.randperm(images + 5, device=DEV, generator=g)[:images]
    dz = (torch.randn(images, 20, 20, 32, device=DEV, generator=g) * torch.exp2(-10 * torch.rand(images, 1, 1, 1, device=DEV, generator=g)) * scale
          * (torch.rand(images, 20, 20, 32, device=DEV, generator=g) > 0.5))
    x = obs[inds].double().permute(0, 3, 1, 2) / 255.0
    ref = torch.nn.grad.conv2d_weight(x, (32, 4, 8, 8), dz.double().permute(0, 3, 1, 2), stride=4)
    dW, db = cnn.conv_wgrad(obs, dz, 1, inds, amax=(None, _rec_of(dz)))
    dWb, dbb = cnn.conv_wgrad(obs, dz, 1, inds)
    e_h, e_b = _close(dW, ref, f"conv1 wgrad f16x2, {images} images"), _close(dWb, ref, f"conv1 wgrad bf16x3, {images} images")
    assert e_h <= max(4.0 * e_b, 2e-6), f"f16x2 {e_h:.2e} vs bf16x3 {e_b:.2e}"
    _close(db, dz.double().sum((0, 1, 2)), f"conv1 bias gradient, {images} images", tol=2e-5)
    assert torch.equal(db, dbb) or (db.double() - dbb.double()).abs().max().item() <= 2e-6 * max(dbb.abs().max().item(), 1e-30)
    dW2, db2 = cnn.conv_wgrad(obs, dz, 1, inds, amax=(None, _rec_of(dz)))
    assert torch.equal(dW, dW2) and torch.equal(db, db2)          # run to run


@pytest.mark.parametrize("M", [1024, 4096])
def test_fc_weight_gradient_f16x2_against_float64(M):
    lib = cnn._lib.load()
    assert chr(lib.mi355ppo_fc_wgrad_kernel(M, 512, 3136)) == "W"
    g = torch.Generator(device=DEV).manual_seed(M + 1)
    a = torch.relu(torch.randn(M, 3136, device=DEV, generator=g)) * torch.exp(torch.randn(M, 3136, device=DEV, generator=g))
    dz = torch.randn(M, 516, device=DEV, generator=g)[:, :512] * torch.exp2(-12 * torch.rand(M, 1, device=DEV, generator=g)) * 1e-4
    ref = dz.double().t() @ a.double()
    got = cnn.fc_wgrad(dz, a, amax=(_rec_of(dz), _rec_of(a)))
    e_h = _close(got, ref, f"fc wgrad f16x2 M={M}")
    e_b = _close(cnn.fc_wgrad(dz, a), ref, f"fc wgrad bf16x3 M={M}")
    assert e_h <= max(4.0 * e_b, 2e-6), f"f16x2 {e_h:.2e} vs bf16x3 {e_b:.2e}"
    chw = cnn.fc_wgrad(dz, a, 64, amax=(_rec_of(dz), _rec_of(a)))
    assert torch.equal(chw, got.view(512, 49, 64).permute(0, 2, 1).reshape(512, 3136))


@pytest.mark.parametrize("M", [1000, 1024, 4096, 8192, 9008])
def test_kernel_h_fc_weight_gradient_against_float64_and_kernel_w(monkeypatch, M):
    """Kernel H (csrc/gemmh.hip, round 6: both operands of Linear(3136, 512)'s weight gradient through a workgroup-wide LDS ring, split once,
    fragments by LDS transpose reads; 8 slabs of contiguous rows; the default from 4,096 rows on) against float64 with kernel W's bar, against
    kernel W itself (another order of the same exact products), with a padded dz pitch, a batch that is not a multiple of the slot size, the
    (h, w, c) -> (c, h, w) column order, and run twice (deterministic)."""
    lib = cnn._lib.load()
    g = torch.Generator(device=DEV).manual_seed(M + 7)
    a = torch.relu(torch.randn(M, 3136, device=DEV, generator=g)) * torch.exp(torch.randn(M, 3136, device=DEV, generator=g))
    dz = torch.randn(M, 516, device=DEV, generator=g)[:, :512] * torch.exp2(-12 * torch.rand(M, 1, device=DEV, generator=g)) * 1e-4
    ref = dz.double().t() @ a.double()
    rz, ra = _rec_of(dz), _rec_of(a)
    monkeypatch.setenv("MI355PPO_FC_H", "0")
    assert chr(lib.mi355ppo_fc_wgrad_kernel_f16x2(M, 512, 3136)) in "WY"
    w = cnn.fc_wgrad(dz, a, amax=(rz, ra))
    e_w = _close(w, ref, f"fc wgrad kernel W/Y M={M}")
    monkeypatch.setenv("MI355PPO_FC_H", "1")
    monkeypatch.setenv("MI355PPO_FC_H", "min:1")
    assert chr(lib.mi355ppo_fc_wgrad_kernel_f16x2(M, 512, 3136)) == "H"
    got = cnn.fc_wgrad(dz, a, amax=(rz, ra), out=torch.full((512, 3136), float("nan"), device=DEV))
    e_h = _close(got, ref, f"fc wgrad kernel H M={M}")
    assert e_h <= max(4.0 * e_w, 2e-6), f"kernel H {e_h:.2e} vs kernel W {e_w:.2e}"
    assert torch.equal(got, cnn.fc_wgrad(dz, a, amax=(rz, ra)))
    chw = cnn.fc_wgrad(dz, a, 64, amax=(rz, ra))
    assert torch.equal(chw, got.view(512, 49, 64).permute(0, 2, 1).reshape(512, 3136))
    monkeypatch.delenv("MI355PPO_FC_H")
    assert chr(lib.mi355ppo_fc_wgrad_kernel_f16x2(M, 512, 3136)) == ("H" if M >= 4096 else ("W" if M % 16 == 0 and M >= 1024 else "Y"))


@pytest.mark.parametrize("images", [5, 1024])
def test_conv1q_amax_and_heads_backward_amax(images):
    """The two producers outside kernel Z: kernel Q's layer-1 forward and the heads' backward -- same results as their plain entry points,
    records = the exact maxima."""
    W, b = _params(1, 3)
    g = torch.Generator(device=DEV).manual_seed(images)
    obs = torch.randint(0, 256, (images, 84, 84, 4), dtype=torch.uint8, device=DEV, generator=g)
    pack = cnn.repack_weights(W, 1, cnn.MODE_FWD_Q)
    ref = cnn.conv_fwd(obs, pack, b, 1, None, variant=cnn.VARIANT_Q)
    out, r = torch.empty_like(ref), cnn.new_amax(1, DEV)[0]
    cnn.conv1q_fwd_amax(obs, pack, b, None, out, None, r)
    assert torch.equal(out, ref) and cnn.amax_value(r) == ref.max().item()
    bits, r2 = torch.empty(cnn.mask_words(ref.numel()), dtype=torch.int32, device=DEV), cnn.new_amax(1, DEV)[0]
    cnn.conv1q_fwd_amax(obs, pack, b, None, out, bits, r2)
    assert torch.equal(out, ref) and torch.equal(cnn.unpack_mask_bits(bits, ref.shape), ref > 0) and cnn.amax_value(r2) == ref.max().item()
    # heads backward
    lib = cnn._lib.load()
    M, A = images, 4
    h = torch.relu(torch.randn(M, 512, device=DEV, generator=g))
    Wa, Wc = torch.randn(A, 512, device=DEV, generator=g) * 0.05, torch.randn(1, 512, device=DEV, generator=g) * 0.05
    gl, gv = torch.randn(M, A, device=DEV, generator=g) * 1e-5, torch.randn(M, 1, device=DEV, generator=g) * 1e-5
    ws = torch.empty(lib.mi355ppo_heads_bwd_workspace_bytes(M, A), dtype=torch.uint8, device=DEV)
    P = cnn._ptr
    res = []
    for with_amax in (False, True):
        dh = torch.zeros((M, 516), device=DEV)[:, :512]
        o = [torch.empty_like(Wa), torch.empty(A, device=DEV), torch.empty_like(Wc), torch.empty(1, device=DEV), torch.empty(512, device=DEV)]
        rr = cnn.new_amax(1, DEV)[0]
        if with_amax:
            st = lib.mi355ppo_heads_bwd_relu_amax_f32(P(h), P(Wa), P(Wc), P(gl), P(gv), P(dh), 516, *[P(t) for t in o], M, A, 512, P(ws), ws.numel(), P(rr), None)
        else:
            st = lib.mi355ppo_heads_bwd_relu_f32(P(h), P(Wa), P(Wc), P(gl), P(gv), P(dh), 516, *[P(t) for t in o], M, A, 512, P(ws), ws.numel(), None)
        assert st == 0, lib.mi355ppo_last_error()
        res.append((dh, o, rr))
    assert torch.equal(res[0][0], res[1][0]) and all(torch.equal(x, y) for x, y in zip(res[0][1], res[1][1]))
    assert cnn.amax_value(res[1][2]) == res[1][0].abs().max().item()


def test_an_understated_record_is_loud():
    """A record below the tensor's maximum makes the f16 conversion overflow: infinities / NaNs in the result, not a plausible number."""
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(9000, 3136, device=DEV, generator=g).abs()
    W = torch.randn(512, 3136, device=DEV, generator=g) / 56.0
    r = _rec_of(a * 2.0 ** -6)                                  # 64 x too small: s a reaches 2^20 > 65,504
    h = cnn.fc_fwd_relu_packed(a, cnn.fc_pack_f16x2(W), torch.zeros(512, device=DEV), 512, amax=(r, None))
    assert not torch.isfinite(h).all()


def test_f16x2_at_the_full_minibatch_size_against_float64_on_the_device():
    """Config-C minibatch (32,768 images): every f16x2 launch of one update at its bench size, chained the way the learner chains them (each
    kernel's record feeds the next), against float64 on the device (a strided slab for the convolutions).  Bars as in
    test_gpu_cnn.py::test_conv_forward_and_data_gradient_kernels_at_the_full_minibatch_size_... / test_fc_kernels_at_the_full_minibatch_size_..."""
    M = 32768
    g = torch.Generator(device=DEV).manual_seed(11)
    idx = torch.arange(0, M, 16, device=DEV)[:2048]
    sl = torch.unique(torch.cat([idx, torch.arange(0, 64, device=DEV), torch.arange(M - 64, M, device=DEV)]))
    (W1, b1), (W2, b2), (W3, b3) = (_params(l, 70 + l) for l in (1, 2, 3))
    rec = cnn.new_amax(cnn.N_REC, DEV)
    obs = torch.randint(0, 256, (M, 84, 84, 4), dtype=torch.uint8, device=DEV, generator=g)
    inds = torch.randperm(M, device=DEV, generator=g)
    a1 = torch.empty((M, 20, 20, 32), device=DEV)
    mb1 = torch.empty(cnn.mask_words(a1.numel()), dtype=torch.int32, device=DEV)
    cnn.conv1q_fwd_amax(obs, cnn.repack_weights(W1, 1, cnn.MODE_FWD_Q), b1, inds, a1, mb1, rec[cnn.REC_A1])
    assert cnn.amax_value(rec[cnn.REC_A1]) == a1.max().item()
    del obs
    a2 = torch.empty((M, 9, 9, 64), device=DEV)
    mb2 = torch.empty(cnn.mask_words(a2.numel()), dtype=torch.int32, device=DEV)
    cnn.conv_fwd_packed(a1, cnn.conv_zpack_f16x2(W2, 2, cnn.MODE_FWD), b2, 2, a2, bits=mb2, amax=(rec[cnn.REC_A1], rec[cnn.REC_A2]))
    _close(a2[sl], torch.relu(_conv64(a1[sl].double(), W2.double(), b2.double(), 2)), "conv2 fwd f16x2 at 32768")
    assert cnn.amax_value(rec[cnn.REC_A2]) == a2.max().item()
    a3 = torch.empty((M, 7, 7, 64), device=DEV)
    mb3 = torch.empty(cnn.mask_words(a3.numel()), dtype=torch.int32, device=DEV)
    cnn.conv_fwd_packed(a2, cnn.conv_zpack_f16x2(W3, 3, cnn.MODE_FWD), b3, 3, a3, bits=mb3, amax=(rec[cnn.REC_A2], rec[cnn.REC_A3]))
    _close(a3[sl], torch.relu(_conv64(a2[sl].double(), W3.double(), b3.double(), 1)), "conv3 fwd f16x2 at 32768")
    assert cnn.amax_value(rec[cnn.REC_A3]) == a3.max().item()
    # FC layer
    Wfc = torch.randn(512, 3136, device=DEV, generator=g) / 56.0
    bfc = torch.randn(512, device=DEV, generator=g) * 0.1
    af = a3.view(M, 3136)
    h = cnn.fc_fwd_relu_packed(af, cnn.fc_pack_f16x2(Wfc), bfc, 512, amax=(rec[cnn.REC_A3], None))
    _close(h, torch.relu(af.double() @ Wfc.double().t() + bfc.double()), "fc fwd f16x2 at 32768")
    dh = (torch.randn(M, 516, device=DEV, generator=g)[:, :512] * 3e-5 * (h > 0))
    cnn.absmax(dh.contiguous(), rec[cnn.REC_DH])
    Wt = torch.empty((3136, 516), device=DEV)[:, :512]
    Wt.copy_(Wfc.t())
    dz3 = torch.empty((M, 3136), device=DEV)
    cnn.fc_dgrad_mask_packed(dh, cnn.fc_pack_f16x2(Wt), af, out=dz3, bits=mb3, amax=(rec[cnn.REC_DH], rec[cnn.REC_DZ3]))
    _close(dz3, (dh.double() @ Wfc.double()) * (af > 0), "fc dgrad f16x2 at 32768")
    assert cnn.amax_value(rec[cnn.REC_DZ3]) == dz3.abs().max().item()
    ref = dh.double().t() @ af.double()
    got = cnn.fc_wgrad(dh, af, amax=(rec[cnn.REC_DH], rec[cnn.REC_A3]))
    scale = ref.abs().max().item()
    err, err_t = (got.double() - ref).abs().max().item(), ((dh.t() @ af).double() - ref).abs().max().item()
    assert err <= max(2e-5 * scale, 4.0 * err_t), f"fc dW f16x2 at 32768: err {err:.3e}, library f32 err {err_t:.3e}, scale {scale:.3e}"
    del ref, got, h
    # conv3 gradients
    dz3 = dz3.view(M, 7, 7, 64)
    dz2 = torch.empty((M, 9, 9, 64), device=DEV)
    cnn.conv_dgrad_packed(dz3, cnn.conv_zpack_f16x2(W3, 3, cnn.MODE_DGRAD_S1), None, 3, dz2, bits=mb2, amax=(rec[cnn.REC_DZ3], rec[cnn.REC_DZ2]))
    x = a2[sl].double().requires_grad_(True)
    _conv64(x, W3.double(), None, 1).backward(dz3[sl].double())
    _close(dz2[sl], x.grad * (a2[sl] > 0), "conv3 dgrad f16x2 at 32768")
    assert cnn.amax_value(rec[cnn.REC_DZ2]) == dz2.abs().max().item()
    dW3, db3 = cnn.conv_wgrad(a2, dz3, 3, amax=(rec[cnn.REC_A2], rec[cnn.REC_DZ3]))
    dW3b, _ = cnn.conv_wgrad(a2, dz3, 3)
    ref = torch.nn.grad.conv2d_weight(a2.double().permute(0, 3, 1, 2), (64, 64, 3, 3), dz3.double().permute(0, 3, 1, 2), stride=1)
    e_h, e_b = _close(dW3, ref, "conv3 wgrad f16x2 at 32768"), _close(dW3b, ref, "conv3 wgrad bf16x3 at 32768")
    assert e_h <= max(4.0 * e_b, 2e-6), (e_h, e_b)
    del x, ref
    # conv2 gradients
    dz1 = torch.empty((M, 20, 20, 32), device=DEV)
    cnn.conv_dgrad_packed(dz2, cnn.conv_zpack_f16x2(W2, 2, cnn.MODE_DGRAD_S2), None, 2, dz1, bits=mb1, amax=(rec[cnn.REC_DZ2], rec[cnn.REC_DZ1]))
    assert cnn.amax_value(rec[cnn.REC_DZ1]) == dz1.abs().max().item()
    x = a1[sl].double().requires_grad_(True)
    _conv64(x, W2.double(), None, 2).backward(dz2[sl].double())
    _close(dz1[sl], x.grad * (a1[sl] > 0), "conv2 dgrad f16x2 at 32768")
    assert ((a1 > 0) | (dz1 == 0)).all()
    del x
    dW2, _ = cnn.conv_wgrad(a1, dz2, 2, amax=(rec[cnn.REC_A1], rec[cnn.REC_DZ2]))
    dW2b, _ = cnn.conv_wgrad(a1, dz2, 2)
    # (a float64 weight gradient over 32,768 x 81 pixels on the device: 1.7 GB of float64 im2col columns per 2,048 images -- in chunks)
    ref = torch.zeros(64, 32, 4, 4, dtype=torch.float64, device=DEV)
    for lo in range(0, M, 2048):
        ref += torch.nn.grad.conv2d_weight(a1[lo:lo + 2048].double().permute(0, 3, 1, 2), (64, 32, 4, 4), dz2[lo:lo + 2048].double().permute(0, 3, 1, 2), stride=2)
    e_h, e_b = _close(dW2, ref, "conv2 wgrad f16x2 at 32768"), _close(dW2b, ref, "conv2 wgrad bf16x3 at 32768")
    assert e_h <= max(4.0 * e_b, 2e-6), (e_h, e_b)


def test_nature_packs_f16x2_equal_the_per_matrix_route():
    """``mi355ppo_nature_packs_f16x2_f32`` (through ``_Buffers._repack_all``) against repack_weights + fc_pack_f16x2 per matrix; the
    weights' records = their exact maxima."""
    torch.manual_seed(3)
    W1, W2, W3 = (torch.randn(s, device=DEV) * 0.05 for s in ((32, 4, 8, 8), (64, 32, 4, 4), (64, 64, 3, 3)))
    Wfc = torch.randn(512, 3136, device=DEV) * 0.02
    bufs = cnn._Buffers()
    bufs.cache_weights, bufs.pack_params = True, (W1, W2, W3, Wfc)
    assert bufs.f16(W2)
    got = {"c2f": bufs.conv_zpack(W2, 2, cnn.MODE_FWD), "c3f": bufs.conv_zpack(W3, 3, cnn.MODE_FWD), "c3d": bufs.conv_zpack(W3, 3, cnn.MODE_DGRAD_S1),
           "c2d": bufs.conv_zpack(W2, 2, cnn.MODE_DGRAD_S2), "fcf": bufs.fc_pack_fwd(Wfc), "fcd": bufs.fc_pack_dgrad(Wfc)}
    ref = {"c2f": cnn.conv_zpack_f16x2(W2, 2, cnn.MODE_FWD), "c3f": cnn.conv_zpack_f16x2(W3, 3, cnn.MODE_FWD),
           "c3d": cnn.conv_zpack_f16x2(W3, 3, cnn.MODE_DGRAD_S1), "c2d": cnn.conv_zpack_f16x2(W2, 2, cnn.MODE_DGRAD_S2),
           "fcf": cnn.fc_pack_f16x2(cnn.fc_weight_hwc(Wfc).contiguous()), "fcd": cnn.fc_pack_f16x2(cnn.fc_weight_hwc(Wfc).t().contiguous())}
    for k in ref:
        assert torch.equal(got[k], ref[k]), k
    assert [cnn.amax_value(bufs.w_amax[i]) for i in range(3)] == [W2.abs().max().item(), W3.abs().max().item(), Wfc.abs().max().item()]
    assert torch.equal(bufs.weights(W1, 1, cnn.MODE_FWD_Q).view(torch.uint8), cnn.repack_weights(W1, 1, cnn.MODE_FWD_Q).view(torch.uint8))


def test_trunk_autograd_under_f16x2_against_float64():
    """The learner's route: ``NatureTrunk`` -> ``LinearReLUHwcFn`` -> ``HeadsFn`` forward and backward with every amax record produced and
    consumed by the kernels themselves, against float64 autograd of the reference layers (ppo_atari_multigpu.py:136-149)."""
    from types import SimpleNamespace

    from cleanrl_amd import envs as E
    from cleanrl_amd.agents import AtariAgent

    torch.manual_seed(5)
    spaces = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    agent = AtariAgent(spaces).to(DEV)
    M = 512
    obs = torch.randint(0, 256, (M, 84, 84, 4), dtype=torch.uint8, device=DEV)
    logits, value = agent.heads_u8(obs)
    assert agent._trunk.bufs.f16(obs)
    gl, gv = torch.randn_like(logits) / M, torch.randn_like(value) / M
    for p in agent.parameters():
        p.grad = None
    torch.autograd.backward([logits, value], [gl, gv])
    got = {n: p.grad.clone() for n, p in agent.named_parameters()}
    ref_agent = AtariAgent(spaces).to(DEV).double()
    ref_agent.load_state_dict({k: v.double() for k, v in agent.state_dict().items()})
    l64, v64 = ref_agent.heads(obs.double().permute(0, 3, 1, 2) / 255.0)
    _close(logits, l64, "logits under f16x2")
    _close(value, v64, "value under f16x2")
    torch.autograd.backward([l64, v64], [gl.double(), gv.double()])
    bad = []
    for n, p in ref_agent.named_parameters():
        scale = p.grad.abs().max().item()
        err = (got[n].double() - p.grad).abs().max().item()
        if not (np.isfinite(err) and err <= 5e-5 * scale):
            bad.append(f"grad of {n}: err {err:.3e}, scale {scale:.3e}")
    assert not bad, "; ".join(bad)


@pytest.mark.parametrize("images", [32768, 3073, 3074, 3075])
def test_kernels_r_and_g_equal_kernel_z_bit_for_bit_at_the_bench_size(images):
    """Every tensor of one minibatch update at BASELINE configs[2]'s size (32,768 images; and at 3,073 / 3,074 / 3,075: kernel RB's three-image groups
    with a last group of one, two and three images) hashed on the torch-free driver (tools/conv_traffic:
    order-independent 64-bit hash of the bit patterns) with kernels R / RB / G switched off (kernel Z: the oracle kernel) and on (the default): the
    forwards, the data gradients and -- fed by them -- every weight gradient must come out bit-identical.  (Round-5 review: this comparison
    lived outside the suite.)"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "conv_traffic")
    if not os.path.exists(exe):
        pytest.skip("tools/conv_traffic not built (python -m cleanrl_amd.build)")

    def hashes(extra):
        env = {k: v for k, v in os.environ.items() if not k.startswith("MI355PPO_")}
        env.update({"CONV_TRAFFIC_HASH": "1", "CONV_TRAFFIC_F16": "1"}, **extra)
        out = subprocess.run([exe, str(images), "1"], cwd=root, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return dict(ln.split()[1:3] for ln in out.stdout.splitlines() if ln.startswith("hash "))

    z = hashes({"MI355PPO_CONV_R": "0", "MI355PPO_FC_G": "0"})
    d = hashes({})
    assert len(z) >= 14 and set(z) == set(d)
    assert z == d, {k: (z[k], d[k]) for k in z if z[k] != d[k]}
