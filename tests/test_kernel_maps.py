"""Lane-level emulation (numpy, CPU) of the register-streaming weight-gradient kernel ``conv_wgrad_taps_kernel``
(cleanrl_amd/csrc/conv.hip, kernel T): who loads which element, which MFMA tile it feeds and where the tile's columns land
in dW.  The kernel itself is checked on the GPU against float64 convolutions (tests/test_gpu_cnn.py); this restates its
INDEX MAPS -- including the staged paired-load variant of layer 2, which has not run on a GPU yet -- so that a mapping
mistake shows up here, on the CPU, as a wrong weight gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

GEOM = {2: dict(H=20, W=20, Cin=32, KH=4, KW=4, SS=2, GY=9, GX=9, KHW=2, CSPLIT=1),
        3: dict(H=9, W=9, Cin=64, KH=3, KW=3, SS=1, GY=7, GX=7, KHW=3, CSPLIT=2)}


def _emulate(layer, src, dz, pair_loads):
    """src (images, H, W, Cin), dz (images, GY, GX, 64) -> dW (64, Cin, KH, KW), one workgroup, as the kernel walks it."""
    g = GEOM[layer]
    H, W, Cin, KH, KW, SS, GY, GX, KHW, CSPLIT = (g[k] for k in ("H", "W", "Cin", "KH", "KW", "SS", "GY", "GX", "KHW", "CSPLIT"))
    images, K = src.shape[0], KH * KW * Cin
    npairs = (images + 1) // 2
    dWt = np.zeros((64, K))                                   # [cout][(kh, kw, cin)]: the partial the kernel writes
    li = np.arange(32)
    for wave in range(4):
        ci, sub = wave & 1, wave >> 1
        kh0 = sub * KHW if CSPLIT == 1 else 0
        c0 = sub * 32 if CSPLIT == 2 else 0
        acc = np.zeros((KHW * KW, 32, 32))                    # [tile][i = cout within ci][j = lane li]
        for pair in range(npairs):
            for s in range(GY * GX):
                gy, gx = divmod(s, GX)
                # A operand: lane (li, lh) holds dz[image 2*pair + lh][pixel s][ci*32 + li]; 0 for a missing second image
                a = np.zeros((32, 2))
                for lh in range(2):
                    img = 2 * pair + lh
                    if img < images:
                        a[:, lh] = dz[img, gy, gx, ci * 32 + li]
                for r in range(KHW):
                    y = gy * SS + kh0 + r
                    if pair_loads:                            # one 8-byte load per lane covers two source columns
                        for pp in range(2):
                            for which in range(2):
                                b = np.zeros((2, 32))
                                for lh in range(2):
                                    img = min(2 * pair + lh, images - 1)
                                    flat = src[img, y, 2 * (gx + pp): 2 * (gx + pp) + 2, :].reshape(-1)     # 64 floats
                                    b[lh] = flat[2 * li + which]
                                acc[r * KW + 2 * pp + which] += a @ b
                    else:
                        for c in range(KW):
                            b = np.zeros((2, 32))
                            for lh in range(2):
                                img = min(2 * pair + lh, images - 1)
                                b[lh] = src[img, y, gx * SS + c, c0 + li]
                            acc[r * KW + c] += a @ b
        for t in range(KHW * KW):                             # partial write: tile t -> dW columns
            if pair_loads:
                kcol = ((kh0 + t // KW) * KW + 2 * ((t % KW) >> 1) + (li >> 4)) * Cin + 2 * (li & 15) + (t & 1)
            else:
                kcol = ((kh0 + t // KW) * KW + t % KW) * Cin + c0 + li
            dWt[ci * 32:(ci + 1) * 32, kcol] = acc[t]
    return dWt.reshape(64, KH, KW, Cin).transpose(0, 3, 1, 2)


@pytest.mark.parametrize("layer,pair_loads,images", [(2, False, 3), (2, True, 3), (2, True, 2), (3, False, 3)])
def test_kernel_T_index_maps_give_the_weight_gradient(layer, pair_loads, images):
    g = GEOM[layer]
    rs = np.random.RandomState(layer * 10 + images + int(pair_loads))
    src = rs.standard_normal((images, g["H"], g["W"], g["Cin"]))
    dz = rs.standard_normal((images, g["GY"], g["GX"], 64))
    x = torch.from_numpy(src).permute(0, 3, 1, 2)
    W = torch.zeros(64, g["Cin"], g["KH"], g["KW"], dtype=torch.float64, requires_grad=True)
    out = F.conv2d(x, W, None, stride=g["SS"])
    (ref,) = torch.autograd.grad(out, W, torch.from_numpy(dz).permute(0, 3, 1, 2))
    got = _emulate(layer, src, dz, pair_loads)
    assert np.abs(got - ref.numpy()).max() <= 1e-9 * np.abs(ref.numpy()).max()


# ---------------------------------------------------------------------------------------------------------------------
# Kernel P (cleanrl_amd/csrc/conv1p.hip): layer-1 weight gradient on the bf16 pipe.  Index maps restated: the transposed
# LDS staging (4-chunk quads + the q = 20 chunk), the 15 pixel groups -> 8 steps, the dz window / slot mapping of the
# short group, the shifted read for tap columns 4..7, the three-term bf16 split, and the partial's column order.
def _p_row(g): return g // 3 if g < 15 else 4
def _p_ox0(g): return 8 * (g % 3) if g < 15 else 16
def _p_nv(g): return (8 if g % 3 < 2 else 4) if g < 15 else 0


def _bf16_terms(x):
    """x (f32) -> (hi, mid, lo) as the kernel forms them: truncations to the top 16 bits of x, x - hi, (x - hi) - mid."""
    x = np.asarray(x, np.float32)
    trunc = lambda v: (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    hi = trunc(x)
    r = (x - hi).astype(np.float32)
    mid = trunc(r)
    lo = (r - mid).astype(np.float32)
    assert np.array_equal(trunc(lo), lo), "the third term must be a bf16 value"
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    return hi, mid, lo


def _emulate_kernel_p(frames, dz):
    """frames (images, 84, 84, 4) uint8, dz (images, 20, 20, 32) f32 -> partial [32][256] (float64 accumulation of the exact
    products), one workgroup of four waves."""
    images = frames.shape[0]
    part = np.zeros((32, 256))
    LINE, ROWLDS = 32, 512
    for wave in range(4):
        acc = np.zeros((8, 32, 32))                               # [tap row r][channel n][column li = kw*4 + c]
        tt = np.full(24 * ROWLDS, 0xEE, np.uint8)                                    # never-written bytes must not matter

        def stage_piece(img, P):
            """Piece P of image `img`'s slab = source rows 8P .. 8P+7 -> the wave's LDS lines (40 quads + the 8 chunks q = 20)."""
            slab = frames[img].reshape(84, 336)[20 * wave:20 * wave + 24]            # 24 source rows of 336 bytes
            for ident in range(40):                                                  # quads: q = 4Q .. 4Q+3 of source row R
                R, Q = 8 * P + ident // 5, ident % 5
                st = [slab[R, (4 * Q + k) * 16:(4 * Q + k) * 16 + 16].reshape(4, 4) for k in range(4)]   # [chunk k][pixel t][channel]
                for t in range(4):
                    for ch in range(4):
                        base = R * ROWLDS + (t * 4 + ch) * LINE + 4 * Q
                        tt[base:base + 4] = [st[k][t, ch] for k in range(4)]
            for R in range(8 * P, 8 * P + 8):                                        # the q = 20 chunk: byte 0 of a dword store
                st20 = slab[R, 320:336].reshape(4, 4)
                for t in range(4):
                    for ch in range(4):
                        base = R * ROWLDS + (t * 4 + ch) * LINE + 20
                        tt[base:base + 4] = [st20[t, ch], 0, 0, 0]

        for P in range(3):
            stage_piece(0, P)
        for img in range(images):
            nxt = min(img + 1, images - 1)
            for s in range(8):
                a_terms = np.zeros((3, 32, 16))                   # [term][channel n][slot]
                b_vals = np.zeros((8, 32, 16))                    # [tap row][column li][slot]
                for lh in range(2):
                    g = 2 * s + lh
                    row, ox0, nv = _p_row(g), _p_ox0(g), _p_nv(g)
                    lox = ox0 if nv == 8 else 12
                    orow = 5 * wave + row
                    ring = dz[img, orow, lox:lox + 8, :]          # (8 pixels, 32 channels): all inside the row
                    for j in range(8):
                        if nv == 8:
                            x = ring[j]
                        elif nv == 4 and j < 4:
                            x = ring[j + 4]
                        else:
                            x = np.zeros(32, np.float32)
                        for term, v in enumerate(_bf16_terms(x)):
                            a_terms[term, :, 8 * lh + j] = v
                    for li in range(32):
                        kw, c = li >> 2, li & 3
                        line = ((kw & 3) * 4 + c) * LINE
                        sh = kw >> 2
                        for r in range(8):
                            lp = (4 * row + r) * ROWLDS + line + ox0
                            w = tt[lp:lp + 12]
                            b_vals[r, li, 8 * lh:8 * lh + 8] = w[sh:sh + 8]
                for r in range(8):
                    for term in range(3):
                        acc[r] += a_terms[term] @ b_vals[r].T
                # the NEXT image's slab replaces the LDS rows piece by piece, as soon as their last reader of this image has issued
                # (conv1p.hip: after steps 2, 5 and 7; LDS instructions of one wave execute in order)
                if s in (2, 5, 7):
                    stage_piece(nxt, (2, 5, 7).index(s))
        for r in range(8):
            part[:, r * 32:(r + 1) * 32] += acc[r]
    return part


def test_kernel_p_index_maps_give_the_layer1_weight_gradient():
    rs = np.random.RandomState(5)
    images = 3                                                    # first image, a middle one, the last (whose prefetch re-reads itself)
    frames = rs.randint(0, 256, size=(images, 84, 84, 4)).astype(np.uint8)
    dz = (rs.standard_normal((images, 20, 20, 32)) * np.exp(rs.uniform(-8, 2, size=(images, 20, 20, 32)))).astype(np.float32)
    part = _emulate_kernel_p(frames, dz)
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).double()
    ref = torch.nn.grad.conv2d_weight(x, (32, 4, 8, 8), torch.from_numpy(dz).permute(0, 3, 1, 2).double(), stride=4)   # (n, c, r, kw)
    ref = ref.permute(0, 2, 3, 1).reshape(32, 256).numpy()                                                             # [n][(r, kw, c)]
    assert np.abs(part - ref).max() <= 1e-9 * np.abs(ref).max()


def test_kernel_z_coalesced_load_lds_transposition_delivers_the_mfma_fragments():
    """Kernel Z (csrc/gemmz.hip) lane by lane: load u of a k-step reads row 16 u + (lane >> 2), floats 4 (lane & 3) .. + 3 of the
    step's 16 (four lanes = one row's 64 contiguous bytes), writes them to the wave's LDS tile at (row * 20 + 4 (lane & 3))
    floats; fragment i is read back at ((32 i + lane % 32) * 20 + 8 (lane // 32)) floats, 8 floats: lane (li, lh) must hold
    A[32 i + li][8 lh .. 8 lh + 7] of the step -- the A operand layout of v_mfma_f32_32x32x16_bf16.  Also: the 80-byte row pitch
    keeps every 16-byte LDS access of a wave in distinct bank groups per hardware lane group (MI355X_MICROARCH.md, LDS table)."""
    MT, PITCH = 2, 20
    rows = 32 * MT
    rs = np.random.RandomState(3)
    A = rs.standard_normal((rows, 16))                       # one k-step of the wave's A rows
    lds = np.full(rows * PITCH, np.nan)
    for lane in range(64):
        for u in range(rows // 16):
            row, c = 16 * u + (lane >> 2), lane & 3
            lds[row * PITCH + 4 * c: row * PITCH + 4 * c + 4] = A[row, 4 * c: 4 * c + 4]
    for i in range(MT):
        for lane in range(64):
            li, lh = lane & 31, lane >> 5
            base = (32 * i + li) * PITCH + 8 * lh
            assert np.array_equal(lds[base: base + 8], A[32 * i + li, 8 * lh: 8 * lh + 8])
    # bank check (64 banks of 4 bytes): ds_read_b128 is serviced in the lane groups below, ds_write_b128 in groups of 8 lanes
    read_groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    read_groups += [[x + 32 for x in grp] for grp in read_groups]
    for grp in read_groups:
        banks = [((lane & 31) * PITCH + 8 * (lane >> 5) + d) % 64 for lane in grp for d in range(4)]
        assert len(set(banks)) == len(banks) == 64
    for g8 in range(8):
        banks = [((lane >> 2) * PITCH + 4 * (lane & 3) + d) % 64 for lane in range(8 * g8, 8 * g8 + 8) for d in range(4)]
        assert len(set(banks)) == len(banks) == 32


# ---------------------------------------------------------------------------------------------------------------------
# Kernel Z, border-class rows of the data gradients (csrc/gemmz.hip: ZAxisDgrad3 / ZAxisDgrad2, ZClassOrder, cls_pixel, the tap
# cursors): emulated row by row -- image groups, class order, row -> (image, grid pixel) with the kernel's multiply-high division, the class's
# valid tap window -- against the float64 transposed convolution.  The tables are read from the source, so the test follows it.
def _axis_tables(name):
    import os
    import re

    src = open(os.path.join(os.path.dirname(__file__), "..", "cleanrl_amd", "csrc", "gemmz.hip")).read()
    body = re.search(r"struct %s \{[^}]*?static constexpr int NC = (\d+), (.*?);\s*\};" % name, src, re.S)
    nc = int(body.group(1))
    arrs = {k: [int(x) for x in v.split(",")] for k, v in re.findall(r"(\w+)\[\d+\] = \{([^}]*)\}", body.group(2))}
    assert all(len(arrs[k]) == nc for k in ("G0", "NG", "T0", "T1"))
    return nc, arrs


def _umulhi_div(r, d):
    """The kernel's r / d: multiply-high by floor(2^32 / d) + 1 (d > 1), exact while r * d < 2^32."""
    if d == 1:
        return r
    magic = ((1 << 32) // d + 1) & 0xFFFFFFFF
    return (r * magic) >> 32


@pytest.mark.parametrize("layer,images", [(3, 1), (3, 70), (2, 3)])
def test_kernel_z_border_class_rows_give_the_data_gradient(layer, images):
    H, C, KT, G, OFF, DH, DM, name = ((7, 64, 3, 9, -2, 9, 1, "ZAxisDgrad3") if layer == 3 else (9, 64, 2, 10, -1, 20, 2, "ZAxisDgrad2"))
    NC, ax = _axis_tables(name)
    N = 64 if layer == 3 else 128                          # layer 2: 4 stride-parity classes x 32 input channels
    rs = np.random.RandomState(layer)
    dz = rs.standard_normal((images, H, H, C))             # the GEMM's source: the incoming gradient, channels last
    Bm = rs.standard_normal((N, KT, KT, C))                # B[n][(r, c, cout)]: the repacked (flipped) weights
    ROWS = 64
    weight = lambda k: ax["NG"][k // NC] * ax["NG"][k % NC] * (ax["T1"][k // NC] - ax["T0"][k // NC]) * (ax["T1"][k % NC] - ax["T0"][k % NC])
    order = list(range(NC * NC))
    for i in range(len(order)):                            # ZClassOrder's selection sort, verbatim (ties keep table order)
        for j in range(i + 1, len(order)):
            if weight(order[j]) > weight(order[i]):
                order[i], order[j] = order[j], order[i]
    out = np.full((images, G, G, N), np.nan)
    tiles = 0
    for grp in range((images + ROWS - 1) // ROWS):          # a group = ROWS images = G * G tiles; classes heaviest first inside it
        group_tiles = 0
        for c in order:
            cy, cx = c // NC, c % NC
            nx, npix = ax["NG"][cx], ax["NG"][cy] * ax["NG"][cx]
            group_tiles += npix                             # the class owns npix whole tiles of the group
            for r in range(ROWS * npix):
                loc = _umulhi_div(r, npix)
                img = grp * ROWS + loc
                if img >= images:                           # last group: rows of images past the batch are dropped at the store
                    continue
                p = r - loc * npix
                py = _umulhi_div(p, nx)
                gy, gx = ax["G0"][cy] + py, ax["G0"][cx] + (p - py * nx)
                acc = np.zeros(N)
                for ty in range(ax["T0"][cy], ax["T1"][cy]):   # the cursor's walk: valid tap rows x valid tap columns (x channel chunks)
                    for tx in range(ax["T0"][cx], ax["T1"][cx]):
                        sy, sx = gy + OFF + ty, gx + OFF + tx
                        assert 0 <= sy < H and 0 <= sx < H     # every tap of the class window is valid for every row of the class
                        acc += Bm[:, ty, tx, :] @ dz[img, sy, sx, :]
                assert np.isnan(out[img, gy, gx, 0])            # every (image, pixel) exactly once
                out[img, gy, gx] = acc
        assert group_tiles == G * G
        tiles += group_tiles
    assert not np.isnan(out).any()
    # the same sum over ALL taps with zero padding (what the un-classed kernel multiplied)
    ref = np.zeros_like(out)
    for ty in range(KT):
        for tx in range(KT):
            for gy in range(G):
                for gx in range(G):
                    sy, sx = gy + OFF + ty, gx + OFF + tx
                    if 0 <= sy < H and 0 <= sx < H:
                        ref[:, gy, gx, :] += dz[:, sy, sx, :] @ Bm[:, ty, tx, :].T
    assert np.abs(out - ref).max() <= 1e-9 * np.abs(ref).max()
    valid = sum(weight(k) for k in range(NC * NC))
    assert valid * (1.65 if layer == 3 else 1.23) == pytest.approx(G * G * KT * KT, rel=0.01)      # the padded windows' MFMA overhead
    assert tiles == (images + ROWS - 1) // ROWS * G * G


@pytest.mark.parametrize("total", [1, 7, 8, 9, 10, 63, 64, 65, 1000, 5184])
def test_kernel_z_xcd_aware_workgroup_order_is_a_bijection_with_contiguous_ranges(total):
    """gemmz.hip's workgroup remap: launch index L -> XCD L % 8 (round-robin dispatch) runs logical index start[x] + L // 8 -- every
    logical index exactly once, and XCD x owns one contiguous range."""
    q, rem = total >> 3, total & 7
    per_xcd = {}
    for L in range(total):
        x = L & 7
        logical = x * q + min(x, rem) + (L >> 3)
        per_xcd.setdefault(x, []).append(logical)
    allv = sorted(v for vs in per_xcd.values() for v in vs)
    assert allv == list(range(total))
    for x, vs in per_xcd.items():
        assert vs == list(range(vs[0], vs[0] + len(vs)))


def _z_supertile_order(gx, gy, super_rows):
    """The workgroup -> (column block, row block) map of kernel Z for GEMM rows with stacked waves (gemmz.hip, the FC data
    gradient), restated: launch index L runs on XCD L % 8; XCD x takes a contiguous range of the logical order; the logical
    order walks supertiles of `super_rows` row blocks x all column blocks, row block fastest."""
    out = []
    total = gx * gy
    for L in range(total):
        x, q, rem = L & 7, total >> 3, total & 7
        logical = x * q + min(x, rem) + (L >> 3)
        per = super_rows * gx
        sup, r = divmod(logical, per)
        rows = min(gy - sup * super_rows, super_rows)
        bx = r // rows
        by = sup * super_rows + (r - bx * rows)
        out.append((L & 7, logical, bx, by))
    return out


@pytest.mark.parametrize("gx,gy", [(25, 128), (25, 16), (25, 32), (25, 1), (25, 3), (7, 5), (1, 9), (25, 129)])
def test_kernel_z_supertile_order_is_a_bijection_and_keeps_a_supertile_on_one_xcd(gx, gy):
    order = _z_supertile_order(gx, gy, 4)
    tiles = {(bx, by) for _, _, bx, by in order}
    assert tiles == {(bx, by) for bx in range(gx) for by in range(gy)}          # every tile exactly once
    assert len(order) == gx * gy
    if (gx * gy) % 8 == 0 and (gx * gy // 8) % (4 * gx) == 0:                    # (config C: 3,200 workgroups = 8 x 4 supertiles)
        for xcd in range(8):
            sups = {by // 4 for x, _, _, by in order if x == xcd}
            assert len(sups) == gy // 4 // 8                                    # whole supertiles, none shared between XCDs
        # consecutive workgroups of an XCD (its launch order) walk a supertile row block fastest: 32 in flight = 8 column blocks
        mine = sorted((logical, bx, by) for x, logical, bx, by in order if x == 0)
        first32 = mine[:32]
        assert {by for _, _, by in first32} == {0, 1, 2, 3} and {bx for _, bx, _ in first32} == set(range(8))
