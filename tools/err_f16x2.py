"""CPU study (numpy, no GPU): error against float64 of an f32 GEMM carried out on a 16-bit matrix pipe with f32 accumulation,
for the operand splits kernel Z could use -- the one it uses (three bf16 terms, six term pairs) and the two-term f16 split the
round-4 review asks for (hi = f16(s x), lo = f16(s x - hi), products hi hi + hi lo + lo hi; s = a power of two per tensor that puts
the tensor's largest magnitude in [2^14, 2^15)), with the f16 conversions rounding to nearest or truncating (v_cvt_pkrtz_f16_f32).
The matrix instruction is modelled as: the 16 products of a k-step exact, their sum exact, one rounding into the f32 accumulator
(term pairs in sequence within a k-step) -- the same model for every row, so the comparison between rows is what carries.
Operands as in tools/err_pairs.py (post-ReLU log-normal activations; weights ~ N(0, 1 / K)), plus a gradient-like operand with a
wide dynamic range.  One JSON line per (case, split) into profiles/r05_err_f16x2.jsonl.

Reference arithmetic being matched: cleanrl/ppo_atari_multigpu.py:136-148,358 (f32 Conv2d / Linear forward + backward)."""
import json
import sys

import numpy as np


def split_bf16x3(x):
    xb = x.view(np.uint32)
    t8 = (xb & np.uint32(0xFFFF0000)).view(np.float32)
    t16 = (xb & np.uint32(0xFFFFFF00)).view(np.float32)
    return [t8, t16 - t8, x - t16]


def f16_rtz(x):
    """f32 -> f16 truncating toward zero (v_cvt_pkrtz_f16_f32), subnormal results included."""
    h = x.astype(np.float16)                      # round to nearest even
    over = np.abs(h.astype(np.float32)) > np.abs(x)
    hb = h.view(np.uint16).copy()
    hb[over] -= 1                                 # one f16 step toward zero (sign-magnitude encoding)
    return hb.view(np.float16)


def pow2_scale(x, top=15):
    m = float(np.abs(x).max())
    if m == 0.0:
        return 1.0
    return float(2.0 ** (top - 1 - int(np.floor(np.log2(m)))))      # s * max in [2^(top-1), 2^top)


def split_f16x2(x, rtz, scale=True, top=15):
    s = np.float32(pow2_scale(x, top) if scale else 1.0)
    xs = x * s
    cv = f16_rtz if rtz else (lambda v: v.astype(np.float16))
    hi = cv(xs).astype(np.float32)
    lo = cv(xs - hi).astype(np.float32)
    return [hi, lo], float(s)


def gemm_terms(at, bt, pairs, kstep=16):
    """C = sum over `pairs` (i, j) of at[i] @ bt[j].T, accumulated the way the kernel does: k-step by k-step, pair by pair, one f32
    rounding per (k-step, pair)."""
    M, K = at[0].shape
    N = bt[0].shape[0]
    acc = np.zeros((M, N), np.float32)
    a64 = [t.astype(np.float64) for t in at]
    b64 = [t.astype(np.float64) for t in bt]
    for s in range(0, K, kstep):
        for (i, j) in pairs:
            acc = (acc.astype(np.float64) + a64[i][:, s:s + kstep] @ b64[j][:, s:s + kstep].T).astype(np.float32)
    return acc


def gemm_f32_seq(a, b, kstep=16):
    """f32 products, f32 running sum in k order (what an f32 FMA loop / the f32 MFMA does up to its internal order)."""
    return gemm_terms([a], [b], [(0, 0)], kstep=1) if a.shape[1] <= 64 else _f32_chunks(a, b)


def _f32_chunks(a, b):
    M, K = a.shape
    acc = np.zeros((M, b.shape[0]), np.float32)
    for k in range(K):
        acc = acc + np.outer(a[:, k], b[:, k]).astype(np.float32)
    return acc


P6 = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
P3 = [(0, 0), (0, 1), (1, 0)]
P4 = [(0, 0), (0, 1), (1, 0), (1, 1)]


def study(name, a, b, out):
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    scale = np.abs(ref).max()

    def row(split, c, extra=None):
        e = np.abs(c.astype(np.float64) - ref)
        r = {"case": name, "split": split, "M": a.shape[0], "K": a.shape[1], "N": b.shape[0], "mean_err": float(e.mean() / scale),
             "max_err": float(e.max() / scale), "bias": float((c.astype(np.float64) - ref).mean() / scale)}
        if extra:
            r.update(extra)
        out.append(r)
        print(json.dumps(r), flush=True)

    row("f32_sequential", _f32_chunks(a, b))
    row("f32_blas", a @ b.T)
    row("bf16x3_6pairs", gemm_terms(split_bf16x3(a), split_bf16x3(b), P6))
    for rtz in (False, True):
        (at, sa), (bt, sb) = split_f16x2(a, rtz), split_f16x2(b, rtz)
        c = gemm_terms(at, bt, P3) * np.float32(1.0 / (sa * sb))
        row("f16x2_3pairs_" + ("rtz" if rtz else "rtn"), c, {"scale_a": sa, "scale_b": sb})
        c = gemm_terms(at, bt, P4) * np.float32(1.0 / (sa * sb))
        row("f16x2_4pairs_" + ("rtz" if rtz else "rtn"), c)
    (at, sa), (bt, sb) = split_f16x2(a, False, scale=False), split_f16x2(b, False, scale=False)
    row("f16x2_3pairs_rtn_unscaled", gemm_terms(at, bt, P3))
    # mixed: A truncated in the kernel (one packed convert), B (packed ahead, off the hot path) rounded to nearest
    (at, sa), (bt, sb) = split_f16x2(a, True), split_f16x2(b, False)
    row("f16x2_3pairs_A_rtz_B_rtn", gemm_terms(at, bt, P3) * np.float32(1.0 / (sa * sb)))


def main():
    rng = np.random.default_rng(5)
    out = []
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    # FC forward: post-ReLU log-normal activations x weights (tools/err_pairs.py's operands)
    a = (np.maximum(rng.standard_normal((M, 3136)), 0) * np.exp(rng.standard_normal((M, 3136)))).astype(np.float32)
    W = (rng.standard_normal((512, 3136)) / 56.0).astype(np.float32)
    study("fc_fwd K=3136", a, W, out)
    # conv2 forward as a GEMM over K = 512
    a2 = (np.maximum(rng.standard_normal((4 * M, 512)), 0) * np.exp(rng.standard_normal((4 * M, 512)))).astype(np.float32)
    W2 = (rng.standard_normal((64, 512)) / 512 ** 0.5).astype(np.float32)
    study("conv2_fwd K=512", a2, W2, out)
    # a data gradient: dz with a wide dynamic range (per-row scales spread over 2^-20 .. 1, 60 % zeros from the ReLU mask) x weights
    dz = (rng.standard_normal((4 * M, 576)) * np.exp2(-20 * rng.random((4 * M, 1))) * (rng.random((4 * M, 576)) < 0.4) * 1e-3).astype(np.float32)
    W3 = (rng.standard_normal((64, 576)) / 24.0).astype(np.float32)
    study("conv3_dgrad K=576 wide-range dz", dz, W3, out)
    # a weight gradient: both operands activations, reduction over the batch (K = rows); 2,048 rows of the slab
    dzw = (rng.standard_normal((64, 2048)) * np.exp2(-12 * rng.random((1, 2048))) * 1e-3).astype(np.float32)
    aw = (np.maximum(rng.standard_normal((576, 2048)), 0) * np.exp(rng.standard_normal((576, 2048)))).astype(np.float32)
    study("conv3_wgrad K=2048 (batch)", dzw, aw, out)
    with open("profiles/r05_err_f16x2.jsonl", "w") as f:
        for r in out:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
