"""GPU parity tests (run with `-m gpu` on an MI355X): every HIP kernel, called through the C ABI via
cleanrl_amd.ops, against (a) the committed goldens minted from the reference's own lines and (b) the
oracle on seeded inputs.  Tolerances are stated per test.  /root/reference is never touched here."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from cleanrl_amd import ops, synthetic
from oracle import c_oracle, torch_oracle as TO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def G(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


# ================================================================================== K1  GAE
@pytest.mark.parametrize("variant", [0, 1, 3, 6])
@pytest.mark.parametrize("case", sorted(load_golden("gae")))
def test_gae_goldens_bit_exact(case, variant):
    g = load_golden("gae")[case]
    T_, N = g["rewards"].shape
    if variant == 3 and N % 4 != 0:
        pytest.skip("float4 variant needs N % 4 == 0")
    adv, ret = ops.gae(G(g["rewards"]), G(g["dones"]), G(g["values"]), G(g["next_done"]), G(g["next_value"]),
                       float(g["gamma"]), float(g["gae_lambda"]), variant=variant)
    assert np.array_equal(adv.cpu().numpy(), g["advantages"]), "advantages differ bitwise from the reference"
    assert np.array_equal(ret.cpu().numpy(), g["returns"]), "returns differ bitwise from the reference"


@pytest.mark.parametrize("T_,N", [(1, 1), (2, 3), (63, 17), (64, 16), (65, 15), (129, 33), (128, 1024), (128, 1023),
                                  (257, 4100), (37, 16384), (16, 65536), (128, 262144), (2048, 64)])
def test_gae_vs_c_oracle_bit_exact_all_variants(T_, N):
    s = synthetic.rollout_scalars(T_, N, 4, seed=T_ + N, done_p=0.05)
    s["next_done"] = (torch.rand(N) < 0.3).float()
    adv_o, ret_o = c_oracle.gae(*(s[k].numpy() for k in ("rewards", "dones", "values", "next_done", "next_value")),
                                0.99, 0.95)
    args = [s[k].to(DEV) for k in ("rewards", "dones", "values", "next_done", "next_value")]
    for variant in [0, 1, 3, 6]:
        if variant == 3 and N % 4 != 0:
            continue
        adv, ret = ops.gae(*args, 0.99, 0.95, variant=variant)
        assert np.array_equal(adv.cpu().numpy(), adv_o), f"variant {variant}"
        assert np.array_equal(ret.cpu().numpy(), ret_o), f"variant {variant}"


def test_gae_properties_at_scale():
    """Size-independent properties at a size the CPU oracle would not finish quickly (128 x 2^21)."""
    T_, N = 128, 1 << 21
    g = torch.Generator(device=DEV).manual_seed(3)
    rewards = torch.randn(T_, N, device=DEV, generator=g)
    values = torch.randn(T_, N, device=DEV, generator=g)
    dones = (torch.rand(T_, N, device=DEV, generator=g) < 0.01).float()
    nd = torch.zeros(N, device=DEV)
    nv = torch.randn(N, device=DEV, generator=g)
    adv, ret = ops.gae(rewards, dones, values, nd, nv, 0.99, 0.95)
    # returns == advantages + values, bitwise (reference :301)
    assert torch.equal(ret, adv + values)
    # lambda = 0 collapses to the one-step TD error, bitwise
    adv0, _ = ops.gae(rewards, dones, values, nd, nv, 0.99, 0.0)
    nextv = torch.cat([values[1:], nv[None]], 0)
    nnt = 1.0 - torch.cat([dones[1:], nd[None]], 0)
    assert torch.equal(adv0, rewards + 0.99 * nextv * nnt - values)
    # column independence: any column slice recomputed alone is bitwise identical
    cols = slice(12345, 12345 + 1000)
    adv_s, _ = ops.gae(rewards[:, cols].contiguous(), dones[:, cols].contiguous(), values[:, cols].contiguous(),
                       nd[cols].contiguous(), nv[cols].contiguous(), 0.99, 0.95)
    assert torch.equal(adv_s, adv[:, cols])
    # an episode boundary cuts the recurrence: rows before a done do not see later rewards
    r2 = rewards.clone()
    r2[100:] += 5.0
    d2 = dones.clone()
    d2[100] = 1.0
    a_ref, _ = ops.gae(rewards, d2, values, nd, nv, 0.99, 0.95)
    a_mod, _ = ops.gae(r2, d2, values, nd, nv, 0.99, 0.95)
    assert torch.equal(a_ref[:99], a_mod[:99])


# ======================================================================== K2  Categorical
@pytest.mark.parametrize("case", sorted(load_golden("categorical")))
def test_categorical_goldens_noise_mode(case):
    g = load_golden("categorical")[case]
    a64, af, lp, ent = ops.categorical_sample(G(g["logits"]), noise_exp1=G(g["noise_exp1"]),
                                              action_f32_out=torch.empty(g["logits"].shape[0], device=DEV))
    a64, af, lp, ent = a64.cpu().numpy(), af.cpu().numpy(), lp.cpu().numpy(), ent.cpu().numpy()
    assert np.array_equal(a64.astype(np.float32), af)
    mism = a64 != g["action"]
    if mism.any():    # device expf vs CPU expf differ by ~1 ulp: the argmax may flip only on a near-tie
        p = g["probs"] / g["noise_exp1"]
        top2 = np.sort(p[mism], axis=-1)[:, -2:]
        assert np.all(top2[:, 1] / top2[:, 0] < 1 + 1e-5)
    assert mism.mean() < 1e-3
    # tolerance: log_prob / entropy rtol 2e-6 + atol 2e-6 (one ulp of exp/log at |x| ~ 1..30)
    np.testing.assert_allclose(lp[~mism], g["logprob"][~mism], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ent, g["entropy"], rtol=2e-6, atol=2e-6)
    lp2, ent2 = ops.categorical_logprob_entropy(G(g["logits"]), G(g["action"]))
    np.testing.assert_allclose(lp2.cpu().numpy(), g["logprob"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ent2.cpu().numpy(), g["entropy"], rtol=2e-6, atol=2e-6)
    lp3, _ = ops.categorical_logprob_entropy(G(g["logits"]), G(g["action"], torch.float32))
    assert torch.equal(lp2, lp3)


@pytest.mark.parametrize("A", [2, 4, 6, 9, 18, 33])
def test_categorical_philox_distribution_and_determinism(A):
    B = 1 << 18
    logits = (torch.randn(A) * 1.5).to(DEV)
    batch = logits[None].expand(B, A).contiguous()
    a1, _, lp1, _ = ops.categorical_sample(batch, seed=7, offset=1)
    a2, _, lp2, _ = ops.categorical_sample(batch, seed=7, offset=1)
    a3, _, _, _ = ops.categorical_sample(batch, seed=7, offset=2)
    assert torch.equal(a1, a2) and torch.equal(lp1, lp2)            # counter-based: reproducible
    assert not torch.equal(a1, a3)                                   # new offset -> new stream
    p = torch.softmax(logits.double(), 0).cpu().numpy()
    counts = np.bincount(a1.cpu().numpy(), minlength=A)
    chi2 = ((counts - B * p) ** 2 / (B * p)).sum()
    from scipy.stats import chi2 as chi2_dist
    assert chi2_dist.sf(chi2, A - 1) > 1e-4, f"chi2={chi2:.1f}"
    # geometry independence: row i of a bigger batch draws the same sample
    a_small, _, _, _ = ops.categorical_sample(batch[:1000].contiguous(), seed=7, offset=1)
    assert torch.equal(a_small, a1[:1000])


# ============================================================================= K2'  Normal
@pytest.mark.parametrize("case", sorted(load_golden("normal")))
def test_normal_goldens_noise_mode(case):
    g = load_golden("normal")[case]
    act, lp, ent = ops.normal_sample(G(g["mean"]), G(g["logstd"]), noise=G(g["noise"]))
    # tolerance: action rtol 1e-6 (device expf for std), log_prob/entropy sums rtol 1e-5, atol 1e-5
    np.testing.assert_allclose(act.cpu().numpy(), g["action"], rtol=1e-6, atol=1e-6)
    lp2, ent2 = ops.normal_logprob_entropy(G(g["mean"]), G(g["logstd"]), G(g["action"]))
    np.testing.assert_allclose(lp2.cpu().numpy(), g["logprob_sum"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent2.cpu().numpy(), g["entropy_sum"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), g["entropy_sum"], rtol=1e-5, atol=1e-5)


def test_normal_philox_moments():
    B, D = 1 << 17, 6
    mean = torch.linspace(-1, 1, D, device=DEV)[None].expand(B, D).contiguous()
    logstd = torch.linspace(-1, 0.5, D, device=DEV)
    act, lp, _ = ops.normal_sample(mean, logstd, seed=3, offset=9)
    z = (act - mean) / logstd.exp()
    assert z.mean().abs().item() < 0.01 and abs(z.std().item() - 1) < 0.01
    assert abs((z ** 4).mean().item() - 3.0) < 0.1                    # kurtosis of a normal
    lp_ref, _ = TO.normal_logprob_entropy(mean.cpu(), logstd.cpu().reshape(1, -1), act.cpu())
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.numpy(), rtol=1e-5, atol=1e-5)
    act2, _, _ = ops.normal_sample(mean, logstd, seed=3, offset=9)
    assert torch.equal(act, act2)


# ================================================================================= K3  loss
SCALARS = list(ops.LOSS_SCALAR_NAMES)


def _kw(g):
    return dict(clip_coef=float(g["clip_coef"]), ent_coef=float(g["ent_coef"]), vf_coef=float(g["vf_coef"]),
                norm_adv=bool(g["norm_adv"]), clip_vloss=bool(g["clip_vloss"]))


@pytest.mark.parametrize("case", sorted(load_golden("loss_categorical")))
def test_loss_categorical_goldens(case):
    g = load_golden("loss_categorical")[case]
    sc, dl, dv = ops.ppo_loss_categorical(G(g["new_logits"]), G(g["new_value"]), G(g["mb_inds"]), G(g["b_actions"]),
                                          G(g["b_logprobs"]), G(g["b_advantages"]), G(g["b_returns"]), G(g["b_values"]),
                                          **_kw(g))
    sc = sc.cpu().numpy()
    # tolerance: scalars rtol 1e-5 (tree vs cascade reductions); grads rtol 1e-4, atol 1e-5 * max|grad|
    for i, k in enumerate(SCALARS):
        np.testing.assert_allclose(sc[i], g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(dl.cpu().numpy(), g["dlogits"], rtol=1e-4, atol=1e-5 * np.abs(g["dlogits"]).max())
    np.testing.assert_allclose(dv.cpu().numpy(), g["dvalue"], rtol=1e-4, atol=1e-5 * np.abs(g["dvalue"]).max())


@pytest.mark.parametrize("case", sorted(load_golden("loss_normal")))
def test_loss_normal_goldens(case):
    g = load_golden("loss_normal")[case]
    sc, dm, dls, dv = ops.ppo_loss_normal(G(g["new_mean"]), G(g["logstd"]), G(g["new_value"]), G(g["mb_inds"]),
                                          G(g["b_actions"]), G(g["b_logprobs"]), G(g["b_advantages"]), G(g["b_returns"]),
                                          G(g["b_values"]), **_kw(g))
    sc = sc.cpu().numpy()
    for i, k in enumerate(SCALARS):
        np.testing.assert_allclose(sc[i], g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    for got, k in ((dm, "dmean"), (dls, "dlogstd"), (dv, "dvalue")):
        np.testing.assert_allclose(got.cpu().numpy(), g[k], rtol=1e-4, atol=1e-5 * np.abs(g[k]).max(), err_msg=k)


@pytest.mark.parametrize("M,A,flags", [(32768, 4, (1, 1)), (8192, 4, (1, 0)), (4096, 6, (0, 1)), (1000, 18, (1, 1)),
                                       (257, 2, (0, 0)), (5, 3, (1, 1)), (131072, 4, (1, 1)), (1, 4, (0, 1)),
                                       (300001, 4, (1, 1)), (1200007, 4, (1, 1)), (900001, 6, (1, 0)), (700003, 18, (1, 1))])
def test_loss_categorical_vs_c_oracle_at_config_sizes(M, A, flags):
    """BASELINE config sizes (C: M=32768 of B=131072, D: 8192, B: 4096) against the scalar C oracle; the ragged large
    sizes cover the persistent row pass (more than one sweep per lane) and the 1024-partial statistics."""
    rs = np.random.RandomState(M + A)
    Bf = 4 * M
    logits = rs.standard_normal((M, A)).astype(np.float32)
    value = rs.standard_normal(M).astype(np.float32)
    inds = rs.permutation(Bf)[:M].astype(np.int64)
    b_actions = rs.randint(0, A, Bf).astype(np.float32)
    b_logprobs = (-np.log(A) + rs.standard_normal(Bf) * 0.3).astype(np.float32)
    b_adv = (rs.standard_normal(Bf) * 2 + 0.5).astype(np.float32)
    b_val = rs.standard_normal(Bf).astype(np.float32)
    b_ret = (b_val + b_adv).astype(np.float32)
    kw = dict(clip_coef=0.1, ent_coef=0.01, vf_coef=0.5, norm_adv=bool(flags[0]), clip_vloss=bool(flags[1]))
    sc_o, dl_o, dv_o = c_oracle.loss_categorical(logits, value, inds, b_actions, b_logprobs, b_adv, b_ret, b_val, **kw)
    args = [G(x) for x in (logits, value, inds, b_actions, b_logprobs, b_adv, b_ret, b_val)]
    sc, dl, dv = ops.ppo_loss_categorical(*args, **kw)
    np.testing.assert_allclose(sc.cpu().numpy(), sc_o, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), dl_o, rtol=1e-4, atol=1e-5 * np.abs(dl_o).max())
    np.testing.assert_allclose(dv.cpu().numpy(), dv_o, rtol=1e-4, atol=1e-5 * np.abs(dv_o).max())
    # deterministic: bitwise identical on a second run
    sc2, dl2, dv2 = ops.ppo_loss_categorical(*args, **kw)
    assert torch.equal(sc, sc2) and torch.equal(dl, dl2) and torch.equal(dv, dv2)
    # identity indices == explicit arange
    ar = torch.arange(M, device=DEV)
    a_id = [args[0], args[1], None] + [t[inds] if False else t for t in args[3:]]
    sc3, dl3, _ = ops.ppo_loss_categorical(*a_id, **kw)
    sc4, dl4, _ = ops.ppo_loss_categorical(args[0], args[1], ar, *args[3:], **kw)
    assert torch.equal(sc3, sc4) and torch.equal(dl3, dl4)
    if flags[0]:
        # statistics hoisted out of the call (one launch for every minibatch of an epoch): same gradients
        md = ops.adv_stats(args[5], args[2], M)
        mean, std = float(b_adv[inds].astype(np.float64).mean()), float(b_adv[inds].astype(np.float64).std(ddof=1)) if M > 1 else float("nan")
        if M > 1:
            np.testing.assert_allclose(md.cpu().numpy(), [[mean, np.float32(std) + np.float32(1e-8)]], rtol=2e-6, atol=1e-7)
        sc5, dl5, dv5 = ops.ppo_loss_categorical(*args, adv_mean_den=md[0], **kw)
        np.testing.assert_allclose(sc5.cpu().numpy(), sc_o, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dl5.cpu().numpy(), dl_o, rtol=1e-4, atol=1e-5 * np.abs(dl_o).max())
        # scalar fold deferred: two calls leave their partial sums in two slots, one launch folds both -> the same bits
        slots = ops.LossSlots(3, DEV)
        none6, dl6, _ = ops.ppo_loss_categorical(*args, adv_mean_den=md[0], slot=(slots, 2), **kw)
        ops.ppo_loss_categorical(*args, slot=(slots, 1), **kw)
        table = torch.zeros(3, 7, device=DEV)
        slots.fold(2, table, first=1)
        assert none6 is None and torch.equal(dl6, dl5)
        assert torch.equal(table[2], sc5) and torch.equal(table[1], sc) and float(table[0].abs().sum()) == 0.0


@pytest.mark.parametrize("M,A,flags", [(32768, 4, (1, 1)), (8192, 4, (1, 0)), (4096, 6, (0, 1)), (1000, 18, (1, 1)), (5, 3, (1, 1)),
                                       (1, 4, (0, 1)), (300001, 4, (1, 1)), (700003, 18, (1, 1))])
def test_loss_categorical_on_packed_behaviour_rows_is_bit_identical(M, A, flags):
    """Round 3: the five behaviour scalars of a row as ONE 32-byte packed row (mi355ppo_batch_pack_f32) -- the packed K3 call
    and the packed advantage statistics run the same arithmetic on the same values, so every output is bit-identical to the
    five-array call (which the tests above hold to the reference's lines and to the C oracle); identity indices included."""
    rs = np.random.RandomState(7 * M + A)
    Bf = 4 * M
    logits = G(rs.standard_normal((M, A)).astype(np.float32))
    value = G(rs.standard_normal(M).astype(np.float32))
    inds = G(rs.permutation(Bf)[:M].astype(np.int64))
    b_actions = G(rs.randint(0, A, Bf).astype(np.float32))
    b_logprobs = G((-np.log(A) + rs.standard_normal(Bf) * 0.3).astype(np.float32))
    b_adv = G((rs.standard_normal(Bf) * 2 + 0.5).astype(np.float32))
    b_val = G(rs.standard_normal(Bf).astype(np.float32))
    b_ret = b_val + b_adv
    kw = dict(clip_coef=0.1, ent_coef=0.01, vf_coef=0.5, norm_adv=bool(flags[0]), clip_vloss=bool(flags[1]))
    pack = ops.batch_pack(b_actions, b_logprobs, b_adv, b_ret, b_val)
    want = torch.zeros(Bf, 8, device=DEV)
    for j, t in enumerate((b_actions, b_logprobs, b_adv, b_ret, b_val)):
        want[:, j] = t
    assert torch.equal(pack, want)                                                  # the row layout of the header
    md = ops.adv_stats(b_adv, inds, M) if flags[0] else None
    if flags[0]:
        assert torch.equal(ops.adv_stats_packed(pack, inds, M), md)
        assert torch.equal(ops.adv_stats_packed(pack, None, M), ops.adv_stats(b_adv, None, M))       # every minibatch of an unpermuted epoch
    row = md[0] if md is not None else None
    sc, dl, dv = ops.ppo_loss_categorical(logits, value, inds, b_actions, b_logprobs, b_adv, b_ret, b_val, adv_mean_den=row, **kw)
    scp, dlp, dvp = ops.ppo_loss_categorical_packed(logits, value, inds, pack, adv_mean_den=row, **kw)
    assert torch.equal(scp, sc) and torch.equal(dlp, dl) and torch.equal(dvp, dv)
    # without a caller-supplied statistics row the wrapper computes it from the packed rows: same bits again
    scq, dlq, _ = ops.ppo_loss_categorical_packed(logits, value, inds, pack, **kw)
    assert torch.equal(scq, sc) and torch.equal(dlq, dl)
    # identity indices; and the deferred scalar fold
    row_id = ops.adv_stats(b_adv[:M], None, M)[0] if flags[0] else None
    sc_i, dl_i, dv_i = ops.ppo_loss_categorical(logits, value, None, b_actions, b_logprobs, b_adv, b_ret, b_val, adv_mean_den=row_id, **kw)
    sc_j, dl_j, dv_j = ops.ppo_loss_categorical_packed(logits, value, None, pack, adv_mean_den=row_id, **kw)
    assert torch.equal(sc_j, sc_i) and torch.equal(dl_j, dl_i) and torch.equal(dv_j, dv_i)
    slots = ops.LossSlots(2, DEV)
    none, dl_s, _ = ops.ppo_loss_categorical_packed(logits, value, inds, pack, adv_mean_den=row, slot=(slots, 1), **kw)
    table = torch.zeros(2, 7, device=DEV)
    slots.fold(1, table, first=1)
    assert none is None and torch.equal(dl_s, dl) and torch.equal(table[1], sc)


def test_loss_autograd_function_matches_torch_autograd_on_device():
    """PPOLossCategorical plugged under a tiny network == the reference op chain differentiated by autograd."""
    torch.manual_seed(0)
    M, A, Bf = 512, 4, 2048
    net = torch.nn.Linear(16, A + 1).to(DEV)
    x = torch.randn(M, 16, device=DEV)
    inds = torch.randperm(Bf, device=DEV)[:M]
    b_actions = torch.randint(0, A, (Bf,), device=DEV).float()
    b_logprobs = -1.4 + 0.2 * torch.randn(Bf, device=DEV)
    b_adv = torch.randn(Bf, device=DEV)
    b_val = torch.randn(Bf, device=DEV)
    b_ret = b_val + b_adv
    out = net(x)
    loss, scalars = ops.PPOLossCategorical.apply(out[:, :A].contiguous(), out[:, A].contiguous(), inds, b_actions,
                                                 b_logprobs, b_adv, b_ret, b_val, 0.1, 0.01, 0.5, True, True)
    loss.backward()
    g_hip = net.weight.grad.clone()
    net.zero_grad()
    out = net(x)
    lp, ent = TO.categorical_logprob_entropy(out[:, :A], b_actions[inds])
    ref = TO.ppo_loss(lp, ent, out[:, A], b_logprobs[inds], b_adv[inds], b_ret[inds], b_val[inds], 0.1, 0.01, 0.5, True, True)
    ref["loss"].backward()
    np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=1e-5)
    np.testing.assert_allclose(g_hip.cpu().numpy(), net.weight.grad.cpu().numpy(), rtol=1e-4,
                               atol=1e-5 * net.weight.grad.abs().max().item())


# ================================================================================== K5  obs
def test_obs_convert_all_256_values_bit_exact_and_gather():
    src = torch.arange(256, dtype=torch.uint8).repeat(2, 3).reshape(2, 768)       # every byte value
    out = ops.obs_u8_to_f32(src.to(DEV))
    assert torch.equal(out.cpu(), src.float() / 255.0), "x/255 must be the correctly rounded quotient"
    out_raw = ops.obs_u8_to_f32(src.to(DEV), scale_255=False)
    assert torch.equal(out_raw.cpu(), src.float())
    frames = torch.from_numpy(synthetic.atari_frames(300, seed=5))
    inds = torch.from_numpy(np.random.RandomState(1).permutation(300)[:77].astype(np.int64))
    got = ops.obs_u8_to_f32(frames.to(DEV), inds.to(DEV))
    assert got.shape == (77, 4, 84, 84)
    assert torch.equal(got.cpu(), frames.float()[inds] / 255.0)                   # reference: b_obs[mb_inds]; x/255.0
    ref_c = c_oracle.obs_u8_to_f32(frames.numpy(), inds.numpy())
    assert np.array_equal(got.cpu().numpy(), ref_c)
    # ragged row length (not a multiple of the 1024-dword workgroup tile), identity indices
    odd = torch.randint(0, 256, (5, 4 * 1237), dtype=torch.uint8)
    assert torch.equal(ops.obs_u8_to_f32(odd.to(DEV)).cpu(), odd.float() / 255.0)


def test_obs_convert_full_minibatch_roundtrip_property():
    """Config C minibatch (32768 rows of 28,224 B): u8 -> f32 -> round(x*255) is the identity, and a
    checksum over the gathered rows equals the checksum of the source rows."""
    R, M = 4096, 32768
    g = torch.Generator(device=DEV).manual_seed(0)
    src = torch.randint(0, 256, (R, 4, 84, 84), dtype=torch.uint8, device=DEV, generator=g)
    inds = torch.randint(0, R, (M,), device=DEV, generator=g)
    out = ops.obs_u8_to_f32(src, inds)
    back = (out * 255.0).round().to(torch.uint8)
    assert torch.equal(back, src[inds])
    row_sums = src.reshape(R, -1).sum(1, dtype=torch.int64)
    assert torch.equal((out.reshape(M, -1).double().sum(1) * 255.0).round().long(), row_sums[inds])


# ============================================================================ a8/a9 optimiser
@pytest.mark.parametrize("n,world", [(9219, 1), (1686693, 1), (1686693, 2), (7, 1)])
def test_clip_adam_matches_torch_clip_grad_norm_and_adam(n, world):
    torch.manual_seed(n)
    sizes = []
    left = n
    while left > 0:
        s = min(left, max(1, n // 7 + 3))
        sizes.append(s)
        left -= s
    p0 = torch.randn(n, device=DEV) * 0.1
    params = [torch.nn.Parameter(p0[o:o + s].clone()) for o, s in zip(np.cumsum([0] + sizes[:-1]), sizes)]
    opt = torch.optim.Adam(params, lr=2.5e-4, eps=1e-5)
    p = p0.clone()
    g = torch.zeros(n, device=DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        grads = torch.randn(n, device=DEV) * (0.01 if step != 2 else 5.0)   # step 2 is clipped hard
        g.copy_(grads)
        for prm, o, s in zip(params, np.cumsum([0] + sizes[:-1]), sizes):
            prm.grad = (grads[o:o + s] / world).clone()                       # :372  / world_size
        tn_ref = torch.nn.utils.clip_grad_norm_(params, 0.5)
        opt.step()
        tn = ops.clip_adam_(p, g, m, v, step, 2.5e-4, 0.5, grad_scale=1.0 / world)
        # tolerance: total norm rtol 1e-5; parameters rtol 1e-5, atol 1e-7 (f32 Adam, op order of torch)
        np.testing.assert_allclose(tn.item(), tn_ref.item(), rtol=1e-5)
        ref = torch.cat([q.detach() for q in params])
        np.testing.assert_allclose(p.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-7)
        assert torch.count_nonzero(g).item() == 0, "gradient buffer must be zeroed for the next backward"


def test_kernels_are_stream_capturable():
    """No allocation / sync inside any entry point: the whole hot path records into a HIP graph and replays."""
    T_, N, A, M = 16, 256, 4, 1024
    s = {k: v.to(DEV) for k, v in synthetic.rollout_scalars(T_, N, A, seed=2).items()}
    adv, ret = torch.empty_like(s["rewards"]), torch.empty_like(s["rewards"])
    logits = torch.randn(M, A, device=DEV)
    value = torch.randn(M, device=DEV)
    inds = torch.randperm(T_ * N, device=DEV)[:M]
    sc, dl, dv = torch.empty(7, device=DEV), torch.empty(M, A, device=DEV), torch.empty(M, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gae(s["rewards"], s["dones"], s["values"], s["next_done"], s["next_value"], 0.99, 0.95, adv, ret)  # warm ws
        ops.ppo_loss_categorical(logits, value, inds, s["actions"], s["logprobs"], adv, ret, s["values"], 0.1, 0.01, 0.5,
                                 scalars_out=sc, dlogits_out=dl, dvalue_out=dv)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    expect = (adv.clone(), sc.clone(), dl.clone())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.gae(s["rewards"], s["dones"], s["values"], s["next_done"], s["next_value"], 0.99, 0.95, adv, ret)
        ops.ppo_loss_categorical(logits, value, inds, s["actions"], s["logprobs"], adv, ret, s["values"], 0.1, 0.01, 0.5,
                                 scalars_out=sc, dlogits_out=dl, dvalue_out=dv)
    adv.zero_(); sc.zero_(); dl.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(adv, expect[0]) and torch.equal(sc, expect[1]) and torch.equal(dl, expect[2])


def test_obs_relayout_nchw_to_nhwc_u8():
    frames = torch.from_numpy(synthetic.atari_frames(37, seed=9)).to(DEV)
    out = ops.obs_nchw_to_nhwc_u8(frames)
    assert out.shape == (37, 84, 84, 4) and torch.equal(out, frames.permute(0, 2, 3, 1).contiguous())
    odd = torch.randint(0, 256, (5, 3, 7, 9), dtype=torch.uint8, device=DEV)      # generic-C path
    assert torch.equal(ops.obs_nchw_to_nhwc_u8(odd), odd.permute(0, 2, 3, 1).contiguous())
    # the relayout followed by the streaming convert is exactly the reference's x/255 on a channels-last view
    x = ops.obs_u8_to_f32(out).permute(0, 3, 1, 2)
    # (CPU torch divides; torch's GPU kernel multiplies by 1/255 and is 1 ulp off for 126 byte values -- the CPU
    # quotient is the oracle's and the kernel's definition, see obs.hip)
    assert x.is_contiguous(memory_format=torch.channels_last) and torch.equal(x.cpu(), frames.cpu().float() / 255.0)


def test_init_accepts_the_mi355x_and_rejects_a_missing_ordinal():
    from cleanrl_amd import _lib

    lib = _lib.load()
    assert lib.mi355ppo_init(0) == 0, lib.mi355ppo_last_error()
    assert lib.mi355ppo_init(torch.cuda.device_count()) == -1 and b"out of range" in lib.mi355ppo_last_error()
    _lib.require_device(0)
