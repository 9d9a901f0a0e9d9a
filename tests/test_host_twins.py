"""The ``*_cpu`` host-pointer twins of the C ABI (csrc/host_twins.hip) against the goldens minted from the
reference's own lines -- the same fixtures the GPU kernels are held to in tests/test_gpu_kernels.py, so a twin and its device
kernel are pinned to one set of numbers.  CPU only: these calls compute on host pointers and need no GPU.

Bars: GAE bit-exact (mul / add / sub only, same op order); actions exact from supplied noise except on near-ties; log-prob,
entropy, loss scalars and gradients at a few f32 ulp of the result's scale (libm vs torch's vectorised expf / logf; f64 sums in
row order vs torch's f32 tree sums)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden
from cleanrl_amd import _lib, host_ops as H

T = torch.from_numpy
SCALARS = ["loss", "pg_loss", "v_loss", "entropy", "old_approx_kl", "approx_kl", "clipfrac"]


@pytest.mark.parametrize("case", sorted(load_golden("gae")))
def test_gae_twin_bit_exact(case):
    g = load_golden("gae")[case]
    adv, ret = H.gae(T(g["rewards"]), T(g["dones"]), T(g["values"]), T(g["next_done"]), T(g["next_value"]), float(g["gamma"]),
                     float(g["gae_lambda"]))
    assert np.array_equal(adv.numpy(), g["advantages"])
    assert np.array_equal(ret.numpy(), g["returns"])


@pytest.mark.parametrize("case", sorted(load_golden("categorical")))
def test_categorical_twins(case):
    g = load_golden("categorical")[case]
    act, lp, ent = H.categorical_sample(T(g["logits"]), T(g["noise_exp1"]))
    mism = act.numpy() != g["action"]
    assert mism.mean() <= 2e-3                               # argmax(p / q) may flip only on near-ties of the two quotients
    ok = ~mism
    np.testing.assert_allclose(lp.numpy()[ok], g["logprob"][ok], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ent.numpy(), g["entropy"], rtol=2e-6, atol=2e-6)
    lp2, ent2 = H.categorical_logprob_entropy(T(g["logits"]), T(g["action"]))
    np.testing.assert_allclose(lp2.numpy(), g["logprob"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ent2.numpy(), g["entropy"], rtol=2e-6, atol=2e-6)
    # backward of (log_prob, entropy): against torch autograd of torch.distributions.Categorical on the same rows
    logits = T(g["logits"]).clone().requires_grad_(True)
    d = torch.distributions.Categorical(logits=logits)
    gl, ge = torch.linspace(-1, 1, logits.shape[0]), torch.linspace(0.5, -0.5, logits.shape[0])
    (d.log_prob(T(g["action"])) * gl + d.entropy() * ge).sum().backward()
    got = H.categorical_logprob_entropy_bwd(T(g["logits"]), T(g["action"]), gl, ge)
    # (peaked rows: 1 - p cancels in both derivations, so the bar is absolute on the gradient's scale, max |g_logprob| = 1)
    np.testing.assert_allclose(got.numpy(), logits.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_categorical_twin_philox_stream_is_a_valid_sampler():
    """No noise supplied: the draws come from the Philox stream of the device kernel (same counter layout).  Chi-square of
    200,000 draws against the row's probabilities, and the stream is a function of (seed, offset, row) only."""
    logits = torch.tensor([[0.3, -1.2, 0.9, 0.0]]).repeat(200000, 1)
    a1, lp, _ = H.categorical_sample(logits, None, seed=7, offset=3)
    a2, _, _ = H.categorical_sample(logits[:1000], None, seed=7, offset=3)
    assert torch.equal(a1[:1000], a2)                         # geometry-independent
    a3, _, _ = H.categorical_sample(logits[:1000], None, seed=7, offset=4)
    assert not torch.equal(a2, a3)
    p = torch.softmax(logits[0], 0).numpy()
    counts = np.bincount(a1.numpy(), minlength=4)
    chi2 = ((counts - p * len(a1)) ** 2 / (p * len(a1))).sum()
    assert chi2 < 16.27                                        # chi-square, 3 dof, p = 1e-3
    np.testing.assert_allclose(lp[:8].numpy(), torch.log_softmax(logits[:8], 1).gather(1, a1[:8, None])[:, 0].numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("case", sorted(load_golden("normal")))
def test_normal_twins(case):
    g = load_golden("normal")[case]
    act, lp, ent = H.normal_sample(T(g["mean"]), T(g["logstd"]), T(g["noise"]))
    np.testing.assert_allclose(act.numpy(), g["action"], rtol=1e-6, atol=1e-6)          # exp(logstd): libm vs torch, then mul + add
    lp2, ent2 = H.normal_logprob_entropy(T(g["mean"]), T(g["logstd"]), T(g["action"]))
    scale = np.abs(g["logprob_sum"]).max()
    np.testing.assert_allclose(lp2.numpy(), g["logprob_sum"], rtol=2e-6, atol=2e-6 * scale)
    np.testing.assert_allclose(ent2.numpy(), g["entropy_sum"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ent.numpy(), g["entropy_sum"], rtol=2e-6, atol=2e-6)
    mean = T(g["mean"]).clone().requires_grad_(True)
    logstd = T(g["logstd"]).clone().requires_grad_(True)
    d = torch.distributions.Normal(mean, torch.exp(logstd.reshape(1, -1).expand_as(mean)))
    B = mean.shape[0]
    gl, ge = torch.linspace(-1, 1, B), torch.linspace(0.5, -0.5, B)
    (d.log_prob(T(g["action"])).sum(1) * gl + d.entropy().sum(1) * ge).sum().backward()
    dmean, dls_rows = H.normal_logprob_entropy_bwd(T(g["mean"]), T(g["logstd"]), T(g["action"]), gl, ge)
    np.testing.assert_allclose(dmean.numpy(), mean.grad.numpy(), rtol=1e-4, atol=1e-5 * np.abs(mean.grad.numpy()).max())
    np.testing.assert_allclose(dls_rows.sum(0).numpy(), logstd.grad.numpy(), rtol=2e-4, atol=1e-4 * np.abs(logstd.grad.numpy()).max())


def test_normal_twin_box_muller_moments():
    mean, logstd = torch.zeros(100000, 4), torch.tensor([0.0, -1.0, 0.5, 0.2])
    act, lp, _ = H.normal_sample(mean, logstd, None, seed=11, offset=0)
    sd = torch.exp(logstd)
    assert (act.mean(0).abs() < 5 * sd / np.sqrt(100000)).all()
    np.testing.assert_allclose(act.std(0).numpy(), sd.numpy(), rtol=2e-2)
    ref = torch.distributions.Normal(mean, sd.expand_as(mean)).log_prob(act).sum(1)
    np.testing.assert_allclose(lp.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def _kw(g):
    return dict(clip_coef=float(g["clip_coef"]), ent_coef=float(g["ent_coef"]), vf_coef=float(g["vf_coef"]),
                norm_adv=bool(g["norm_adv"]), clip_vloss=bool(g["clip_vloss"]))


@pytest.mark.parametrize("case", sorted(load_golden("loss_categorical")))
def test_loss_categorical_twin_forward_and_autograd(case):
    g = load_golden("loss_categorical")[case]
    logits = T(g["new_logits"]).clone().requires_grad_(True)
    value = T(g["new_value"]).clone().requires_grad_(True)
    loss, sc = H.ppo_loss_categorical(logits, value, T(g["mb_inds"]), T(g["b_actions"]), T(g["b_logprobs"]), T(g["b_advantages"]),
                                      T(g["b_returns"]), T(g["b_values"]), **_kw(g))
    for i, k in enumerate(SCALARS):
        np.testing.assert_allclose(sc[i].item(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)      # the GPU kernel's bars (test_gpu_kernels.py)
    loss.backward()
    gmax = np.abs(g["dlogits"]).max()
    np.testing.assert_allclose(logits.grad.numpy(), g["dlogits"], rtol=2e-4, atol=2e-5 * gmax)
    np.testing.assert_allclose(value.grad.numpy(), g["dvalue"], rtol=2e-4, atol=2e-5 * np.abs(g["dvalue"]).max())


@pytest.mark.parametrize("case", sorted(load_golden("loss_normal")))
def test_loss_normal_twin_forward_and_autograd(case):
    g = load_golden("loss_normal")[case]
    mean = T(g["new_mean"]).clone().requires_grad_(True)
    logstd = T(g["logstd"]).clone().requires_grad_(True)
    value = T(g["new_value"]).clone().requires_grad_(True)
    loss, sc = H.ppo_loss_normal(mean, logstd, value, T(g["mb_inds"]), T(g["b_actions"]), T(g["b_logprobs"]), T(g["b_advantages"]),
                                 T(g["b_returns"]), T(g["b_values"]), **_kw(g))
    for i, k in enumerate(SCALARS):
        np.testing.assert_allclose(sc[i].item(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)
    loss.backward()
    for got, k in ((mean.grad, "dmean"), (logstd.grad, "dlogstd"), (value.grad, "dvalue")):
        np.testing.assert_allclose(got.numpy().reshape(g[k].shape), g[k], rtol=2e-4, atol=2e-5 * np.abs(g[k]).max(), err_msg=k)


def test_clip_adam_twin_reproduces_the_reference_update_steps():
    """a8 on the golden of ppo.py's three optimizer steps (update_step.npz::ppo_mlp_3steps): gradients from torch autograd of the
    reference loss, then ``mi355ppo_clip_adam_f32_cpu`` on flat buffers against the parameters the reference's
    clip_grad_norm_ + Adam.step left."""
    from oracle import torch_oracle as TO

    g = load_golden("update_step")["ppo_mlp_3steps"]
    dims = [(64, 4), (64,), (64, 64), (64,), (1, 64), (1,), (64, 4), (64,), (64, 64), (64,), (2, 64), (2,)]

    def forward(p, x):
        w, o = [], 0
        for d in dims:
            n = int(np.prod(d))
            w.append(p[o:o + n].reshape(d))
            o += n
        h = torch.tanh(x @ w[0].T + w[1]); h = torch.tanh(h @ w[2].T + w[3]); v = h @ w[4].T + w[5]
        a = torch.tanh(x @ w[6].T + w[7]); a = torch.tanh(a @ w[8].T + w[9]); logits = a @ w[10].T + w[11]
        return logits, v.reshape(-1)

    p = T(g["init_params"]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    M = 128
    for k in range(3):
        idx = T(g["perm"][k * M:(k + 1) * M])
        pp = p.clone().requires_grad_(True)
        logits, val = forward(pp, T(g["b_obs"])[idx])
        loss, _ = H.ppo_loss_categorical(logits, val, idx, T(g["b_actions"]), T(g["b_logprobs"]), T(g["b_advantages"]),
                                         T(g["b_returns"]), T(g["b_values"]), 0.2, 0.01, 0.5, True, True)
        np.testing.assert_allclose(loss.item(), g["losses"][k], rtol=2e-5)
        loss.backward()
        grads = pp.grad.clone()
        H.clip_adam_(p, grads, m, v, k + 1, float(g["lr"]), 0.5)
        assert not grads.any()                                                     # zeroed for the next backward, as on the device
        np.testing.assert_allclose(p.numpy(), g[f"params_after_{k + 1}"], rtol=1e-5, atol=2e-7)
    assert TO is not None


def test_obs_twin_is_exact_and_gathers():
    rs = np.random.RandomState(0)
    src = T(rs.randint(0, 256, size=(9, 4, 6, 6), dtype=np.uint8))
    inds = torch.tensor([8, 0, 3, 3])
    out = H.obs_u8_to_f32(src, inds)
    assert torch.equal(out, src[inds].float() / 255.0)                             # correctly rounded division, as torch's
    assert torch.equal(H.obs_u8_to_f32(src, None, scale_255=False), src.float())


def test_twins_validate_and_refuse_device_tensors():
    lib = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mi355ppo_gae_f32_cpu(None, p, p, p, p, p, p, 4, 4, 0.99, 0.95) == -1
    assert b"null" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_gae_f32_cpu(p, p, p, p, p, p, p, 0, 4, 0.99, 0.95) == -1
    assert lib.mi355ppo_categorical_sample_f32_cpu(p, None, 0, 0, p, None, p, None, 4, 65) == -1
    assert lib.mi355ppo_loss_categorical_fwd_bwd_f32_cpu(p, p, None, p, p, p, p, p, 1, 4, 0.1, 0.01, 0.5, 1, 1, None, p, p, p) == -1
    assert b"norm_adv" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_clip_adam_f32_cpu(p, p, p, p, 16, 1.0, 0.5, 1e-3, 0.9, 0.999, 1e-5, 0, None) == -1
    meta = torch.zeros(4, 4, device="meta")
    with pytest.raises(TypeError, match="CPU tensors only"):
        H.gae(meta, meta, meta, meta[0], meta[0], 0.99, 0.95)
