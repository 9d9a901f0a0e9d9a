"""GPU tests of K7, the fused MLP kernel family (csrc/mlp.hip), through the C ABI: forward, rollout step (sampling), and the whole
minibatch body -- gather, forwards, distribution, PPO loss, backward, weight gradients -- against (a) float64 CPU torch autograd
over the reference's own formulas (oracle/torch_oracle.py, pinned to the reference-line goldens), (b) the unfused kernels K2 /
K2' / K3 this library already has (bit-equal where the arithmetic is the same), and (c) the full-size whole-iteration goldens of
BASELINE configs[4] (E: ppo_continuous_action.py, 320 updates) and configs[0] (A: ppo.py, 16 updates)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from cleanrl_amd import envs as E, learner_smoke, ops, synthetic
from cleanrl_amd.agents import ContinuousAgent, MlpAgent, layer_init
from cleanrl_amd.learner import PPOLearner
from oracle import torch_oracle as TO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _nets(O, nout, seed, std3=0.5):
    torch.manual_seed(seed)
    mk = lambda n, s: nn.Sequential(layer_init(nn.Linear(O, 64)), nn.Tanh(), layer_init(nn.Linear(64, 64)), nn.Tanh(),      # noqa: E731
                                    layer_init(nn.Linear(64, n), std=s))
    critic, actor = mk(1, 1.0), mk(nout, std3)
    with torch.no_grad():
        for m in list(critic) + list(actor):
            if isinstance(m, nn.Linear):
                m.bias.normal_(0, 0.1)                        # the reference's zero biases would hide a bias mix-up
    return actor, critic


def _dev(seq):
    d = nn.Sequential(*[type(m)(m.in_features, m.out_features) if isinstance(m, nn.Linear) else nn.Tanh() for m in seq]).to(DEV)
    d.load_state_dict(seq.state_dict())
    for p in d.parameters():
        p.grad = torch.zeros_like(p)
    return d


@pytest.mark.parametrize("O,nout,B", [(4, 2, 1), (4, 2, 128), (17, 6, 64), (17, 6, 700), (27, 8, 33), (32, 1, 5), (8, 4, 4097),
                                      (376, 17, 1), (376, 17, 700), (111, 8, 33), (33, 2, 64), (17, 9, 40), (512, 20, 129), (40, 18, 5)])      # (second line: the WIDE kernels, round 6)
def test_mlp_forward_matches_float64(O, nout, B):
    actor, critic = _nets(O, nout, seed=B)
    x = torch.randn(B, O)
    pa, pc = ops.MlpNetPtrs(_dev(actor)), ops.MlpNetPtrs(_dev(critic))
    out, val = ops.mlp_forward(x.to(DEV), pa, pc)
    ref_o, ref_v = actor.double()(x.double()), critic.double()(x.double())
    np.testing.assert_allclose(out.cpu().numpy(), ref_o.detach().numpy(), rtol=0, atol=3e-6 * max(1.0, float(ref_o.abs().max())))
    np.testing.assert_allclose(val.cpu().numpy(), ref_v.detach().numpy()[:, 0], rtol=0, atol=3e-6 * max(1.0, float(ref_v.abs().max())))
    out2, val2 = ops.mlp_forward(x.to(DEV), pa, pc)
    assert torch.equal(out, out2) and torch.equal(val, val2)


def test_mlp_refuses_unsupported_shapes_loudly():
    actor, critic = _nets(513, 2, seed=0)
    da, dc = _dev(actor), _dev(critic)
    assert ops.mlp_supported(376, 17) and ops.mlp_supported(512, 20) and not ops.mlp_supported(513, 2) and not ops.mlp_supported(17, 21)
    with pytest.raises(Exception, match="obs_dim=513"):
        ops.mlp_forward(torch.randn(3, 513, device=DEV), ops.MlpNetPtrs(da), ops.MlpNetPtrs(dc))


@pytest.mark.parametrize("O,A,B", [(4, 2, 4), (4, 2, 300), (12, 7, 65), (128, 18, 65), (40, 3, 300), (10, 18, 33)])
def test_mlp_act_categorical_is_the_unfused_pair_of_kernels(O, A, B):
    """One launch == mlp_forward then K2 on its logits: same logits bits, same Philox stream, same row math."""
    actor, critic = _nets(O, A, seed=3)
    pa, pc = ops.MlpNetPtrs(_dev(actor)), ops.MlpNetPtrs(_dev(critic))
    x = torch.randn(B, O, device=DEV)
    logits, value = ops.mlp_forward(x, pa, pc)
    a64, af, lp, ent, val, lg = ops.mlp_act_categorical(x, pa, pc, seed=11, offset=5, want_entropy=True, want_logits=True,
                                                        action_f32_out=torch.empty(B, device=DEV))
    r64, _, rlp, rent = ops.categorical_sample(logits, seed=11, offset=5)
    assert torch.equal(lg, logits) and torch.equal(val, value)
    assert torch.equal(a64, r64) and torch.equal(af, r64.float()) and torch.equal(lp, rlp) and torch.equal(ent, rent)
    # the stream position in device memory (captured launches): offset 2 + base 3 == offset 5
    base = torch.full((1,), 3, dtype=torch.int64, device=DEV)
    b64 = ops.mlp_act_categorical(x, pa, pc, seed=11, offset=2, offset_base=base)[0]
    assert torch.equal(b64, r64)
    # caller-supplied Exp(1) draws: torch's multinomial for that noise (parity mode)
    noise = torch.empty(B, A, device=DEV).exponential_()
    n64 = ops.mlp_act_categorical(x, pa, pc, noise_exp1=noise)[0]
    assert torch.equal(n64.cpu(), TO.categorical_sample_from_noise(logits.cpu(), noise.cpu()))


@pytest.mark.parametrize("O,D,B", [(17, 6, 64), (5, 3, 1), (27, 8, 129), (376, 17, 129), (376, 17, 1), (111, 8, 40)])
def test_mlp_act_normal_is_the_unfused_pair_of_kernels(O, D, B):
    actor, critic = _nets(O, D, seed=4)
    pa, pc = ops.MlpNetPtrs(_dev(actor)), ops.MlpNetPtrs(_dev(critic))
    x = torch.randn(B, O, device=DEV)
    logstd = torch.randn(1, D, device=DEV) * 0.3
    mean, value = ops.mlp_forward(x, pa, pc)
    act, lp, ent, val, mu = ops.mlp_act_normal(x, pa, pc, logstd, seed=9, offset=4, want_entropy=True, want_mean=True)
    ract, rlp, rent = ops.normal_sample(mean, logstd, seed=9, offset=4)
    assert torch.equal(mu, mean) and torch.equal(val, value)
    assert torch.equal(act, ract) and torch.equal(lp, rlp) and torch.equal(ent, rent)
    base = torch.full((1,), 1, dtype=torch.int64, device=DEV)
    assert torch.equal(ops.mlp_act_normal(x, pa, pc, logstd, seed=9, offset=3, offset_base=base)[0], ract)
    z = torch.randn(B, D, device=DEV)
    nact = ops.mlp_act_normal(x, pa, pc, logstd, noise=z)[0]
    assert torch.equal(nact, ops.normal_sample(mean, logstd, noise=z)[0])


def _behaviour(Bf, nout, normal, seed):
    g = torch.Generator().manual_seed(seed)
    acts = torch.randn(Bf, nout, generator=g) if normal else torch.randint(0, nout, (Bf,), generator=g).float()
    return acts, torch.randn(Bf, generator=g) - (3.0 if normal else 1.0), torch.randn(Bf, generator=g), torch.randn(Bf, generator=g), \
        torch.randn(Bf, generator=g)


@pytest.mark.parametrize("normal", [False, True])
@pytest.mark.parametrize("O,nout,Bf,M,rpb", [(4, 2, 512, 128, 0), (17, 6, 5000, 4096, 0), (17, 6, 300, 131, 4), (27, 8, 200, 200, 64),
                                            (9, 3, 64, 1, 0), (17, 6, 70000, 32768, 0),
                                            (376, 17, 2048, 64, 0), (376, 17, 5000, 4096, 0), (111, 8, 300, 131, 4), (33, 2, 200, 200, 64),
                                            (512, 20, 700, 333, 0), (20, 12, 64, 1, 0)])      # (second half: the WIDE kernels)
def test_mlp_ppo_minibatch_against_float64_autograd(normal, O, nout, Bf, M, rpb):
    """The fused minibatch body against float64 torch autograd over the reference's loss lines (oracle/torch_oracle.ppo_loss ==
    ppo.py:253-285, pinned to the reference-line goldens): the seven scalars and every parameter gradient of both networks (and of
    the shared actor_logstd).  Gradients are ADDED to what the .grad tensors hold."""
    actor, critic = _nets(O, nout, seed=M + nout)
    da, dc = _dev(actor), _dev(critic)
    pa, pc = ops.MlpNetPtrs(da), ops.MlpNetPtrs(dc)
    g = torch.Generator().manual_seed(7)
    obs = torch.randn(Bf, O, generator=g)
    acts, old_lp, adv, ret, old_v = _behaviour(Bf, nout, normal, 8)
    idx = torch.randperm(Bf, generator=g)[:M] if M < Bf else None
    logstd = (torch.randn(1, nout, generator=g) * 0.2) if normal else None
    clip, entc, vfc = 0.2, 0.01, 0.5
    norm_adv = M > 1
    # ---- float64 reference
    ra, rc = actor.double(), critic.double()
    rows = idx if idx is not None else torch.arange(Bf)
    x64 = obs.double()[rows]
    outa, v = ra(x64), rc(x64)
    if normal:
        ls64 = logstd.double().clone().requires_grad_(True)
        lp, ent = TO.normal_logprob_entropy(outa, ls64, acts.double()[rows])
    else:
        lp, ent = TO.categorical_logprob_entropy(outa, acts.long()[rows])
    ref = TO.ppo_loss(lp, ent, v, old_lp.double()[rows], adv.double()[rows], ret.double()[rows], old_v.double()[rows], clip, entc, vfc, norm_adv, True)
    ref["loss"].backward()
    # ---- fused kernel; the .grad tensors start at a known non-zero value (accumulation semantics)
    for p in list(da.parameters()) + list(dc.parameters()):
        p.grad.fill_(0.25)
    dls = logstd.to(DEV) if normal else None
    gls = torch.full((1, nout), 0.25, device=DEV) if normal else None
    D = lambda t: t.to(DEV).contiguous()          # noqa: E731
    sc = ops.mlp_ppo_fwd_bwd(D(obs), None if idx is None else D(idx), pa, pc, D(acts), D(old_lp), D(adv), D(ret), D(old_v), clip, entc, vfc,
                             norm_adv, True, logstd=dls, logstd_grad=gls, rows_per_block=rpb)
    want = [float(ref[k]) for k in ("loss", "pg_loss", "v_loss", "entropy", "old_approx_kl", "approx_kl", "clipfrac")]
    np.testing.assert_allclose(sc.cpu().numpy(), want, rtol=2e-5, atol=2e-6)
    worst = 0.0
    for (name, pd), pr in zip(list(da.named_parameters()) + list(dc.named_parameters()), list(ra.parameters()) + list(rc.parameters())):
        got = (pd.grad - 0.25).cpu().double().numpy()
        scale = max(float(pr.grad.abs().max()), 1e-6)
        err = np.abs(got - pr.grad.numpy()).max() / scale
        worst = max(worst, err)
        assert err < 2e-5 + 2e-6 / scale, (name, err, scale)          # f32 sums of up to 32,768 terms; 0.25 + g loses 2^-24 * 0.25
    if normal:
        got = (gls - 0.25).cpu().double().numpy().reshape(-1)
        np.testing.assert_allclose(got, ls64.grad.numpy().reshape(-1), rtol=0, atol=2e-5 * max(float(ls64.grad.abs().max()), 1e-2))
    # deterministic: a second call adds exactly the same numbers
    before = [p.grad.clone() for p in list(da.parameters()) + list(dc.parameters())]
    ops.mlp_ppo_fwd_bwd(D(obs), None if idx is None else D(idx), pa, pc, D(acts), D(old_lp), D(adv), D(ret), D(old_v), clip, entc, vfc,
                        norm_adv, True, logstd=dls, logstd_grad=gls, rows_per_block=rpb)
    for p, b0 in zip(list(da.parameters()) + list(dc.parameters()), before):
        assert torch.allclose(p.grad - b0, b0 - 0.25, rtol=0, atol=1e-6)


def test_mlp_ppo_rows_per_block_changes_only_the_summation_order():
    actor, critic = _nets(17, 6, seed=2)
    acts, old_lp, adv, ret, old_v = _behaviour(3000, 6, True, 3)
    obs, idx = torch.randn(3000, 17), torch.randperm(3000)[:2048]
    logstd = torch.zeros(1, 6, device=DEV)
    res = []
    for rpb in (4, 16, 64):
        da, dc = _dev(actor), _dev(critic)
        gls = torch.zeros(1, 6, device=DEV)
        D = lambda t: t.to(DEV).contiguous()      # noqa: E731
        sc = ops.mlp_ppo_fwd_bwd(D(obs), D(idx), ops.MlpNetPtrs(da), ops.MlpNetPtrs(dc), D(acts), D(old_lp), D(adv), D(ret), D(old_v), 0.2, 0.0, 0.5,
                                 logstd=logstd, logstd_grad=gls, rows_per_block=rpb)
        res.append((sc.cpu(), torch.cat([p.grad.reshape(-1) for p in list(da.parameters()) + list(dc.parameters())] + [gls.reshape(-1)]).cpu()))
    for sc, g in res[1:]:
        np.testing.assert_allclose(sc.numpy(), res[0][0].numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(g.numpy(), res[0][1].numpy(), rtol=0, atol=2e-6 * float(res[0][1].abs().max()))


def test_clip_adam_sched_is_bit_identical_to_the_eager_step():
    n = 11085
    g = torch.Generator().manual_seed(1)
    mk = lambda: [torch.randn(n, generator=g).to(DEV) for _ in range(2)] + [torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)]      # noqa: E731
    a = mk()
    b = [t.clone() for t in a]
    sched = torch.zeros(2, device=DEV)
    for step in (1, 2, 7, 320):
        lr = 3e-4 * (1 - step / 400)
        grad = torch.randn(n, generator=g).to(DEV)
        a[1].copy_(grad); b[1].copy_(grad)
        ops.clip_adam_(a[0], a[1], a[2], a[3], step, lr, 0.5)
        sched.copy_(torch.tensor(ops.adam_schedule(lr, step)))
        ops.clip_adam_sched_(b[0], b[1], b[2], b[3], sched, 0.5)
        for x, y in zip(a, b):
            assert torch.equal(x, y), step


# ----------------------------------------------------------------------------------------------------- whole iterations, full size
def _teacher_forced(L, g, obs_seq, step_done, rewards):
    T = L.T
    L.observe(0, obs_seq[0], step_done[0])
    worst_v, worst_lp = 0.0, 0.0
    for step in range(T):
        L.act(step)
        worst_v = max(worst_v, float(np.abs(L.values[step].cpu().numpy() - g["values"][step]).max()))
        L.actions[step].copy_(torch.from_numpy(g["actions"][step]))
        L.logprobs[step].copy_(torch.from_numpy(g["logprobs"][step]))
        L.values[step].copy_(torch.from_numpy(g["values"][step]))
        L.store_reward(step, rewards[step])
        L.observe(step + 1, obs_seq[step + 1], step_done[step + 1])
    return worst_v


def test_config_e_whole_iteration_320_updates_on_the_fused_mlp_kernels():
    """BASELINE configs[4] at its full size -- 64 envs x 2048 steps, 10 epochs x 32 minibatches = 320 updates of 4,096 rows,
    obs 17 / act 6 -- against one whole iteration of cleanrl/ppo_continuous_action.py's own lines :232-309
    (tests/golden/continuous_iteration_cfgE.npz, oracle/mint_full_size.py; the reference's sampled actions forced, observations
    regenerated from the seed on both sides).  The HIP path: fused MLP rollout step (values), K1, the fused MLP minibatch body,
    K6.  Checked: rollout values, log-probs of the forced actions, GAE, the seven scalars of ALL 320 minibatches, the clipped
    gradient at updates 1 / 160 / 320 (whole vector), actor_logstd before every step, the parameters after update 320."""
    g = load_golden("continuous_iteration_cfgE")["mujoco_T2048_N64"]
    T, N = g["values"].shape
    OBS, ACT = 17, 6
    obs_seq, step_done, rewards = synthetic.continuous_inputs(T, N, OBS, int(g["input_seed"]))
    assert abs(float(obs_seq.astype(np.float64).sum()) - float(g["obs_checksum"])) < 1e-6
    env = SimpleNamespace(single_observation_space=E.Box(-np.inf, np.inf, (OBS,)), single_action_space=E.Box(-1.0, 1.0, (ACT,)))
    torch.manual_seed(int(g["init_seed"]))
    agent = ContinuousAgent(env).to(DEV)
    args = learner_smoke.default_args(num_steps=T, num_minibatches=32, update_epochs=10, clip_coef=0.2, ent_coef=0.0, learning_rate=3e-4)
    L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, DEV, sample_seed=1)
    assert L.hip and L.mlp is not None and not L.discrete and L.minibatch_size == 4096
    assert [n for n, _ in agent.named_parameters()] == [str(x) for x in g["param_names"]]
    # same construction order as the reference Agent => the same initial weights (orthogonal_'s QR may differ in the last bits from
    # one host CPU to the next: close, then forced, so that the trajectory starts from the golden's exact parameters)
    np.testing.assert_allclose(L.flat.params.cpu().numpy(), g["init_params"], rtol=0, atol=2e-5)
    L.flat.params.copy_(torch.from_numpy(g["init_params"]))
    worst_v = _teacher_forced(L, g, obs_seq, step_done, rewards)
    assert worst_v <= 2e-5 * max(1.0, float(np.abs(g["values"]).max())), worst_v
    # the forced actions' log-probs through the fused forward + K2'
    mean, _ = L.ops.mlp_forward(torch.from_numpy(obs_seq[:T].reshape(T * N, OBS)).to(DEV), *L.mlp)
    lp, _ = L.ops.normal_logprob_entropy(mean, agent.actor_logstd.detach(), L.actions.reshape(T * N, ACT).contiguous())
    np.testing.assert_allclose(lp.cpu().numpy(), g["logprobs"].reshape(-1), rtol=1e-4, atol=2e-5)
    L.finish_rollout()
    np.testing.assert_allclose(L.advantages.cpu().numpy(), g["advantages"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(L.returns.cpu().numpy(), g["returns"], rtol=1e-4, atol=1e-4)
    L.advantages.copy_(torch.from_numpy(g["advantages"]))
    L.returns.copy_(torch.from_numpy(g["returns"]))
    keep = tuple(int(k) for k in g["grad_updates"])
    seen, logstd, count = {}, [], [0]
    real = L.optimizer_step_hip

    def spy(lr):
        count[0] += 1
        logstd.append(agent.actor_logstd.detach().reshape(-1).clone())
        if count[0] in keep:
            seen[count[0]] = L.flat.grads.clone()
        real(lr)

    L.optimizer_step_hip = spy
    np.random.seed(int(g["shuffle_seed"]))
    m = L.update(float(g["lr"]))
    assert m["num_updates"] == 320
    sc, ref = L._scalars[:320].cpu().numpy().astype(np.float64), g["scalars"].astype(np.float64)
    # 320 Adam steps amplify f32 round-off: the bars widen with the update index (first epoch: same parameters to ~1e-6)
    atol = np.array([2e-4, 5e-5, 2e-4, 1e-4, 2e-5, 2e-5, 2.5e-3])
    grow = 1.0 + np.arange(320)[:, None] / 32.0
    err, bar = np.abs(sc - ref), (1e-3 * np.abs(ref) + atol) * grow
    assert (err <= bar).all(), "minibatch scalars off the reference's lines: worst err/bar per column %s at updates %s" % (
        (err / bar).max(0).round(3), (err / bar).argmax(0) + 1)
    np.testing.assert_allclose(sc[:32], ref[:32], rtol=2e-4, atol=3e-5)          # first epoch: tight
    problems = []
    for k, (b_el, b_cos) in zip(keep, ((1e-5, 1e-10), (1e-5, 1e-10), (2e-3, 1e-6))):      # measured 8e-7 / 7e-7 / 1.6e-4; the reference's own
                                                                                             # distance from itself at 4 CPU threads: 8e-7 / 6e-7 / 1.5e-4
        gh = seen[k].cpu().numpy().astype(np.float64)
        n = np.linalg.norm(gh)
        clipped = gh * min(1.0, args.max_grad_norm / (n + 1e-6))
        want = g[f"mb{k}_grad"].astype(np.float64)
        worst = np.abs(clipped - want).max() / np.abs(want).max()
        c = float(clipped @ want / (np.linalg.norm(clipped) * np.linalg.norm(want)))
        if worst > b_el or 1 - c > b_cos:
            problems.append(f"update {k}: max|dg|/absmax {worst:.2e}, 1-cosine {1 - c:.2e}")
    got_ls = torch.stack(logstd).cpu().numpy()
    ls_err = np.abs(got_ls - g["logstd_before_step"])
    move = np.abs(g["logstd_before_step"][-1] - g["logstd_before_step"][0]).max()
    if ls_err.max() > 1e-4 * move:
        problems.append(f"actor_logstd trajectory: worst {ls_err.max():.2e} against a total movement of {move:.2e}")
    delta = L.flat.params.cpu().numpy() - g["init_params"]
    want = g["final_params"] - g["init_params"]
    c = float(delta.astype(np.float64) @ want.astype(np.float64) / (np.linalg.norm(delta) * np.linalg.norm(want)))
    close = np.isclose(delta, want, rtol=5e-2, atol=5e-5)
    if c < 0.999999 or close.mean() < 0.999 or abs(np.linalg.norm(delta) / np.linalg.norm(want) - 1) > 1e-4:
        problems.append(f"320-step parameter move: cosine {c:.6f}, length ratio {np.linalg.norm(delta) / np.linalg.norm(want):.5f}, "
                        f"{close.mean():.4f} of the parameters within 5 %")
    print("config E whole iteration vs the reference's lines: values %.2e; scalars worst err/bar per column %s (first epoch worst abs %s); "
          "logstd %.2e of %.2e; 320-step move cosine %.8f, length ratio %.6f, within 5 %% %.4f" % (
              worst_v, (err / bar).max(0).round(3), err[:32].max(0), ls_err.max(), move, c, np.linalg.norm(delta) / np.linalg.norm(want), close.mean()))
    for k in keep:
        gh = seen[k].cpu().numpy().astype(np.float64)
        clipped = gh * min(1.0, args.max_grad_norm / (np.linalg.norm(gh) + 1e-6))
        want_g = g[f"mb{k}_grad"].astype(np.float64)
        print(f"  update {k}: max|dg|/absmax {np.abs(clipped - want_g).max() / np.abs(want_g).max():.2e}, "
              f"1-cosine {1 - float(clipped @ want_g / (np.linalg.norm(clipped) * np.linalg.norm(want_g))):.2e}")
    assert not problems, "\n".join(problems)
    L.flat.check_views()


def test_config_a_whole_iteration_on_the_gpu_fused_mlp_kernels():
    """BASELINE configs[0]'s script at its own size (4 envs x 128 steps, 16 updates of 128 rows; cleanrl/ppo.py:217-294,
    tests/golden/ppo_iteration_cfgA.npz) on the GPU path: the Categorical MLP agent on the fused kernels.  (The configuration
    itself is the reference's CPU case -- tests/test_host_logic.py holds the --no-cuda path to the same golden.)"""
    g = load_golden("ppo_iteration_cfgA")["cartpole_T128_N4"]
    T, N = g["values"].shape
    OBS, A = 4, 2
    obs_seq, step_done, rewards = synthetic.continuous_inputs(T, N, OBS, int(g["input_seed"]), done_p=1.0 / 30.0, unit_rewards=True)
    assert abs(float(obs_seq.astype(np.float64).sum()) - float(g["obs_checksum"])) < 1e-6
    env = SimpleNamespace(single_observation_space=E.Box(-np.inf, np.inf, (OBS,)), single_action_space=E.Discrete(A))
    torch.manual_seed(int(g["init_seed"]))
    agent = MlpAgent(env).to(DEV)
    args = learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=4, clip_coef=0.2, ent_coef=0.01, learning_rate=2.5e-4)
    L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, DEV, sample_seed=1)
    assert L.hip and L.mlp is not None and L.discrete and L.minibatch_size == 128
    np.testing.assert_allclose(L.flat.params.cpu().numpy(), g["init_params"], rtol=0, atol=2e-5)
    L.flat.params.copy_(torch.from_numpy(g["init_params"]))
    worst_v = _teacher_forced(L, g, obs_seq, step_done, rewards)
    assert worst_v <= 1e-5, worst_v
    L.finish_rollout()
    np.testing.assert_allclose(L.advantages.cpu().numpy(), g["advantages"], rtol=1e-4, atol=1e-4)
    L.advantages.copy_(torch.from_numpy(g["advantages"]))
    L.returns.copy_(torch.from_numpy(g["returns"]))
    keep = tuple(int(k) for k in g["grad_updates"])
    seen, count = {}, [0]
    real = L.optimizer_step_hip

    def spy(lr):
        count[0] += 1
        if count[0] in keep:
            seen[count[0]] = L.flat.grads.clone()
        real(lr)

    L.optimizer_step_hip = spy
    np.random.seed(int(g["shuffle_seed"]))
    m = L.update(float(g["lr"]))
    assert m["num_updates"] == 16
    sc, ref = L._scalars[:16].cpu().numpy().astype(np.float64), g["scalars"].astype(np.float64)
    np.testing.assert_allclose(sc[:, :4], ref[:, :4], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(sc[:, 4:6], ref[:, 4:6], rtol=2e-2, atol=2e-5)
    np.testing.assert_allclose(sc[:, 6], ref[:, 6], atol=1.0 / 128 + 1e-6)          # one row of 128 across the clip boundary
    for k, (b_el, b_cos) in zip(keep, ((1e-4, 1e-8), (5e-3, 1e-5), (1e-2, 5e-5))):
        gh = seen[k].cpu().numpy().astype(np.float64)
        clipped = gh * min(1.0, args.max_grad_norm / (np.linalg.norm(gh) + 1e-6))
        want = g[f"mb{k}_grad"].astype(np.float64)
        c = float(clipped @ want / (np.linalg.norm(clipped) * np.linalg.norm(want)))
        assert np.abs(clipped - want).max() <= b_el * np.abs(want).max() and 1 - c <= b_cos, (k, np.abs(clipped - want).max() / np.abs(want).max(), 1 - c)
    delta, want = L.flat.params.cpu().numpy() - g["init_params"], g["final_params"] - g["init_params"]
    assert np.isclose(delta, want, rtol=5e-2, atol=2e-5).mean() > 0.98
    L.flat.check_views()


def test_captured_continuous_rollout_replays_bit_identically_to_the_eager_loop():
    """capture_rollout on the MLP / continuous-action path: every env step = fused act kernel + the stand-in env's step kernel,
    captured; two iterations (rollout, update, rollout) equal the eager loop bit for bit."""
    N, T = 16, 24

    def make(graph, per):
        torch.manual_seed(4)
        env = E.DeviceSyntheticContinuousVecEnv(N, DEV, seed=6, horizon=10)
        agent = ContinuousAgent(env).to(DEV)
        args = learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=2, clip_coef=0.2, ent_coef=0.0, learning_rate=3e-4)
        L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, DEV, sample_seed=8)
        L.observe(0, env.obs(), L.dones[0])
        if graph:
            L.capture_rollout(env, steps_per_graph=per)
        return L, env

    def rollout(L, env):
        if getattr(L, "_rollout_graphs", None):
            L.replay_rollout()
        else:
            for step in range(T):
                action = L.act(step)
                obs_dst, done_dst = L._slot(step + 1)
                env.step_into(action, obs_dst, L.rewards[step], done_dst)
        L.finish_rollout()

    for per in (1, 5, T):
        (Le, enve), (Lg, envg) = make(False, 0), make(True, per)
        for it in range(2):
            rollout(Le, enve)
            rollout(Lg, envg)
            for name in ("obs", "actions", "logprobs", "values", "rewards", "dones", "advantages", "returns", "boot_obs", "boot_done"):
                assert torch.equal(getattr(Le, name), getattr(Lg, name)), (per, it, name)
            assert Le.dones.sum() > 0                      # truncations happened (horizon 10)
            np.random.seed(50 + it)
            Le.update(3e-4)
            np.random.seed(50 + it)
            Lg.update(3e-4)
            Le.start_iteration(); Lg.start_iteration()
            assert torch.equal(Le.flat.params, Lg.flat.params)
