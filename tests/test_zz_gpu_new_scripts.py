"""GPU tests of the drop-ins added after the round's GPU minutes were spent: the recurrent learner (ppo_atari_lstm.py),
teacher-forced on the rollout the reference's own lines produced (tests/golden/lstm_iteration.npz), ppo_procgen.py and
ppo_rnd_envpool.py.

Strict tests (no xfail).  Their host paths are proven against the reference lines (tests/test_lstm_script.py,
tests/test_procgen_script.py, tests/test_rnd_script.py) and the kernels they call -- K1, K2, K3, K5, K6 -- are the ones the other
GPU tests cover."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from cleanrl_amd import envs as E
from cleanrl_amd.agents import AtariLSTMAgent
from cleanrl_amd.learner_lstm import LSTMPPOLearner
from cleanrl_amd.learner_smoke import default_args

pytestmark = pytest.mark.gpu
# First GPU run (round 2, profiles/r02_*): every test of this file takes 0.6-4.3 s on a fresh box (MIOpen's immediate-mode
# kernels for the 1-channel conv1 / IMPALA stacks need no tuning), so they always run -- no opt-in any more.
DEV = torch.device("cuda:0")


def _learner(g, T, N):
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (1, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    n = torch.get_num_threads()
    torch.set_num_threads(1)          # as when minted: orthogonal_'s QR of the LSTM weights rounds per thread count
    torch.manual_seed(int(g["init_seed"]))
    agent = AtariLSTMAgent(envs).to(DEV)
    torch.set_num_threads(n)
    args = default_args(num_steps=T, num_minibatches=2, update_epochs=2)
    return agent, LSTMPPOLearner(agent, args, envs.single_observation_space, envs.single_action_space, N, DEV, sample_seed=1)


def test_lstm_hip_path_teacher_forced_against_reference_iteration():
    g = load_golden("lstm_iteration")["lstm_T8_N4"]
    T, N = g["rewards"].shape
    agent, L = _learner(g, T, N)
    assert L.hip and L.obs.dtype == torch.uint8 and not L.fused_cnn
    stride = int(g["stride"])
    # same seed -> same weights, up to LAPACK: orthogonal_'s QR of the LSTM matrices takes another code path on the GPU box's
    # host CPU than on the minting machine (1 of 20,695 sampled weights off by 2e-6 there)
    np.testing.assert_allclose(L.flat.params[::stride].cpu().numpy(), g["init_params_sub"], rtol=1e-4, atol=1e-5)
    frames, step_done = g["frames_u8"], g["step_done"]
    # rollout: the reference's actions are forced (the HIP sampler draws from Philox, not from torch's generator), the
    # network outputs along the way must agree with what the reference Agent computed
    L.observe(0, frames[0], step_done[0])
    for step in range(T):
        L.act(step)
        # values: conv stack + LSTM on MIOpen/hipBLASLt vs CPU torch -> 1e-4 of the value scale
        np.testing.assert_allclose(L.values[step].cpu().numpy(), g["values"][step], rtol=1e-3, atol=2e-4)
        L.actions[step].copy_(torch.from_numpy(g["actions"][step]))
        L.logprobs[step].copy_(torch.from_numpy(g["logprobs"][step]))
        L.values[step].copy_(torch.from_numpy(g["values"][step]))
        L.store_reward(step, g["rewards"][step])
        L.observe(step + 1, frames[step + 1], step_done[step + 1])
    np.testing.assert_allclose(L.next_lstm_state[0].cpu().numpy(), g["next_h"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(L.next_lstm_state[1].cpu().numpy(), g["next_c"], rtol=1e-3, atol=2e-4)
    L.finish_rollout()
    # K1 is bit-exact given its inputs; the bootstrap value comes from the device network -> 1e-4
    np.testing.assert_allclose(L.advantages.cpu().numpy(), g["advantages"], rtol=1e-3, atol=3e-4)
    np.testing.assert_allclose(L.returns.cpu().numpy(), g["returns"], rtol=1e-3, atol=3e-4)
    L.advantages.copy_(torch.from_numpy(g["advantages"]))
    L.returns.copy_(torch.from_numpy(g["returns"]))
    np.random.seed(int(g["shuffle_seed"]))
    m = L.update(float(g["lr"]))
    assert m["num_updates"] == 4
    np.testing.assert_allclose(m["loss"], float(g["last_loss"]), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(m["value_loss"], float(g["last_v_loss"]), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(m["entropy"], float(g["last_entropy"]), rtol=1e-3)
    delta = L.flat.params[::stride].cpu().numpy() - g["init_params_sub"]
    want = g["final_params_sub"] - g["init_params_sub"]
    close = np.isclose(delta, want, rtol=5e-2, atol=2e-5)
    # four Adam steps of ~lr each: ill-conditioned (tiny-gradient) parameters aside, the update must be the reference's
    assert close.mean() > 0.98, f"only {close.mean():.4f} of sampled parameters match the reference update"
    L.flat.check_views()


def test_ppo_atari_lstm_script_runs_on_gpu():
    from cleanrl_amd import ppo_atari_lstm

    L = ppo_atari_lstm.main(["--num-envs", "8", "--num-steps", "16", "--total-timesteps", "256", "--num-minibatches", "4"])
    assert L.hip and L.obs.dtype == torch.uint8 and tuple(L.obs.shape[2:]) == (84, 84, 1)
    assert np.isfinite(L.last_metrics["loss"]) and L.last_metrics["num_updates"] == 16


def test_ppo_procgen_script_runs_on_gpu_without_relayout():
    """ppo_procgen.py drop-in on the HIP path: pixel-interleaved frames go straight into the uint8 rollout rows."""
    from cleanrl_amd import ppo_procgen

    L = ppo_procgen.main(["--num-envs", "8", "--num-steps", "16", "--total-timesteps", "256", "--num-minibatches", "4"])
    assert L.hip and L.obs.dtype == torch.uint8 and L.hwc_frames and not L.relayout and L.stage_obs is None
    assert tuple(L.obs.shape[2:]) == (64, 64, 3) and np.isfinite(L.last_metrics["loss"])


def test_ppo_rnd_envpool_script_runs_on_gpu():
    """ppo_rnd_envpool.py drop-in on the HIP path: two K1 launches per rollout, K3 on the combined advantage, one flat
    buffer over agent + predictor parameters."""
    from cleanrl_amd import ppo_rnd_envpool

    L = ppo_rnd_envpool.main(["--num-envs", "8", "--num-steps", "16", "--total-timesteps", "256", "--num-minibatches", "4",
                              "--num-iterations-obs-norm-init", "1"])
    assert L.hip and L.obs.dtype == torch.uint8 and L.flat.numel == sum(p.numel() for p in L.combined_parameters)
    assert np.isfinite(L.last_metrics["loss"]) and np.isfinite(L.last_metrics["fwd_loss"])
    assert L.int_returns.abs().sum().item() > 0 and L.curiosity_rewards.min().item() >= 0.0
    L.flat.check_views()


def test_ppg_procgen_script_runs_on_gpu():
    """ppg_procgen.py drop-in on the HIP path: policy phase on K1/K2/K3/K5/K6 with Adam eps 1e-8, the auxiliary buffer in
    HBM, the auxiliary phase accumulating into the flat gradient buffer."""
    from cleanrl_amd import ppg_procgen

    L = ppg_procgen.main(["--num-envs", "8", "--num-steps", "16", "--total-timesteps", "512", "--num-minibatches", "4",
                          "--n-iteration", "2", "--e-auxiliary", "1", "--num-aux-rollouts", "4"])
    assert L.hip and L.adam_eps == 1e-8 and L.aux_obs.is_cuda and L.aux_obs.dtype == torch.uint8
    assert np.isfinite(L.last_metrics["loss"]) and all(np.isfinite(v) for v in L.last_aux.values())
    L.flat.check_views()


def test_ppo_pettingzoo_ma_atari_script_runs_on_gpu():
    """ppo_pettingzoo_ma_atari.py drop-in on the HIP path: K5 without the /255, frame channels scaled afterwards."""
    from cleanrl_amd import ppo_pettingzoo_ma_atari

    L = ppo_pettingzoo_ma_atari.main(["--num-envs", "8", "--num-steps", "16", "--total-timesteps", "256", "--num-minibatches", "4"])
    assert L.hip and L.partial_scale and L.obs.dtype == torch.uint8 and tuple(L.obs.shape[2:]) == (84, 84, 6)
    assert np.isfinite(L.last_metrics["loss"])


# ------------------------------------------------------------------------------------------------------------------------
# Teacher-forced HIP-path runs of the other drop-ins against the goldens minted from the reference's own lines (the same
# fixtures the host-path tests consume).  The device network (MIOpen / hipBLASLt convolutions, another summation order than
# the minting CPU) is held to 1e-3 of its outputs; what the reference sampled is then forced so that every later stage is
# compared on identical inputs.
def _delta_matches(got_now, init_sub, final_sub, what, frac=0.98, rtol=5e-2, atol=2e-5):
    delta, want = got_now - init_sub, final_sub - init_sub
    close = np.isclose(delta, want, rtol=rtol, atol=atol)
    assert close.mean() > frac, f"{what}: only {close.mean():.4f} of sampled parameters moved as the reference's did"


def _init_1thread(seed, build):
    n = torch.get_num_threads()
    torch.set_num_threads(1)          # as when minted
    torch.manual_seed(seed)
    out = build()
    torch.set_num_threads(n)
    return out


def test_rnd_hip_path_teacher_forced_against_reference_iteration(monkeypatch):
    from cleanrl_amd.agents import RNDAgent, RNDModel
    from cleanrl_amd.learner_rnd import RNDPPOLearner

    g = load_golden("rnd_iteration")["rnd_T8_N4"]
    T, N = g["rewards"].shape
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8),
                           single_action_space=E.Discrete(int(g["n_actions"])))
    agent, rnd_model = _init_1thread(int(g["init_seed"]), lambda: (RNDAgent(envs), RNDModel(4, envs.single_action_space.n)))
    agent, rnd_model = agent.to(DEV), rnd_model.to(DEV)
    args = default_args(num_steps=T, num_minibatches=2, update_epochs=1, gamma=0.999, int_gamma=0.99, clip_coef=0.1,
                        ent_coef=0.001, update_proportion=0.25, int_coef=1.0, ext_coef=2.0, learning_rate=1e-4)
    L = RNDPPOLearner(agent, rnd_model, args, envs.single_observation_space, envs.single_action_space, N, DEV, sample_seed=1)
    assert L.hip and L.obs.dtype == torch.uint8
    stride = int(g["stride"])
    np.testing.assert_allclose(L.flat.params[::stride].cpu().numpy(), g["init_params_sub"], rtol=1e-4, atol=1e-5)
    L.obs_rms.mean, L.obs_rms.var, L.obs_rms.count = g["obs_mean0"].copy(), g["obs_var0"].copy(), float(g["obs_count0"])
    frames, step_done = g["frames_u8"], g["step_done"]
    F = lambda k: torch.from_numpy(g[k]).to(DEV)
    L.observe(0, frames[0], step_done[0])
    for step in range(T):
        L.act(step)
        np.testing.assert_allclose(L.values[step].cpu().numpy(), g["ext_values"][step], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(L.int_values[step].cpu().numpy(), g["int_values"][step], rtol=1e-3, atol=2e-4)
        L.actions[step].copy_(F("actions")[step]); L.logprobs[step].copy_(F("logprobs")[step])
        L.values[step].copy_(F("ext_values")[step]); L.int_values[step].copy_(F("int_values")[step])
        L.store_reward(step, g["rewards"][step])
        L.observe(step + 1, frames[step + 1], step_done[step + 1])
        L.curiosity(step)
        np.testing.assert_allclose(L.curiosity_rewards[step].cpu().numpy(), g["raw_curiosity"][step], rtol=2e-3, atol=1e-5)
        L.curiosity_rewards[step].copy_(F("raw_curiosity")[step])
    L.finish_rollout()
    assert abs(L.reward_rms.var - float(g["reward_var"])) <= 1e-6 * float(g["reward_var"])
    # two K1 streams on forced inputs: only the bootstrap values come from the device network
    for mine, gold in ((L.curiosity_rewards, "scaled_curiosity"), (L.advantages, "ext_advantages"), (L.returns, "ext_returns"),
                       (L.int_advantages, "int_advantages"), (L.int_returns, "int_returns")):
        np.testing.assert_allclose(mine.cpu().numpy(), g[gold], rtol=1e-3, atol=3e-4, err_msg=gold)
        mine.copy_(F(gold))
    np.random.seed(int(g["shuffle_seed"]))
    torch.manual_seed(int(g["mask_seed"]))
    # the distillation mask (:473 torch.rand(..., device=device)) is forced too: drawn from the CPU generator as when minted
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *a, device=None, **kw: real_rand(*a, **kw).to(device) if device is not None
                        else real_rand(*a, **kw))
    m = L.update(float(g["lr"]))
    monkeypatch.undo()
    assert m["num_updates"] == 2
    np.testing.assert_allclose(L.obs_rms.mean, g["obs_mean1"], rtol=1e-6, atol=1e-6)
    for key, gold in (("policy_loss", "last_pg_loss"), ("value_loss", "last_v_loss"), ("entropy", "last_entropy")):
        ref = float(np.asarray(g[gold]).reshape(-1)[0])
        assert abs(m[key] - ref) <= 2e-3 * max(1.0, abs(ref)), (key, m[key], ref)
    _delta_matches(L.flat.params[::stride].cpu().numpy(), g["init_params_sub"], g["final_params_sub"], "RND update", frac=0.9)
    L.flat.check_views()


def test_ppg_hip_path_teacher_forced_against_reference_phase(capsys):
    from cleanrl_amd.agents import PPGAgent
    from cleanrl_amd.learner_ppg import PPGLearner

    g = load_golden("ppg_phase")["ppg_T8_N4"]
    T, N = g["rewards"].shape
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (64, 64, 3), np.uint8), single_action_space=E.Discrete(15))
    agent = _init_1thread(int(g["init_seed"]), lambda: PPGAgent(envs)).to(DEV)
    args = default_args(num_steps=T, num_minibatches=2, gamma=0.999, clip_coef=0.2, adv_norm_fullbatch=True, e_policy=1,
                        e_auxiliary=2, beta_clone=1.0, num_aux_rollouts=2, n_aux_grad_accum=1, aux_batch_rollouts=N, n_iteration=1,
                        learning_rate=5e-4)
    L = PPGLearner(agent, args, envs.single_observation_space, envs.single_action_space, N, DEV, sample_seed=1)
    assert L.hip and L.adam_eps == 1e-8 and L.obs.dtype == torch.uint8
    stride = int(g["stride"])
    np.testing.assert_allclose(L.flat.params[::stride].cpu().numpy(), g["init_params_sub"], rtol=1e-6, atol=1e-7)
    frames, step_done = g["frames_u8"], g["step_done"]
    F = lambda k: torch.from_numpy(g[k]).to(DEV)
    L.observe(0, frames[0], step_done[0])
    for step in range(T):
        L.act(step)
        np.testing.assert_allclose(L.values[step].cpu().numpy(), g["values"][step], rtol=1e-3, atol=2e-4)
        L.actions[step].copy_(F("actions")[step]); L.logprobs[step].copy_(F("logprobs")[step]); L.values[step].copy_(F("values")[step])
        L.store_reward(step, g["rewards"][step])
        L.observe(step + 1, frames[step + 1], step_done[step + 1])
    L.finish_rollout()
    np.testing.assert_allclose(L.returns.cpu().numpy(), g["returns"], rtol=1e-3, atol=3e-4)
    L.returns.copy_(F("returns"))
    L.advantages.copy_(L.returns - L.values)
    np.random.seed(int(g["shuffle_seed"]))
    m = L.update(float(g["lr"]))                                      # policy phase on K3 / K6 (Adam eps 1e-8)
    np.testing.assert_allclose(L.advantages.reshape(-1).cpu().numpy(), g["b_advantages"], rtol=1e-4, atol=1e-5)
    assert abs(m["loss"] - float(g["policy_loss"])) <= 2e-3 * max(1.0, abs(float(g["policy_loss"])))
    _delta_matches(L.flat.params[::stride].cpu().numpy(), g["init_params_sub"], g["policy_params_sub"], "PPG policy phase", frac=0.9)
    assert torch.equal(L.aux_obs[:, :N].cpu(), torch.from_numpy(g["frames_u8"][:T]))
    aux = L.aux_phase()
    assert "aux epoch 2" in capsys.readouterr().out
    for key in ("kl_loss", "aux_value_loss", "real_value_loss"):
        ref = float(g[key])
        assert abs(aux[key] - ref) <= 5e-2 * max(abs(ref), 1e-3), (key, aux[key], ref)
    _delta_matches(L.flat.params[::stride].cpu().numpy(), g["init_params_sub"], g["final_params_sub"], "PPG auxiliary phase", frac=0.85)
    L.flat.check_views()


@pytest.mark.parametrize("which", ["procgen", "ma_atari"])
def test_procgen_and_ma_atari_hip_minibatch_steps_against_reference_lines(which):
    """ppo_procgen.py / ppo_pettingzoo_ma_atari.py: two consecutive minibatch updates on the HIP path (K5 on pixel-interleaved
    rows -- with only the frame channels scaled for the two-player agent --, K3, K6) against the reference's lines."""
    from cleanrl_amd.agents import MAAtariAgent, ProcgenAgent
    from cleanrl_amd.learner import PPOLearner

    if which == "procgen":
        g, shape, nact, cls, nmb, clip = load_golden("procgen_update")["impala_2steps"], (64, 64, 3), 15, ProcgenAgent, 3, 0.2
    else:
        g, shape, nact, cls, nmb, clip = load_golden("ma_atari_update")["ma_2steps"], (84, 84, 6), 6, MAAtariAgent, 2, 0.1
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, shape, np.uint8), single_action_space=E.Discrete(nact))
    agent = _init_1thread(int(g["init_seed"]), lambda: cls(envs)).to(DEV)
    stride = int(g["stride"])
    B = g["b_actions"].shape[0]
    args = default_args(num_steps=B // 4, num_minibatches=nmb, clip_coef=clip)
    L = PPOLearner(agent, args, envs.single_observation_space, envs.single_action_space, 4, DEV, sample_seed=1)
    assert L.hip and L.hwc_frames and not L.relayout and L.obs.dtype == torch.uint8
    np.testing.assert_allclose(L.flat.params[::stride].cpu().numpy(), g["init_params_sub"], rtol=1e-4, atol=1e-5)
    F = lambda k: torch.from_numpy(g[k]).to(DEV)
    L.obs.view((B,) + shape).copy_(F("b_obs_u8"))
    b_obs = L.obs.reshape((-1,) + L.obs_shape)
    from cleanrl_amd import ops as O

    with torch.no_grad():                              # the device forward on the stored rows vs the reference's values
        if L.partial_scale:
            x = agent.scale_frames_(O.obs_u8_to_f32(b_obs, None, None, False)).permute(0, 3, 1, 2)
        else:
            x = O.obs_u8_to_f32(b_obs).permute(0, 3, 1, 2)
        logits, v = agent.heads(x)
        lp = torch.log_softmax(logits, 1).gather(1, F("b_actions").long().view(-1, 1)).view(-1)
    np.testing.assert_allclose(lp.cpu().numpy(), g["logprob_all"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(v.view(-1).cpu().numpy(), g["value_all"], rtol=1e-3, atol=2e-4)
    M = 16
    sc = torch.zeros(2, 7, device=DEV)
    prev = g["init_params_sub"]
    for k in range(2):
        idx = torch.from_numpy(np.ascontiguousarray(g["perm"][k * M:(k + 1) * M])).to(DEV)
        L._minibatch_hip(idx, b_obs, F("b_actions"), F("b_logprobs"), F("b_advantages"), F("b_returns"), F("b_values"), float(g["lr"]),
                         sc[k])
        ref = float(g["losses"][k])
        assert abs(sc[k, 0].item() - ref) <= 2e-3 * max(1.0, abs(ref)), (k, sc[k, 0].item(), ref)
        now = L.flat.params[::stride].cpu().numpy()
        _delta_matches(now, prev, g[f"params_sub_after_{k + 1}"], f"{which} minibatch step {k + 1}", frac=0.9)
        prev = g[f"params_sub_after_{k + 1}"]
        L.flat.params[::stride].copy_(torch.from_numpy(prev).to(DEV))     # teacher-force the sampled parameters for step 2
    L.flat.check_views()
