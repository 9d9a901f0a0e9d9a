"""ppg_procgen.py drop-in: the PPG learner's host path against a whole phase of the reference's own lines
(tests/golden/ppg_phase.npz, minted by oracle/mint_goldens.py::mint_ppg_phase from cleanrl/ppg_procgen.py:100-211 Agent,
the GAE lines, :336-398 policy-phase update with full-batch advantage normalisation, :416-418 aux-buffer storage and
:421-474 the auxiliary phase), plus the script's CLI."""
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from cleanrl_amd import envs as E
from cleanrl_amd.agents import PPGAgent
from cleanrl_amd.learner_ppg import PPGLearner, flatten01, unflatten01
from cleanrl_amd.learner_smoke import default_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)          # as when the goldens were minted
    yield
    torch.set_num_threads(n)


def _flat(agent):
    return torch.cat([p.detach().reshape(-1) for p in agent.parameters()])


def test_ppg_phase_matches_the_reference_lines(one_thread, capsys):
    g = load_golden("ppg_phase")["ppg_T8_N4"]
    T, N = g["rewards"].shape
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (64, 64, 3), np.uint8), single_action_space=E.Discrete(15))
    torch.manual_seed(int(g["init_seed"]))
    agent = PPGAgent(envs)
    stride = int(g["stride"])
    assert torch.equal(_flat(agent)[::stride], torch.from_numpy(g["init_params_sub"]))     # norm-scaled init: no LAPACK
    args = default_args(num_steps=T, num_minibatches=2, gamma=0.999, clip_coef=0.2, adv_norm_fullbatch=True, e_policy=1,
                        e_auxiliary=2, beta_clone=1.0, num_aux_rollouts=2, n_aux_grad_accum=1, aux_batch_rollouts=N, n_iteration=1,
                        learning_rate=5e-4)
    L = PPGLearner(agent, args, envs.single_observation_space, envs.single_action_space, N, torch.device("cpu"))
    assert args.norm_adv is False and args.update_epochs == 1 and L.optimizer.defaults["eps"] == 1e-8
    frames, step_done = g["frames_u8"], g["step_done"]
    L.observe(0, frames[0], step_done[0])
    torch.manual_seed(int(g["sample_seed"]))
    for step in range(T):
        L.act(step)
        L.store_reward(step, g["rewards"][step])
        L.observe(step + 1, frames[step + 1], step_done[step + 1])
    for mine, gold in ((L.actions, "actions"), (L.logprobs, "logprobs"), (L.values, "values")):
        assert torch.equal(mine, torch.from_numpy(g[gold])), gold
    L.finish_rollout()
    assert torch.equal(L.returns, torch.from_numpy(g["returns"]))
    np.random.seed(int(g["shuffle_seed"]))
    m = L.update(float(g["lr"]))                                      # policy phase
    assert torch.equal(L.advantages.reshape(-1), torch.from_numpy(g["b_advantages"]))     # full-batch normalisation
    # (the policy phase's loss runs through the fused host twin of K3, mi355ppo_loss_categorical_fwd_bwd_f32_cpu: libm expf / logf
    # and f64 row-order sums instead of torch's kernels -- a few ulp per gradient, which Adam's normalisation turns into at most
    # 1.2e-6 here; one Adam step moves a parameter by lr = 5e-4)
    assert (_flat(agent)[::stride] - torch.from_numpy(g["policy_params_sub"])).abs().max().item() <= 4e-6
    assert abs(m["loss"] - float(g["policy_loss"])) <= 1e-6 * max(1.0, abs(float(g["policy_loss"])))
    assert torch.equal(L.aux_obs[:, :N], torch.from_numpy(g["frames_u8"][:T])) and torch.equal(L.aux_returns[:, :N], L.returns)
    aux = L.aux_phase()                                               # auxiliary phase (continues the numpy shuffle stream)
    assert "aux epoch 2" in capsys.readouterr().out
    assert (_flat(agent)[::stride] - torch.from_numpy(g["final_params_sub"])).abs().max().item() <= 4e-6
    assert abs(_flat(agent).double().sum().item() - float(g["final_checksum"])) <= 1e-4
    for key in ("kl_loss", "aux_value_loss", "real_value_loss"):
        ref = float(g[key])
        assert abs(aux[key] - ref) <= 2e-6 * max(1.0, abs(ref)), (key, aux[key], ref)
    assert L._aux_update == 0


def test_flatten_unflatten_roundtrip():
    a = torch.rand(7, 3, 5, 5, 2)                                     # ppg_procgen.py:115-119
    assert torch.equal(unflatten01(flatten01(a), a.shape[:2]), a)


def test_ppg_procgen_cli_runs_on_cpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "cleanrl_amd", "ppg_procgen.py"), "--no-cuda", "--num-envs", "4",
                          "--num_steps", "8", "--total-timesteps", "128", "--num-minibatches", "2", "--n-iteration", "2",
                          "--e-auxiliary", "1", "--num-aux-rollouts", "4"], capture_output=True, text=True, cwd="/tmp", timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("SPS:") == 4 and out.stdout.count("aux epoch 1") == 2      # 2 phases x 2 policy iterations
