"""ppo_atari_lstm.py drop-in: the recurrent learner's host path against a whole iteration of the reference's own lines
(tests/golden/lstm_iteration.npz, minted by oracle/mint_goldens.py::mint_lstm_iteration from cleanrl/ppo_atari_lstm.py
:117-165 Agent, :240-249 action logic, :267-284 GAE, :287-357 env-wise minibatch update), plus the script's CLI."""
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from cleanrl_amd import envs as E
from cleanrl_amd.agents import AtariLSTMAgent
from cleanrl_amd.learner_lstm import LSTMPPOLearner
from cleanrl_amd.learner_smoke import default_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat(agent):
    return torch.cat([p.detach().reshape(-1) for p in agent.parameters()])


@pytest.fixture
def one_thread():
    """The goldens were minted with one CPU thread (orthogonal_'s LAPACK QR and the conv / GEMM reductions round
    differently with another thread count -- 1 ulp, enough to flip a sampled action)."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def test_lstm_iteration_matches_the_reference_lines(one_thread):
    g = load_golden("lstm_iteration")["lstm_T8_N4"]
    T, N = g["rewards"].shape
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (1, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    torch.manual_seed(int(g["init_seed"]))
    agent = AtariLSTMAgent(envs)
    stride = int(g["stride"])
    init = _flat(agent)
    assert torch.equal(init[::stride], torch.from_numpy(g["init_params_sub"]))            # same seed -> same weights
    assert abs(init.double().sum().item() - float(g["init_checksum"])) < 1e-9
    keys = list(agent.state_dict().keys())
    assert keys[8:12] == ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0"]
    args = default_args(num_steps=T, num_minibatches=2, update_epochs=2)
    L = LSTMPPOLearner(agent, args, envs.single_observation_space, envs.single_action_space, N, torch.device("cpu"))
    frames, step_done = g["frames_u8"], g["step_done"]
    # rollout through the learner's API, the sampler on the reference's stream
    L.observe(0, frames[0], step_done[0])
    torch.manual_seed(int(g["sample_seed"]))
    for step in range(T):
        L.act(step)
        L.store_reward(step, g["rewards"][step])
        L.observe(step + 1, frames[step + 1], step_done[step + 1])
    assert torch.equal(L.actions, torch.from_numpy(g["actions"]))
    assert torch.equal(L.logprobs, torch.from_numpy(g["logprobs"]))
    assert torch.equal(L.values, torch.from_numpy(g["values"]))
    assert torch.equal(L.dones, torch.from_numpy(step_done[:T]))
    assert torch.equal(L.next_lstm_state[0], torch.from_numpy(g["next_h"]))
    assert torch.equal(L.next_lstm_state[1], torch.from_numpy(g["next_c"]))
    assert L.initial_lstm_state[0].abs().max().item() == 0.0                               # snapshot taken at step 0
    L.finish_rollout()
    assert torch.equal(L.advantages, torch.from_numpy(g["advantages"]))
    assert torch.equal(L.returns, torch.from_numpy(g["returns"]))
    np.random.seed(int(g["shuffle_seed"]))
    m = L.update(float(g["lr"]))
    final = _flat(agent)
    want = torch.from_numpy(g["final_params_sub"])
    # same torch ops in the same order on the same machine class: equal to the last bits of f32 round-off
    assert (final[::stride] - want).abs().max().item() <= 1e-7
    assert abs(final.double().sum().item() - float(g["final_checksum"])) <= 1e-4
    assert m["num_updates"] == 4
    for key, gold in (("loss", "last_loss"), ("policy_loss", "last_pg_loss"), ("value_loss", "last_v_loss"),
                      ("entropy", "last_entropy"), ("approx_kl", "last_approx_kl")):
        assert abs(m[key] - float(g[gold])) <= 1e-6 * max(1.0, abs(float(g[gold]))), key
    assert abs(m["clipfrac"] - float(np.mean(g["clipfracs"]))) < 1e-7
    # the next iteration snapshots the carried state before its first action
    L.start_iteration()
    L.act(0)
    assert torch.equal(L.initial_lstm_state[0], torch.from_numpy(g["next_h"]))


def test_env_wise_minibatch_indices_are_time_major():
    """:306 ``flatinds[:, mbenvinds].ravel()``: the device-side construction used on the HIP path gives the same rows."""
    T, N = 5, 6
    flatinds = np.arange(T * N).reshape(T, N)
    envinds = np.random.RandomState(0).permutation(N)
    flat_dev = torch.arange(T * N).reshape(T, N)
    for start in range(0, N, 2):
        mb = envinds[start:start + 2]
        want = flatinds[:, mb].ravel()
        got = flat_dev.index_select(1, torch.from_numpy(mb)).reshape(-1).numpy()
        assert np.array_equal(got, want)


def test_ppo_atari_lstm_cli_runs_on_cpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "cleanrl_amd", "ppo_atari_lstm.py"), "--no-cuda", "--num-envs", "4",
                          "--num_steps", "8", "--total-timesteps", "64", "--num-minibatches", "2", "--update-epochs", "1",
                          "--seed", "3"], capture_output=True, text=True, cwd="/tmp", timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    sps = [ln for ln in out.stdout.splitlines() if ln.startswith("SPS:")]
    assert len(sps) == 2 and all(int(ln.split()[1]) > 0 for ln in sps)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "cleanrl_amd", "ppo_atari_lstm.py"), "--no-cuda", "--num-envs", "3",
                          "--num-steps", "4", "--total-timesteps", "12", "--num-minibatches", "2"], capture_output=True,
                         text=True, cwd="/tmp", timeout=600)
    assert bad.returncode != 0 and "divisible" in bad.stderr                               # :296 assert
