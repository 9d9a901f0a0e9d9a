"""ppo_pettingzoo_ma_atari.py drop-in: the 6-channel agent and the learner's host path (pixel-interleaved frames, only the
four frame channels divided by 255) against the reference's own lines (tests/golden/ma_atari_update.npz, minted by
oracle/mint_goldens.py::mint_ma_atari_update from cleanrl/ppo_pettingzoo_ma_atari.py:86-118 Agent and :250-290 update),
the argparse flag surface of the reference (strtobool flags included), the stand-in env and the CLI."""
import ast
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from cleanrl_amd import cli, envs as E
from cleanrl_amd.agents import MAAtariAgent
from cleanrl_amd.learner import PPOLearner
from cleanrl_amd.learner_smoke import default_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/cleanrl/ppo_pettingzoo_ma_atari.py"


@pytest.fixture
def one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)          # as when the goldens were minted
    yield
    torch.set_num_threads(n)


def _flat(agent):
    return torch.cat([p.detach().reshape(-1) for p in agent.parameters()])


def test_ma_atari_host_minibatch_steps_match_the_reference_lines(one_thread):
    g = load_golden("ma_atari_update")["ma_2steps"]
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (84, 84, 6), np.uint8), single_action_space=E.Discrete(6))
    torch.manual_seed(int(g["init_seed"]))
    agent = MAAtariAgent(envs)
    stride = int(g["stride"])
    assert torch.equal(_flat(agent)[::stride], torch.from_numpy(g["init_params_sub"]))
    B = g["b_actions"].shape[0]
    args = default_args(num_steps=B // 4, num_minibatches=2, clip_coef=0.1)
    L = PPOLearner(agent, args, envs.single_observation_space, envs.single_action_space, 4, torch.device("cpu"))
    assert L.hwc_frames and L.partial_scale and not L.relayout and tuple(L.obs.shape[2:]) == (84, 84, 6)
    b_obs = torch.from_numpy(g["b_obs_u8"]).float()
    keep = b_obs.clone()
    with torch.no_grad():
        _, lp, _, v = agent.get_action_and_value(b_obs, torch.from_numpy(g["b_actions"]).long())
    assert torch.equal(lp, torch.from_numpy(g["logprob_all"])) and torch.equal(v.view(-1), torch.from_numpy(g["value_all"]))
    T = lambda k: torch.from_numpy(g[k])
    M = 16
    for k in range(2):
        sc = L._minibatch_host(g["perm"][k * M:(k + 1) * M], b_obs, T("b_actions"), T("b_logprobs"), T("b_advantages"),
                               T("b_returns"), T("b_values"), float(g["lr"]))
        assert abs(sc[0].item() - float(g["losses"][k])) <= 1e-6 * max(1.0, abs(float(g["losses"][k])))
        assert (_flat(agent)[::stride] - T(f"params_sub_after_{k + 1}")).abs().max().item() <= 1e-7
    assert torch.equal(b_obs, keep)                                   # the stored observations are never scaled in place


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference tree not present on this box")
def test_flag_surface_matches_the_reference_argparse():
    """Every ``parser.add_argument`` of the reference exists as a field with the same default, under both spellings."""
    from cleanrl_amd.ppo_pettingzoo_ma_atari import Args

    tree = ast.parse(open(REF).read())
    ref = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            flag = ast.literal_eval(node.args[0])
            default = [k.value for k in node.keywords if k.arg == "default"][0]
            try:
                ref[flag.lstrip("-").replace("-", "_")] = ast.literal_eval(default)
            except ValueError:
                pass                                                   # exp_name: os.path.basename(__file__).rstrip(".py")
    assert len(ref) == 24
    a = cli.parse(Args, [])
    for name, default in ref.items():
        assert getattr(a, name) == default, (name, getattr(a, name), default)
    assert a.exp_name == "ppo_pettingzoo_ma_atari"
    b = cli.parse(Args, ["--cuda", "False", "--capture_video", "--anneal-lr", "false", "--norm-adv", "True", "--env-id", "surround_v2"])
    assert (b.cuda, b.capture_video, b.anneal_lr, b.norm_adv, b.env_id) == (False, True, False, True, "surround_v2")


def test_synthetic_two_player_env_contract():
    env = E.SyntheticMAAtariVecEnv(6, seed=4)
    obs = env.reset()
    assert obs.shape == (6, 84, 84, 6) and obs.dtype == np.uint8
    assert np.array_equal(obs[0, ..., :4], obs[1, ..., :4])                     # the two players of a game see the same frames
    assert (obs[0, ..., 4] == 1).all() and (obs[0, ..., 5] == 0).all() and (obs[1, ..., 5] == 1).all()
    ends = 0
    for _ in range(400):
        obs, r, d, info = env.step(np.zeros(6, np.int64))
        assert np.array_equal(r[0::2], -r[1::2]) and np.array_equal(d[0::2], d[1::2]) and len(info) == 6
        ends += int(d.sum())
    assert ends >= 2


def test_ppo_pettingzoo_ma_atari_cli_runs_on_cpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "cleanrl_amd", "ppo_pettingzoo_ma_atari.py"), "--cuda", "False",
                          "--num-envs", "4", "--num-steps", "8", "--total-timesteps", "64", "--num-minibatches", "2",
                          "--update-epochs", "1"], capture_output=True, text=True, cwd="/tmp", timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("SPS:") == 2
