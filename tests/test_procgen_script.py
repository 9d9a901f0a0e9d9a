"""ppo_procgen.py drop-in: the IMPALA-CNN agent and the learner's host path on pixel-interleaved frames against the
reference's own lines (tests/golden/procgen_update.npz, minted by oracle/mint_goldens.py::mint_procgen_update from
cleanrl/ppo_procgen.py:86-158 Agent and :284-324 minibatch update), the stand-in env's contract, and the CLI."""
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from cleanrl_amd import envs as E
from cleanrl_amd.agents import ProcgenAgent
from cleanrl_amd.learner import PPOLearner
from cleanrl_amd.learner_smoke import default_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)          # as when the goldens were minted (conv / GEMM reductions round per thread count)
    yield
    torch.set_num_threads(n)


def _flat(agent):
    return torch.cat([p.detach().reshape(-1) for p in agent.parameters()])


def test_procgen_host_minibatch_steps_match_the_reference_lines(one_thread):
    g = load_golden("procgen_update")["impala_2steps"]
    envs = SimpleNamespace(single_observation_space=E.Box(0, 255, (64, 64, 3), np.uint8), single_action_space=E.Discrete(15))
    torch.manual_seed(int(g["init_seed"]))
    agent = ProcgenAgent(envs)
    stride = int(g["stride"])
    # conv layers keep torch's default (kaiming-uniform) init -> no LAPACK in the way: exact
    assert torch.equal(_flat(agent)[::stride], torch.from_numpy(g["init_params_sub"]))
    keys = list(agent.state_dict().keys())
    assert keys[:4] == ["network.0.conv.weight", "network.0.conv.bias", "network.0.res_block0.conv0.weight",
                        "network.0.res_block0.conv0.bias"] and keys[-4:] == ["actor.weight", "actor.bias", "critic.weight", "critic.bias"]
    assert sum(p.numel() for p in agent.parameters()) == 626256
    B = g["b_actions"].shape[0]
    args = default_args(num_steps=B // 4, num_minibatches=3, clip_coef=0.2)
    L = PPOLearner(agent, args, envs.single_observation_space, envs.single_action_space, 4, torch.device("cpu"))
    assert L.hwc_frames and not L.relayout and tuple(L.obs.shape[2:]) == (64, 64, 3)
    b_obs = torch.from_numpy(g["b_obs_u8"]).float()
    with torch.no_grad():
        _, lp, _, v = agent.get_action_and_value(b_obs, torch.from_numpy(g["b_actions"]).long())
    assert torch.equal(lp, torch.from_numpy(g["logprob_all"])) and torch.equal(v.view(-1), torch.from_numpy(g["value_all"]))
    T = lambda k: torch.from_numpy(g[k])
    M = 16
    for k in range(2):
        sc = L._minibatch_host(g["perm"][k * M:(k + 1) * M], b_obs, T("b_actions"), T("b_logprobs"), T("b_advantages"),
                               T("b_returns"), T("b_values"), float(g["lr"]))
        assert abs(sc[0].item() - float(g["losses"][k])) <= 1e-6 * max(1.0, abs(float(g["losses"][k])))
        got = _flat(agent)[::stride]
        assert (got - T(f"params_sub_after_{k + 1}")).abs().max().item() <= 1e-7
    assert abs(_flat(agent).double().sum().item() - float(g["final_checksum"])) <= 1e-4


def test_synthetic_procgen_env_contract():
    a, b = E.SyntheticProcgenVecEnv(6, seed=2), E.SyntheticProcgenVecEnv(6, seed=2)
    oa, ob = a.reset(), b.reset()
    assert oa.shape == (6, 64, 64, 3) and oa.dtype == np.uint8 and np.array_equal(oa, ob)
    ends = 0
    for _ in range(400):
        oa, ra, da, ia = a.step(np.zeros(6, np.int64))
        ob, rb, db, ib = b.step(np.ones(6, np.int64))
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db)
        assert ra.min() >= 0.0 and ra.max() <= 10.0 and isinstance(ia, list) and len(ia) == 6
        for i, item in enumerate(ia):
            assert ("episode" in item) == bool(da[i])
            ends += int(da[i])
    assert ends > 3 and a.ret_rms.count > 2000


def test_ppo_procgen_cli_runs_on_cpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "cleanrl_amd", "ppo_procgen.py"), "--no-cuda", "--num-envs", "4",
                          "--num_steps", "8", "--total-timesteps", "64", "--num-minibatches", "2", "--update-epochs", "1"],
                         capture_output=True, text=True, cwd="/tmp", timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    sps = [ln for ln in out.stdout.splitlines() if ln.startswith("SPS:")]
    assert len(sps) == 2 and all(int(ln.split()[1]) > 0 for ln in sps)
