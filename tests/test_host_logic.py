"""Host-side logic on CPU: environments, flat buffers, the learner's host path against the goldens minted
from the reference's lines (including the 2-rank data-parallel step over gloo)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden
from cleanrl_amd import envs as E
from cleanrl_amd.agents import AtariAgent, ContinuousAgent, MlpAgent
from cleanrl_amd.flat import FlatParams
from cleanrl_amd.learner import PPOLearner
from cleanrl_amd.learner_smoke import default_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cartpole_env_contract():
    env = E.CartPoleVecEnv(3, seed=0)
    obs, info = env.reset(seed=5)
    assert obs.shape == (3, 4) and obs.dtype == np.float32 and np.abs(obs).max() <= 0.05
    total, ends = 0, 0
    for _ in range(600):
        obs, r, term, trunc, infos = env.step(np.ones(3, np.int64))      # always push right -> falls quickly
        total += 1
        if "final_info" in infos:
            for fi in infos["final_info"]:
                if fi:
                    ends += 1
                    assert 5 <= fi["episode"]["l"][0] <= 60
    assert ends > 20 and (r == 1).all()


def test_synthetic_atari_env_is_deterministic_and_frame_stacked():
    a, b = E.SyntheticAtariVecEnv(5, seed=3), E.SyntheticAtariVecEnv(5, seed=3)
    oa, _ = a.reset(seed=3)
    ob, _ = b.reset(seed=3)
    assert oa.dtype == np.uint8 and oa.shape == (5, 4, 84, 84) and np.array_equal(oa, ob)
    prev = oa
    for _ in range(20):
        oa, ra, term, trunc, _ = a.step(np.zeros(5, np.int64))
        ob, rb, _, _, _ = b.step(np.ones(5, np.int64))
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb)
        assert set(np.unique(ra)) <= {-1.0, 0.0, 1.0}
        cont = ~term
        assert np.array_equal(oa[cont][:, :3], prev[cont][:, 1:])        # FrameStack: 3 of 4 planes carry over
        prev = oa
    g = E.SyntheticAtariVecEnv(2, seed=1, api="gym")
    o = g.reset()
    o, r, d, info = g.step(np.zeros(2, np.int64))
    assert set(info) >= {"lives", "r", "l", "reward", "terminated"}


def test_same_seed_same_initial_weights_as_reference_layout():
    """Construction order mirrors the reference, so parameters()/state_dict keys line up with its Agent."""
    env = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    torch.manual_seed(1)
    agent = AtariAgent(env)
    keys = list(agent.state_dict().keys())
    assert keys == ["network.0.weight", "network.0.bias", "network.2.weight", "network.2.bias", "network.4.weight",
                    "network.4.bias", "network.7.weight", "network.7.bias", "actor.weight", "actor.bias", "critic.weight",
                    "critic.bias"]
    assert sum(p.numel() for p in agent.parameters()) == 1686693
    g = load_golden("update_step")["multigpu_cnn_world2"]
    flat = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
    # identical init for seed 1 (orthogonal_'s LAPACK QR rounds differently with another thread count: 1 ulp)
    np.testing.assert_allclose(flat[::int(g["stride"])].numpy(), g["init_params_sub"], rtol=1e-5, atol=1e-6)


def test_flat_params_views_survive_backward():
    env = SimpleNamespace(single_observation_space=E.Box(-1, 1, (4,)), single_action_space=E.Discrete(2))
    agent = MlpAgent(env)
    ref = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).clone()
    flat = FlatParams(agent)
    assert torch.equal(flat.params, ref)
    logits, v = agent.heads(torch.randn(8, 4))
    (logits.sum() + v.sum()).backward()
    flat.check_views()
    assert flat.grads.abs().sum() > 0
    g1 = flat.grads.clone()
    logits, v = agent.heads(torch.randn(8, 4))
    (logits.sum() + v.sum()).backward()
    flat.check_views()                                   # accumulation happened in place
    assert not torch.equal(flat.grads, g1)


def _host_learner_from_golden(g, world_size=1):
    env = SimpleNamespace(single_observation_space=E.Box(-1, 1, (4,)), single_action_space=E.Discrete(2))
    agent = MlpAgent(env)
    with torch.no_grad():
        off = 0
        for p in agent.parameters():
            p.copy_(torch.from_numpy(g["init_params"][off:off + p.numel()]).view_as(p))
            off += p.numel()
    args = default_args(num_steps=128, num_minibatches=4, clip_coef=0.2)
    return agent, PPOLearner(agent, args, env.single_observation_space, env.single_action_space, 4, torch.device("cpu"))


def test_host_minibatch_step_matches_reference_goldens():
    """learner._minibatch_host == the reference's ppo.py:250-290 executed verbatim (3 consecutive updates)."""
    g = load_golden("update_step")["ppo_mlp_3steps"]
    agent, learner = _host_learner_from_golden(g)
    T_ = torch.from_numpy
    for k in range(3):
        sc = learner._minibatch_host(g["perm"][k * 128:(k + 1) * 128], T_(g["b_obs"]), T_(g["b_actions"]),
                                     T_(g["b_logprobs"]), T_(g["b_advantages"]), T_(g["b_returns"]), T_(g["b_values"]),
                                     float(g["lr"]))
        np.testing.assert_allclose(sc[0].item(), g["losses"][k], rtol=1e-6)
        flat = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
        np.testing.assert_allclose(flat.numpy(), g[f"params_after_{k + 1}"], rtol=1e-6, atol=1e-8)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = load_golden("update_step")["multigpu_cnn_world2"]
    env = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    torch.manual_seed(int(g["init_seed"]))
    agent = AtariAgent(env)
    args = default_args(num_steps=8, num_minibatches=2, clip_coef=0.1)
    learner = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, 8, torch.device("cpu"),
                         world_size=world)
    T_ = torch.from_numpy
    learner._minibatch_host(g["mb_inds"], T_(g[f"obs_u8_rank{rank}"]).float(), T_(g[f"b_actions_rank{rank}"]),
                            T_(g[f"b_logprobs_rank{rank}"]), T_(g[f"b_advantages_rank{rank}"]),
                            T_(g[f"b_returns_rank{rank}"]), T_(g[f"b_values_rank{rank}"]), float(g["lr"]))
    flat = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
    q.put((rank, flat[::int(g["stride"])].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_dp_step_matches_reference_collective_block():
    """world_size=2 over gloo: per-rank minibatch, flat-grad SUM all-reduce, /world_size, clip, Adam ==
    ppo_atari_multigpu.py:320-377 executed verbatim against a 2-rank stand-in collective (golden)."""
    g = load_golden("update_step")["multigpu_cnn_world2"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0], res[1]), "replicas diverged"
    # first Adam step moves every parameter by ~lr*sign(g): compare the update itself
    delta = res[0] - g["init_params_sub"]
    close = np.isclose(delta, g["delta_sub"], rtol=1e-3, atol=2e-6)
    assert close.mean() > 0.999, f"only {close.mean():.4f} of sampled parameters match the reference update"


def test_learner_host_path_learns_cartpole():
    """End-to-end on CPU (BASELINE config A shape: N=4, T=128): PPO must learn CartPole-v1."""
    from cleanrl_amd import ppo

    learner = ppo.main(["--no-cuda", "--total-timesteps", "60000", "--seed", "1"])
    # evaluate the greedy policy for a few episodes
    env = E.CartPoleVecEnv(8, seed=123)
    obs, _ = env.reset(seed=123)
    lengths = []
    steps = np.zeros(8)
    for _ in range(2000):
        with torch.no_grad():
            logits, _ = learner.agent.heads(torch.from_numpy(obs))
        obs, r, term, trunc, infos = env.step(logits.argmax(-1).numpy())
        if "final_info" in infos:
            lengths += [fi["episode"]["l"][0] for fi in infos["final_info"] if fi]
    assert len(lengths) > 0 and np.mean(lengths) > 150, f"mean greedy episode length {np.mean(lengths):.1f}"


def test_whole_atari_iteration_matches_the_reference_lines():
    """ppo_atari_envpool.py (config B's script), one whole iteration through the learner's API on CPU: sampled actions,
    log-probs, values, GAE bit-equal to the reference's own lines (:223-232, :250-263); parameters after the 2 x 2
    minibatch updates (:266-326) within 1e-7 (tests/golden/atari_iteration.npz)."""
    g = load_golden("atari_iteration")["atari_T8_N4"]
    T, N = g["rewards"].shape
    n = torch.get_num_threads()
    torch.set_num_threads(1)                       # as when minted
    try:
        env = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
        torch.manual_seed(int(g["init_seed"]))
        agent = AtariAgent(env)
        stride = int(g["stride"])
        flat = lambda: torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
        assert torch.equal(flat()[::stride], torch.from_numpy(g["init_params_sub"]))
        args = default_args(num_steps=T, num_minibatches=2, update_epochs=2)
        L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, torch.device("cpu"))
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
        assert torch.equal(L.advantages, torch.from_numpy(g["advantages"])) and torch.equal(L.returns, torch.from_numpy(g["returns"]))
        np.random.seed(int(g["shuffle_seed"]))
        m = L.update(float(g["lr"]))
        assert (flat()[::stride] - torch.from_numpy(g["final_params_sub"])).abs().max().item() <= 1e-7
        assert abs(flat().double().sum().item() - float(g["final_checksum"])) <= 1e-4
        assert m["num_updates"] == 4 and abs(m["loss"] - float(g["last_loss"])) <= 1e-6
        assert abs(m["value_loss"] - float(g["last_v_loss"])) <= 1e-6 and abs(m["entropy"] - float(g["last_entropy"])) <= 1e-6
    finally:
        torch.set_num_threads(n)


def test_whole_continuous_iteration_matches_the_reference_lines():
    """ppo_continuous_action.py (BASELINE configs[4]'s script), one whole iteration through the learner's API on CPU against
    tests/golden/continuous_iteration.npz (the script's own lines :134-141, :232-246, :248-309 exec'd on HalfCheetah-shaped
    synthetic inputs; 2 minibatches x 3 epochs = six Adam steps).  Sampled actions, log-probs and values come out of the same
    torch ops as the reference's (bit-equal); GAE runs through ``mi355ppo_gae_f32_cpu`` (bit-equal); the update runs through the
    fused Normal loss twin ``mi355ppo_loss_normal_fwd_bwd_f32_cpu`` -- a few ulp per gradient (libm vs torch's expf / logf, f64
    row-order sums): parameters after the six steps within 5e-7 (measured 3e-8) where the update moves them by 1e-3 on average,
    every minibatch's loss scalars, the first step's clipped gradient and the shared ``actor_logstd`` after every step held too."""
    g = load_golden("continuous_iteration")["mujoco_T16_N4"]
    T, N = g["rewards"].shape
    OBS, ACT = g["obs_seq"].shape[-1], g["actions"].shape[-1]
    n = torch.get_num_threads()
    torch.set_num_threads(1)                       # as when minted
    try:
        env = SimpleNamespace(single_observation_space=E.Box(-np.inf, np.inf, (OBS,)), single_action_space=E.Box(-1.0, 1.0, (ACT,)))
        torch.manual_seed(int(g["init_seed"]))
        agent = ContinuousAgent(env)
        flat = lambda: torch.cat([p.detach().reshape(-1) for p in agent.parameters()])      # noqa: E731
        np.testing.assert_allclose(flat().numpy(), g["init_params"], rtol=0, atol=3e-7)     # (orthogonal_: LAPACK QR moves by ulps with the host CPU)
        args = default_args(num_steps=T, num_minibatches=2, update_epochs=3, clip_coef=0.2, ent_coef=0.0, learning_rate=3e-4)
        L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, torch.device("cpu"))
        obs_seq, step_done = g["obs_seq"], g["step_done"]
        L.observe(0, obs_seq[0], step_done[0])
        torch.manual_seed(int(g["sample_seed"]))
        for step in range(T):
            L.act(step)
            L.store_reward(step, g["rewards"][step])
            L.observe(step + 1, obs_seq[step + 1], step_done[step + 1])
        for mine, gold in ((L.actions, "actions"), (L.logprobs, "logprobs"), (L.values, "values")):
            np.testing.assert_allclose(mine.numpy(), g[gold], rtol=0, atol=2e-6, err_msg=gold)
            getattr(L, gold).copy_(torch.from_numpy(g[gold]))             # teacher-forced from here: the update sees the golden rollout
        L.finish_rollout()
        np.testing.assert_allclose(L.advantages.numpy(), g["advantages"], rtol=0, atol=5e-6)   # bootstrap value from the agent
        np.testing.assert_allclose(L.returns.numpy(), g["returns"], rtol=0, atol=5e-6)
        L.advantages.copy_(torch.from_numpy(g["advantages"]))
        L.returns.copy_(torch.from_numpy(g["returns"]))
        # record what the reference recorded: scalars of every minibatch, actor_logstd after every optimizer step
        scal, logstd, grads = [], [], []
        real_mb, real_step = L._minibatch_host, L.optimizer.step

        def mb(*a, **kw):
            out = real_mb(*a, **kw)
            scal.append(out.clone())
            logstd.append(agent.actor_logstd.detach().clone().reshape(-1))
            return out

        def step(*a, **kw):
            if not grads:
                grads.append(torch.cat([p.grad.reshape(-1) for p in agent.parameters()]).clone())      # after clip_grad_norm_
            return real_step(*a, **kw)

        L._minibatch_host, L.optimizer.step = mb, step
        np.random.seed(int(g["shuffle_seed"]))
        m = L.update(float(g["lr"]))
        assert m["num_updates"] == 6 and len(scal) == 6
        names = list(g["scalar_names"])
        mine = torch.stack(scal).numpy()              # loss, pg, v, entropy, old_kl, kl, clipfrac  (ops.LOSS_SCALAR_NAMES order)
        col = {"loss": 0, "pg_loss": 1, "v_loss": 2, "entropy_loss": 3, "old_approx_kl": 4, "approx_kl": 5}
        for j, name in enumerate(names):
            np.testing.assert_allclose(mine[:, col[name]], g["scalars"][:, j], rtol=2e-5, atol=2e-6, err_msg=name)
        np.testing.assert_allclose(mine[:, 6], g["clipfracs"], atol=1e-6)
        # the clipped gradient of the first optimizer step: direction and scale
        g1, ref1 = grads[0].double().numpy(), g["mb1_grad_sub"].astype(np.float64)
        cos = float(g1 @ ref1 / (np.linalg.norm(g1) * np.linalg.norm(ref1)))
        assert cos > 1 - 1e-9 and abs(np.linalg.norm(g1) - float(g["mb1_grad_norm"])) <= 1e-5 * float(g["mb1_grad_norm"])
        np.testing.assert_allclose(g1, ref1, rtol=0, atol=5e-6 * float(g["mb1_grad_absmax"]))
        # six Adam steps of ~lr = 3e-4 each
        np.testing.assert_allclose(torch.stack(logstd).numpy(), g["logstd_after_step"], rtol=0, atol=2e-8)       # measured 9e-10
        moved = np.abs(g["final_params"] - g["init_params"])
        assert moved.mean() > 2e-4
        np.testing.assert_allclose(flat().numpy(), g["final_params"], rtol=0, atol=5e-7)       # measured 3e-8; the update moves 1e-3
    finally:
        torch.set_num_threads(n)


def test_whole_ppo_iteration_matches_the_reference_lines():
    """cleanrl/ppo.py (BASELINE configs[0]: the CPU configuration), one whole iteration through the learner's API on CPU against
    tests/golden/ppo_iteration.npz (the script's own lines exec'd on CartPole-shaped synthetic inputs; 4 minibatches x 4 epochs
    = sixteen Adam steps).  This is the path config A runs: torch's Categorical for the rollout (bit-equal), ``mi355ppo_gae_f32_cpu``
    (bit-equal) and the fused Categorical loss twin ``mi355ppo_loss_categorical_fwd_bwd_f32_cpu`` for the update."""
    g = load_golden("ppo_iteration")["cartpole_T16_N4"]
    T, N = g["rewards"].shape
    OBS = g["obs_seq"].shape[-1]
    n = torch.get_num_threads()
    torch.set_num_threads(1)                       # as when minted
    try:
        env = SimpleNamespace(single_observation_space=E.Box(-np.inf, np.inf, (OBS,)), single_action_space=E.Discrete(2))
        torch.manual_seed(int(g["init_seed"]))
        agent = MlpAgent(env)
        flat = lambda: torch.cat([p.detach().reshape(-1) for p in agent.parameters()])      # noqa: E731
        np.testing.assert_allclose(flat().numpy(), g["init_params"], rtol=0, atol=3e-7)
        args = default_args(num_steps=T, num_minibatches=4, update_epochs=4, clip_coef=0.2, ent_coef=0.01, learning_rate=2.5e-4)
        L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, torch.device("cpu"))
        obs_seq, step_done = g["obs_seq"], g["step_done"]
        L.observe(0, obs_seq[0], step_done[0])
        torch.manual_seed(int(g["sample_seed"]))
        for step in range(T):
            L.act(step)
            L.store_reward(step, g["rewards"][step])
            L.observe(step + 1, obs_seq[step + 1], step_done[step + 1])
        assert torch.equal(L.actions, torch.from_numpy(g["actions"]))                        # same sampler stream, same draws
        for gold in ("logprobs", "values"):
            np.testing.assert_allclose(getattr(L, gold).numpy(), g[gold], rtol=0, atol=2e-6, err_msg=gold)
            getattr(L, gold).copy_(torch.from_numpy(g[gold]))
        L.finish_rollout()
        np.testing.assert_allclose(L.advantages.numpy(), g["advantages"], rtol=0, atol=5e-6)
        L.advantages.copy_(torch.from_numpy(g["advantages"]))
        L.returns.copy_(torch.from_numpy(g["returns"]))
        scal = []
        real_mb = L._minibatch_host
        L._minibatch_host = lambda *a, **kw: (scal.append(real_mb(*a, **kw).clone()), scal[-1])[1]
        np.random.seed(int(g["shuffle_seed"]))
        m = L.update(float(g["lr"]))
        assert m["num_updates"] == 16 and len(scal) == 16
        mine = torch.stack(scal).numpy()
        np.testing.assert_allclose(mine[:, :6], g["scalars"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(mine[:, 6], g["clipfracs"], atol=1e-6)
        moved = np.abs(g["final_params"] - g["init_params"]).mean()
        assert moved > 5e-4
        np.testing.assert_allclose(flat().numpy(), g["final_params"], rtol=0, atol=5e-7)       # measured 3e-8; sixteen Adam steps move 1e-3
    finally:
        torch.set_num_threads(n)


def test_update_graph_policy_over_rccl_is_self_checked_graphs_in_the_reference_arrangement(monkeypatch):
    """learner.update_graph_policy / early_bucket_policy (round 6, the one place runner.train and bench.py ask): graphs on one GPU and over gloo as they
    are; over nccl (RCCL) graphs only behind the captured-vs-eager self-check, and by default WITHOUT the early bucket (the reference's single all-reduce
    behind the backward: every capture begin / end and every collective on the calling thread); MI355PPO_UPDATE_GRAPHS=1 opts in to the early bucket,
    0 switches the graphs off everywhere."""
    import torch.distributed as dist

    from cleanrl_amd import learner as L

    monkeypatch.delenv("MI355PPO_UPDATE_GRAPHS", raising=False)
    assert L.update_graph_policy(1) == "capture" and L.early_bucket_policy(1)
    for backend, want, early_auto in (("gloo", "capture", True), ("nccl", "capture+check", False)):
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "get_backend", lambda b=backend: b)
        monkeypatch.delenv("MI355PPO_UPDATE_GRAPHS", raising=False)
        assert L.update_graph_policy(8) == want and L.update_graph_policy(1) == "capture"
        assert L.early_bucket_policy(8) is early_auto and L.early_bucket_policy(1)
        monkeypatch.setenv("MI355PPO_UPDATE_GRAPHS", "auto")
        assert L.update_graph_policy(2) == want and L.early_bucket_policy(2) is early_auto
        monkeypatch.setenv("MI355PPO_UPDATE_GRAPHS", "1")
        assert L.update_graph_policy(4) == want and L.early_bucket_policy(4) and L.update_graph_policy(1) == "capture"
        monkeypatch.setenv("MI355PPO_UPDATE_GRAPHS", "0")
        assert L.update_graph_policy(4) == "off" and L.update_graph_policy(1) == "off" and L.early_bucket_policy(4) is early_auto


def test_all_ranks_agree_takes_every_rank_back_when_one_capture_failed():
    """learner.all_ranks_agree over a real gloo group of three CPU ranks: everybody agrees only if nobody failed (the ADVICE item of round 5:
    runner.train used to fall back per rank, so ranks could run different routes through the update)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "MI355PPO_UPDATE_GRAPHS")}
    for bad, want in ((-1, "True"), (1, "False")):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", "--nproc-per-node", "3", "--local-addr",
                              "127.0.0.1", os.path.join("tests", "dp_agree_worker.py"), str(bad)], cwd=root, capture_output=True, text=True,
                             timeout=300, env=env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        for r in range(3):
            assert f"rank {r} agreed={want} policy=capture" in out.stdout, out.stdout
