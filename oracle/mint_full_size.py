"""Mint the whole-iteration goldens of BASELINE.json's configs at their FULL size by executing the reference's own lines.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Build container only (needs ``/root/reference``):

    python -m oracle.mint_full_size [C] [D] [E] [A]

=====  ==========================================  =========================================  ==========================
config  reference lines                            shape                                      fixture
=====  ==========================================  =========================================  ==========================
C       cleanrl/ppo_atari.py:234-311               1024 envs x 128 steps, 16 updates x 32,768   atari_iteration_cfgC.npz
D       cleanrl/ppo_atari_multigpu.py:287-377      2 ranks x (256 envs x 128 steps), 16 x 8,192  atari_iteration_cfgD.npz
E       cleanrl/ppo_continuous_action.py:232-309   64 envs x 2048 steps, 320 updates x 4,096    continuous_iteration_cfgE.npz
A       cleanrl/ppo.py:217-294                     4 envs x 128 steps, 16 updates x 128         ppo_iteration_cfgA.npz
=====  ==========================================  =========================================  ==========================

As in ``mint_goldens.mint_atari_iteration_config_b``: the reference ``Agent``'s own action logic fills the rollout on seeded
synthetic inputs that BOTH sides regenerate from the seed (``cleanrl_amd.synthetic``; a checksum is stored), then the GAE lines
and the flatten + epochs x minibatches lines are ``exec``'d verbatim.  Recorded: the rollout tensors, the GAE output, the
scalars of EVERY minibatch, the clipped flat gradient ``optimizer.step()`` saw at three updates, the final parameters.

C and D are 28 / 14 TFLOP of f32 convolutions on the CPU: they run with 8 torch threads (B was minted with one); the thread
count is stored.  D's two ranks run as two Python threads over the reference's collective block (:360-374) with an in-process
SUM all-reduce standing in for ``torch.distributed``; every rank has its own ``numpy`` legacy ``RandomState`` (the reference's
ranks are processes with ``np.random.seed(args.seed + rank)``, :206-210).
"""
from __future__ import annotations

import os
import sys
import textwrap
import threading
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import synthetic  # noqa: E402  (input generators only)
from oracle import ref_extract as R  # noqa: E402
from oracle.mint_goldens import SCALAR_KEYS, _flat, _grad_record, _GradSpy, _np, _save  # noqa: E402


def _update_ranges(lines):
    g0 = R._find(lines, "# bootstrap value if not done") + 1
    g1 = R._find(lines, "returns = advantages + values", g0)
    u0 = R._find(lines, "# flatten the batch", g1) + 1
    u1 = R._find(lines, "y_pred, y_true = b_values.cpu().numpy()", u0)
    return g0, g1, u0, u1


def _atari_inputs(T, N, frame_seed):
    frames = torch.from_numpy(synthetic.atari_frames((T + 1) * N, seed=frame_seed)).view(T + 1, N, 4, 84, 84)
    rs = np.random.RandomState(frame_seed + 1)
    step_done = torch.from_numpy((rs.random_sample((T + 1, N)) < 1.0 / 50.0).astype(np.float32))
    step_done[0] = 0.0
    rewards = torch.from_numpy(rs.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(T, N), p=[0.05, 0.9, 0.05]).astype(np.float32))
    return frames, step_done, rewards


def _atari_rollout(agent, frames, step_done, T, N, sample_seed):
    obs = torch.zeros((T, N, 4, 84, 84))
    actions, logprobs, dones, values = (torch.zeros((T, N)) for _ in range(4))
    torch.manual_seed(sample_seed)                              # the sampler's stream
    for step in range(T):
        obs[step], dones[step] = frames[step].float(), step_done[step]
        with torch.no_grad():
            action, logprob, _, value = agent.get_action_and_value(obs[step])
            values[step] = value.flatten()
        actions[step], logprobs[step] = action, logprob
    return obs, actions, logprobs, dones, values


def _record_common(d, init, final, optimizer, agent, keep, scalars, extra_stride=53):
    sub = slice(0, None, 89)
    d.update(init_params_sub=init[sub], final_params_sub=final[sub], stride=np.int64(89),
             init_checksum=np.float64(init.double().sum().item()), final_checksum=np.float64(final.double().sum().item()),
             scalars=np.array(scalars, np.float32), scalar_names=np.array(list(SCALAR_KEYS) + ["clipfrac"]),
             grad_updates=np.array(keep, np.int64))
    for k, gk in zip(keep, optimizer.grads):
        d.update(_grad_record(f"mb{k}_grad", gk, agent.parameters(), stride=extra_stride))


# ------------------------------------------------------------------------------------------------------------- config C
def mint_config_c(threads=8, save=True, T=128, N=1024, name="atari_iteration_cfgC"):
    """BASELINE configs[2] (the configuration the metric is quoted on): ppo_atari.py, 1024 envs x 128 steps, 4 epochs x 4
    minibatches = 16 updates of 32,768 rows."""
    torch.set_num_threads(threads)
    script = "ppo_atari.py"
    lines = R._read(script)
    A, frame_seed = 4, 91
    torch.manual_seed(21)
    Agent, _ = R.load_agent_class(script)
    envs = R.fake_envs((4, 84, 84), n_actions=A)
    agent = Agent(envs)
    args = R.make_args(num_steps=T, num_envs=N, num_minibatches=4, update_epochs=4, batch_size=T * N, minibatch_size=T * N // 4)
    ns, scalars = {}, []

    def on_step(k):
        scalars.append([float(ns[key]) for key in SCALAR_KEYS] + [float(ns["clipfracs"][-1])])
        print(f"  [{name}] update {k}/16 loss {scalars[-1][0]:.6f}  ({time.time() - t0:.0f} s)", flush=True)

    keep = (1, 8, 16)
    optimizer = _GradSpy(R.make_optimizer(agent, 2.5e-4), agent.parameters(), keep_steps=keep, on_step=on_step)
    init = _flat(agent.parameters()).clone()
    t0 = time.time()
    frames, step_done, rewards = _atari_inputs(T, N, frame_seed)
    obs, actions, logprobs, dones, values = _atari_rollout(agent, frames, step_done, T, N, 23)
    print(f"  [{name}] rollout done ({time.time() - t0:.0f} s)", flush=True)
    next_obs, next_done = frames[T].float(), step_done[T]
    ns.update(args=args, agent=agent, optimizer=optimizer, envs=envs, obs=obs, actions=actions, logprobs=logprobs,
              rewards=rewards, dones=dones, values=values, next_obs=next_obs, next_done=next_done, device=torch.device("cpu"),
              np=np, torch=torch, nn=nn)
    g0, g1, u0, u1 = _update_ranges(lines)
    exec(textwrap.dedent("\n".join(lines[g0:g1 + 1])), ns)
    np.random.seed(6)
    exec(textwrap.dedent("\n".join(lines[u0:u1])), ns)
    assert optimizer.n_steps == 16 and len(optimizer.grads) == 3
    final = _flat(agent.parameters())
    d = dict(frame_seed=np.int64(frame_seed), frames_checksum=np.int64(frames.sum(dtype=torch.int64).item()),
             frames_first_row=frames[0, 0, 0, 0].clone(), step_done=step_done, rewards=rewards, actions=actions.to(torch.uint8),
             logprobs=logprobs, values=values, advantages=ns["advantages"], returns=ns["returns"], init_seed=np.int64(21),
             sample_seed=np.int64(23), shuffle_seed=np.int64(6), lr=np.float64(2.5e-4), torch_threads=np.int64(threads),
             lines=np.array([g0 + 1, g1 + 1, u0 + 1, u1], np.int64))
    _record_common(d, init, final, optimizer, agent, keep, scalars)
    if save:
        _save(name, {f"atari_T{T}_N{N}": d})
    return _np(d)


# ------------------------------------------------------------------------------------------------------------- config D
class _ThreadDist:
    """In-process stand-in for ``torch.distributed`` inside ppo_atari_multigpu.py:360-374: a SUM all-reduce over the ranks,
    each a Python thread executing the same reference lines.  The sum is formed in rank order (rank 0's tensor + rank 1's)."""

    class ReduceOp:
        SUM = "sum"

    def __init__(self, world):
        self.world, self.slots = world, [None] * world
        self.b1, self.b2 = threading.Barrier(world), threading.Barrier(world)

    def handle(self, rank):
        parent = self

        class _H:
            ReduceOp = _ThreadDist.ReduceOp

            @staticmethod
            def all_reduce(tensor, op=None):
                parent.slots[rank] = tensor
                parent.b1.wait()
                total = parent.slots[0].clone()
                for r in range(1, parent.world):
                    total.add_(parent.slots[r])
                parent.b2.wait()                       # everybody has read every slot
                tensor.copy_(total)

        return _H


def mint_config_d(threads=4, save=True, T=128, N=256, world=2, name="atari_iteration_cfgD"):
    """BASELINE configs[3] at its per-GPU size with the smallest world that has a collective: ppo_atari_multigpu.py, 2 ranks x
    (256 envs x 128 steps), 16 updates of 8,192 local rows, gradients SUM-all-reduced and divided by world_size (:360-374)."""
    torch.set_num_threads(threads)
    script = "ppo_atari_multigpu.py"
    lines = R._read(script)
    A, seed = 4, 1
    Agent, _ = R.load_agent_class(script)
    envs = R.fake_envs((4, 84, 84), n_actions=A)
    g0, g1, u0, u1 = _update_ranges(lines)
    tdist = _ThreadDist(world)
    keep = (1, 8, 16)
    ranks, errors = [None] * world, []
    t0 = time.time()

    def run(rank):
        try:
            # seeding protocol of :206-212,231: same torch seed for the replicas' initial weights, per-rank streams afterwards
            torch.manual_seed(21)                                  # (serialised below: the global torch RNG is shared by the threads)
            agent = Agent(envs)
            args = R.make_args(num_steps=T, local_num_envs=N, num_minibatches=4, update_epochs=4, local_batch_size=T * N,
                               local_minibatch_size=T * N // 4, world_size=world)
            ns, scalars = {}, []

            def on_step(k):
                scalars.append([float(ns[key]) for key in SCALAR_KEYS] + [float(ns["clipfracs"][-1])])
                if rank == 0:
                    print(f"  [{name}] update {k}/16 loss {scalars[-1][0]:.6f}  ({time.time() - t0:.0f} s)", flush=True)

            optimizer = _GradSpy(R.make_optimizer(agent, 2.5e-4), agent.parameters(), keep_steps=keep, on_step=on_step)
            init = _flat(agent.parameters()).clone()
            frames, step_done, rewards = _atari_inputs(T, N, 300 + 10 * rank)
            ranks[rank] = dict(agent=agent, args=args, ns=ns, scalars=scalars, optimizer=optimizer, init=init, frames=frames,
                               step_done=step_done, rewards=rewards)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            raise

    for r in range(world):                 # construction + rollout serially: they draw from torch's global generator
        run(r)
        st = ranks[r]
        st["rollout"] = _atari_rollout(st["agent"], st["frames"], st["step_done"], T, N, 23 + r)
        print(f"  [{name}] rank {r} rollout done ({time.time() - t0:.0f} s)", flush=True)
    assert torch.equal(ranks[0]["init"], ranks[1]["init"])

    def update(rank):
        try:
            st = ranks[rank]
            obs, actions, logprobs, dones, values = st["rollout"]
            rs = np.random.RandomState(seed + rank)                # np.random.seed(args.seed) with args.seed += local_rank
            np_proxy = SimpleNamespace(arange=np.arange, random=rs)
            ns = st["ns"]
            ns.update(args=st["args"], agent=st["agent"], optimizer=st["optimizer"], envs=envs, obs=obs, actions=actions,
                      logprobs=logprobs, rewards=st["rewards"], dones=dones, values=values, next_obs=st["frames"][T].float(),
                      next_done=st["step_done"][T], device=torch.device("cpu"), np=np_proxy, torch=torch, nn=nn,
                      dist=tdist.handle(rank))
            exec(textwrap.dedent("\n".join(lines[g0:g1 + 1])), ns)
            exec(textwrap.dedent("\n".join(lines[u0:u1])), ns)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            tdist.b1.abort(); tdist.b2.abort()
            raise

    ths = [threading.Thread(target=update, args=(r,)) for r in range(world)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    finals = [_flat(ranks[r]["agent"].parameters()) for r in range(world)]
    assert torch.equal(finals[0], finals[1]), "the replicas diverged"
    for r in range(world):
        assert ranks[r]["optimizer"].n_steps == 16
    assert torch.equal(ranks[0]["optimizer"].grads[0], ranks[1]["optimizer"].grads[0])
    d = dict(world_size=np.int64(world), init_seed=np.int64(21), lr=np.float64(2.5e-4), torch_threads=np.int64(threads),
             lines=np.array([g0 + 1, g1 + 1, u0 + 1, u1], np.int64))
    for r in range(world):
        st = ranks[r]
        obs, actions, logprobs, dones, values = st["rollout"]
        d.update({f"frame_seed_rank{r}": np.int64(300 + 10 * r), f"frames_checksum_rank{r}": np.int64(st["frames"].sum(dtype=torch.int64).item()),
                  f"shuffle_seed_rank{r}": np.int64(seed + r), f"sample_seed_rank{r}": np.int64(23 + r),
                  f"step_done_rank{r}": st["step_done"], f"rewards_rank{r}": st["rewards"], f"actions_rank{r}": actions.to(torch.uint8),
                  f"logprobs_rank{r}": logprobs, f"values_rank{r}": values, f"advantages_rank{r}": st["ns"]["advantages"],
                  f"returns_rank{r}": st["ns"]["returns"], f"scalars_rank{r}": np.array(st["scalars"], np.float32)})
    st = ranks[0]
    _record_common(d, st["init"], finals[0], st["optimizer"], st["agent"], keep, st["scalars"])
    if save:
        _save(name, {f"atari_T{T}_N{N}_world{world}": d})
    return _np(d)


# ------------------------------------------------------------------------------------------------------------- config E
def mint_config_e(save=True, T=2048, N=64, name="continuous_iteration_cfgE", threads=1):
    """BASELINE configs[4]: ppo_continuous_action.py at its defaults -- 64 envs (the BASELINE's num_envs) x 2048 steps, 32
    minibatches x 10 epochs = 320 updates of 4,096 rows, obs 17 / act 6 (HalfCheetah-v4's shapes), clip 0.2, ent 0, lr 3e-4."""
    torch.set_num_threads(threads)
    script = "ppo_continuous_action.py"
    lines = R._read(script)
    OBS, ACT, seed = 17, 6, 401
    torch.manual_seed(31)
    Agent, _ = R.load_agent_class(script)
    envs = R.fake_envs((OBS,), action_shape=(ACT,))
    agent = Agent(envs)
    args = R.make_args(num_steps=T, num_envs=N, num_minibatches=32, update_epochs=10, batch_size=T * N, minibatch_size=T * N // 32,
                       clip_coef=0.2, ent_coef=0.0, learning_rate=3e-4)
    ns, scalars, logstd = {}, [], []
    keep = (1, 160, 320)

    def on_step(k):          # right before optimizer.step() of update k
        scalars.append([float(ns[key]) for key in SCALAR_KEYS] + [float(ns["clipfracs"][-1])])
        logstd.append(agent.actor_logstd.detach().clone().reshape(-1))

    optimizer = _GradSpy(R.make_optimizer(agent, 3e-4), agent.parameters(), keep_steps=keep, on_step=on_step)
    names = [n for n, _ in agent.named_parameters()]
    init = _flat(agent.parameters()).clone()
    obs_seq, step_done, rewards = (torch.from_numpy(x) for x in synthetic.continuous_inputs(T, N, OBS, seed))
    obs = torch.zeros((T, N, OBS))
    actions = torch.zeros((T, N, ACT))
    logprobs, dones, values = (torch.zeros((T, N)) for _ in range(3))
    torch.manual_seed(33)
    for step in range(T):
        obs[step], dones[step] = obs_seq[step], step_done[step]
        with torch.no_grad():
            action, logprob, _, value = agent.get_action_and_value(obs[step])
            values[step] = value.flatten()
        actions[step], logprobs[step] = action, logprob
    ns.update(args=args, agent=agent, optimizer=optimizer, envs=envs, obs=obs, actions=actions, logprobs=logprobs, rewards=rewards,
              dones=dones, values=values, next_obs=obs_seq[T], next_done=step_done[T], device=torch.device("cpu"), np=np,
              torch=torch, nn=nn)
    g0, g1, u0, u1 = _update_ranges(lines)
    exec(textwrap.dedent("\n".join(lines[g0:g1 + 1])), ns)
    np.random.seed(7)
    exec(textwrap.dedent("\n".join(lines[u0:u1])), ns)
    assert optimizer.n_steps == 320
    final = _flat(agent.parameters())
    d = dict(input_seed=np.int64(seed), obs_checksum=np.float64(obs_seq.double().sum().item()), actions=actions, logprobs=logprobs,
             values=values, advantages=ns["advantages"], returns=ns["returns"], init_params=init, final_params=final,
             param_names=np.array(names), init_seed=np.int64(31), sample_seed=np.int64(33), shuffle_seed=np.int64(7),
             lr=np.float64(3e-4), scalars=np.array(scalars, np.float32), scalar_names=np.array(list(SCALAR_KEYS) + ["clipfrac"]),
             grad_updates=np.array(keep, np.int64), torch_threads=np.int64(threads), logstd_before_step=torch.stack(logstd),
             lines=np.array([g0 + 1, g1 + 1, u0 + 1, u1], np.int64))
    for k, gk in zip(keep, optimizer.grads):
        d[f"mb{k}_grad"] = gk.clone()                 # 11,085 elements: whole
    if save:
        _save(name, {f"mujoco_T{T}_N{N}": d})
    return _np(d)


# ------------------------------------------------------------------------------------------------------------- config A
def mint_config_a(save=True, T=128, N=4, name="ppo_iteration_cfgA", threads=1):
    """BASELINE configs[0]: cleanrl/ppo.py at its defaults -- CartPole-v1's shapes (obs 4, 2 actions), 4 envs x 128 steps,
    4 minibatches x 4 epochs = 16 updates of 128 rows."""
    torch.set_num_threads(threads)
    script = "ppo.py"
    lines = R._read(script)
    OBS, A, seed = 4, 2, 501
    torch.manual_seed(41)
    Agent, _ = R.load_agent_class(script)
    envs = R.fake_envs((OBS,), n_actions=A)
    agent = Agent(envs)
    args = R.make_args(num_steps=T, num_envs=N, num_minibatches=4, update_epochs=4, batch_size=T * N, minibatch_size=T * N // 4,
                       clip_coef=0.2, ent_coef=0.01, learning_rate=2.5e-4)
    ns, scalars = {}, []
    keep = (1, 8, 16)

    def on_step(k):
        scalars.append([float(ns[key]) for key in SCALAR_KEYS] + [float(ns["clipfracs"][-1])])

    optimizer = _GradSpy(R.make_optimizer(agent, 2.5e-4), agent.parameters(), keep_steps=keep, on_step=on_step)
    names = [n for n, _ in agent.named_parameters()]
    init = _flat(agent.parameters()).clone()
    obs_seq, step_done, rewards = (torch.from_numpy(x) for x in synthetic.continuous_inputs(T, N, OBS, seed, done_p=1.0 / 30.0,
                                                                                           unit_rewards=True))
    obs = torch.zeros((T, N, OBS))
    actions, logprobs, dones, values = (torch.zeros((T, N)) for _ in range(4))
    torch.manual_seed(43)
    for step in range(T):
        obs[step], dones[step] = obs_seq[step], step_done[step]
        with torch.no_grad():
            action, logprob, _, value = agent.get_action_and_value(obs[step])
            values[step] = value.flatten()
        actions[step], logprobs[step] = action, logprob
    ns.update(args=args, agent=agent, optimizer=optimizer, envs=envs, obs=obs, actions=actions, logprobs=logprobs, rewards=rewards,
              dones=dones, values=values, next_obs=obs_seq[T], next_done=step_done[T], device=torch.device("cpu"), np=np,
              torch=torch, nn=nn)
    g0, g1, u0, u1 = _update_ranges(lines)
    exec(textwrap.dedent("\n".join(lines[g0:g1 + 1])), ns)
    np.random.seed(8)
    exec(textwrap.dedent("\n".join(lines[u0:u1])), ns)
    assert optimizer.n_steps == 16
    final = _flat(agent.parameters())
    d = dict(input_seed=np.int64(seed), obs_checksum=np.float64(obs_seq.double().sum().item()), actions=actions, logprobs=logprobs,
             values=values, advantages=ns["advantages"], returns=ns["returns"], init_params=init, final_params=final,
             param_names=np.array(names), init_seed=np.int64(41), sample_seed=np.int64(43), shuffle_seed=np.int64(8),
             lr=np.float64(2.5e-4), scalars=np.array(scalars, np.float32), scalar_names=np.array(list(SCALAR_KEYS) + ["clipfrac"]),
             grad_updates=np.array(keep, np.int64), torch_threads=np.int64(threads),
             lines=np.array([g0 + 1, g1 + 1, u0 + 1, u1], np.int64))
    for k, gk in zip(keep, optimizer.grads):
        d[f"mb{k}_grad"] = gk.clone()
    if save:
        _save(name, {f"cartpole_T{T}_N{N}": d})
    return _np(d)


def main():
    assert R.available(), "needs /root/reference (build container only)"
    torch.use_deterministic_algorithms(True)
    which = [a.upper() for a in sys.argv[1:]] or ["A", "E", "D", "C"]
    for w in which:
        t = time.time()
        {"A": mint_config_a, "E": mint_config_e, "D": mint_config_d, "C": mint_config_c}[w]()
        print(f"config {w}: {time.time() - t:.0f} s", flush=True)


if __name__ == "__main__":
    main()
