"""The reference scripts' ``__main__`` skeleton, shared by the five drop-in scripts.

Keeps, in the reference's order (ppo_atari_multigpu.py:162-403): runtime-derived sizes, optional
``init_process_group``, run name + writer + hyper-parameter table (rank 0), the seeding protocol
(``seed += rank``; ``random``/``numpy`` per-rank; torch seeded identically on all ranks for model init,
then per-rank after the agent exists), device selection (``--device-ids`` / ``cuda:{rank}`` / ``cuda`` /
``cpu``), env construction, the iteration loop with LR annealing, episodic-return prints
(``global_step=..., episodic_return=...``) and the ``SPS: <int>`` line per iteration.
"""
from __future__ import annotations

import os
import random
import time
import warnings

import numpy as np
import torch
import torch.distributed as dist

from .learner import PPOLearner
from .logger import make_writer


def setup_distributed(args):
    """ppo_atari_multigpu.py:166-183.  Returns (local_rank, world_size)."""
    local_rank = int(os.getenv("LOCAL_RANK", "0"))
    world_size = int(os.getenv("WORLD_SIZE", "1"))
    args.world_size = world_size
    args.local_batch_size = int(args.local_num_envs * args.num_steps)
    args.local_minibatch_size = int(args.local_batch_size // args.num_minibatches)
    args.num_envs = args.local_num_envs * world_size
    args.batch_size = int(args.num_envs * args.num_steps)
    args.minibatch_size = int(args.batch_size // args.num_minibatches)
    args.num_iterations = args.total_timesteps // args.batch_size
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.backend == "nccl" and torch.cuda.is_available() and args.cuda:
            torch.cuda.set_device(local_rank if not args.device_ids else args.device_ids[local_rank])
        dist.init_process_group(args.backend, rank=local_rank, world_size=world_size)
    else:
        warnings.warn(
            "\nNot using distributed mode!\nIf you want to use distributed mode, please execute this script with "
            "'torchrun'.\nE.g., `torchrun --standalone --nnodes=1 --nproc_per_node=2 ppo_atari_multigpu.py`\n")
    return local_rank, world_size


def select_device(args, local_rank: int = 0, world_size: int = 1, multigpu: bool = False) -> torch.device:
    use_cuda = torch.cuda.is_available() and args.cuda
    if not multigpu:
        return torch.device("cuda" if use_cuda else "cpu")                      # ppo.py:159
    device_ids = getattr(args, "device_ids", [])
    if len(device_ids) > 0:                                                      # ppo_atari_multigpu.py:214-222
        assert len(device_ids) == world_size, "you must specify the same number of device ids as `--nproc_per_node`"
        return torch.device(f"cuda:{device_ids[local_rank]}" if use_cuda else "cpu")
    if torch.cuda.device_count() < world_size:
        return torch.device("cuda" if use_cuda else "cpu")
    return torch.device(f"cuda:{local_rank}" if use_cuda else "cpu")


def seed_everything(args, local_rank: int = 0, multigpu: bool = False) -> None:
    if multigpu:                                   # ppo_atari_multigpu.py:206-212
        args.seed += local_rank
        random.seed(args.seed)
        np.random.seed(args.seed)
        torch.manual_seed(args.seed - local_rank)  # same init weights on every rank
    else:                                          # ppo.py:153-157
        random.seed(args.seed)
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    torch.backends.cudnn.deterministic = args.torch_deterministic


def open_writer(args, run_name: str, enabled: bool = True):
    if not enabled:
        return None
    if args.track:
        import wandb

        wandb.init(project=args.wandb_project_name, entity=args.wandb_entity, sync_tensorboard=True, config=vars(args),
                   name=run_name, monitor_gym=True, save_code=True)
    writer = make_writer(run_name)
    writer.add_text("hyperparameters",
                    "|param|value|\n|-|-|\n%s" % ("\n".join([f"|{key}|{value}|" for key, value in vars(args).items()])))
    return writer


def _env_step(envs, act_np, env_api: str, writer, global_step: int, avg_returns: list):
    """One ``envs.step`` in the reference's API of choice -> (next_obs, reward, next_done); prints / logs episodic returns
    exactly where the reference does."""
    if env_api == "gym":
        next_obs, reward, next_done, info = envs.step(act_np)
        if writer is not None:                          # ppo_atari_envpool.py:241-247
            for idx, d in enumerate(next_done):
                if d and info["lives"][idx] == 0:
                    print(f"global_step={global_step}, episodic_return={info['r'][idx]}")
                    avg_returns[:] = (avg_returns + [info["r"][idx]])[-20:]
                    writer.add_scalar("charts/avg_episodic_return", np.average(avg_returns), global_step)
                    writer.add_scalar("charts/episodic_return", info["r"][idx], global_step)
                    writer.add_scalar("charts/episodic_length", info["l"][idx], global_step)
    elif env_api == "pettingzoo":                       # ppo_pettingzoo_ma_atari.py:203-213: envs alternate players
        next_obs, reward, next_done, info = envs.step(act_np)
        if writer is not None:
            for idx, item in enumerate(info):
                player_idx = idx % 2
                if "episode" in item.keys():
                    print(f"global_step={global_step}, {player_idx}-episodic_return={item['episode']['r']}")
                    writer.add_scalar(f"charts/episodic_return-player{player_idx}", item["episode"]["r"], global_step)
                    writer.add_scalar(f"charts/episodic_length-player{player_idx}", item["episode"]["l"], global_step)
    elif env_api == "procgen":                          # ppo_procgen.py:241-250: 4-tuple, one info dict per env
        next_obs, reward, next_done, info = envs.step(act_np)
        if writer is not None:
            for item in info:
                if "episode" in item.keys():
                    print(f"global_step={global_step}, episodic_return={item['episode']['r']}")
                    writer.add_scalar("charts/episodic_return", item["episode"]["r"], global_step)
                    writer.add_scalar("charts/episodic_length", item["episode"]["l"], global_step)
                    break
    else:
        next_obs, reward, terminations, truncations, infos = envs.step(act_np)
        next_done = np.logical_or(terminations, truncations)
        if writer is not None and "final_info" in infos:            # :277-282
            for info in infos["final_info"]:
                if info and "episode" in info:
                    print(f"global_step={global_step}, episodic_return={info['episode']['r']}")
                    writer.add_scalar("charts/episodic_return", info["episode"]["r"], global_step)
                    writer.add_scalar("charts/episodic_length", info["episode"]["l"], global_step)
    return next_obs, reward, next_done


def train(args, envs, agent, device, writer, local_rank: int = 0, world_size: int = 1, env_api: str = "gymnasium",
          local_num_envs=None, verbose_rank_line: bool = False, learner_cls=PPOLearner):
    """The hot loop of every PPO script (ppo.py:178-309).  Returns the learner (for tests / evaluation).

    ``envs`` may be a LIST of vector envs ("env groups", the scripts' ``--env-groups K``): the rollout then runs as K
    overlapped lanes (cleanrl_amd/pipeline.py) -- same per-env trajectories as the serial loop, host stepping of one group
    hidden behind the GPU work and the PCIe copies of the others."""
    groups = list(envs) if isinstance(envs, (list, tuple)) else None
    if groups is not None and len(groups) == 1:
        envs, groups = groups[0], None
    space_env = groups[0] if groups is not None else envs
    local_num_envs = local_num_envs or args.num_envs
    learner = learner_cls(agent, args, space_env.single_observation_space, space_env.single_action_space, local_num_envs, device,
                          world_size=world_size, sample_seed=args.seed)
    global_step = 0
    start_time = time.time()
    avg_returns = []
    grouped = None
    if groups is not None:
        from .pipeline import GroupedRollout

        grouped = GroupedRollout(learner, len(groups), frame_delta=bool(getattr(args, "frame_delta", True)))
        for g, ge in enumerate(groups):
            # gymnasium's SyncVectorEnv seeds env i of a group with seed + i: offset every group by its first env's index, so the
            # union over the groups is seed .. seed + N - 1, the single-vector-env case (no two envs share a seed)
            o = (ge.reset() if env_api in ("gym", "procgen", "pettingzoo")
                 else ge.reset(seed=args.seed + g * (local_num_envs // len(groups)))[0])
            grouped.first_observation(g, o)
    else:
        if env_api in ("gym", "procgen", "pettingzoo"):             # envpool / procgen / supersuit: reset() returns obs only (:214)
            next_obs = envs.reset()
        else:
            next_obs, _ = envs.reset(seed=args.seed)
        learner.observe(0, next_obs, np.zeros(local_num_envs, np.float32))
    # The update as captured hipGraphs -- one per (epoch, minibatch) slot: gather, forward, fused loss, backward, clip + Adam -- when
    # nothing in it needs the host: the plain PPO learner on the fused kernels, no KL early stop, no noise drawn inside the update (RPO);
    # with world > 1 a slot is the graphs between its collectives (learner._SlotGraphs).  With host envs every env step synchronises on the actions, so the update starts with an empty GPU queue
    # and the host only microseconds ahead: ~55 launches per minibatch become one replay (MI355PPO_UPDATE_GRAPHS=0: eager).
    # Policy (MI355PPO_UPDATE_GRAPHS = auto | 0 | 1) and the all-ranks agreement live in ONE place: learner.update_graph_policy /
    # PPOLearner.capture_update_agreed -- over RCCL the graphs are used only behind a captured-vs-eager self-check, and ranks never end up on different routes.
    # (MI355PPO_ALLREDUCE=peer: the gradient exchange is five launches over HIP IPC segments inside the slot's one graph -- dp_comm.py.)
    if (learner.hip and type(learner) is PPOLearner and args.target_kl is None
            and getattr(agent, "rpo_alpha", None) is None and learner.batch_size % max(learner.minibatch_size, 1) == 0
            and (learner.fused_cnn or learner.mlp is not None)):
        learner.capture_update_agreed(log=lambda m: print(m, flush=True))
    metrics = {}
    action = None
    for iteration in range(1, args.num_iterations + 1):
        lrnow = args.learning_rate
        if args.anneal_lr:                                      # :251-254
            frac = 1.0 - (iteration - 1.0) / args.num_iterations
            lrnow = frac * args.learning_rate
        if grouped is not None:
            base = global_step

            def step_fn(g, act_np, step, _base=base):
                return _env_step(groups[g], act_np, env_api, writer, _base + (step + 1) * args.num_envs, avg_returns)

            grouped.run(step_fn)
            global_step += args.num_envs * args.num_steps
        else:
            for step in range(0, args.num_steps):
                global_step += args.num_envs
                action = learner.act(step)
                act_np = action.cpu().numpy()                       # :269  D2H + sync, as the reference
                next_obs, reward, next_done = _env_step(envs, act_np, env_api, writer, global_step, avg_returns)
                learner.store_reward(step, reward)
                learner.observe(step + 1, next_obs, next_done)
        if verbose_rank_line:                                   # ppo_atari_multigpu.py:284-286
            asum = action.sum() if action is not None else learner.actions[-1].sum()
            print(f"local_rank: {local_rank}, action.sum(): {asum}, iteration: {iteration}, "
                  f"agent.actor.weight.sum(): {agent.actor.weight.sum()}")
        learner.finish_rollout()
        metrics = learner.update(lrnow)
        learner.start_iteration()
        if writer is not None:                                  # :386-397
            writer.add_scalar("charts/learning_rate", lrnow, global_step)
            writer.add_scalar("losses/value_loss", metrics["value_loss"], global_step)
            writer.add_scalar("losses/policy_loss", metrics["policy_loss"], global_step)
            writer.add_scalar("losses/entropy", metrics["entropy"], global_step)
            writer.add_scalar("losses/old_approx_kl", metrics["old_approx_kl"], global_step)
            writer.add_scalar("losses/approx_kl", metrics["approx_kl"], global_step)
            writer.add_scalar("losses/clipfrac", metrics["clipfrac"], global_step)
            writer.add_scalar("losses/explained_variance", metrics["explained_variance"], global_step)
            print("SPS:", int(global_step / (time.time() - start_time)))
            writer.add_scalar("charts/SPS", int(global_step / (time.time() - start_time)), global_step)
    learner.last_metrics = metrics
    if getattr(learner, "_peer", None) is not None:          # every rank is done with every segment before anybody unmaps / frees
        torch.cuda.synchronize(learner.device)
        torch.distributed.barrier()
        learner._peer.close()
    return learner
