"""Drop-in for ``cleanrl/ppo_atari_multigpu.py``: data-parallel PPO, one learner process per GPU.

    torchrun --standalone --nnodes=1 --nproc_per_node=8 cleanrl_amd/ppo_atari_multigpu.py \\
        --env-id BreakoutNoFrameskip-v4 --local-num-envs 256 --num-steps 128 --backend nccl

Each rank owns ``--local-num-envs`` environments, its own rollout storage, GAE and shuffles (per-rank
numpy seed) and an identically initialised model replica.  The only exchange is one SUM all-reduce of
the flat f32 gradient per minibatch step (ppo_atari_multigpu.py:360-374) -- issued on the persistent
flat gradient buffer over RCCL/xGMI (``--backend nccl`` *is* RCCL on ROCm) -- followed by the fused
``/world_size`` -> clip -> Adam kernel.  ``--num-envs`` is accepted and overwritten, as in the reference.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass, field
from typing import List, Literal

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, runner  # noqa: E402
from cleanrl_amd.agents import AtariAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402
from cleanrl_amd.ppo_atari import make_atari_env_groups  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__)[: -len(".py")]
    env_id: str = "BreakoutNoFrameskip-v4"
    total_timesteps: int = 10000000
    clip_coef: float = 0.1
    local_num_envs: int = 8
    """number of parallel environments on this rank"""
    device_ids: List[int] = field(default_factory=lambda: [])
    """GPU index per rank (defaults to cuda:{rank})"""
    backend: Literal["gloo", "nccl", "mpi"] = "gloo"
    """torch.distributed backend ("nccl" = RCCL over xGMI on ROCm)"""

    # to be filled in runtime
    local_batch_size: int = 0
    """per-rank rollout batch size (computed at run time)"""
    local_minibatch_size: int = 0
    """per-rank minibatch size (computed at run time)"""
    num_envs: int = 0
    """global number of environments (computed at run time)"""
    world_size: int = 0
    """number of ranks (computed at run time)"""


def main(argv=None):
    args = cli.parse(Args, argv)
    local_rank, world_size = runner.setup_distributed(args)
    run_name = f"{args.env_id}__{args.exp_name}__{args.seed}__{int(time.time())}"
    writer = runner.open_writer(args, run_name, enabled=(local_rank == 0))
    if local_rank == 0:
        print(args)
    # CRUCIAL (reference :207): a different seed per data-parallel worker, the same torch seed for model init
    runner.seed_everything(args, local_rank, multigpu=True)
    device = runner.select_device(args, local_rank, world_size, multigpu=True)
    envs = make_atari_env_groups(args, run_name, args.local_num_envs, args.seed)
    assert hasattr(envs[0].single_action_space, "n"), "only discrete action space is supported"
    agent = Agent(envs[0]).to(device)
    torch.manual_seed(args.seed)                      # :231 per-rank action sampling from here on
    learner = runner.train(args, envs, agent, device, writer, local_rank=local_rank, world_size=world_size,
                           local_num_envs=args.local_num_envs, verbose_rank_line=True)
    for e in envs:
        e.close()
    if local_rank == 0:
        writer.close()
    if world_size > 1:
        torch.distributed.destroy_process_group()
    return learner


if __name__ == "__main__":
    main()
