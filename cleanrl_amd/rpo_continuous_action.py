"""Drop-in for ``cleanrl/rpo_continuous_action.py`` (Robust Policy Optimization: PPO for continuous control whose
re-evaluated action means are perturbed by ``U(-rpo_alpha, rpo_alpha)``, rpo_continuous_action.py:138-142).

    python cleanrl_amd/rpo_continuous_action.py --env-id HalfCheetah-v4 --num-envs 64 --rpo-alpha 0.5

Same hot path and the same libmi355ppo kernels as ``ppo_continuous_action.py`` (the GAE and
loss blocks appear verbatim in this script, :233-246, :265-300): rollout storage, fused GAE, Normal sample/log_prob and
the fused continuous-action loss; the perturbation is added to the network's mean before the loss kernel, and since it
carries no gradient the kernel's ``dmean`` is the gradient of the unperturbed mean.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, envs as E, runner  # noqa: E402
from cleanrl_amd.agents import ContinuousAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402
from cleanrl_amd.ppo_continuous_action import make_envs  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__)[: -len(".py")]
    env_id: str = "HalfCheetah-v4"
    total_timesteps: int = 8000000
    learning_rate: float = 3e-4
    num_envs: int = 1
    num_steps: int = 2048
    num_minibatches: int = 32
    update_epochs: int = 10
    clip_coef: float = 0.2
    ent_coef: float = 0.0
    rpo_alpha: float = 0.5
    """the alpha parameter for RPO"""


def main(argv=None):
    args = cli.parse(Args, argv)
    args.batch_size = int(args.num_envs * args.num_steps)
    args.minibatch_size = int(args.batch_size // args.num_minibatches)
    args.num_iterations = args.total_timesteps // args.batch_size
    run_name = f"{args.env_id}__{args.exp_name}__{args.seed}__{int(time.time())}"
    writer = runner.open_writer(args, run_name)
    runner.seed_everything(args)
    device = runner.select_device(args)
    envs = make_envs(args, run_name)
    assert isinstance(envs.single_action_space, E.Box) or not hasattr(envs.single_action_space, "n"), \
        "only continuous action space is supported"
    agent = Agent(envs, args.rpo_alpha).to(device)
    learner = runner.train(args, envs, agent, device, writer)
    envs.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
