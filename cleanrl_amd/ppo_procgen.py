"""Drop-in for ``cleanrl/ppo_procgen.py`` (Procgen, IMPALA-CNN, 64 envs x 256 steps, 8 minibatches x 3 epochs).

    python cleanrl_amd/ppo_procgen.py --env-id starpilot --num-envs 64 --num-steps 256 --seed 1

The PPO hot path is the one of the other scripts -- uint8 rollout rows in HBM (procgen's frames arrive pixel-interleaved,
which is the rollout buffer's own layout: no relayout), the gather+convert kernel, GAE, Categorical sampling, the fused
clipped-surrogate loss (forward+backward), fused clip+Adam; the IMPALA-CNN's 3x3 convolutions, pooling and residual adds
stay on MIOpen.  Without the ``procgen`` package the synthetic (N,64,64,3) uint8 stand-in environment is used.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, envs as E, runner  # noqa: E402
from cleanrl_amd.agents import ProcgenAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__)[: -len(".py")]
    env_id: str = "starpilot"
    total_timesteps: int = int(25e6)
    learning_rate: float = 5e-4
    num_envs: int = 64
    num_steps: int = 256
    anneal_lr: bool = False
    gamma: float = 0.999
    num_minibatches: int = 8
    update_epochs: int = 3


def have_procgen() -> bool:
    try:
        import gym  # noqa: F401
        import procgen  # noqa: F401
    except Exception:
        return False
    return True


def make_procgen_envs(args, run_name):
    """ppo_procgen.py:189-198."""
    if have_procgen() and not args.synthetic_env:
        import gym
        import numpy as np
        from procgen import ProcgenEnv

        envs = ProcgenEnv(num_envs=args.num_envs, env_name=args.env_id, num_levels=0, start_level=0, distribution_mode="easy")
        envs = gym.wrappers.TransformObservation(envs, lambda obs: obs["rgb"])
        envs.single_action_space = envs.action_space
        envs.single_observation_space = envs.observation_space["rgb"]
        envs.is_vector_env = True
        envs = gym.wrappers.RecordEpisodeStatistics(envs)
        if args.capture_video:
            envs = gym.wrappers.RecordVideo(envs, f"videos/{run_name}")
        envs = gym.wrappers.NormalizeReward(envs, gamma=args.gamma)
        envs = gym.wrappers.TransformReward(envs, lambda reward: np.clip(reward, -10, 10))
        return envs
    print("[cleanrl_amd] procgen not installed: using the synthetic (N,64,64,3) uint8 Procgen stand-in", file=sys.stderr)
    return E.SyntheticProcgenVecEnv(args.num_envs, seed=args.seed, gamma=args.gamma)


def main(argv=None):
    args = cli.parse(Args, argv)
    args.batch_size = int(args.num_envs * args.num_steps)
    args.minibatch_size = int(args.batch_size // args.num_minibatches)
    args.num_iterations = args.total_timesteps // args.batch_size
    run_name = f"{args.env_id}__{args.exp_name}__{args.seed}__{int(time.time())}"
    writer = runner.open_writer(args, run_name)
    runner.seed_everything(args)
    device = runner.select_device(args)
    envs = make_procgen_envs(args, run_name)
    assert hasattr(envs.single_action_space, "n"), "only discrete action space is supported"
    agent = Agent(envs).to(device)
    learner = runner.train(args, envs, agent, device, writer, env_api="procgen")
    envs.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
