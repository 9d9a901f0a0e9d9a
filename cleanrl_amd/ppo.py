"""Drop-in for ``cleanrl/ppo.py`` (classic control, MLP actor-critic).

    python cleanrl_amd/ppo.py --env-id CartPole-v1 --num-envs 4 --num-steps 128 --seed 1 [--no-cuda]

Same flags, defaults, stdout lines and scalar tags as the reference; the rollout-storage -> GAE ->
minibatch-update hot loop runs in ``cleanrl_amd.learner`` (HIP kernels on a GPU, the reference's torch
semantics with ``--no-cuda``).  Without gymnasium the built-in numpy CartPole-v1 is used.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, envs as E, runner  # noqa: E402
from cleanrl_amd.agents import MlpAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__)[: -len(".py")]
    env_id: str = "CartPole-v1"
    total_timesteps: int = 500000
    num_envs: int = 4
    clip_coef: float = 0.2


def make_envs(args, run_name):
    """ppo.py:73-91,162-165: SyncVectorEnv of RecordEpisodeStatistics(gym.make(env_id))."""
    if E.have_gymnasium() and not args.synthetic_env:
        import gymnasium as gym

        def make_env(env_id, idx, capture_video):
            def thunk():
                if capture_video and idx == 0:
                    env = gym.make(env_id, render_mode="rgb_array")
                    env = gym.wrappers.RecordVideo(env, f"videos/{run_name}")
                else:
                    env = gym.make(env_id)
                return gym.wrappers.RecordEpisodeStatistics(env)

            return thunk

        return gym.vector.SyncVectorEnv([make_env(args.env_id, i, args.capture_video) for i in range(args.num_envs)])
    if args.env_id != "CartPole-v1":
        raise SystemExit(f"gymnasium is not installed and the built-in environments only cover CartPole-v1, not {args.env_id}")
    print("[cleanrl_amd] gymnasium not installed: using the built-in numpy CartPole-v1", file=sys.stderr)
    return E.CartPoleVecEnv(args.num_envs, seed=args.seed)


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
    assert isinstance(envs.single_action_space, E.Discrete) or hasattr(envs.single_action_space, "n"), \
        "only discrete action space is supported"
    agent = Agent(envs).to(device)
    learner = runner.train(args, envs, agent, device, writer)
    envs.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
