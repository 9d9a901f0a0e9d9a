"""Drop-in for ``cleanrl/ppo_pettingzoo_ma_atari.py`` (two-player PettingZoo Atari through supersuit, self-play PPO with
one shared NatureCNN policy on (84, 84, 6) observations).

    python cleanrl_amd/ppo_pettingzoo_ma_atari.py --env-id pong_v3 --num-envs 16 --num-steps 128 --seed 1

The PPO hot path of the other scripts; the observations arrive pixel-interleaved (the rollout buffer's own layout) and only
their four frame channels are divided by 255, the two agent-indicator planes pass through.  The reference script parses its
flags with argparse + strtobool (``--cuda False``); both that form and ``--no-cuda`` are accepted.  Without
pettingzoo / supersuit the synthetic (N,84,84,6) uint8 two-player stand-in environment is used.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, envs as E, runner  # noqa: E402
from cleanrl_amd.agents import MAAtariAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__).rstrip(".py")      # (:18 -- rstrip, as the reference)
    env_id: str = "pong_v3"
    total_timesteps: int = 20000000
    num_envs: int = 16
    clip_coef: float = 0.1


def have_pettingzoo() -> bool:
    try:
        import pettingzoo  # noqa: F401
        import supersuit  # noqa: F401
    except Exception:
        return False
    return True


def make_envs(args, run_name):
    """ppo_pettingzoo_ma_atari.py:151-167."""
    if have_pettingzoo() and not args.synthetic_env:
        import importlib

        import gym
        import supersuit as ss

        env = importlib.import_module(f"pettingzoo.atari.{args.env_id}").parallel_env()
        env = ss.max_observation_v0(env, 2)
        env = ss.frame_skip_v0(env, 4)
        env = ss.clip_reward_v0(env, lower_bound=-1, upper_bound=1)
        env = ss.color_reduction_v0(env, mode="B")
        env = ss.resize_v1(env, x_size=84, y_size=84)
        env = ss.frame_stack_v1(env, 4)
        env = ss.agent_indicator_v0(env, type_only=False)
        env = ss.pettingzoo_env_to_vec_env_v1(env)
        envs = ss.concat_vec_envs_v1(env, args.num_envs // 2, num_cpus=0, base_class="gym")
        envs.single_observation_space = envs.observation_space
        envs.single_action_space = envs.action_space
        envs.is_vector_env = True
        envs = gym.wrappers.RecordEpisodeStatistics(envs)
        if args.capture_video:
            envs = gym.wrappers.RecordVideo(envs, f"videos/{run_name}")
        return envs
    print("[cleanrl_amd] pettingzoo/supersuit not installed: using the synthetic (N,84,84,6) uint8 two-player stand-in",
          file=sys.stderr)
    return E.SyntheticMAAtariVecEnv(args.num_envs, seed=args.seed)


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
    assert hasattr(envs.single_action_space, "n"), "only discrete action space is supported"
    agent = Agent(envs).to(device)
    learner = runner.train(args, envs, agent, device, writer, env_api="pettingzoo")
    envs.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
