"""Drop-in for ``cleanrl/ppo_atari.py`` (Atari, NatureCNN, gymnasium SyncVectorEnv).

    python cleanrl_amd/ppo_atari.py --env-id BreakoutNoFrameskip-v4 --num-envs 1024 --num-steps 128 --seed 1

Observations are kept as uint8 in HBM and converted by the fused gather+convert kernel; sampling, GAE,
the clipped-surrogate loss (forward+backward) and clip+Adam are libmi355ppo kernels.  Without
gymnasium/ale_py the synthetic (N,4,84,84) uint8 stand-in environment is used (no emulator in this image).
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, envs as E, runner  # noqa: E402
from cleanrl_amd.agents import AtariAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__)[: -len(".py")]
    env_id: str = "BreakoutNoFrameskip-v4"
    total_timesteps: int = 10000000
    num_envs: int = 8
    clip_coef: float = 0.1


def make_atari_envs(args, run_name, num_envs, seed):
    """ppo_atari.py:80-100: NoopReset(30) / MaxAndSkip(4) / EpisodicLife / FireReset / ClipReward /
    Resize 84x84 / GrayScale / FrameStack(4) around gym.make, in a SyncVectorEnv."""
    if E.have_gymnasium() and not args.synthetic_env:
        import gymnasium as gym
        from cleanrl_amd.atari_wrappers import (ClipRewardEnv, EpisodicLifeEnv, FireResetEnv, MaxAndSkipEnv,
                                                NoopResetEnv)

        def make_env(env_id, idx, capture_video):
            def thunk():
                if capture_video and idx == 0:
                    env = gym.make(env_id, render_mode="rgb_array")
                    env = gym.wrappers.RecordVideo(env, f"videos/{run_name}")
                else:
                    env = gym.make(env_id)
                env = gym.wrappers.RecordEpisodeStatistics(env)
                env = NoopResetEnv(env, noop_max=30)
                env = MaxAndSkipEnv(env, skip=4)
                env = EpisodicLifeEnv(env)
                if "FIRE" in env.unwrapped.get_action_meanings():
                    env = FireResetEnv(env)
                env = ClipRewardEnv(env)
                env = gym.wrappers.ResizeObservation(env, (84, 84))
                env = gym.wrappers.GrayScaleObservation(env)
                env = gym.wrappers.FrameStack(env, 4)
                return env

            return thunk

        return gym.vector.SyncVectorEnv([make_env(args.env_id, i, args.capture_video) for i in range(num_envs)])
    print("[cleanrl_amd] gymnasium/ale_py not installed: using the synthetic (N,4,84,84) uint8 Atari stand-in",
          file=sys.stderr)
    return E.SyntheticAtariVecEnv(num_envs, seed=seed, n_actions=4)


def make_atari_env_groups(args, run_name, num_envs, seed):
    """``--env-groups K`` vector envs of num_envs / K envs each (a list of one = the reference's single vector env); group g
    is seeded ``seed + g * (num_envs / K)`` so that no two envs of the rank share a seed."""
    from cleanrl_amd.pipeline import split_env_groups

    k = max(int(getattr(args, "env_groups", 1)), 1)
    return split_env_groups(lambda g, n: make_atari_envs(args, run_name, n, seed + g * n), num_envs, k)


def main(argv=None):
    args = cli.parse(Args, argv)
    args.batch_size = int(args.num_envs * args.num_steps)
    args.minibatch_size = int(args.batch_size // args.num_minibatches)
    args.num_iterations = args.total_timesteps // args.batch_size
    run_name = f"{args.env_id}__{args.exp_name}__{args.seed}__{int(time.time())}"
    writer = runner.open_writer(args, run_name)
    runner.seed_everything(args)
    device = runner.select_device(args)
    envs = make_atari_env_groups(args, run_name, args.num_envs, args.seed)
    assert hasattr(envs[0].single_action_space, "n"), "only discrete action space is supported"
    agent = Agent(envs[0]).to(device)
    learner = runner.train(args, envs, agent, device, writer)
    for e in envs:
        e.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
