"""Drop-in for ``cleanrl/ppo_atari_lstm.py`` (Atari, one 84x84 frame per step, NatureCNN -> LSTM(512,128)).

    python cleanrl_amd/ppo_atari_lstm.py --env-id BreakoutNoFrameskip-v4 --num-envs 8 --num-steps 128 --seed 1

Same hot path as the feed-forward scripts -- uint8 rollout rows in HBM, the gather+convert kernel, GAE, Categorical
sampling, the fused clipped-surrogate loss (forward+backward), fused clip+Adam -- with the reference's recurrent
structure on top: the LSTM state is carried through the rollout, reset where ``done`` is 1, and every minibatch is all
``num_steps`` steps of ``num_envs // num_minibatches`` environments, unrolled from the state the rollout started from
(``cleanrl_amd/learner_lstm.py``).  Without gymnasium/ale_py the synthetic (N,1,84,84) uint8 stand-in environment is used.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, envs as E, runner  # noqa: E402
from cleanrl_amd.agents import AtariLSTMAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402
from cleanrl_amd.learner_lstm import LSTMPPOLearner  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__)[: -len(".py")]
    env_id: str = "BreakoutNoFrameskip-v4"
    total_timesteps: int = 10000000
    num_envs: int = 8
    clip_coef: float = 0.1


def make_atari_envs(args, run_name, num_envs, seed):
    """ppo_atari_lstm.py:84-108: the ppo_atari.py wrapper stack with FrameStack(1) -- memory lives in the LSTM."""
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
                env = gym.wrappers.FrameStack(env, 1)
                return env

            return thunk

        return gym.vector.SyncVectorEnv([make_env(args.env_id, i, args.capture_video) for i in range(num_envs)])
    print("[cleanrl_amd] gymnasium/ale_py not installed: using the synthetic (N,1,84,84) uint8 Atari stand-in",
          file=sys.stderr)
    return E.SyntheticAtariVecEnv(num_envs, seed=seed, n_actions=4, frames=1)


def main(argv=None):
    args = cli.parse(Args, argv)
    args.batch_size = int(args.num_envs * args.num_steps)
    args.minibatch_size = int(args.batch_size // args.num_minibatches)
    args.num_iterations = args.total_timesteps // args.batch_size
    run_name = f"{args.env_id}__{args.exp_name}__{args.seed}__{int(time.time())}"
    writer = runner.open_writer(args, run_name)
    runner.seed_everything(args)
    device = runner.select_device(args)
    envs = make_atari_envs(args, run_name, args.num_envs, args.seed)
    assert hasattr(envs.single_action_space, "n"), "only discrete action space is supported"
    agent = Agent(envs).to(device)
    learner = runner.train(args, envs, agent, device, writer, learner_cls=LSTMPPOLearner)
    envs.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
