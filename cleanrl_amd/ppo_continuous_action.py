"""Drop-in for ``cleanrl/ppo_continuous_action.py`` (MuJoCo-style continuous control, Normal policy).

    python cleanrl_amd/ppo_continuous_action.py --env-id HalfCheetah-v4 --num-envs 64 --seed 1

``Normal(mean, exp(logstd))`` sample + log_prob and the continuous-action loss (forward+backward, incl.
the shared ``actor_logstd`` gradient) are libmi355ppo kernels.  ``--save-model`` stores
``agent.state_dict()`` under ``runs/<run_name>/`` as the reference does (:326-329); the state dict has
the reference's parameter names.  Without gymnasium/mujoco a HalfCheetah-shaped stand-in task is used.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, envs as E, runner  # noqa: E402
from cleanrl_amd.agents import ContinuousAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__)[: -len(".py")]
    save_model: bool = False
    """save the final model into `runs/{run_name}`"""
    upload_model: bool = False
    """upload the saved model to the Hugging Face hub (needs network + huggingface_hub)"""
    hf_entity: str = ""
    """Hugging Face user or organisation for `--upload-model`"""
    env_id: str = "HalfCheetah-v4"
    total_timesteps: int = 1000000
    learning_rate: float = 3e-4
    num_envs: int = 1
    num_steps: int = 2048
    num_minibatches: int = 32
    update_epochs: int = 10
    clip_coef: float = 0.2
    ent_coef: float = 0.0


def make_envs(args, run_name):
    """ppo_continuous_action.py:85-103,178-181."""
    if E.have_gymnasium() and not args.synthetic_env:
        import gymnasium as gym

        def make_env(env_id, idx, capture_video, gamma):
            def thunk():
                if capture_video and idx == 0:
                    env = gym.make(env_id, render_mode="rgb_array")
                    env = gym.wrappers.RecordVideo(env, f"videos/{run_name}")
                else:
                    env = gym.make(env_id)
                env = gym.wrappers.FlattenObservation(env)
                env = gym.wrappers.RecordEpisodeStatistics(env)
                env = gym.wrappers.ClipAction(env)
                env = gym.wrappers.NormalizeObservation(env)
                env = gym.wrappers.TransformObservation(env, lambda obs: np.clip(obs, -10, 10))
                env = gym.wrappers.NormalizeReward(env, gamma=gamma)
                env = gym.wrappers.TransformReward(env, lambda reward: np.clip(reward, -10, 10))
                return env

            return thunk

        return gym.vector.SyncVectorEnv([make_env(args.env_id, i, args.capture_video, args.gamma)
                                         for i in range(args.num_envs)])
    print("[cleanrl_amd] gymnasium/mujoco not installed: using the HalfCheetah-shaped (obs 17, act 6) stand-in task",
          file=sys.stderr)
    return E.NormalizeVecEnv(E.SyntheticContinuousVecEnv(args.num_envs, seed=args.seed), args.gamma)


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
    agent = Agent(envs).to(device)
    learner = runner.train(args, envs, agent, device, writer)
    if args.save_model:
        model_path = f"runs/{run_name}/{args.exp_name}.cleanrl_model"
        torch.save(agent.state_dict(), model_path)
        print(f"model saved to {model_path}")
        if args.upload_model:
            raise SystemExit("--upload-model needs network access and cleanrl_utils.huggingface; not available here")
    envs.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
