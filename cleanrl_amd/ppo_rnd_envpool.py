"""Drop-in for ``cleanrl/ppo_rnd_envpool.py`` (Atari via EnvPool, PPO + Random Network Distillation).

    python cleanrl_amd/ppo_rnd_envpool.py --env-id MontezumaRevenge-v5 --num-envs 128 --num-steps 128 --seed 1

The PPO hot path of the other scripts with RND's additions (``cleanrl_amd/learner_rnd.py``): a second, non-episodic
value stream -- the GAE kernel runs twice, once per stream -- an intrinsic reward from a frozen random target network and
a trained predictor on the newest frame, and a minibatch loss on the combined advantage.  Without envpool the synthetic
(N,4,84,84) uint8 stand-in environment (gym API) is used.
"""
from __future__ import annotations

import os
import sys
import time
from collections import deque
from dataclasses import dataclass

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, envs as E, runner  # noqa: E402
from cleanrl_amd.agents import RNDAgent as Agent, RNDModel  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402
from cleanrl_amd.learner_rnd import RNDPPOLearner  # noqa: E402
from cleanrl_amd.ppo_atari_envpool import RecordEpisodeStatistics  # noqa: E402


@dataclass
class Args(PPOArgs):
    exp_name: str = os.path.basename(__file__)[: -len(".py")]
    env_id: str = "MontezumaRevenge-v5"
    total_timesteps: int = 2000000000
    learning_rate: float = 1e-4
    num_envs: int = 128
    gamma: float = 0.999
    clip_coef: float = 0.1
    ent_coef: float = 0.001

    # RND arguments
    update_proportion: float = 0.25
    """proportion of exp used for predictor update"""
    int_coef: float = 1.0
    """coefficient of the intrinsic advantage"""
    ext_coef: float = 2.0
    """coefficient of the extrinsic advantage"""
    int_gamma: float = 0.99
    """intrinsic reward discount rate"""
    num_iterations_obs_norm_init: int = 50
    """number of rollout lengths of random actions used to initialise the observation normalisation"""


def make_envs(args):
    """ppo_rnd_envpool.py:278-291."""
    if E.have_envpool() and not args.synthetic_env:
        import envpool

        envs = envpool.make(args.env_id, env_type="gym", num_envs=args.num_envs, episodic_life=True, reward_clip=True,
                            seed=args.seed, repeat_action_probability=0.25)
        envs.num_envs = args.num_envs
        envs.single_action_space = envs.action_space
        envs.single_observation_space = envs.observation_space
        return RecordEpisodeStatistics(envs)
    print("[cleanrl_amd] envpool not installed: using the synthetic (N,4,84,84) uint8 Atari stand-in (gym API)",
          file=sys.stderr)
    return E.SyntheticAtariVecEnv(args.num_envs, seed=args.seed, n_actions=18, api="gym")


def init_obs_normalisation(args, envs, learner) -> None:
    """:325-336: ``num_steps * num_iterations_obs_norm_init`` random-action steps; the newest frame of every observation
    feeds ``obs_rms`` in batches of one rollout length."""
    print("Start to initialize observation normalization parameter.....")
    next_ob = []
    for step in range(args.num_steps * args.num_iterations_obs_norm_init):
        acs = np.random.randint(0, envs.single_action_space.n, size=(args.num_envs,))
        s, r, d, _ = envs.step(acs)
        next_ob.append(np.asarray(s)[:, 3, :, :].reshape([-1, 1, 84, 84]).astype(np.float64))
        if len(next_ob) == args.num_steps:
            learner.obs_rms.update(np.concatenate(next_ob))
            next_ob = []
    print("End to initialize...")


def main(argv=None):
    args = cli.parse(Args, argv)
    args.batch_size = int(args.num_envs * args.num_steps)
    args.minibatch_size = int(args.batch_size // args.num_minibatches)
    args.num_iterations = args.total_timesteps // args.batch_size
    run_name = f"{args.env_id}__{args.exp_name}__{args.seed}__{int(time.time())}"
    writer = runner.open_writer(args, run_name)
    runner.seed_everything(args)
    device = runner.select_device(args)
    envs = make_envs(args)
    assert hasattr(envs.single_action_space, "n"), "only discrete action space is supported"
    agent = Agent(envs).to(device)
    rnd_model = RNDModel(4, envs.single_action_space.n).to(device)
    learner = RNDPPOLearner(agent, rnd_model, args, envs.single_observation_space, envs.single_action_space, args.num_envs,
                            device, sample_seed=args.seed)
    avg_returns = deque(maxlen=20)
    global_step = 0
    start_time = time.time()
    next_obs = envs.reset()
    init_obs_normalisation(args, envs, learner)
    learner.observe(0, next_obs, np.zeros(args.num_envs, np.float32))
    metrics = {}
    for update in range(1, args.num_iterations + 1):
        lrnow = args.learning_rate
        if args.anneal_lr:                                       # :340-343
            frac = 1.0 - (update - 1.0) / args.num_iterations
            lrnow = frac * args.learning_rate
        for step in range(0, args.num_steps):
            global_step += 1 * args.num_envs
            action = learner.act(step)
            next_obs, reward, done, info = envs.step(action.cpu().numpy())
            learner.store_reward(step, reward)
            learner.observe(step + 1, next_obs, done)
            curiosity = learner.curiosity(step)
            for idx, d in enumerate(done):                       # :372-388
                if d and info["lives"][idx] == 0:
                    avg_returns.append(info["r"][idx])
                    epi_ret = np.average(avg_returns)
                    print(f"global_step={global_step}, episodic_return={info['r'][idx]}, "
                          f"curiosity_reward={np.mean(curiosity.cpu().numpy())}")
                    writer.add_scalar("charts/avg_episodic_return", epi_ret, global_step)
                    writer.add_scalar("charts/episodic_return", info["r"][idx], global_step)
                    writer.add_scalar("charts/episode_curiosity_reward", curiosity[idx].item(), global_step)
                    writer.add_scalar("charts/episodic_length", info["l"][idx], global_step)
        learner.finish_rollout()
        metrics = learner.update(lrnow)
        learner.start_iteration()
        writer.add_scalar("charts/learning_rate", lrnow, global_step)           # :527-536
        writer.add_scalar("losses/value_loss", metrics["value_loss"], global_step)
        writer.add_scalar("losses/policy_loss", metrics["policy_loss"], global_step)
        writer.add_scalar("losses/entropy", metrics["entropy"], global_step)
        writer.add_scalar("losses/old_approx_kl", metrics["old_approx_kl"], global_step)
        writer.add_scalar("losses/fwd_loss", metrics["fwd_loss"], global_step)
        writer.add_scalar("losses/approx_kl", metrics["approx_kl"], global_step)
        print("SPS:", int(global_step / (time.time() - start_time)))
        writer.add_scalar("charts/SPS", int(global_step / (time.time() - start_time)), global_step)
    learner.last_metrics = metrics
    envs.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
