"""Drop-in for ``cleanrl/ppg_procgen.py`` (Phasic Policy Gradient on Procgen, IMPALA-CNN).

    python cleanrl_amd/ppg_procgen.py --env-id starpilot --num-envs 64 --num-steps 256 --seed 1

Policy phase = the PPO hot path of the other scripts (uint8 rollout rows, gather+convert, GAE, Categorical sampling, the
fused clipped-surrogate loss, fused clip + Adam) for ``n_iteration`` rollouts, every rollout also kept in the auxiliary
buffer (in HBM on a GPU); auxiliary phase = ``e_auxiliary`` epochs of whole-rollout minibatches distilling the value
function into the encoder under a KL leash on the policy (``cleanrl_amd/learner_ppg.py``).  Without the ``procgen``
package the synthetic (N,64,64,3) uint8 stand-in environment is used.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cleanrl_amd import cli, runner  # noqa: E402
from cleanrl_amd.agents import PPGAgent as Agent  # noqa: E402
from cleanrl_amd.args import PPOArgs  # noqa: E402
from cleanrl_amd.learner_ppg import PPGLearner  # noqa: E402
from cleanrl_amd.ppo_procgen import make_procgen_envs  # noqa: E402


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
    adv_norm_fullbatch: bool = True
    """normalise advantages over the full batch (PPG), not per minibatch"""

    # PPG specific arguments
    n_iteration: int = 32
    """N_pi: policy updates (rollouts) per policy phase"""
    e_policy: int = 1
    """E_pi: epochs per policy update"""
    v_value: int = 1
    """E_V: value epochs per policy update (only 1 is supported, as in the reference)"""
    e_auxiliary: int = 6
    """E_aux: epochs of the auxiliary phase"""
    beta_clone: float = 1.0
    """behaviour-cloning (KL) coefficient"""
    num_aux_rollouts: int = 4
    """rollouts (envs) per auxiliary minibatch"""
    n_aux_grad_accum: int = 1
    """auxiliary minibatches per optimiser step"""

    # to be filled in runtime
    num_phases: int = 0
    """number of phases (computed at run time)"""
    aux_batch_rollouts: int = 0
    """rollouts in the auxiliary buffer (computed at run time)"""


def main(argv=None):
    args = cli.parse(Args, argv)
    args.batch_size = int(args.num_envs * args.num_steps)
    args.minibatch_size = int(args.batch_size // args.num_minibatches)
    args.num_iterations = args.total_timesteps // args.batch_size
    args.num_phases = int(args.num_iterations // args.n_iteration)
    args.aux_batch_rollouts = int(args.num_envs * args.n_iteration)
    assert args.v_value == 1, "Multiple value epoch (v_value != 1) is not supported yet"
    run_name = f"{args.env_id}__{args.exp_name}__{args.seed}__{int(time.time())}"
    writer = runner.open_writer(args, run_name)
    runner.seed_everything(args)
    device = runner.select_device(args)
    envs = make_procgen_envs(args, run_name)
    assert hasattr(envs.single_action_space, "n"), "only discrete action space is supported"
    agent = Agent(envs).to(device)
    learner = PPGLearner(agent, args, envs.single_observation_space, envs.single_action_space, args.num_envs, device,
                         sample_seed=args.seed)
    global_step = 0
    start_time = time.time()
    next_obs = envs.reset()
    import numpy as np

    learner.observe(0, next_obs, np.zeros(args.num_envs, np.float32))
    metrics = {}
    for phase in range(1, args.num_phases + 1):
        for update in range(1, args.n_iteration + 1):                      # POLICY PHASE (:283-414)
            lrnow = args.learning_rate
            if args.anneal_lr:
                frac = 1.0 - (update - 1.0) / args.num_iterations
                lrnow = frac * args.learning_rate
            for step in range(0, args.num_steps):
                global_step += 1 * args.num_envs
                action = learner.act(step)
                next_obs, reward, done, info = envs.step(action.cpu().numpy())
                learner.store_reward(step, reward)
                learner.observe(step + 1, next_obs, done)
                for item in info:
                    if "episode" in item.keys():
                        print(f"global_step={global_step}, episodic_return={item['episode']['r']}")
                        writer.add_scalar("charts/episodic_return", item["episode"]["r"], global_step)
                        writer.add_scalar("charts/episodic_length", item["episode"]["l"], global_step)
                        break
            learner.finish_rollout()
            metrics = learner.update(lrnow)
            learner.start_iteration()
            writer.add_scalar("charts/learning_rate", lrnow, global_step)
            for key, tag in (("value_loss", "value_loss"), ("policy_loss", "policy_loss"), ("entropy", "entropy"),
                             ("old_approx_kl", "old_approx_kl"), ("approx_kl", "approx_kl"), ("clipfrac", "clipfrac"),
                             ("explained_variance", "explained_variance")):
                writer.add_scalar(f"losses/{tag}", metrics[key], global_step)
            print("SPS:", int(global_step / (time.time() - start_time)))
            writer.add_scalar("charts/SPS", int(global_step / (time.time() - start_time)), global_step)
        aux = learner.aux_phase()                                          # AUXILIARY PHASE (:416-474)
        writer.add_scalar("losses/aux/kl_loss", aux["kl_loss"], global_step)
        writer.add_scalar("losses/aux/aux_value_loss", aux["aux_value_loss"], global_step)
        writer.add_scalar("losses/aux/real_value_loss", aux["real_value_loss"], global_step)
    learner.last_metrics = metrics
    envs.close()
    writer.close()
    return learner


if __name__ == "__main__":
    main()
