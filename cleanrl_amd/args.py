"""The flag surface shared by the PPO scripts (reference: cleanrl/ppo.py:19-70).

Field names, types and defaults ARE the drop-in contract: every field is a ``--flag`` (see ``cli.py``).
Per-script defaults (env id, number of envs, clip coefficient, ...) are overridden in each script's
``Args`` subclass exactly where the reference scripts differ.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class PPOArgs:
    exp_name: str = "ppo"
    """experiment name (used in the run name)"""
    seed: int = 1
    """experiment seed"""
    torch_deterministic: bool = True
    """sets `torch.backends.cudnn.deterministic` (on ROCm: MIOpen deterministic mode)"""
    cuda: bool = True
    """use the GPU (HIP kernels) when one is present; `--no-cuda` selects the CPU host path"""
    track: bool = False
    """track the run with Weights and Biases (needs the `wandb` package)"""
    wandb_project_name: str = "cleanRL"
    """W&B project"""
    wandb_entity: str = None
    """W&B entity (team)"""
    capture_video: bool = False
    """record videos of the agent into `videos/` (needs gymnasium's RecordVideo)"""

    # Algorithm specific arguments
    env_id: str = "CartPole-v1"
    """environment id"""
    total_timesteps: int = 500000
    """total environment steps of the run"""
    learning_rate: float = 2.5e-4
    """Adam learning rate"""
    num_envs: int = 4
    """number of parallel environments"""
    num_steps: int = 128
    """steps per environment per rollout"""
    anneal_lr: bool = True
    """linearly anneal the learning rate to 0 over the run"""
    gamma: float = 0.99
    """discount factor"""
    gae_lambda: float = 0.95
    """lambda of generalised advantage estimation"""
    num_minibatches: int = 4
    """minibatches per epoch"""
    update_epochs: int = 4
    """epochs over the rollout per policy update"""
    norm_adv: bool = True
    """normalise advantages per minibatch"""
    clip_coef: float = 0.2
    """surrogate clipping coefficient"""
    clip_vloss: bool = True
    """clip the value loss as in the PPO paper's implementation"""
    ent_coef: float = 0.01
    """entropy bonus coefficient"""
    vf_coef: float = 0.5
    """value-loss coefficient"""
    max_grad_norm: float = 0.5
    """global gradient-norm clip"""
    target_kl: float = None
    """stop the epoch loop early when approx_kl exceeds this"""

    # to be filled in runtime
    batch_size: int = 0
    """rollout batch size (computed at run time)"""
    minibatch_size: int = 0
    """minibatch size (computed at run time)"""
    num_iterations: int = 0
    """number of rollout/update iterations (computed at run time)"""

    # additions of this implementation (not in the reference)
    synthetic_env: bool = False
    """force the built-in stand-in environment even when gymnasium/envpool are importable"""
    env_groups: int = 1
    """split the (local) envs into this many independently stepped vector envs whose host stepping, PCIe copies and GPU
    work overlap (cleanrl_amd/pipeline.py); 1 = the reference's serial rollout loop"""
    frame_delta: bool = True
    """with --env-groups > 1 on FrameStack(4) image envs: send only the newest frame of every env that was not reset"""
