"""Worker of ``test_gpu_multirank.py``: ONE rank of the data-parallel update step, launched twice by
``torch.distributed.run`` with both ranks on ``cuda:0`` (backend gloo accepts device tensors; RCCL refuses two ranks on one
device).  Each rank feeds ITS half of the golden ``update_step.npz::multigpu_cnn_world2`` through the learner's own
``_minibatch_hip`` -- conv/heads kernels, K3, ``dist.all_reduce(flat.grads)`` across the two processes, the fused
``/world_size`` -> clip -> Adam kernel -- and dumps what it ended with.  Not a test module itself."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden  # noqa: E402
from cleanrl_amd import envs as E, learner_smoke, ops  # noqa: E402
from cleanrl_amd.agents import AtariAgent  # noqa: E402
from cleanrl_amd.learner import PPOLearner  # noqa: E402


def main(out_dir):
    rank, world = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("update_step")["multigpu_cnn_world2"]
    env = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    torch.manual_seed(int(g["init_seed"]))                      # same init on every rank (ppo_atari_multigpu.py:211)
    agent = AtariAgent(env).to(dev)
    args = learner_smoke.default_args(num_steps=8, num_minibatches=2, clip_coef=0.1)
    L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, 8, dev, world_size=world)
    G = lambda k: torch.from_numpy(g[f"{k}_rank{rank}"]).to(dev)
    idx = torch.from_numpy(g["mb_inds"]).to(dev)
    sc = torch.zeros(7, device=dev)
    seen = {}
    real_step = L.optimizer_step_hip

    def spy(lr):
        seen["reduced"] = L.flat.grads.clone()                  # what the all-reduce left in the flat buffer (:367)
        real_step(lr)

    L.optimizer_step_hip = spy
    L._minibatch_hip(idx, ops.obs_nchw_to_nhwc_u8(G("obs_u8")), G("b_actions"), G("b_logprobs"), G("b_advantages"),
                     G("b_returns"), G("b_values"), float(g["lr"]), sc)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=L.flat.params.cpu().numpy(), reduced=seen["reduced"].cpu().numpy(),
             total_norm=L._total_norm.cpu().numpy(), loss=sc[0].item(), world=world,
             backend=np.bytes_(dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
