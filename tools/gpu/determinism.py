"""Run-to-run determinism probe: one PPO update (16 minibatches) of the Atari learner on seeded synthetic rollout data; prints a
checksum of the parameters after every minibatch.  Launched several times -- alone and two at a time on one GPU -- by
tools/gpu/r4_determinism.sh: equal lines = bit-identical trajectories."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from cleanrl_amd import envs as E, learner_smoke  # noqa: E402
from cleanrl_amd.agents import AtariAgent  # noqa: E402
from cleanrl_amd.learner import PPOLearner  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = 128
    dev = torch.device("cuda:0")
    env = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    torch.manual_seed(3)
    agent = AtariAgent(env).to(dev)
    args = learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=4)
    L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, dev, sample_seed=1)
    g = torch.Generator(device=dev).manual_seed(5)
    L.obs.copy_(torch.randint(0, 256, L.obs.shape, dtype=torch.uint8, device=dev, generator=g))
    L.actions.copy_(torch.randint(0, 4, (T, N), device=dev, generator=g).float())
    L.logprobs.copy_(-1.386 + 0.01 * torch.randn((T, N), device=dev, generator=g))
    L.values.copy_(torch.randn((T, N), device=dev, generator=g))
    L.advantages.copy_(torch.randn((T, N), device=dev, generator=g))
    L.returns.copy_(L.advantages + L.values)
    sums = []
    real = L.optimizer_step_hip
    names = [n for n, _ in agent.named_parameters()]
    hashes = []

    def h(t):          # order-independent exact hash of a tensor's bits
        v = t.detach().contiguous().view(torch.int32).to(torch.int64)
        w = torch.arange(1, 2 * v.numel(), 2, device=v.device, dtype=torch.int64)
        return int((v.reshape(-1) * w).sum().item()) & (2**64 - 1)

    def spy(lr):
        gsum = L.flat.grads.double().sum().item()
        if len(sums) < 2:       # per-parameter gradient hashes of the first two minibatches + the trunk's buffers
            row = {n: h(L.flat.grads[o:o + k]) for n, (o, k) in zip(names, L.flat.segments)}
            def walk(prefix, val):
                if isinstance(val, torch.Tensor):
                    if val.is_cuda and val.numel() > 256 and val.element_size() == 4:
                        row[prefix] = h(val)
                elif isinstance(val, (list, tuple)):
                    for i2, v2 in enumerate(val):
                        walk(f"{prefix}[{i2}]", v2)
                elif isinstance(val, dict):
                    for k2, v2 in val.items():
                        walk(f"{prefix}[{str(k2)[:40]}]", v2)

            walk("bufs", vars(agent._trunk.bufs))
            hashes.append(row)
        real(lr)
        sums.append((gsum, L.flat.params.double().sum().item()))

    L.optimizer_step_hip = spy
    np.random.seed(7)
    L.update(2.5e-4)
    torch.cuda.synchronize()
    for i, row in enumerate(hashes):
        for k, v in row.items():
            print(f"H mb{i + 1} {k} {v:016x}")
    print(f"N={N} pid={os.getpid()} " + " ".join(f"{a:.17g}/{b:.17g}" for a, b in sums[:3] + sums[7:8] + sums[-1:]), flush=True)


if __name__ == "__main__":
    main()
