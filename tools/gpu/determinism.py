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
        if t.numel() < 256 or t.element_size() != 4:
            t = t.detach().float().reshape(-1)
            t = torch.cat([t, torch.zeros(256, device=t.device)])
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
    if os.environ.get("DET_HEADS") == "1":      # hash the heads' backward outputs of EVERY call; report the first call that differs
        import json

        from cleanrl_amd import cnn

        real_bwd = cnn.HeadsFn.backward
        calls, rows = [0], []
        ref_path = os.environ.get("DET_REF")
        ref = json.load(open(ref_path)) if ref_path and os.path.exists(ref_path) else None
        state = {"reported": False}

        def bwd(ctx, dlogits, dvalue):
            out = real_bwd(ctx, dlogits, dvalue)
            calls[0] += 1
            hh, Wa, Wc = ctx.saved_tensors
            row = {"in": [h(hh), h(Wc), h(dlogits), h(dvalue)], "out": [h(out[0]), h(out[1]), h(out[3]), h(out[4])]}
            rows.append(row)
            if ref is not None and not state["reported"] and calls[0] <= len(ref):
                r = ref[calls[0] - 1]
                if r["in"] != row["in"]:
                    state["reported"] = True
                    print(f"FIRST MISMATCH call {calls[0]}: INPUTS differ (something upstream)", flush=True)
                elif r["out"] != row["out"]:
                    state["reported"] = True
                    names = ["dz", "dWa", "dWc", "dbc"]
                    print(f"FIRST MISMATCH call {calls[0]}: heads backward outputs differ on identical inputs: "
                          f"{[n for n, x, y in zip(names, r['out'], row['out']) if x != y]}", flush=True)
            return out

        cnn.HeadsFn.backward = staticmethod(bwd)
    np.random.seed(7)
    for _ in range(int(os.environ.get("DET_UPDATES", "1"))):
        L.update(2.5e-4)
    torch.cuda.synchronize()
    if os.environ.get("DET_HEADS") == "1":
        if ref is None and ref_path:
            json.dump(rows, open(ref_path, "w"))
        elif not state["reported"]:
            print(f"NO MISMATCH in {calls[0]} heads-backward calls", flush=True)
    for i, row in enumerate(hashes):
        for k, v in row.items():
            print(f"H mb{i + 1} {k} {v:016x}")
    print(f"N={N} pid={os.getpid()} " + " ".join(f"{a:.17g}/{b:.17g}" for a, b in sums[:3] + sums[7:8] + sums[-1:]), flush=True)


if __name__ == "__main__":
    main()
