"""PCIe-inclusive mode: the same learner fed by HOST vector envs (the reference's arrangement: envs on the host, one D2H of
the actions and one H2D of the frames per step).  Frames come from envs.SyntheticAtariVecEnv (numpy, host cores) and go
through the env-group lanes of cleanrl_amd/pipeline.py.

    python tools/host_env_bench.py [--num-envs 1024] [--iters 2] [--groups 1 2 4 8]      -> one JSON line per group count
This number is NOT bench.py's `value` (which starts with inputs resident in HBM); bench.py reports it as `pcie_inclusive`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanrl_amd import learner_smoke  # noqa: E402
from cleanrl_amd.agents import AtariAgent  # noqa: E402
from cleanrl_amd.envs import SyntheticAtariVecEnv  # noqa: E402
from cleanrl_amd.learner import PPOLearner  # noqa: E402


def run(num_envs=1024, num_steps=128, iters=2, groups=4, frame_delta=True, device=None, graphs=None, workers=None, pin=True, static_frames=False):
    """PCIe-inclusive env-steps/s of the learner fed by HOST vector envs (numpy stand-ins on host threads): ``groups`` = 1 is
    the reference's serial arrangement (act -> D2H -> envs.step -> H2D of the full stacks), ``groups`` > 1 the overlapped
    env-group lanes of cleanrl_amd/pipeline.py.  Returns the dict that main() prints."""
    from cleanrl_amd.pipeline import GroupedRollout, split_env_groups

    dev = device or torch.device("cuda:0")
    N, T = num_envs, num_steps
    workers = groups > 1 if workers is None else workers
    if workers:          # every group's vector env in its own process, frames through shared memory (cleanrl_amd/env_workers.py)
        from cleanrl_amd.env_workers import ProcessVecEnv

        envs = split_env_groups(lambda g, n: ProcessVecEnv(("cleanrl_amd.envs", "SyntheticAtariVecEnv",
                                                            dict(num_envs=n, seed=1 + g * n, api="gym", static_frames=static_frames))), N, groups)
    else:
        envs = split_env_groups(lambda g, n: SyntheticAtariVecEnv(n, seed=1 + g * n, api="gym", static_frames=static_frames), N, groups)
    torch.manual_seed(1)
    np.random.seed(1)
    agent = AtariAgent(envs[0]).to(dev)
    args = learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=4, clip_coef=0.1)
    L = PPOLearner(agent, args, envs[0].single_observation_space, envs[0].single_action_space, N, dev, sample_seed=1)
    roll = GroupedRollout(L, groups, frame_delta=frame_delta)
    for g, e in enumerate(envs):
        roll.first_observation(g, e.reset())
    graphs = groups > 1 if graphs is None else graphs
    if graphs:
        roll.capture()                                # one hipGraph per (lane, step) for the policy forward + sampling + D2H
    one_thread = bool(graphs and workers)             # envs in processes + captured lane steps: one host thread drives every lane
    pinned = one_thread and pin and all([e.pin() for e in envs])      # frames DMA'd straight from the workers' (registered) shared memory
    t_env = [0.0] * groups

    def step_fn(g, actions, step):
        e0 = time.perf_counter()
        o, r, d, _ = envs[g].step(actions)
        t_env[g] += time.perf_counter() - e0
        return o, r, d

    t_roll = t_upd = 0.0
    for it in range(iters + 1):
        if it == 1:                                   # iteration 0 is warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t_roll = t_upd = 0.0
            t_env[:] = [0.0] * groups
        r0 = time.perf_counter()
        if one_thread:
            roll.run_async(envs)
        else:
            roll.run(step_fn)
        L.finish_rollout()
        torch.cuda.synchronize()
        t_roll += time.perf_counter() - r0
        u0 = time.perf_counter()
        L.update(args.learning_rate)
        L.start_iteration()
        torch.cuda.synchronize()
        t_upd += time.perf_counter() - u0
    el = time.perf_counter() - t0
    for e in envs:
        e.close()
    delta = roll.lanes[0].delta
    kind = ", STATIC frames: an env that costs almost nothing -- the pipeline ceiling" if static_frames else ""
    return {"mode": f"host envs (numpy stand-ins{kind}{', one worker process per group' if workers else ''}) in {groups} env group(s): pinned uint8 staging, one stream + host thread per group"
                    + (", newest-frame-only H2D" if delta else ", full-stack H2D") + (", captured lane steps" if graphs else "")
                    + (", one driver thread" + (", frames DMA'd from the workers' pinned shared memory" if pinned else "") if one_thread else ""),
            "num_envs": N, "num_steps": T, "iters": iters, "env_groups": groups, "sps": N * T * iters / el,
            "ms_per_iter": el / iters * 1e3, "rollout_ms": t_roll / iters * 1e3,
            "host_env_ms_per_group": [t / iters * 1e3 for t in t_env], "update_ms": t_upd / iters * 1e3,
            "h2d_bytes_per_step": N * (7056 if delta else 28224),
            **({"lane_step_us": lane_step_us(roll.async_stats)} if one_thread else {})}


def stand_in_step_us(n: int, reps: int = 50) -> float:
    """Microseconds ONE step of the numpy stand-in takes for a group of n envs in this process (mostly np.take of n four-frame stacks): the
    env cost inside `env_wait_us`, to be read beside it."""
    env = SyntheticAtariVecEnv(n, seed=1, api="gym")
    out = np.zeros((n, 4, 84, 84), np.uint8)
    env.reset(out=out)
    act = np.zeros(n, np.int64)
    for _ in range(3):
        env.step(act, out=out)
    t0 = time.perf_counter()
    for _ in range(reps):
        env.step(act, out=out)
    return (time.perf_counter() - t0) / reps * 1e6


def lane_step_us(async_stats: dict) -> dict:
    """Microseconds PER LANE STEP the driver thread spends waiting on the GPU / on the env workers / in its own Python, from the pipeline's summed
    seconds (``GroupedRollout.async_stats``: keys ``*_s``).  The keys say what the values are: ``gpu_wait_us``, ``env_wait_us``, ``host_us``."""
    n = max(async_stats.get("lane_steps", 0), 1)
    return {(k[:-2] + "_us" if k.endswith("_s") else k): (v / n * 1e6 if k != "lane_steps" else v) for k, v in async_stats.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=1024)
    ap.add_argument("--num-steps", type=int, default=128)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--groups", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--no-graphs", action="store_true", help="eager lane steps (the arrangement before GroupedRollout.capture)")
    ap.add_argument("--no-pin", action="store_true", help="do not register the workers' shared memory as pinned host memory (stage through the lanes' buffers)")
    ap.add_argument("--no-workers", action="store_true", help="step every group's envs in the training process (threads only)")
    ap.add_argument("--static-frames", action="store_true", help="envs that cost almost nothing (frames left as they are): the pipeline's own ceiling")
    a = ap.parse_args()
    for k in a.groups:
        print(json.dumps(run(a.num_envs, a.num_steps, a.iters, k, frame_delta=k > 1, graphs=False if a.no_graphs else None,
                             workers=False if a.no_workers else None, pin=not a.no_pin, static_frames=a.static_frames)), flush=True)


if __name__ == "__main__":
    main()
