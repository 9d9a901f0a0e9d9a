"""Shared body of the full-size whole-iteration GPU tests (configs C and D; config B's test predates it and keeps its own copy):
teacher-force the HIP learner through ONE iteration of the reference's own lines (oracle/mint_full_size.py goldens) and measure
how far every recorded quantity is from the reference.  Not a test module itself."""
from types import SimpleNamespace

import numpy as np
import torch

from cleanrl_amd import envs as E, learner_smoke, synthetic
from cleanrl_amd.agents import AtariAgent
from cleanrl_amd.learner import PPOLearner

SCALAR_NAMES = ["loss", "pg_loss", "v_loss", "entropy_loss", "old_approx_kl", "approx_kl", "clipfrac"]


def cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def run_atari_iteration(g, dev, rank=0, world=1, T=128, N=None, graphs=False):
    """-> dict of what the HIP path produced.  ``g`` is the golden case; with world > 1 its per-rank arrays carry the suffix
    ``_rank<r>`` and torch.distributed must be initialised (the learner all-reduces the flat gradient).  ``graphs``: the update
    replays captured hipGraphs (``capture_update``: bench.py's and runner.train's default on one GPU) -- the optimizer step then
    lives inside the graphs, so the pre-Adam gradients are not observable and only scalars and parameters are returned."""
    sfx = f"_rank{rank}" if world > 1 else ""
    G = lambda k: g[k + sfx]                                                  # noqa: E731
    T, N = G("rewards").shape
    frames = synthetic.atari_frames((T + 1) * N, seed=int(G("frame_seed"))).reshape(T + 1, N, 4, 84, 84)
    assert int(frames.sum(dtype=np.int64)) == int(G("frames_checksum")), "the seeded frames differ from the golden's"
    env = SimpleNamespace(single_observation_space=E.Box(0, 255, (4, 84, 84), np.uint8), single_action_space=E.Discrete(4))
    torch.manual_seed(int(g["init_seed"]))
    agent = AtariAgent(env).to(dev)
    args = learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=4)
    L = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, dev, world_size=world, sample_seed=1 + rank)
    assert L.hip and L.fused_cnn and L.minibatch_size == T * N // 4
    stride = int(g["stride"])
    out = {"init_err": float(np.abs(L.flat.params[::stride].cpu().numpy() - g["init_params_sub"]).max())}
    step_done = G("step_done")
    actions, logprobs, values = G("actions").astype(np.float32), G("logprobs"), G("values")
    L.observe(0, frames[0], step_done[0])
    worst_value = 0.0
    for step in range(T):
        L.act(step)
        worst_value = max(worst_value, float(np.abs(L.values[step].cpu().numpy() - values[step]).max()))
        L.actions[step].copy_(torch.from_numpy(actions[step]))
        L.logprobs[step].copy_(torch.from_numpy(logprobs[step]))
        L.values[step].copy_(torch.from_numpy(values[step]))
        L.store_reward(step, G("rewards")[step])
        L.observe(step + 1, frames[step + 1], step_done[step + 1])
    del frames
    out["worst_value"], out["value_scale"] = worst_value, max(1.0, float(np.abs(values).max()))
    L.finish_rollout()
    out["adv_err"] = float(np.abs(L.advantages.cpu().numpy() - G("advantages")).max())
    out["ret_err"] = float(np.abs(L.returns.cpu().numpy() - G("returns")).max())
    L.advantages.copy_(torch.from_numpy(G("advantages")))
    L.returns.copy_(torch.from_numpy(G("returns")))
    keep = tuple(int(k) for k in g["grad_updates"])
    seen, count = {}, [0]
    real = L.optimizer_step_hip

    def spy(lr):
        count[0] += 1
        if count[0] in keep:
            seen[count[0]] = L.flat.grads.clone()          # after the all-reduce (SUM over the ranks), before /world, clip, Adam
        real(lr)

    if graphs:
        L.capture_update()
        assert L._update_graphs is not None
    else:
        L.optimizer_step_hip = spy
    np.random.seed(int(G("shuffle_seed")))
    m = L.update(float(g["lr"]))
    out["num_updates"] = m["num_updates"]
    out["scalars"] = L._scalars[:m["num_updates"]].cpu().numpy().astype(np.float64)
    sizes = [p.numel() for p in agent.parameters()]
    for k in (keep if not graphs else ()):
        gh = seen[k].cpu().numpy().astype(np.float64) / world
        n = np.linalg.norm(gh)
        clipped = gh * min(1.0, args.max_grad_norm / (n + 1e-6))             # clip_grad_norm_(0.5) as the reference's step saw it
        s = int(g[f"mb{k}_grad_stride"])
        out[f"grad{k}_sub"] = clipped[::s].astype(np.float32)
        out[f"grad{k}_norm"] = float(np.linalg.norm(clipped))
        out[f"grad{k}_tensor_norms"] = np.array([np.linalg.norm(c) for c in np.split(clipped, np.cumsum(sizes)[:-1])])
        out[f"grad{k}_mag_sub"] = np.abs(gh[::stride]).astype(np.float32)
    out["final_params_sub"] = L.flat.params[::stride].cpu().numpy()
    out["params_checksum"] = float(L.flat.params.double().sum().item())
    L.flat.check_views()
    return out


def check_atari_iteration(out, g, bars, sfx="", clip_rows=None, report=None):
    """``bars[k]`` = (max element / absmax, 1 - cosine, whole norm, per-tensor norms) for the gradient at update k.
    ``report`` (a list) receives one line per measured quantity, pass or fail."""
    problems = []
    report = report if report is not None else []
    if out["init_err"] > 2e-6:
        problems.append(f"initial parameters differ from the reference Agent's: {out['init_err']:.2e}")
    if out["worst_value"] > 5e-5 * out["value_scale"]:
        problems.append(f"rollout values: {out['worst_value']:.2e} > 5e-5 x {out['value_scale']:.2f}")
    if out["adv_err"] > 2e-4 or out["ret_err"] > 2e-4:
        problems.append(f"GAE from the kernels' own values: advantages {out['adv_err']:.2e}, returns {out['ret_err']:.2e}")
    assert out["num_updates"] == 16
    assert [str(x) for x in g["scalar_names"]] == SCALAR_NAMES
    sc, ref = out["scalars"], g["scalars" + sfx].astype(np.float64)
    rows = clip_rows or out["scalars"].shape[0]
    # rtol 1e-3 of each scalar plus an absolute floor per column (pg_loss and the KL estimates sit near 0; clipfrac counts rows)
    atol = np.array([2e-4, 2e-4, 2e-4, 1e-4, 2e-5, 2e-5, 2.5e-3])
    err, bar = np.abs(sc - ref), 1e-3 * np.abs(ref) + atol
    if not (err <= bar).all():
        problems.append("minibatch scalars off the reference's lines: worst err/bar per column %s at updates %s" % (
            (err / bar).max(0).round(3), (err / bar).argmax(0) + 1))
    for k in (int(x) for x in g["grad_updates"]):
        if f"grad{k}_sub" not in out:              # (update graphs: the optimizer step sits inside the replayed graph)
            continue
        want = g[f"mb{k}_grad_sub"]
        worst = np.abs(out[f"grad{k}_sub"] - want).max() / float(g[f"mb{k}_grad_absmax"])
        c = cos(out[f"grad{k}_sub"], want)
        nrm = out[f"grad{k}_norm"] / float(g[f"mb{k}_grad_norm"]) - 1.0
        per_rel = np.abs(out[f"grad{k}_tensor_norms"] / g[f"mb{k}_grad_tensor_norms"] - 1.0)
        b = bars[k]
        report.append(f"update {k}: max|dg|/absmax {worst:.2e}, 1-cosine {1 - c:.2e}, norm {nrm:+.2e}, per-tensor norms max {per_rel.max():.2e} "
                      f"(reference's clipped norm {float(g[f'mb{k}_grad_norm']):.4f})")
        if worst > b[0] or 1.0 - c > b[1] or abs(nrm) > b[2] or per_rel.max() > b[3]:
            problems.append(f"update {k}: max|dg|/absmax {worst:.2e}, 1-cosine {1 - c:.2e}, norm {nrm:+.2e}, per-tensor norms {per_rel.round(5)}")
    delta = out["final_params_sub"] - g["init_params_sub"]
    want = g["final_params_sub"] - g["init_params_sub"]
    close = np.isclose(delta, want, rtol=5e-2, atol=2e-5)
    order = np.argsort(out["grad16_mag_sub"] if "grad16_mag_sub" in out else np.abs(want))
    deciles = [float(close[p].mean()) for p in np.array_split(order, 10)]
    report.append(f"values {out['worst_value']:.2e}, GAE {out['adv_err']:.2e}; scalars worst err/bar per column {(err / bar).max(0).round(3)}; "
                  f"16-step move: cosine {cos(delta, want):.7f}, length ratio {np.linalg.norm(delta) / np.linalg.norm(want):.5f}, "
                  f"within 5 % {close.mean():.4f}; params checksum {out['params_checksum']!r}")
    if close.mean() <= 0.98 or min(deciles[2:]) <= 0.99:
        problems.append(f"only {close.mean():.4f} of sampled parameters match after 16 updates; by |g| decile: {deciles}")
    if cos(delta, want) <= 0.9999 or abs(np.linalg.norm(delta) / np.linalg.norm(want) - 1.0) > 2e-3:
        problems.append(f"16-step parameter move: cosine {cos(delta, want):.6f}, length ratio {np.linalg.norm(delta) / np.linalg.norm(want):.5f}")
    return problems
