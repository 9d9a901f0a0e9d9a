"""Kernel micro-benchmarks on one MI355X (tuning aid; writes JSON lines to stdout).

    python tools/kbench.py [gae] [loss] [obs] [adam] [cnn] [sample]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cleanrl_amd import ops, synthetic  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5, flush=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        if flush is not None:
            flush()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = np.array(ts)
    return float(np.median(ts)), float(ts.min())


def timeit_batch(fn, reps=20, warm=3):
    """us per call measured over a back-to-back batch (amortises event overhead; includes launch gaps)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


_flush_buf = None


def flush_caches():
    global _flush_buf
    if _flush_buf is None:
        _flush_buf = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    _flush_buf.add_(1)


def out(**kw):
    print(json.dumps(kw), flush=True)


def bench_gae():
    sizes = [(128, 128), (128, 256), (128, 1024), (2048, 64), (128, 4096), (128, 16384), (128, 65536), (128, 262144),
             (128, 1 << 20), (128, 1 << 22)]
    if __import__("os").environ.get("KBENCH_GAE_SIZES"):
        sizes = [tuple(int(x) for x in t.split("x")) for t in __import__("os").environ["KBENCH_GAE_SIZES"].split(",")]
    for T, N in sizes:
        s = {k: v.to(DEV) for k, v in synthetic.rollout_scalars(T, min(N, 4096), 4, seed=1).items()}
        if N > 4096:
            rep = N // 4096
            s = {k: (v.repeat(1, rep) if v.dim() == 2 else v.repeat(rep)).contiguous() for k, v in s.items()}
        adv, ret = torch.empty_like(s["rewards"]), torch.empty_like(s["rewards"])
        nbytes = 20 * T * N + 8 * N
        for variant in [1, 3, 6]:      # (2 / 4 / 5: the tile kernels retired in round 6)
            f = lambda: ops.gae(s["rewards"], s["dones"], s["values"], s["next_done"], s["next_value"], 0.99, 0.95, adv, ret, variant=variant)
            med, mn = timeit(f, iters=30)
            cold, _ = timeit(f, iters=10, flush=flush_caches) if nbytes < (1 << 28) else (med, mn)
            bb = timeit_batch(f)
            out(k="gae", T=T, N=N, variant=variant, us_med=med, us_min=mn, us_cold=cold, us_b2b=bb, GBps=nbytes / med / 1e3, GBps_cold=nbytes / cold / 1e3)


def bench_loss():
    for M, A in [(4096, 4), (8192, 4), (32768, 4), (32768, 18), (131072, 4), (1 << 20, 4)]:
        Bf = 4 * M
        g = torch.Generator(device=DEV).manual_seed(0)
        logits = torch.randn(M, A, device=DEV, generator=g)
        value = torch.randn(M, device=DEV, generator=g)
        inds = torch.randperm(Bf, device=DEV)[:M]
        ba = torch.randint(0, A, (Bf,), device=DEV).float()
        bl = torch.randn(Bf, device=DEV) * 0.3 - 1.4
        badv, bval = torch.randn(Bf, device=DEV), torch.randn(Bf, device=DEV)
        bret = badv + bval
        sc, dl, dv = torch.empty(7, device=DEV), torch.empty_like(logits), torch.empty(M, device=DEV)
        f = lambda: ops.ppo_loss_categorical(logits, value, inds, ba, bl, badv, bret, bval, 0.1, 0.01, 0.5, scalars_out=sc, dlogits_out=dl, dvalue_out=dv)
        med, mn = timeit(f)
        nbytes = (8 * A + 28 + 8) * M
        out(k="loss_cat", M=M, A=A, us_med=med, us_min=mn, us_b2b=timeit_batch(f), GBps=nbytes / med / 1e3)


def bench_obs():
    R = 32768
    src = torch.randint(0, 256, (R, 4, 84, 84), dtype=torch.uint8, device=DEV)
    for M in [1024, 8192, 32768]:
        inds = torch.randint(0, R, (M,), device=DEV)
        dst = torch.empty(M, 4, 84, 84, device=DEV)
        f = lambda: ops.obs_u8_to_f32(src, inds, dst)
        med, mn = timeit(f, iters=20)
        nbytes = M * 28224 * 5
        out(k="obs", M=M, us_med=med, us_min=mn, GBps=nbytes / med / 1e3, frac_of_8TBps=nbytes / med / 1e3 / 8000)
        f2 = lambda: torch.div(src[inds].float(), 255.0)
        med2, _ = timeit(f2, iters=5, warm=2)
        out(k="obs_torch_u8_gather_float_div", M=M, us_med=med2)
    del src
    torch.cuda.empty_cache()
    srcf = torch.rand(8192, 4, 84, 84, device=DEV)
    inds = torch.randint(0, 8192, (8192,), device=DEV)
    f3 = lambda: srcf[inds] / 255.0            # the reference's f32 path: b_obs[mb_inds] ; x / 255.0
    med3, _ = timeit(f3, iters=5, warm=2)
    out(k="obs_reference_f32_gather_div", M=8192, us_med=med3)


def bench_adam():
    n = 1686693
    p, g, m, v = (torch.randn(n, device=DEV) for _ in range(4))
    v.abs_()
    f = lambda: ops.clip_adam_(p, g.normal_(), m, v, 3, 2.5e-4, 0.5)
    f0 = lambda: g.normal_()
    med, _ = timeit(f)
    med0, _ = timeit(f0)
    out(k="clip_adam", n=n, us_med=med - med0, GBps=36 * n / max(med - med0, 1e-3) / 1e3)
    params = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in (8192, 32, 32768, 64, 36864, 64, 1605632, 512, 2048, 4, 512, 1)]
    opt = torch.optim.Adam(params, lr=2.5e-4, eps=1e-5)
    for q in params:
        q.grad = torch.randn_like(q)

    def ft():
        torch.nn.utils.clip_grad_norm_(params, 0.5)
        opt.step()
    med, _ = timeit(ft, iters=20)
    out(k="torch_clip_grad_norm_plus_adam", us_med=med)


def bench_sample():
    for B, A in [(1024, 4), (1024, 18), (131072, 4)]:
        logits = torch.randn(B, A, device=DEV)
        af, lp = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
        f = lambda: ops.categorical_sample(logits, seed=1, offset=2, action_f32_out=af, logprob_out=lp)
        med, mn = timeit(f)
        out(k="cat_sample", B=B, A=A, us_med=med, us_b2b=timeit_batch(f))

        def ft():
            d = torch.distributions.Categorical(logits=logits)
            a = d.sample()
            return a, d.log_prob(a), d.entropy()
        med, _ = timeit(ft, iters=20)
        out(k="torch_categorical", B=B, A=A, us_med=med)


def bench_cnn():
    import torch.nn as nn
    net = nn.Sequential(nn.Conv2d(4, 32, 8, stride=4), nn.ReLU(), nn.Conv2d(32, 64, 4, stride=2), nn.ReLU(),
                        nn.Conv2d(64, 64, 3, stride=1), nn.ReLU(), nn.Flatten(), nn.Linear(3136, 512), nn.ReLU(),
                        nn.Linear(512, 5)).to(DEV)
    for mb in [1024, 8192, 32768]:
        x = torch.rand(mb, 4, 84, 84, device=DEV)
        for cl in (False, True):
            xx = x.contiguous(memory_format=torch.channels_last) if cl else x
            nn_ = net.to(memory_format=torch.channels_last) if cl else net.to(memory_format=torch.contiguous_format)

            def fwd():
                with torch.no_grad():
                    return nn_(xx)

            def fb():
                nn_.zero_grad(set_to_none=True)
                nn_(xx).sum().backward()
            t0 = time.time()
            medf, _ = timeit(fwd, iters=5, warm=2)
            medb, _ = timeit(fb, iters=5, warm=2)
            flops_f = mb * 18.7e6
            out(k="cnn", mb=mb, channels_last=cl, fwd_us=medf, fwdbwd_us=medb, fwd_TFs=flops_f / medf / 1e6,
                fwdbwd_TFs=3 * flops_f / medb / 1e6, wall_s=time.time() - t0)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gae", "loss", "obs", "adam", "sample", "cnn"]
    out(k="device", name=torch.cuda.get_device_name(0), torch=torch.__version__, hip=torch.version.hip)
    for w in which:
        globals()["bench_" + w]()
