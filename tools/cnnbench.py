"""Time the f32-MFMA NatureCNN kernels against torch's (MIOpen) convolutions on one MI355X.
    python tools/cnnbench.py [images ...]      -> JSON lines (us, TFLOP/s, fraction of the 157.3 TF f32-MFMA peak)"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanrl_amd import cnn  # noqa: E402

DEV = torch.device("cuda:0")
PEAK = 157.3
SPEC = cnn.LAYERS


def bench(fn, reps=10, warm=3):
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


def out(**kw):
    print(json.dumps(kw), flush=True)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 32768]
    for M in sizes:
        obs = torch.randint(0, 256, (M, 84, 84, 4), dtype=torch.uint8, device=DEV)
        inds = torch.randperm(M, device=DEV)
        acts = {0: obs}
        for layer in (1, 2, 3):
            cin, cout, k, s, hin, hout = SPEC[layer]
            W = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5
            b = torch.randn(cout, device=DEV) * 0.1
            flops = 2.0 * M * hout * hout * cout * cin * k * k
            bt = cnn.repack_weights(W, layer)
            src = acts[layer - 1]
            dst = torch.empty((M, hout, hout, cout), device=DEV)
            for v in ((2,) if os.environ.get("CNNBENCH_ONLY") == "fwd" else (2, 4)):
                us = bench(lambda: cnn.conv_fwd(src, bt, b, layer, inds if layer == 1 else None, dst, variant=v))
                out(k="fwd", variant=v, layer=layer, M=M, us=us, tflops=flops / us / 1e6, frac=flops / us / 1e6 / PEAK)
            if layer == 1:      # kernel Q: the integer matrix pipe (HBM-bound: report GB/s of the algorithmic bytes too)
                pack = cnn.repack_weights(W, 1, cnn.MODE_FWD_Q)
                us = bench(lambda: cnn.conv_fwd(src, pack, b, 1, inds, dst, variant=cnn.VARIANT_Q))
                nbytes = M * (84 * 84 * 4 + 20 * 20 * 32 * 4)
                out(k="fwd", variant=6, layer=1, M=M, us=us, tflops=flops / us / 1e6, frac_of_f32_peak=flops / us / 1e6 / PEAK,
                    algorithmic_GBps=nbytes / us / 1e3, frac_of_8TBps=nbytes / us / 1e3 / 8000)
            acts[layer] = dst
            dz = torch.randn_like(dst)
            if os.environ.get("CNNBENCH_ONLY") == "fwd":
                continue
            us = bench(lambda: cnn.conv_wgrad(src, dz, layer, inds if layer == 1 else None))
            out(k="wgrad", layer=layer, M=M, us=us, tflops=flops / us / 1e6, frac=flops / us / 1e6 / PEAK)
            if layer > 1:
                mode = cnn.MODE_DGRAD_S1 if layer == 3 else cnn.MODE_DGRAD_S2
                btd = cnn.repack_weights(W, layer, mode)
                dsrc = torch.empty_like(src)
                for v in ((2, 5) if layer == 3 else (2, 6)):
                    if v == 5:
                        btd = cnn.repack_weights(W, layer, cnn.MODE_DGRAD_S1_CLASSES)
                    if v == 6:
                        btd = cnn.repack_weights(W, layer, cnn.MODE_DGRAD_S2_CLASSES)
                    us = bench(lambda: cnn.conv_dgrad(dz, btd, src, layer, dsrc, variant=v))
                    out(k="dgrad", variant=v, layer=layer, M=M, us=us, tflops=flops / us / 1e6, frac=flops / us / 1e6 / PEAK,
                        note="flops counted as the forward conv's")
            if os.environ.get("CNNBENCH_TORCH", "0") != "1":
                continue
            # torch / MIOpen on channels-last f32 for comparison
            if layer == 1:
                x = (obs.float() / 255.0).permute(0, 3, 1, 2)
            else:
                x = src.permute(0, 3, 1, 2)
            x = x.detach().requires_grad_(layer > 1)
            Wt = W.clone().to(memory_format=torch.channels_last).requires_grad_(True)
            bt_ = b.clone().requires_grad_(True)
            us = bench(lambda: F.conv2d(x, Wt, bt_, stride=s))
            out(k="torch_fwd", layer=layer, M=M, us=us, tflops=flops / us / 1e6)
            y = F.conv2d(x, Wt, bt_, stride=s)
            gy = torch.randn_like(y)
            ins = (x, Wt, bt_) if layer > 1 else (Wt, bt_)
            us = bench(lambda: torch.autograd.grad(y, ins, gy, retain_graph=True))
            out(k="torch_bwd_all", layer=layer, M=M, us=us)
            del x, y, gy


if __name__ == "__main__":
    main()
