"""Error of kernel P's dW1 / db1 against float64 at 4,096 dense images (every image carries a gradient), for the mode the environment
selects (MI355PPO_P_ZEXT=1: zero-extended subnormal frame operand; 0: converted operand).  Prints one JSON line; with a file argument also
saves dW1 so that two runs can be compared element by element (`--compare a.pt b.pt`)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "--compare":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        d = (a.double() - b.double()).abs()
        print(json.dumps({"compare": [sys.argv[2], sys.argv[3]], "elements": a.numel(), "differing": int((a != b).sum()),
                          "max_abs_diff_over_scale": float(d.max() / a.double().abs().max()), "scale": float(a.abs().max())}))
        return
    from cleanrl_amd import cnn

    dev = torch.device("cuda:0")
    M = 4096
    g = torch.Generator(device=dev).manual_seed(5)
    obs = torch.randint(0, 256, (M, 84, 84, 4), dtype=torch.uint8, device=dev, generator=g)
    dz = torch.randn(M, 20, 20, 32, device=dev, generator=g) * torch.exp(torch.randn(M, 1, 1, 1, device=dev, generator=g) * 3.0) * 1e-3
    dW, db = cnn.conv_wgrad(obs, dz, 1, None)
    x = obs.permute(0, 3, 1, 2).double() / 255.0
    Wd = torch.zeros(32, 4, 8, 8, dtype=torch.float64, device=dev, requires_grad=True)
    bd = torch.zeros(32, dtype=torch.float64, device=dev, requires_grad=True)
    refW, refb = torch.autograd.grad(F.conv2d(x, Wd, bd, stride=4), (Wd, bd), dz.permute(0, 3, 1, 2).double())
    eW = (dW.double() - refW).abs()
    out = {"p_zext": os.environ.get("MI355PPO_P_ZEXT", "1"), "images": M, "dW1_scale": float(refW.abs().max()),
           "dW1_max_err_over_scale": float(eW.max() / refW.abs().max()), "dW1_mean_err_over_scale": float(eW.mean() / refW.abs().max()),
           "db1_max_err_over_scale": float((db.double() - refb).abs().max() / refb.abs().max())}
    # torch's own f32 weight gradient on the device, for calibration
    Wf = torch.zeros(32, 4, 8, 8, device=dev, requires_grad=True)
    tW, = torch.autograd.grad(F.conv2d(obs.permute(0, 3, 1, 2).float() / 255.0, Wf, None, stride=4), (Wf,), dz.permute(0, 3, 1, 2).contiguous())
    out["torch_f32_max_err_over_scale"] = float((tW.double() - refW).abs().max() / refW.abs().max())
    out["torch_f32_mean_err_over_scale"] = float((tW.double() - refW).abs().mean() / refW.abs().max())
    print(json.dumps(out))
    if len(sys.argv) > 1:
        torch.save(dW.cpu(), sys.argv[1])


if __name__ == "__main__":
    main()
