"""Error of the bf16-pipe kernel Z (FC forward; layers 2 / 3 forward) against float64, next to the f32-pipe / library result on the
same inputs.  Run once per setting of MI355PPO_BF16_PAIRS (read once per process): prints one JSON line.  (profiles/r03_err_pairs.jsonl
was taken with round 2's kernels X / C, which multiplied the same term pairs: keys x_* / c_*.)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanrl_amd import cnn  # noqa: E402

DEV = torch.device("cuda:0")
g = torch.Generator(device=DEV).manual_seed(5)
out = {"pairs": os.environ.get("MI355PPO_BF16_PAIRS", "6")}
M = 4096
a = torch.relu(torch.randn(M, 3136, device=DEV, generator=g)) * torch.exp(torch.randn(M, 3136, device=DEV, generator=g))
W = torch.randn(512, 3136, device=DEV, generator=g) / 56.0
b = torch.randn(512, device=DEV, generator=g) * 0.1
ref = torch.relu(a.double() @ W.double().t() + b.double())
x = cnn.fc_fwd_relu_packed(a, cnn.fc_pack(W), b, 512).double()
lib = torch.relu(a @ W.t() + b).double()
s = ref.abs().max().item()
out["fc_fwd"] = {"x_max": (x - ref).abs().max().item() / s, "x_mean": (x - ref).abs().mean().item() / s,
                 "lib_max": (lib - ref).abs().max().item() / s, "lib_mean": (lib - ref).abs().mean().item() / s,
                 "x_bias_mean": (x - ref).mean().item() / s}
spec = {2: (32, 64, 4, 2, 20), 3: (64, 64, 3, 1, 9)}
for layer, (cin, cout, k, st, hin) in spec.items():
    xin = torch.relu(torch.randn(2048, hin, hin, cin, device=DEV, generator=g)) * torch.exp(torch.randn(2048, hin, hin, cin, device=DEV, generator=g))
    Wc = torch.randn(cout, cin, k, k, device=DEV, generator=g) / (cin * k * k) ** 0.5
    bc = torch.randn(cout, device=DEV, generator=g) * 0.1
    cols = torch.nn.functional.unfold(xin.double().permute(0, 3, 1, 2), kernel_size=k, stride=st)
    ref = torch.relu(torch.einsum("nk,bkl->bln", Wc.double().reshape(cout, -1), cols) + bc.double()).reshape(2048, -1, cout)
    f = cnn.conv_fwd(xin, cnn.repack_weights(Wc, layer), bc, layer).double().reshape(2048, -1, cout)
    c = cnn.conv_fwd_packed(xin, cnn.conv_zpack(Wc, layer, cnn.MODE_FWD), bc, layer).double().reshape(2048, -1, cout)
    s = ref.abs().max().item()
    out[f"conv{layer}_fwd"] = {"c_max": (c - ref).abs().max().item() / s, "c_mean": (c - ref).abs().mean().item() / s,
                               "f_max": (f - ref).abs().max().item() / s, "f_mean": (f - ref).abs().mean().item() / s,
                               "c_bias_mean": (c - ref).mean().item() / s}
print(json.dumps(out))
