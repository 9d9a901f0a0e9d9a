"""Stress the heads' backward kernel (csrc/heads.hip) for run-to-run bit-identity while another process competes for the GPU:
    python tools/gpu/heads_stress.py [iterations] [M]
calls mi355ppo_heads_bwd_relu_f32 on fixed inputs `iterations` times and counts the calls whose outputs differ from the first."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from cleanrl_amd import _lib, ops  # noqa: E402
from cleanrl_amd.ops import _ptr, _stream, _workspace  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    A, H = 4, 512
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(1)
    h = torch.relu(torch.randn(M, H, device=dev, generator=g))
    Wa, Wc = torch.randn(A, H, device=dev, generator=g), torch.randn(1, H, device=dev, generator=g)
    dlogits, dvalue = torch.randn(M, A, device=dev, generator=g), torch.randn(M, 1, device=dev, generator=g)
    ws = _workspace(dev, lib.mi355ppo_heads_bwd_workspace_bytes(M, A))
    outs = lambda: [torch.empty(M, 516, device=dev), torch.empty(A, H, device=dev), torch.empty(A, device=dev), torch.empty(1, H, device=dev),      # noqa: E731
                    torch.empty(1, device=dev), torch.empty(H, device=dev)]
    names = ["dz", "dWa", "dba", "dWc", "dbc", "dbh"]

    def call(o):
        for t in o:
            t.zero_()
        st = lib.mi355ppo_heads_bwd_relu_f32(_ptr(h), _ptr(Wa), _ptr(Wc), _ptr(dlogits), _ptr(dvalue), _ptr(o[0]), 516, _ptr(o[1]), _ptr(o[2]),
                                             _ptr(o[3]), _ptr(o[4]), _ptr(o[5]), M, A, H, _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(st, "heads_bwd_relu")

    ref = outs()
    call(ref)
    torch.cuda.synchronize()
    bad = {n: 0 for n in names}
    o = outs()
    t0 = time.time()
    for it in range(iters):
        call(o)
        for n, a, b in zip(names, o, ref):
            if not torch.equal(a, b):
                bad[n] += 1
    print(f"pid={os.getpid()} M={M} iterations={iters} mismatching calls: {bad}  ({time.time() - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
