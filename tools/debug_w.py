import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanrl_amd import cnn, _lib
from cleanrl_amd.ops import _ptr, _stream
DEV = torch.device("cuda:0")
torch.set_printoptions(linewidth=220, precision=1, sci_mode=False)
lib = _lib.load()
M, N, K = 1024, 512, 3136
def run(dz, a):
    ws = torch.full((lib.mi355ppo_fc_wgrad_workspace_bytes(M, N, K) // 4,), -7.0, device=DEV)
    out = torch.empty(N, K, device=DEV)
    lib.mi355ppo_fc_wgrad_f32(_ptr(dz), N, _ptr(a), _ptr(out), M, N, K, 0, _ptr(ws), ws.numel() * 4, _stream(DEV))
    torch.cuda.synchronize()
    return ws[:5 * N * K].view(5, N, K)
def table(p):   # value per (tile i, tile j) of wave 0, k-block 0
    return [[float(p[32 * i + 3, 32 * j + 5]) for j in range(2)] for i in range(4)]
dz = torch.ones(M, N, device=DEV)
print("all ones, slab 0:", table(run(dz, torch.ones(M, K, device=DEV))[0]))
for b in (0, 5, 10, 55, 60):
    a = torch.zeros(M, K, device=DEV); a[16 * b:16 * b + 16] = 1.0
    part = run(dz, a)
    print("block", b, "(step", b // 5, "of slab 0):", table(part[0]), "other slabs max", float(part[1:].abs().max()))
# which rows of the block are seen: a row r only
for b in (0, 5):
    seen = []
    for r in range(16):
        a = torch.zeros(M, K, device=DEV); a[16 * b + r] = 1.0
        seen.append([t[0] for t in table(run(dz, a)[0])][:2])
    print("block", b, "per-row contribution to tiles (0,0),(1,0):", seen)
