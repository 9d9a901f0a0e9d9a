import time, torch
dev = torch.device("cuda:0")
pin = torch.empty(131072, dtype=torch.int64).pin_memory()
d = torch.empty(131072, dtype=torch.int64, device=dev)
x = torch.randn(1 << 20, device=dev)
def t(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return round((t1 - t0) / n * 1e6, 1), round((t2 - t0) / n * 1e6, 1)
print("H2D 1 MB pinned non_blocking, idle stream (host us per call, incl. drain):", t(lambda: d.copy_(pin, non_blocking=True)))
src = torch.arange(131072)
print("CPU copy into pinned:", t(lambda: pin.copy_(src)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x * 2
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    y = x * 2; y = y + 1; y = y * 3; y = y - 1
print("graph replay (4 kernels):", t(lambda: g.replay()))
def seq():
    for _ in range(32): g.replay()
    d.copy_(pin, non_blocking=True)
print("32 replays + H2D:", t(seq, 10))
def seq2():
    for _ in range(32): g.replay()
print("32 replays:", t(seq2, 10))
def seq3():
    for _ in range(32):
        y = x * 2; y = y + 1; y = y * 3; y = y - 1
    d.copy_(pin, non_blocking=True)
print("128 eager kernels + H2D:", t(seq3, 10))
import numpy as np
b = np.arange(131072)
print("np.random.shuffle 131072:", t(lambda: np.random.shuffle(b), 5))
