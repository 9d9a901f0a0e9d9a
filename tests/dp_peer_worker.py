"""Worker of ``test_gpu_multirank.py::test_peer_memory_all_reduce_*``: ONE rank (a process) of the peer-memory gradient exchange
(``cleanrl_amd/dp_comm.py``, ``csrc/dpcomm.hip``), all ranks on ``cuda:0`` -- the peers' segments are HIP IPC mappings of the same HBM.
The process group (gloo) carries the handles and the barriers only.  Reference: the all-reduce block of ppo_atari_multigpu.py:360-367.
Modes: ``raw`` (arithmetic, sizes, back-to-back rounds, a skewed rank, graph replays), ``timeout`` (rank 1 skips a call).  Not a test module."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cleanrl_amd.dp_comm import PeerAllReduce  # noqa: E402

NMAX = 1686693           # the NatureCNN agent's flat gradient (A = 4): 6.75 MB


def rank_input(n, q, it):
    g = np.random.default_rng(1000003 * it + 7919 * q + n)
    return (g.standard_normal(n) * np.float32(10.0) ** g.integers(-3, 3, size=n)).astype(np.float32)


def rank_order_sum(n, world, it):
    s = rank_input(n, 0, it)
    for q in range(1, world):
        s = s + rank_input(n, q, it)            # f32 + f32, rank order: what one rank computes for its slice
    return s


def raw(out_dir, rank, world, dev):
    comm = PeerAllReduce(NMAX, dev, timeout_s=20)
    res = {"sizes": {}, "world": world}
    # 1. sizes around the vector / slice edges, every element against the rank-order f32 sum (bit for bit)
    for n in (1, 2, 3, 4, 5, 7, 8, 4 * world - 1, 4 * world, 4 * world + 1, 1023, 4096, 100003, NMAX):
        x = torch.from_numpy(rank_input(n, rank, 0)).to(dev)
        buf = torch.zeros(n + 8, dtype=torch.float32, device=dev)      # (guard words behind the n floats)
        buf[:n] = x
        comm.all_reduce_sum_(buf[:n])
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        res["sizes"][str(n)] = bool(np.array_equal(got[:n], rank_order_sum(n, world, 0)) and not got[n:].any())
    # 2. 300 rounds back to back with no host synchronisation, rank `it % world` held up by a sleep kernel in front of some of them: the one-buffer
    # protocol (a rank may be a whole round ahead of a peer); inputs differ per round, every round's result is kept and checked at the end
    n, rounds = 50001, 300
    base = torch.from_numpy(rank_input(n, rank, 1)).to(dev)
    outs = torch.zeros((rounds, n + 3), dtype=torch.float32, device=dev)      # (rows 16-byte aligned: the entry point's contract)
    for it in range(rounds):
        if it % 7 == 0 and (it // 7) % world == rank:
            torch.cuda._sleep(2_000_000)                                # ~1 ms of this rank only
        outs[it, :n] = base * float(1 + (it % 5))
        comm.all_reduce_sum_(outs[it, :n])
    torch.cuda.synchronize()
    ins = [rank_input(n, q, 1) for q in range(world)]
    outs_np = outs.cpu().numpy()
    ok = True
    for it in range(rounds):
        s = ins[0] * np.float32(1 + (it % 5))
        for q in range(1, world):
            s = s + ins[q] * np.float32(1 + (it % 5))
        ok = ok and bool(np.array_equal(outs_np[it, :n], s) and not outs_np[it, n:].any())
    res["back_to_back"] = ok
    # 3. the call inside a captured graph, replayed with new inputs (the round counter lives on the device); eager calls in between
    n = 300007
    g_in = torch.zeros(n, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        comm.all_reduce_sum_(g_in)                                      # warm-up on the capture stream (every rank: same sequence)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        g_in.mul_(2.0)
        comm.all_reduce_sum_(g_in)
    ok = True
    for it in range(12):
        g_in.copy_(torch.from_numpy(rank_input(n, rank, 10 + it)).to(dev))
        graph.replay()
        if it % 3 == 2:                                                 # an eager exchange between replays: the same counter
            e = torch.from_numpy(rank_input(777, rank, 50 + it)).to(dev)
            comm.all_reduce_sum_(e)
            torch.cuda.synchronize()
            ok = ok and bool(np.array_equal(e.cpu().numpy(), rank_order_sum(777, world, 50 + it)))
        torch.cuda.synchronize()
        s = rank_input(n, 0, 10 + it) * np.float32(2)
        for q in range(1, world):
            s = s + rank_input(n, q, 10 + it) * np.float32(2)
        ok = ok and bool(np.array_equal(g_in.cpu().numpy(), s))
    res["graph_replays"] = ok
    # 4. time of one exchange of the full gradient (all ranks on ONE device: protocol + launch cost, no fabric)
    x = torch.zeros(NMAX, dtype=torch.float32, device=dev)
    for _ in range(5):
        comm.all_reduce_sum_(x)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(50):
        comm.all_reduce_sum_(x)
    torch.cuda.synchronize()
    res["us_per_exchange_same_device"] = (time.perf_counter() - t0) / 50 * 1e6
    res["status"] = comm.status()
    dist.barrier()
    comm.close()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)


def timeout(out_dir, rank, world, dev):
    """Rank 1 skips the second call: every other rank's wait gives up after the communicator's timeout -- recorded, raised by check(), no hang."""
    comm = PeerAllReduce(4096, dev, timeout_s=1.5)
    x = torch.ones(4096, dtype=torch.float32, device=dev)
    comm.all_reduce_sum_(x)
    torch.cuda.synchronize()
    first_ok = bool((x == world).all().item()) and comm.status() is None
    t0 = time.perf_counter()
    if rank != 1:
        comm.all_reduce_sum_(x)
        comm.all_reduce_sum_(x)            # behind a broken round: returns at once
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    st = comm.status()
    raised = False
    try:
        comm.check()
    except RuntimeError:
        raised = True
    dist.barrier()
    comm.close()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"first_ok": first_ok, "status": st, "raised": raised, "waited_s": waited}, f)


if __name__ == "__main__":
    rank, world = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    {"raw": raw, "timeout": timeout}[sys.argv[2]](sys.argv[1], rank, world, dev)
    dist.barrier()
    dist.destroy_process_group()
