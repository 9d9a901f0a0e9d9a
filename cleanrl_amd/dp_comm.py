"""Host side of the peer-memory gradient exchange (``csrc/dpcomm.hip``, ``include/mi355ppo.h`` section a9/e): the drop-in for
``dist.all_reduce(all_grads_list, op=dist.ReduceOp.SUM)`` of cleanrl/ppo_atari_multigpu.py:360-367 on the persistent flat gradient buffer
that needs no collective library -- every rank's segment is mapped by its peers through HIP IPC (xGMI between the GPUs of a node) and a
call is five small launches on the CURRENT stream, so it may sit inside a hipGraph capture.

The process group of the reference (``dist.init_process_group``, :166-175) stays what it is for: the rendezvous.  Here it carries the
64-byte handles once (``all_gather_object``) and one barrier; the gradients never touch it.

``MI355PPO_ALLREDUCE=peer`` (or ``peer:<seconds>``: the wait timeout, default 20) selects it in ``PPOLearner`` (default ``pg``: the process
group's all-reduce -- RCCL on a multi-GPU node).
No build round could run it ACROSS GPUs (one GPU per box): tests/test_gpu_multirank.py runs 2 and 4 processes on one device."""
from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib

MAX_WORLD = 8           # MI355PPO_DP_MAX_WORLD
HANDLE_BYTES = 64       # MI355PPO_DP_HANDLE_BYTES
_PHASE = {1: "reduce", 2: "collect"}


def exchange_policy(world_size: int) -> str:
    """"peer" or "pg": which transport ``PPOLearner`` gives the flat gradient to when world > 1 (``MI355PPO_ALLREDUCE``; default "pg")."""
    v = _setting()[0]
    return v if world_size > 1 else "pg"


def _setting():
    """``MI355PPO_ALLREDUCE`` = ``pg`` | ``peer`` | ``peer:<seconds>`` -> (route, seconds after which a wait on a peer gives up)."""
    raw = os.environ.get("MI355PPO_ALLREDUCE", "pg").strip().lower()
    v, _, t = raw.partition(":")
    try:
        timeout_s = float(t) if t else 20.0
    except ValueError:
        timeout_s = -1.0
    if v not in ("peer", "pg") or timeout_s <= 0.0 or (t and v != "peer"):
        raise ValueError(f"MI355PPO_ALLREDUCE={raw!r}: 'pg' (the process group's all-reduce), 'peer' (HIP IPC segments, csrc/dpcomm.hip) or 'peer:<seconds>' "
                         "(the time after which a wait on a peer gives up; default 20)")
    return v, timeout_s


class PeerAllReduce:
    """One communicator per process and rank.  Construction is collective over ``group`` (every rank calls it, with the same ``numel``)."""

    def __init__(self, numel: int, device: torch.device, group=None, timeout_s: float | None = None):
        assert dist.is_available() and dist.is_initialized(), "PeerAllReduce: the process group carries the handles (init_process_group first)"
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > MAX_WORLD:
            raise ValueError(f"PeerAllReduce: {self.world} ranks; the segments cover one node (<= {MAX_WORLD} GPUs)")
        self.device = torch.device(device)
        assert self.device.type == "cuda", "PeerAllReduce: device memory only"
        _lib.require_device(self.device.index if self.device.index is not None else torch.cuda.current_device())
        self.lib = _lib.load()
        self.numel = int(numel)
        if timeout_s is None:
            timeout_s = _setting()[1]
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mi355ppo_dp_comm_create(self.world, self.rank, self.numel, float(timeout_s) * 1e3, ctypes.byref(self._comm)),
                       "mi355ppo_dp_comm_create")
            buf = ctypes.create_string_buffer(HANDLE_BYTES)
            _lib.check(self.lib.mi355ppo_dp_comm_handle(self._comm, buf), "mi355ppo_dp_comm_handle")
            handles = [None] * self.world
            dist.all_gather_object(handles, buf.raw, group=group)
            assert all(isinstance(h, bytes) and len(h) == HANDLE_BYTES for h in handles)
            if self.world > 1:
                _lib.check(self.lib.mi355ppo_dp_comm_connect(self._comm, b"".join(handles)), "mi355ppo_dp_comm_connect")
        dist.barrier(group=group)           # every rank has mapped every segment before anybody's first flag arrives

    def all_reduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        """``flat`` (f32, contiguous, on this communicator's device, 16-byte aligned) summed over the ranks in place, enqueued on the
        current stream.  Every rank must issue the same calls in the same order."""
        assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.device == self.device and flat.numel() <= self.numel
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.mi355ppo_dp_allreduce_sum_f32(self._comm, ctypes.c_void_p(flat.data_ptr()), flat.numel(), stream),
                   "mi355ppo_dp_allreduce_sum_f32")
        return flat

    def status(self):
        """None, or (round, peer, phase) of the first wait that gave up -- a host read of a mapped word, no synchronisation."""
        r, p, ph = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        st = self.lib.mi355ppo_dp_comm_status(self._comm, ctypes.byref(r), ctypes.byref(p), ctypes.byref(ph))
        return None if st == 0 else (r.value, p.value, _PHASE.get(ph.value, str(ph.value)))

    def check(self) -> None:
        """Raise if a wait on a peer ever timed out (call it where the host synchronises anyway: the results behind a timeout are garbage)."""
        s = self.status()
        if s is not None:
            raise RuntimeError(f"peer all-reduce: rank {self.rank} gave up waiting for rank {s[1]} in round {s[0]} ({s[2]} phase) -- a peer "
                               "died or the ranks issued different sequences of calls; the communicator is unusable")

    def close(self) -> None:
        if getattr(self, "_comm", None) is not None and self._comm.value:
            with torch.cuda.device(self.device):
                self.lib.mi355ppo_dp_comm_destroy(self._comm)
            self._comm = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
