"""Host vector envs stepped in worker PROCESSES (one per env group), results through shared memory.

The reference steps its envs inside the training process (``envs.step(action.cpu().numpy())``,
cleanrl/ppo_atari_multigpu.py:269-272; ``gym.vector.SyncVectorEnv`` or envpool).  The env-group lanes of ``pipeline.py``
overlap several vector envs from host THREADS, which only helps while ``step`` releases the GIL: Python-level envs
(SyncVectorEnv, the numpy stand-ins of ``envs.py`` -- fancy indexing holds the GIL) serialise again, so four lanes were no
faster than two.  ``ProcessVecEnv`` hosts a vector env in its own process:

* observations, rewards and done flags are written by the worker straight into shared memory (``env.step(..., out=)`` when the
  env supports it, a copy otherwise) -- the parent gets numpy VIEWS, no pickling of frames;
* one byte over a pipe per direction and step; the parent's wait releases the GIL, so the lanes of the other groups run;
* for stacked-frame observations the worker also writes the NEWEST plane of every env contiguously (``newest``): with the shared
  segments registered as pinned host memory (``pin()``), the lanes DMA frames to the GPU straight from what the worker wrote --
  no staging copy in the training process;
* the worker never touches the GPU and imports only what the env's module imports (``spawn`` start method: safe after the HIP
  runtime was initialised in the parent).

``step`` returns the envpool-style ``(obs, reward, done, info)`` for ``api="gym"`` envs and the gymnasium 5-tuple otherwise;
``info`` carries the per-env arrays the scripts log from (``r``, ``l``, ``terminated``, ``lives``, ``reward``) when the env
provides them as arrays of ``num_envs`` elements.  The env is described by ``(module, attribute, kwargs)`` so that the worker
can build it itself (nothing but that tuple is pickled).
"""
from __future__ import annotations

import importlib
import multiprocessing as mp
from multiprocessing import shared_memory
from typing import Optional

import numpy as np

_INFO_KEYS = (("r", np.float32), ("l", np.int32), ("terminated", np.int32), ("lives", np.int32), ("reward", np.float64))


def _attach(names: dict) -> dict:
    """Open the parent's segments in the worker (they stay the parent's to unlink; under ``spawn`` the worker shares the parent's
    resource tracker, where the attachment's registration is the same set entry as the parent's)."""
    return {k: shared_memory.SharedMemory(name=name) for k, name in names.items()}


def _views(segs: dict, n: int, obs_shape, obs_dtype, act_shape, act_dtype):
    """numpy views of the shared segments (both sides call this with the same arguments)."""
    v = {"obs": np.ndarray((n,) + tuple(obs_shape), dtype=obs_dtype, buffer=segs["obs"].buf),
         "reward": np.ndarray((n,), dtype=np.float64, buffer=segs["reward"].buf),
         "done": np.ndarray((2, n), dtype=np.bool_, buffer=segs["done"].buf),          # row 0: done / terminated, row 1: truncated
         "act": np.ndarray((n,) + tuple(act_shape), dtype=act_dtype, buffer=segs["act"].buf)}
    if "newest" in segs:
        v["newest"] = np.ndarray((n,) + tuple(obs_shape[1:]), dtype=obs_dtype, buffer=segs["newest"].buf)
    off = 0
    for key, dt in _INFO_KEYS:
        v["info_" + key] = np.ndarray((n,), dtype=dt, buffer=segs["info"].buf, offset=off)
        off += n * 8
    return v


def _worker(spec, names, n, obs_shape, obs_dtype, act_shape, act_dtype, conn):
    mod, attr, kwargs = spec
    env = getattr(importlib.import_module(mod), attr)(**kwargs)
    segs = _attach(names)
    v = _views(segs, n, obs_shape, obs_dtype, act_shape, act_dtype)
    import inspect

    def _accepts_out(fn):           # decided once from the signature: a TypeError raised INSIDE a step must surface, not re-step
        try:
            ps = inspect.signature(fn).parameters
        except (TypeError, ValueError):
            return False
        return "out" in ps or any(q.kind is inspect.Parameter.VAR_KEYWORD for q in ps.values())

    step_takes_out, reset_takes_out = _accepts_out(env.step), _accepts_out(env.reset)

    def publish(obs):
        if obs is not v["obs"]:
            np.copyto(v["obs"], obs, casting="unsafe")
        if "newest" in v:
            np.copyto(v["newest"], v["obs"][:, -1])

    try:
        while True:
            cmd = conn.recv_bytes()
            if cmd == b"q":
                break
            if cmd == b"r":
                res = env.reset(out=v["obs"]) if reset_takes_out else env.reset()
                publish(res[0] if isinstance(res, tuple) else res)
                conn.send_bytes(b"g" if not isinstance(res, tuple) else b"G")
                continue
            act = v["act"]
            res = env.step(act, out=v["obs"]) if step_takes_out else env.step(act)
            publish(res[0])
            v["reward"][:] = res[1]
            v["done"][0] = res[2]
            info = res[-1] if isinstance(res[-1], dict) else {}
            if len(res) == 5:
                v["done"][1] = res[3]
            mask = 0
            for bit, (key, dt) in enumerate(_INFO_KEYS):
                x = info.get(key)
                if isinstance(x, np.ndarray) and x.shape == (n,):
                    v["info_" + key][:] = x
                    mask |= 1 << bit
            conn.send_bytes(bytes([len(res), mask]))
    finally:
        close = getattr(env, "close", None)
        if close:
            close()
        for s in segs.values():
            s.close()


class ProcessVecEnv:
    """``ProcessVecEnv(("cleanrl_amd.envs", "SyntheticAtariVecEnv", dict(num_envs=256, seed=1, api="gym")))``."""

    def __init__(self, spec, obs_shape=(4, 84, 84), obs_dtype=np.uint8, act_shape=(), act_dtype=np.int64, start_method: str = "spawn",
                 single_observation_space=None, single_action_space=None):
        mod, attr, kwargs = spec
        self.num_envs = n = int(kwargs["num_envs"])
        self._segs = {
            "obs": shared_memory.SharedMemory(create=True, size=max(n * int(np.prod(obs_shape)) * np.dtype(obs_dtype).itemsize, 8)),
            "reward": shared_memory.SharedMemory(create=True, size=n * 8),
            "done": shared_memory.SharedMemory(create=True, size=max(2 * n, 8)),
            "act": shared_memory.SharedMemory(create=True, size=max(n * int(np.prod(act_shape, dtype=np.int64)) * np.dtype(act_dtype).itemsize, 8)),
            "info": shared_memory.SharedMemory(create=True, size=len(_INFO_KEYS) * n * 8),
        }
        if len(obs_shape) == 3 and obs_shape[0] > 1:          # (frames, H, W) stacks: the newest plane of every env, contiguous
            self._segs["newest"] = shared_memory.SharedMemory(create=True, size=n * int(np.prod(obs_shape[1:])) * np.dtype(obs_dtype).itemsize)
        names = {k: s.name for k, s in self._segs.items()}
        self._v = _views(self._segs, n, obs_shape, obs_dtype, act_shape, act_dtype)
        ctx = mp.get_context(start_method)
        self._conn, child = ctx.Pipe()
        self._proc = ctx.Process(target=_worker, args=(spec, names, n, tuple(obs_shape), np.dtype(obs_dtype).str, tuple(act_shape),
                                                       np.dtype(act_dtype).str, child), daemon=True)
        self._proc.start()
        child.close()
        if single_observation_space is None or single_action_space is None:       # (the spaces of the env class, from a throw-away instance of one env)
            probe = getattr(importlib.import_module(mod), attr)(**{**kwargs, "num_envs": 1})
            single_observation_space = single_observation_space or probe.single_observation_space
            single_action_space = single_action_space or probe.single_action_space
            getattr(probe, "close", lambda: None)()
        self.single_observation_space, self.single_action_space = single_observation_space, single_action_space
        self.observation_space, self.action_space = single_observation_space, single_action_space
        self._closed = False
        self._registered = []

    def pin(self) -> bool:
        """Register the observation segments as pinned host memory (``hipHostRegister`` through torch's runtime binding) and expose
        them as uint8 tensors ``obs_t`` / ``newest_t``: H2D copies from them are asynchronous DMA.  False when the runtime refuses
        (the caller then stages through its own pinned buffers)."""
        import torch

        try:
            rt = torch.cuda.cudart()
            for key in ("obs", "newest"):
                if key not in self._segs:
                    continue
                t = torch.frombuffer(self._segs[key].buf, dtype=torch.uint8)
                if int(rt.cudaHostRegister(t.data_ptr(), t.numel(), 0)) != 0:
                    raise RuntimeError("cudaHostRegister")
                self._registered.append(t.data_ptr())
                setattr(self, key + "_t", t.view(self._v[key].shape))
            return all(getattr(self, k + "_t").is_pinned() for k in ("obs", "newest") if k in self._segs)
        except Exception:       # noqa: BLE001 -- any failure means: not pinned
            self.unpin()
            return False

    def unpin(self) -> None:
        if self._registered:
            import torch

            rt = torch.cuda.cudart()
            for ptr in self._registered:
                rt.cudaHostUnregister(ptr)
            self._registered = []
        for key in ("obs_t", "newest_t"):
            if hasattr(self, key):
                delattr(self, key)

    def poll(self) -> bool:
        """True when the result of the last ``step_async`` is ready (``step_wait`` will not block)."""
        return self._conn.poll(0)

    # ---- the vector-env surface the scripts and pipeline.py use
    def reset(self, seed: Optional[int] = None):
        assert seed is None, "seed the env through its constructor kwargs (the worker builds it)"
        self._conn.send_bytes(b"r")
        kind = self._conn.recv_bytes()
        return self._v["obs"] if kind == b"g" else (self._v["obs"], {})

    def step_async(self, actions) -> None:
        np.copyto(self._v["act"], np.asarray(actions).reshape(self._v["act"].shape), casting="unsafe")
        self._conn.send_bytes(b"s")

    def step_wait(self):
        k, mask = self._conn.recv_bytes()                    # blocks in C with the GIL released: the other lanes' threads run
        v = self._v
        info = {key: v["info_" + key] for bit, (key, _) in enumerate(_INFO_KEYS) if mask >> bit & 1}
        if k == 5:
            return v["obs"], v["reward"], v["done"][0], v["done"][1], info
        return v["obs"], v["reward"], v["done"][0], info

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        try:
            self._conn.send_bytes(b"q")
            self._proc.join(timeout=5)
        except (BrokenPipeError, OSError):
            pass
        if self._proc.is_alive():
            self._proc.terminate()
        self.unpin()
        self._v = None                                       # (the views pin the segments' buffers)
        for s in self._segs.values():
            s.close()
            try:
                s.unlink()
            except FileNotFoundError:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass
