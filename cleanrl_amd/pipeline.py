"""Overlapped host-env rollout: env stepping on the host, overlapped with the GPU and with PCIe.

The reference's rollout (ppo_atari_multigpu.py:256-272) is strictly serial per step: policy forward -> D2H of the actions
(sync) -> ``envs.step`` on the host -> H2D of the f32 frames.  Here the local envs are split into K *env groups* (K
independent vector envs of N/K envs each -- rows [g*N/K, (g+1)*N/K) of every rollout buffer).  Each group gets a LANE: a
host thread + its own HIP stream + pinned staging buffers, running the reference's serial step sequence for its own rows.
The lanes interleave by themselves: while lane A's thread is inside ``envs[A].step`` (numpy / envpool release the GIL), the
GPU runs lane B's policy forward and the copy engine moves lane B's frames -- no hand-written schedule, the same
trajectories as the serial loop (a group's envs only ever see their own actions, the policy is fixed during a rollout).

PCIe bytes: frames travel as uint8 through pinned memory, and for FrameStack(4) envs only the NEWEST 84x84 plane of every
env whose stack is the previous one shifted by a frame (``frame_delta``): the device rebuilds the stack from the previous
row (``obs_shift_append_u8``).  An env sends its full stack when it may have been reset -- its done flag is set at this step
(same-step autoreset: gym < 1.0 vector envs) OR was set at the previous one (next-step autoreset: envpool's gym API,
gymnasium >= 1.0, where the call AFTER done returns the fresh stack with done = False) -- and whenever a host-side probe
(one pixel row of each of the three carried planes, 252 bytes per env) finds that the new stack is not the old one shifted
(``full_stack_rows``).  7 KB per env and step instead of the reference's 113 KB of f32.

Sampling stays deterministic: the Philox offset of (step, group) is reserved up front (``_SampleCounter.reserve``), so the
action streams do not depend on thread timing.

Beyond two lanes the host threads used to fight over the GIL: a lane step's policy forward alone is three autograd nodes and a
dozen library / torch calls (4 lanes ran SLOWER than 2, profiles/r02_host_env_bench_groups.jsonl).  ``GroupedRollout.capture()``
records the policy forward + sampling + D2H of the actions of every (lane, step) as one hipGraph on the lane's stream (the
step's rollout rows are baked in; the Philox position lives in device memory), so that a lane step is one graph launch --
the same kernels on the same data in the same order, hence the same actions.
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

StepFn = Callable[[int, np.ndarray, int], tuple]      # (group, actions, step) -> (next_obs, reward, next_done) as numpy

PROBE_ROW = 41        # the pixel row of every plane the shift probe compares


def stack_probe(obs: np.ndarray) -> np.ndarray:
    """(n, 4, 84, 84) uint8 stacks -> (n, 4, 84) copy of one pixel row per plane."""
    return np.array(obs[:, :, PROBE_ROW, :])


def full_stack_rows(done, prev_done, probe, prev_probe) -> np.ndarray:
    """Indices of the envs that must send their whole 4-plane stack this step instead of the newest plane only: flagged done
    now (same-step autoreset), flagged done at the previous step (next-step autoreset: this call returned the fresh stack),
    or planes 0-2 of the new stack are not planes 1-3 of the previous one on the probe row (any other discontinuity)."""
    need = np.asarray(done).astype(bool) | np.asarray(prev_done).astype(bool)
    if prev_probe is not None:
        need = need | (probe[:, :3] != prev_probe[:, 1:]).any(axis=(1, 2))
    return np.flatnonzero(need)


class _Lane:
    """One env group's resources: rows [lo, hi) of the learner's buffers, a stream, pinned staging."""

    def __init__(self, learner, g: int, lo: int, hi: int, frame_delta: bool):
        L = self.L = learner
        self.g, self.lo, self.hi, self.n = g, lo, hi, hi - lo
        self.hip = L.hip
        self.delta = bool(frame_delta and L.hip and L.relayout and tuple(L.frame_shape) == (4, 84, 84))
        if not self.hip:
            return
        dev = L.device
        self.stream = torch.cuda.Stream(device=dev)
        self.evt = torch.cuda.Event()
        n = self.n
        obs_dtype = L.obs.dtype
        self.pin_obs = torch.zeros((n,) + tuple(L.frame_shape), dtype=obs_dtype).pin_memory()
        self.dev_obs = torch.zeros((n,) + tuple(L.frame_shape), dtype=obs_dtype, device=dev) if L.relayout else None
        self.pin_rd = torch.zeros((2, n), dtype=torch.float32).pin_memory()
        self.pin_rd_np = self.pin_rd.numpy()
        self.pin_obs_np = self.pin_obs.numpy()
        act_shape = (n,) + tuple(L.act_shape)
        self.pin_act = (torch.zeros(act_shape, dtype=torch.int64) if L.discrete else torch.zeros(act_shape)).pin_memory()
        self.pin_act_np = self.pin_act.numpy()
        if self.delta:
            self.pin_new = torch.zeros((n, 84, 84), dtype=torch.uint8).pin_memory()
            self.pin_new_np = self.pin_new.numpy()
            self.dev_new = torch.zeros((n, 84, 84), dtype=torch.uint8, device=dev)
            self.prev_done = np.zeros(n, bool)          # done flags of the previous step (next-step autoreset envs)
            self.prev_probe = None                      # stack_probe() of the previous observation

    def rows(self, t):
        return t[self.lo:self.hi]

    # ---- the reference's per-step sequence for this group's rows ----
    def act(self, step: int, rng_offset: Optional[int]) -> np.ndarray:
        L = self.L
        a = L.act(step, rows=(self.lo, self.hi), rng_offset=rng_offset)
        if not self.hip:
            return a.cpu().numpy()
        self.pin_act.copy_(a.view(self.pin_act.shape), non_blocking=True)     # D2H on the lane's stream (:269)
        self.evt.record(self.stream)
        self.evt.synchronize()                    # waits for THIS lane's work only; the other lanes keep the GPU busy
        return self.pin_act_np

    def act_captured(self, step: int) -> np.ndarray:
        """``act`` as one graph launch on the lane's stream (``GroupedRollout.capture``)."""
        self.graphs[step].replay()
        self.evt.record(self.stream)
        self.evt.synchronize()
        return self.pin_act_np

    def observe(self, step: int, obs, done, first: bool = False, pinned_env=None) -> None:
        """rows [lo, hi) of slot ``step`` <- this group's next observation / done flags.  ``pinned_env``: a ``ProcessVecEnv`` whose
        shared segments are registered as pinned host memory (``pin()``): frames then go to the GPU straight from what the worker
        wrote (``newest_t`` / ``obs_t``) instead of through this lane's staging buffers."""
        L = self.L
        obs_dst, done_dst = L._slot(step)
        obs_dst, done_dst = self.rows(obs_dst), self.rows(done_dst)
        if not self.hip:
            obs_dst.copy_(torch.as_tensor(np.asarray(obs), dtype=obs_dst.dtype))
            done_dst.copy_(torch.as_tensor(np.asarray(done), dtype=torch.float32))
            return
        np.copyto(self.pin_rd_np[0], done, casting="unsafe")
        if self.delta and first:
            self.prev_probe, self.prev_done = stack_probe(np.asarray(obs)), np.asarray(done).astype(bool)
        if self.delta and not first:
            prev_rows = self.rows(L._slot(step - 1)[0])
            obs = np.asarray(obs)
            probe = stack_probe(obs)
            reset = full_stack_rows(done, self.prev_done, probe, self.prev_probe)
            self.prev_probe, self.prev_done = probe, np.asarray(done).astype(bool)
            if pinned_env is not None:
                self.dev_new.copy_(pinned_env.newest_t, non_blocking=True)      # DMA from the worker's own (registered) buffer
            else:
                np.copyto(self.pin_new_np, obs[:, 3])                          # the newest plane of every env: 7 KB each
                for j, i in enumerate(reset):
                    np.copyto(self.pin_obs_np[j], obs[i])                      # reset envs: their whole (fresh) stack
                self.dev_new.copy_(self.pin_new, non_blocking=True)
            L.ops.obs_shift_append_u8(prev_rows, self.dev_new, obs_dst)
            if len(reset):
                k = len(reset)
                if pinned_env is not None:
                    for j, i in enumerate(reset):
                        self.dev_obs[j].copy_(pinned_env.obs_t[int(i)], non_blocking=True)
                else:
                    self.dev_obs[:k].copy_(self.pin_obs[:k], non_blocking=True)
                for j, i in enumerate(reset):
                    L.ops.obs_nchw_to_nhwc_u8(self.dev_obs[j:j + 1], obs_dst[int(i):int(i) + 1])
        else:
            np.copyto(self.pin_obs_np, obs, casting="unsafe")
            if L.relayout:
                self.dev_obs.copy_(self.pin_obs, non_blocking=True)
                L.ops.obs_nchw_to_nhwc_u8(self.dev_obs, obs_dst)
            else:
                obs_dst.copy_(self.pin_obs, non_blocking=True)
        done_dst.copy_(self.pin_rd[0], non_blocking=True)

    def store_reward(self, step: int, reward) -> None:
        dst = self.rows(self.L.rewards[step])
        if not self.hip:
            dst.copy_(torch.as_tensor(np.asarray(reward, dtype=np.float32)).view(-1))
            return
        np.copyto(self.pin_rd_np[1], np.asarray(reward).reshape(-1), casting="unsafe")
        dst.copy_(self.pin_rd[1], non_blocking=True)


class GroupedRollout:
    """``GroupedRollout(learner, K)``; ``first_observation(g, obs)`` once, then ``run(step_fn)`` per iteration."""

    def __init__(self, learner, groups: int, frame_delta: bool = True, threads: bool = True):
        N = learner.N
        assert groups >= 1 and N % groups == 0, f"num_envs={N} must be a multiple of the {groups} env groups"
        per = N // groups
        self.L, self.K, self.threads = learner, groups, threads and groups > 1
        self.lanes: List[_Lane] = [_Lane(learner, g, g * per, (g + 1) * per, frame_delta) for g in range(groups)]
        self._rng_base = None           # capture(): 1-element int64 device tensor, the first Philox offset of the current rollout

    def capture(self) -> None:
        """One hipGraph per (lane, step): the policy forward on the lane's rows of slot ``step``, sampling (Philox offset =
        ``step * K + g`` + the rollout's first offset, which lives in device memory), the stores of action / log-prob / value
        and the D2H of the actions into the lane's pinned buffer.  Call once, after ``first_observation`` of every group."""
        L, K, T = self.L, self.K, self.L.T
        assert L.hip and L.discrete, "capture() needs the HIP path and a Discrete action space"
        dev = L.device
        main = torch.cuda.current_stream(dev)
        L.warm_rollout_caches()
        self._rng_base = torch.zeros(1, dtype=torch.int64, device=dev)
        host_rng = L.agent.rng.offset
        pool = None
        for lane in self.lanes:
            lane.stream.wait_stream(main)

            def body(step, lane=lane):
                a = L.act(step, rows=(lane.lo, lane.hi), rng_offset=step * K + lane.g, rng_base=self._rng_base)
                lane.pin_act.copy_(a.view(lane.pin_act.shape), non_blocking=True)

            with torch.cuda.stream(lane.stream):
                body(0)                               # warm-up on the lane's stream: its trunk buffers and workspaces exist
            lane.stream.synchronize()
            lane.graphs = []
            for step in range(T):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=lane.stream):
                    body(step)
                pool = pool or g.pool()
                lane.graphs.append(g)
            main.wait_stream(lane.stream)
        L.agent.rng.offset = host_rng               # (the warm-up steps drew nothing from the host counter, but stay explicit)

    def first_observation(self, g: int, obs, done=None) -> None:
        lane = self.lanes[g]
        done = np.zeros(lane.n, np.float32) if done is None else done
        if lane.hip:
            main = torch.cuda.current_stream(self.L.device)
            # the lane's staging buffers and the learner's rollout buffers were allocated AND zero-filled on the main stream:
            # without this wait that fill can land after the lane's first copy (seen as a rare corrupted slot 0)
            lane.stream.wait_stream(main)
            with torch.cuda.stream(lane.stream):
                lane.observe(0, obs, done, first=True)
            main.wait_stream(lane.stream)
        else:
            lane.observe(0, obs, done, first=True)

    def _lane_loop(self, lane: _Lane, step_fn: StepFn, rng_first: Optional[int], errors: list) -> None:
        L, K, T = self.L, self.K, self.L.T
        try:
            if lane.hip:
                torch.cuda.set_device(L.device)
            ctx = torch.cuda.stream(lane.stream) if lane.hip else _null()
            with ctx:
                captured = lane.hip and getattr(lane, "graphs", None) is not None
                for step in range(T):
                    off = None if rng_first is None else rng_first + step * K + lane.g
                    actions = lane.act_captured(step) if captured else lane.act(step, off)
                    next_obs, reward, next_done = step_fn(lane.g, actions, step)
                    lane.store_reward(step, reward)
                    lane.observe(step + 1, next_obs, next_done)
        except BaseException as e:          # noqa: BLE001 -- re-raised on the caller's thread
            errors.append(e)

    def run_async(self, envs) -> None:
        """One rollout driven from ONE host thread: ``envs[g]`` are ``ProcessVecEnv``s (``step_async`` / ``poll`` / ``step_wait``:
        the envs step in their own processes) and the lanes' policy steps are captured (``capture()``).  The loop polls every
        lane -- actions ready on the GPU -> hand them to the lane's worker; worker done -> store its reward, DMA its frames, launch
        the lane's next captured step -- so no lane waits for another and no two host threads fight over the GIL.  Same
        trajectories as ``run`` (a group's envs only see their own actions; the Philox position of (step, group) is fixed)."""
        import time

        L, K, T = self.L, self.K, self.L.T
        assert L.hip and self._rng_base is not None, "run_async needs the HIP path and capture()"
        rng_first = L.agent.rng.reserve(T * K)
        L.warm_rollout_caches()
        main = torch.cuda.current_stream(L.device)
        self._rng_base.fill_(int(rng_first))
        pinned = [e if getattr(e, "newest_t", None) is not None else None for e in envs]
        step, waiting_env = [0] * K, [False] * K
        stats = self.async_stats = {"gpu_wait_s": 0.0, "env_wait_s": 0.0, "host_s": 0.0, "lane_steps": 0}   # per-phase latencies, summed over lane steps
        mark = [time.perf_counter()] * K
        for lane in self.lanes:
            lane.stream.wait_stream(main)
            with torch.cuda.stream(lane.stream):
                lane.graphs[0].replay()
                lane.evt.record(lane.stream)
        left = K
        while left:
            progressed = False
            for g, lane in enumerate(self.lanes):
                t = step[g]
                if t >= T:
                    continue
                if not waiting_env[g]:
                    if lane.evt.query():                  # this lane's actions are in its pinned buffer
                        now = time.perf_counter()
                        stats["gpu_wait_s"] += now - mark[g]
                        envs[g].step_async(lane.pin_act_np)
                        mark[g] = time.perf_counter()
                        stats["host_s"] += mark[g] - now
                        waiting_env[g] = progressed = True
                elif envs[g].poll():
                    now = time.perf_counter()
                    stats["env_wait_s"] += now - mark[g]
                    res = envs[g].step_wait()
                    with torch.cuda.stream(lane.stream):
                        lane.store_reward(t, res[1])
                        done = res[2] if len(res) == 4 else np.logical_or(res[2], res[3])      # gymnasium API: terminated | truncated
                        lane.observe(t + 1, res[0], done, pinned_env=pinned[g] if lane.delta else None)
                        step[g] = t + 1
                        if t + 1 < T:
                            lane.graphs[t + 1].replay()
                        lane.evt.record(lane.stream)
                    waiting_env[g] = False
                    progressed = True
                    mark[g] = time.perf_counter()
                    stats["host_s"] += mark[g] - now
                    stats["lane_steps"] += 1
                    if t + 1 >= T:
                        left -= 1
            if not progressed:
                time.sleep(0)                             # yield: nothing is ready yet
        for lane in self.lanes:
            main.wait_stream(lane.stream)

    def run(self, step_fn: StepFn) -> None:
        """One rollout of T steps for every group (the learner's ``finish_rollout`` is left to the caller)."""
        L = self.L
        rng_first = L.agent.rng.reserve(L.T * self.K) if L.hip else None
        if L.hip:
            L.warm_rollout_caches()
            main = torch.cuda.current_stream(L.device)
            if self._rng_base is not None:
                self._rng_base.fill_(int(rng_first))  # the captured sampler launches add (step * K + g) to this
            for lane in self.lanes:
                lane.stream.wait_stream(main)      # the lanes read the parameters / slot 0 the main stream wrote
        errors: list = []
        if self.threads:
            ts = [threading.Thread(target=self._lane_loop, args=(lane, step_fn, rng_first, errors), daemon=True) for lane in self.lanes]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        else:
            for lane in self.lanes:
                self._lane_loop(lane, step_fn, rng_first, errors)
        if errors:
            raise errors[0]
        if L.hip:
            for lane in self.lanes:
                main.wait_stream(lane.stream)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def split_env_groups(make_group: Callable[[int, int], object], num_envs: int, groups: int) -> Sequence[object]:
    """``make_group(g, n)`` -> the g-th vector env of n envs; ``groups`` must divide ``num_envs``."""
    assert num_envs % groups == 0, f"--num-envs {num_envs} must be a multiple of --env-groups {groups}"
    return [make_group(g, num_envs // groups) for g in range(groups)]
