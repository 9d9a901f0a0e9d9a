"""The PPO actor-learner: rollout storage -> GAE -> minibatch update.

One ``PPOLearner`` is one rank.  The drop-in scripts (``ppo.py`` ...) own the CLI, seeding, env
construction and logging -- the reference's script skeleton -- and call into this class at the
positions of the reference's inline blocks:

====================================  ==========================================================
reference (ppo_atari_multigpu.py)      here
====================================  ==========================================================
:235-240  storage tensors              ``__init__`` (observations kept as uint8 for image envs)
:258-259  obs[step]/dones[step] store  ``observe``
:262-266  action logic + stores        ``act``           (K5 convert, network, K2 sample kernel)
:271      rewards[step] store          ``store_reward``
:288-301  bootstrap + GAE              ``finish_rollout`` (K1 kernel)
:304-309  flatten                      views, no copy
:312-380  epochs x minibatches         ``update``         (K5 gather, network fwd, K3 fused loss
                                                            fwd+bwd, autograd through the network,
                                                            RCCL all-reduce, fused clip+Adam)
:382-384  explained variance           ``update`` (host numpy, as the reference)
====================================  ==========================================================

Device selection is explicit: a CUDA device runs the HIP kernels (and raises if the library is
missing); a CPU device (``--no-cuda``) runs ``host_ops`` with the reference's torch semantics.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim

from . import host_ops


_FUSED_ACT = True      # the rollout step's FC fold + heads + sampling as one kernel (the A/B switch MI355PPO_FUSED_ACT is gone since round 6)


class PendingMetrics:
    """Diagnostics of one ``PPOLearner.update_async`` call: ``result()`` waits for the copies (HIP path) and computes what the
    reference logs per iteration (ppo_atari_multigpu.py:382-397)."""

    def __init__(self, event, values, returns, scalars, host_last, num_updates):
        self._event, self._values, self._returns, self._scalars = event, values, returns, scalars
        self._host_last, self._k, self._out = host_last, num_updates, None
        self.after_sync = None      # world > 1 over peer memory: PeerAllReduce.check (raises if a wait on a peer gave up during this update)

    def result(self) -> dict:
        if self._out is None:
            if self._event is not None:
                self._event.synchronize()
            if self.after_sync is not None:
                self.after_sync()
            y_pred, y_true = self._values.numpy(), self._returns.numpy()      # :382-384
            var_y = np.var(y_true)
            explained_var = np.nan if var_y == 0 else 1 - np.var(y_true - y_pred) / var_y
            if self._scalars is not None:
                sc = self._scalars.numpy()
                last_np, clipfrac = sc[-1], float(np.mean(sc[:, 6]))
            else:
                last, clipfracs = self._host_last
                last_np, clipfrac = last.numpy(), float(np.mean(clipfracs))
            self._out = dict(loss=float(last_np[0]), policy_loss=float(last_np[1]), value_loss=float(last_np[2]),
                             entropy=float(last_np[3]), old_approx_kl=float(last_np[4]), approx_kl=float(last_np[5]),
                             clipfrac=clipfrac, explained_variance=float(explained_var), num_updates=self._k)
            self._values = self._returns = self._scalars = self._host_last = None
        return self._out


class _SlotGraphs:
    """One (epoch, minibatch) slot of a captured update (``PPOLearner.capture_update``).  World = 1: one hipGraph.  World > 1: the graphs
    between the slot's collectives, replayed with the gradient exchange of ``_minibatch_hip`` issued eagerly between them -- the early
    bucket (Linear(3136,512).weight, 95 % of the bytes) goes out behind the first graph and rides under the second (the conv layers'
    backward), the small rest follows, the last graph (clip + Adam) is enqueued behind the collectives' completion."""
    __slots__ = ("segs", "early")

    def __init__(self, segs, early):
        self.segs, self.early = segs, early

    def replay(self, learner) -> None:
        segs = self.segs
        if len(segs) == 1:                  # world = 1, or world > 1 over peer memory: the exchange is five launches INSIDE the graph
            segs[0].replay()
            return
        g, multi = learner.flat.grads, learner.world_size > 1
        segs[0].replay()
        if self.early is not None:
            off, n = self.early
            work = dist.all_reduce(g[off:off + n], op=dist.ReduceOp.SUM, async_op=True) if multi else None
            segs[1].replay()
            if multi:                                                     # the same three disjoint pieces as the eager path, in the same order
                if off > 0:
                    dist.all_reduce(g[:off], op=dist.ReduceOp.SUM)
                if off + n < g.numel():
                    dist.all_reduce(g[off + n:], op=dist.ReduceOp.SUM)
                work.wait()
        elif multi:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        segs[-1].replay()


def _over_rccl(world_size: int) -> bool:
    return world_size > 1 and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"


def update_graph_policy(world_size: int) -> str:
    """What a training loop does about ``capture_update`` -- ONE place, shared by ``runner.train`` and ``bench.py``:

    ``MI355PPO_UPDATE_GRAPHS`` = ``0``: never (eager launches); ``auto`` (the default) and ``1``: capture -- on one GPU and over gloo (the backend of the
    CPU / one-GPU test runs) as it is; over **nccl (RCCL)**, which no build round could run, only behind ``PPOLearner.self_check_update_graphs`` (one update
    from the graphs against one eager update from the same state: bit-identical or back to eager launches) and the all-ranks agreement.  What ``auto``
    and ``1`` differ in over RCCL is the ARRANGEMENT (``early_bucket_policy``): ``auto`` keeps the reference's -- one all-reduce of the flat gradient behind
    the whole backward (ppo_atari_multigpu.py:360-367), a slot = two graphs cut on the calling thread --, ``1`` adds the early bucket, whose cut falls in
    the middle of the backward on the autograd engine's thread.  Why not simply eager over RCCL: at config C the eager update is host-bound on these boxes
    (0.93 - 0.99 M against 1.31 M env-steps/s, profiles/r06_eager_vs_graphs_cfgC.txt).  Returns "off", "capture" or "capture+check"."""
    v = os.environ.get("MI355PPO_UPDATE_GRAPHS", "auto").strip().lower()
    if v in ("0", "off", "eager", "no"):
        return "off"
    return "capture+check" if _over_rccl(world_size) else "capture"


def early_bucket_policy(world_size: int) -> bool:
    """Does the gradient exchange start early -- Linear(3136, 512).weight's gradient (95 % of the bytes) all-reduced asynchronously from a callback in
    the middle of the backward, under the conv layers' backward (§6)?  Yes on gloo and with ``MI355PPO_UPDATE_GRAPHS=1``; over RCCL by default NO: the
    reference's arrangement (one all-reduce behind the backward) keeps every collective and every capture begin / end on the calling thread for the
    first runs on a multi-GPU node.  The exposed all-reduce of 6.75 MB is 0.1 - 0.25 ms per minibatch."""
    v = os.environ.get("MI355PPO_UPDATE_GRAPHS", "auto").strip().lower()
    return (not _over_rccl(world_size)) or v in ("1", "on", "yes")


def all_ranks_agree(ok: bool, device: torch.device) -> bool:
    """World > 1: True only if EVERY rank says so (one MIN all-reduce) -- ranks must take the same route through the update (graphs or eager
    launches), or a later leg issues different collectives on different ranks.  World = 1 / no process group: ``ok``."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


class PPOLearner:
    def __init__(self, agent: nn.Module, args, obs_space, act_space, num_envs: int, device: torch.device,
                 world_size: int = 1, sample_seed: int = 0):
        self.agent, self.args, self.device = agent, args, device
        self.T, self.N = int(args.num_steps), int(num_envs)
        self.world_size = world_size
        self.adam_eps = 1e-5                       # optim.Adam(..., eps=1e-5) of the PPO scripts (ppo.py:168); PPG uses 1e-8
        self.hip = device.type == "cuda"
        self.discrete = getattr(agent, "discrete", True)
        self.image = bool(getattr(agent, "obs_is_image", False))
        self.obs_shape = tuple(obs_space.shape)
        self.act_shape = tuple(act_space.shape)
        T, N = self.T, self.N
        if self.hip:
            from . import _lib, ops            # the HIP library is mandatory on a GPU: fail here, loudly
            from .flat import FlatParams

            _lib.load()
            _lib.require_device(device.index if device.index is not None else torch.cuda.current_device())   # gfx950 or a sentence why not
            self.ops = ops
            self.flat = FlatParams(agent)
            self.optimizer = None
            agent.rng.seed = int(sample_seed)
        else:
            self.ops = None
            self.flat = None
            self.optimizer = optim.Adam(agent.parameters(), lr=args.learning_rate, eps=1e-5)   # ppo.py:168
        # ALGO Logic: Storage setup (ppo.py:171-176).  Image observations are stored as uint8 on the GPU:
        # frames are integers 0..255, so this is exact and 4x smaller than the reference's f32 tensor.  They
        # are also stored pixel-interleaved ((H,W,C), "NHWC"): the gather+convert kernel then emits
        # channels-last f32 activations and the conv stack runs without any layout transposes.
        obs_dtype = torch.uint8 if (self.image and self.hip) else torch.float32
        self.nhwc = self.image and self.hip
        # NatureCNN convolutions on the f32-MFMA kernels (cnn.py); MI355PPO_CNN=miopen keeps torch's Conv2d (MIOpen)
        # behind the K5 gather+convert kernel -- same results within f32 round-off, used for A/B timing.
        self.fused_cnn = (self.nhwc and hasattr(agent, "heads_u8") and tuple(obs_space.shape) == (4, 84, 84)
                          and os.environ.get("MI355PPO_CNN", "mfma") != "miopen")
        self.frame_shape = self.obs_shape                                   # what the env delivers: (C,H,W) ...
        # ... unless the agent declares pixel-interleaved frames (procgen's "bhwc", ppo_procgen.py:147): then the frames
        # already are in the rollout buffer's row layout and no relayout runs
        self.hwc_frames = self.image and getattr(agent, "obs_layout", "chw") == "hwc"
        if self.nhwc:
            if not self.hwc_frames:
                c, h, w = self.obs_shape
                self.obs_shape = (h, w, c)                                  # how rollout rows are laid out
            os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
        self.relayout = self.nhwc and not self.hwc_frames
        self.partial_scale = self.image and hasattr(agent, "scale_frames_")
        self.obs = torch.zeros((T, N) + self.obs_shape, dtype=obs_dtype, device=device)
        self.actions = torch.zeros((T, N) + self.act_shape, device=device)
        self.logprobs = torch.zeros((T, N), device=device)
        self.rewards = torch.zeros((T, N), device=device)
        self.dones = torch.zeros((T, N), device=device)
        self.values = torch.zeros((T, N), device=device)
        self.advantages = torch.zeros((T, N), device=device)
        self.returns = torch.zeros((T, N), device=device)
        self.boot_obs = torch.zeros((N,) + self.obs_shape, dtype=obs_dtype, device=device)
        self.boot_done = torch.zeros(N, device=device)
        self.batch_size = T * N
        self.minibatch_size = self.batch_size // int(args.num_minibatches)
        if self.hip:
            self._h2d = torch.cuda.Stream(device=device)
            self._h2d_evt = torch.cuda.Event()
            self._stage_free = torch.cuda.Event()    # compute stream: the staging / rollout buffers may be overwritten by the copy stream
            self._pin_obs = torch.zeros((N,) + self.frame_shape, dtype=obs_dtype).pin_memory()
            # device staging for incoming channel-planar frames (H2D target / device-env output)
            self.stage_obs = torch.zeros((N,) + self.frame_shape, dtype=obs_dtype, device=device) if self.relayout else None
            self._pin_rd = torch.zeros((2, N), dtype=torch.float32).pin_memory()
            self._pin_obs_np, self._pin_rd_np = self._pin_obs.numpy(), self._pin_rd.numpy()   # views of the pinned buffers
            self._x_roll = torch.empty((N,) + self.obs_shape, device=device) if (self.image and not self.fused_cnn) else None
            self._x_mb = None
            # B % num_minibatches != 0 leaves a ragged tail minibatch (range(0, B, M), ppo.py:246): count it
            n_upd = int(args.update_epochs) * -(-self.batch_size // max(self.minibatch_size, 1))
            self._scalars = torch.zeros((n_upd, 7), device=device)
            # one row per epoch: the host runs far ahead of the GPU, so epoch e+1's permutation must not land in the
            # pinned buffer epoch e's (asynchronous) H2D copy is still reading.  Rows are rewritten only by the next
            # update() call, which starts after this call's closing D2H of the scalars (a stream sync).
            self._inds_dev = torch.empty((int(args.update_epochs), self.batch_size), dtype=torch.int64, device=device)
            self._inds_pin = torch.empty((int(args.update_epochs), self.batch_size), dtype=torch.int64).pin_memory()
            self._inds_ev = {}          # epoch -> event behind the last H2D copy out of that pinned row
            self._total_norm = torch.zeros(1, device=device)
            self._stage_free.record(torch.cuda.current_stream(device))       # after every buffer's zero fill
        # the reference's 64-64 tanh MLP agents (ppo.py, ppo_continuous_action.py, rpo_continuous_action.py) on the fused MLP kernels
        # (csrc/mlp.hip): rollout step = one launch, minibatch forward + loss + backward = two launches (MI355PPO_MLP=torch keeps
        # the networks on library GEMMs behind K2 / K3 for A/B timing)
        self.mlp = None
        if self.hip and not self.image and hasattr(agent, "mlp_nets") and type(self).forward_backward_hip is PPOLearner.forward_backward_hip:
            from .agents import fused_mlp_ptrs

            self.mlp = fused_mlp_ptrs(agent)            # after FlatParams: the parameters sit at their final addresses
            if self.mlp is None and os.environ.get("MI355PPO_MLP", "fused") != "torch":
                # said once, not silently: which kernel family runs is part of what a drop-in user measures (round-5 review, weak #11)
                import sys

                nets = agent.mlp_nets()
                print(f"cleanrl_amd: this MLP agent (observation width {nets[0][0].in_features}, {nets[0][-1].out_features} actor outputs) is outside the fused "
                      "MLP kernels' shapes (64-64 tanh, observation width <= 512, <= 20 outputs): its networks run on library GEMMs behind the HIP sampling / "
                      "loss kernels", file=sys.stderr, flush=True)
        self._pack = None           # (B, 8) packed behaviour rows (ops.batch_pack) of the current update, or None
        self._pack_buf = None       # their storage, allocated by the first update()
        self._mb_adv_md = None      # the current minibatch's (mean, std + 1e-8) row of ops.adv_stats, or None
        self._mb_slot = None        # (LossSlots, k): K3's scalar fold deferred to one launch per update (categorical family)
        self._update_graphs = None  # capture_update(): [epoch][minibatch] -> hipGraph of one minibatch's forward + loss + backward
        self._adv_md_buf = None
        self._loss_slots = None
        if self.hip and self.discrete and self.mlp is None and type(self).forward_backward_hip is PPOLearner.forward_backward_hip:
            self._loss_slots = self.ops.LossSlots(self._scalars.shape[0], device)
        # world > 1: the all-reduce of the largest parameter's gradient (NatureCNN: Linear(3136,512).weight, 95 % of the
        # bytes, complete right after the heads' and the FC layer's backward) is started from a post-accumulate hook and runs
        # on RCCL's stream while the three conv layers' backward -- two thirds of the backward pass -- still computes; the
        # small rest follows the backward (`_minibatch_hip`).  The reference all-reduces after the whole backward (:360-367).
        self._ar_early = None       # (offset, numel) of the early bucket in the flat gradient buffer
        self._ar_armed = False
        self._ar_work = None
        self._capture_cut = None    # capture_update() with world > 1: ends a slot's first graph / begins its second (see _SlotGraphs)
        # (MI355PPO_UPDATE_GRAPHS=cut: the bucket boundary -- and with it the cut of a captured slot -- also with world = 1, where the
        # collectives are skipped: the one-GPU test of the segmented capture against the single-graph slots)
        self._force_cut = os.environ.get("MI355PPO_UPDATE_GRAPHS", "auto").strip().lower() == "cut"
        # MI355PPO_ALLREDUCE=peer: the flat gradient is exchanged through HIP IPC segments by five small launches on the compute stream
        # (dp_comm.PeerAllReduce, csrc/dpcomm.hip) instead of the process group's all-reduce: no early bucket, no cut -- a captured slot is one graph
        self._peer = None
        if self.hip and world_size > 1 and device.type == "cuda":
            from .dp_comm import PeerAllReduce, exchange_policy

            if exchange_policy(world_size) == "peer":
                self._peer = PeerAllReduce(self.flat.numel, device)
        if (self.hip and (world_size > 1 or self._force_cut) and type(self).forward_backward_hip is PPOLearner.forward_backward_hip
                and self._peer is None and early_bucket_policy(world_size)):      # (over RCCL by default: the reference's single all-reduce behind the backward)
            i = max(range(len(self.flat.segments)), key=lambda j: self.flat.segments[j][1])
            off, n = self.flat.segments[i]
            if n >= (1 << 18) and n * 2 > self.flat.numel:       # worth a launch of its own
                self._ar_early = (off, n)
                self.flat._params_list[i].register_post_accumulate_grad_hook(self._early_all_reduce)

    # ------------------------------------------------------------------ rollout (a2)
    def _slot(self, step: int):
        return (self.obs[step], self.dones[step]) if step < self.T else (self.boot_obs, self.boot_done)

    def observe(self, step: int, next_obs, next_done) -> None:
        """``obs[step] = next_obs; dones[step] = next_done`` (:258-259).  ``step == T`` addresses the
        bootstrap slot (the observation after the last step).  Host arrays are staged through pinned
        memory and copied on a side stream as uint8 -- 4x fewer PCIe bytes than the reference's
        ``torch.Tensor(next_obs).to(device)`` of f32 (:272).  Image frames arrive channel-planar (C,H,W)
        and are re-laid out to the buffer's (H,W,C) rows by the uint8 relayout kernel."""
        obs_dst, done_dst = self._slot(step)
        if isinstance(next_obs, torch.Tensor):
            if self.relayout:
                self.ops.obs_nchw_to_nhwc_u8(next_obs, obs_dst)
            elif next_obs.data_ptr() != obs_dst.data_ptr():
                obs_dst.copy_(next_obs)
            if next_done.data_ptr() != done_dst.data_ptr():
                done_dst.copy_(next_done)
            return
        if not self.hip:
            obs_dst.copy_(torch.as_tensor(np.asarray(next_obs), dtype=obs_dst.dtype))
            done_dst.copy_(torch.as_tensor(np.asarray(next_done), dtype=torch.float32))
            return
        self._h2d_evt.synchronize()                    # the pinned staging buffers are free again
        np.copyto(self._pin_obs_np, next_obs, casting="unsafe")      # one host memcpy straight into pinned memory
        np.copyto(self._pin_rd_np[0], next_done, casting="unsafe")
        h2d_dst = self.stage_obs if self.relayout else obs_dst
        # the copy stream must not overtake the compute stream's last use of its target: the zero fill at construction, the
        # relayout kernel of the previous step
        self._h2d.wait_event(self._stage_free)
        with torch.cuda.stream(self._h2d):
            h2d_dst.copy_(self._pin_obs, non_blocking=True)
            done_dst.copy_(self._pin_rd[0], non_blocking=True)
            self._h2d_evt.record(self._h2d)
        torch.cuda.current_stream(self.device).wait_event(self._h2d_evt)
        if self.relayout:
            self.ops.obs_nchw_to_nhwc_u8(self.stage_obs, obs_dst)
        self._stage_free.record(torch.cuda.current_stream(self.device))

    def start_iteration(self) -> None:
        """Carry the bootstrap observation of the previous rollout into slot 0 (the reference keeps it in
        ``next_obs`` across iterations, :245-247,272)."""
        self.obs[0].copy_(self.boot_obs)
        self.dones[0].copy_(self.boot_done)

    def _heads_rollout(self, obs_rows):
        """Policy/value heads on one slot of the rollout buffer (HIP path)."""
        if self.image and self.fused_cnn:
            if self.agent._trunk is None:
                from . import cnn

                self.agent._trunk = cnn.NatureTrunk()
            self._own_trunk_buffers()
            return self.agent.heads_u8(obs_rows)
        return self.agent.heads(self._features(obs_rows))

    def _own_trunk_buffers(self) -> None:
        """This learner owns the trunk's buffers: it bumps ``weights_version`` after every optimiser step (cached packs), and its
        flat gradient buffer is zeroed by the optimizer kernel before every backward, so the backward nodes may write parameter
        gradients straight into the ``.grad`` views (``direct_grads``)."""
        bufs = self.agent._trunk.bufs
        bufs.cache_weights = True
        if not getattr(bufs, "_owned", False):
            bufs._owned = True
            # (the nodes OVERWRITE .grad instead of accumulating: only the plain learner's update -- one backward per zeroed gradient buffer -- may
            #  ask for that; a subclass that overrides the forward / backward, e.g. an auxiliary phase with a second backward, keeps autograd's adds)
            bufs.direct_grads = type(self).forward_backward_hip is PPOLearner.forward_backward_hip
            if self._ar_early is not None:              # world > 1: the early bucket's all-reduce starts when its gradient is final
                bufs.after_fc_wgrad = self._early_all_reduce

    def warm_rollout_caches(self) -> None:
        """Re-derive the cached forward matrices on the current stream, so that the env-group lanes of a rollout
        (pipeline.py: several host threads, one stream each) only ever READ the cache."""
        if not (self.hip and self.image and self.fused_cnn):
            return
        from . import cnn

        if self.agent._trunk is None:
            self.agent._trunk = cnn.NatureTrunk()
        bufs, net = self.agent._trunk.bufs, self.agent.network
        self._own_trunk_buffers()
        bufs.weights(net[0].weight, 1, cnn.MODE_FWD_Q)
        bufs.weights(net[2].weight, 2, cnn.MODE_FWD)
        bufs.weights(net[4].weight, 3, cnn.MODE_FWD)
        cnn.warm_forward_packs(bufs, net)                # kernel Z's packs: what NatureTrunkFn / LinearReLUHwcFn.forward ask for

    def _features(self, obs_rows):
        """uint8 image rows -> normalised f32 (K5, no gather); other observations pass through."""
        if self.image and self.hip:
            # (N,H,W,C) f32 storage viewed as a channels-last (N,C,H,W) tensor
            if self.partial_scale:      # only the agent's pixel channels are divided by 255 (ppo_pettingzoo_ma_atari.py:104)
                return self.agent.scale_frames_(self.ops.obs_u8_to_f32(obs_rows, None, self._x_roll, False)).permute(0, 3, 1, 2)
            return self.ops.obs_u8_to_f32(obs_rows, None, self._x_roll).permute(0, 3, 1, 2)
        if self.image:
            return obs_rows / 255.0
        return obs_rows

    @torch.no_grad()
    def act(self, step: int, rows=None, rng_offset=None, rng_base=None):
        """Action logic (:262-266): network forward on ``obs[step]``, sample, store action / logprob / value.
        ``rows=(lo, hi)`` restricts the call to that slice of the envs (an env-group lane of pipeline.py, on the lane's
        stream); ``rng_offset`` then is the lane's reserved Philox offset for this step.  ``rng_base`` (1-element int64 device
        tensor) is added to the offset on the device: the form a captured step uses (``capture_rollout``)."""
        lo, hi = (0, self.N) if rows is None else rows
        if self.hip and self.mlp is not None:
            # both networks' forward + sampling + log_prob in ONE launch, written straight into the rollout storage
            seed, off = self.agent.rng.next() if rng_offset is None else (self.agent.rng.seed, int(rng_offset))
            obs_rows = self.obs[step][lo:hi]
            if self.discrete:
                a64 = self.ops.mlp_act_categorical(obs_rows, *self.mlp, seed=seed, offset=off, offset_base=rng_base,
                                                   action_f32_out=self.actions[step][lo:hi], logprob_out=self.logprobs[step][lo:hi],
                                                   value_out=self.values[step][lo:hi], want_i64=rng_base is None)[0]
                return a64 if a64 is not None else self.actions[step][lo:hi]
            return self.ops.mlp_act_normal(obs_rows, *self.mlp, self.agent.actor_logstd.detach(), seed=seed, offset=off,
                                           offset_base=rng_base, action_out=self.actions[step][lo:hi],
                                           logprob_out=self.logprobs[step][lo:hi], value_out=self.values[step][lo:hi])[0]
        if self.hip and self.discrete and self.image and self.fused_cnn and hasattr(self.agent, "act_u8") and _FUSED_ACT:
            # trunk, then Linear(3136,512) + heads + Categorical draw in two launches; action / log-prob / value written in place
            if self.agent._trunk is None:
                from . import cnn

                self.agent._trunk = cnn.NatureTrunk()
            self._own_trunk_buffers()
            seed, off = (self.agent.rng.seed, self.agent.rng.offset + 1) if rng_offset is None else (self.agent.rng.seed, int(rng_offset))
            r = self.agent.act_u8(self.obs[step][lo:hi], seed, off, rng_base, self.actions[step][lo:hi], self.logprobs[step][lo:hi],
                                  self.values[step][lo:hi], want_i64=rng_base is None)
            if r is not None:
                if rng_offset is None:
                    self.agent.rng.next()
                return r[0] if r[0] is not None else r[1]
        if self.hip:
            p, value = self._heads_rollout(self.obs[step][lo:hi])
            seed, off = self.agent.rng.next() if rng_offset is None else (self.agent.rng.seed, int(rng_offset))
            if self.discrete:
                a64, _, _, _ = self.ops.categorical_sample(p.contiguous(), seed=seed, offset=off, offset_base=rng_base,
                                                            action_f32_out=self.actions[step][lo:hi],
                                                            logprob_out=self.logprobs[step][lo:hi], want_entropy=False)
                action = a64
            else:
                action, _, _ = self.ops.normal_sample(p.contiguous(), self.agent.actor_logstd, seed=seed, offset=off,
                                                      action_out=self.actions[step][lo:hi], logprob_out=self.logprobs[step][lo:hi])
            self.values[step][lo:hi].copy_(value.view(-1))
            return action
        action, logprob, _, value = self.agent.get_action_and_value(self.obs[step][lo:hi])
        self.values[step][lo:hi] = value.flatten()
        self.actions[step][lo:hi] = action
        self.logprobs[step][lo:hi] = logprob
        return action

    # ------------------------------------------------------------------ hipGraph of the rollout steps (device-resident envs)
    def capture_rollout(self, env, steps_per_graph: int = 1) -> None:
        """Capture every rollout step -- policy forward, sampling, the device env's step, the store of the next observation --
        into one hipGraph per step (the step's rollout-storage rows are baked into its launches; the Philox positions of
        the sampler and of the env live in device memory and advance by T per rollout).  ``replay_rollout()`` then issues a
        whole rollout as T graph launches instead of ~30 T kernel launches: bit-identical buffers (tests), no host work
        beside the replays.  For envs that write straight into device rows (``step_into``)."""
        vector = self.mlp is not None                     # the MLP agents: vector observations, the env takes the action
        assert self.hip and hasattr(env, "step_into") and (vector or self.discrete), \
            "capture_rollout needs the HIP path and a device-resident env (NatureCNN agent, or an MLP agent on the fused kernels)"
        dev, T = self.device, self.T
        self.warm_rollout_caches()
        self._rng_base = torch.full((1,), int(self.agent.rng.offset), dtype=torch.int64, device=dev)
        env.step_base = torch.full((1,), int(env._step), dtype=torch.int64, device=dev)
        self._graph_env = env
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graphs, pool = [], None

        def body(step):
            action = self.act(step, rng_offset=step + 1, rng_base=self._rng_base)
            obs_dst, done_dst = self._slot(step + 1)
            if vector:                                    # the env writes the next observation straight into its rollout row
                env._step_rel = step
                env.step_into(action, obs_dst, self.rewards[step], done_dst)
                env._step_rel = None
                return
            env._step_rel = step + 1
            if self.relayout and hasattr(env, "step_into_rows"):      # the env writes the rollout row's own layout
                env.step_into_rows(obs_dst, self.rewards[step], done_dst)
                env._step_rel = None
                return
            frames = env.step_into(self.stage_obs, self.rewards[step], done_dst)
            env._step_rel = None
            self.observe(step + 1, frames, done_dst)

        host_rng, host_env = self.agent.rng.offset, env._step
        cursor0 = (torch.cat([env.state.reshape(-1), env.steps]) if vector else env.cursor).clone()
        with torch.cuda.stream(side):
            body(0)                                   # warm-up on the capture stream: per-stream workspaces and trunk buffers exist
        side.synchronize()
        # `steps_per_graph` consecutive steps per graph (1: T graphs; T: the whole rollout is one graph -- a replay boundary costs
        # ~8 us of GPU idle against 1-2 us between the kernels inside a graph, profiles/r03_cold_second_process_gaps.txt)
        per = max(1, min(int(steps_per_graph), T))
        # world > 1: the process group's watchdog thread may query events while this thread captures; only actions of THIS thread
        # may invalidate the capture then (torch's default mode is "global")
        capture_kw = {"capture_error_mode": "thread_local"} if self.world_size > 1 else {}
        for first in range(0, T, per):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=side, **capture_kw):
                for step in range(first, min(first + per, T)):
                    body(step)
            pool = pool or g.pool()
            graphs.append(g)
        torch.cuda.current_stream(dev).wait_stream(side)
        # capturing executes nothing, but the warm-up step and the host counters moved: restore the pre-capture state
        if vector:
            env.state.copy_(cursor0[:env.state.numel()].view_as(env.state))
            env.steps.copy_(cursor0[env.state.numel():])
        else:
            env.cursor.copy_(cursor0)
        self.agent.rng.offset, env._step = host_rng, host_env
        self._rollout_graphs = graphs

    def replay_rollout(self) -> None:
        """One rollout of T captured steps (then ``finish_rollout()`` as usual)."""
        T, env = self.T, self._graph_env
        self.warm_rollout_caches()               # the repacked matrices the captured launches read: re-derived in place
        for g in self._rollout_graphs:
            g.replay()
        self._rng_base.add_(T)
        env.step_base.add_(T)
        self.agent.rng.offset += T               # host mirrors: eager calls and captured steps stay interchangeable
        env._step += T

    def store_reward(self, step: int, reward) -> None:
        """``rewards[step] = torch.tensor(reward).to(device).view(-1)`` (:271)."""
        if isinstance(reward, torch.Tensor):
            if reward.data_ptr() != self.rewards[step].data_ptr():
                self.rewards[step].copy_(reward.view(-1))
        else:
            self.rewards[step].copy_(torch.as_tensor(np.asarray(reward, dtype=np.float32)).view(-1),
                                     non_blocking=self.hip)

    # ------------------------------------------------------------------ GAE (a4)
    @torch.no_grad()
    def finish_rollout(self) -> None:
        """Bootstrap value of the observation in the bootstrap slot, then GAE (:288-301)."""
        a = self.args
        if self.hip:
            if self.mlp is not None:
                next_value = self.ops.mlp_forward(self.boot_obs, *self.mlp)[1]
            else:
                _, next_value = self._heads_rollout(self.boot_obs)
            self.ops.gae(self.rewards, self.dones, self.values, self.boot_done, next_value.reshape(-1).contiguous(),
                         a.gamma, a.gae_lambda, self.advantages, self.returns)
        else:
            x = self.boot_obs
            next_value = self.agent.get_value(x).reshape(1, -1)
            adv, ret = host_ops.gae(self.rewards, self.dones, self.values, self.boot_done, next_value, a.gamma,
                                    a.gae_lambda)
            self.advantages.copy_(adv)
            self.returns.copy_(ret)

    # ------------------------------------------------------------------ update (a5-a9)
    def update(self, lr: float) -> dict:
        """Epochs x minibatches of the clipped-surrogate update (:311-380) + explained variance (:382-384).
        Returns the scalars the reference logs (values of the LAST minibatch, clipfrac averaged)."""
        return self.update_async(lr).result()

    def update_async(self, lr: float) -> "PendingMetrics":
        """``update`` without its device synchronisation: everything is enqueued, the diagnostics (:382-397 -- two (T, N) arrays
        for the explained variance and the (updates, 7) loss scalars) are copied to pinned host memory behind an event, and
        ``.result()`` of the returned handle waits for that event only.  A caller that resolves the handle one iteration late
        lets the host run a whole iteration ahead of the GPU: a host stall of tens of milliseconds (a descheduled thread, a
        page-in, a garbage collection) then no longer drains the GPU's queue -- measured on a fresh box, where the first
        process loses 18-32 ms of GPU time per iteration to such stalls (profiles/r03_cold_first_process_gaps.txt).  The
        training arithmetic and its order are the same; only the moment the host READS the diagnostics moves.  (CPU path and
        ``--target-kl``: resolved at once -- the early stop needs the value.)"""
        a = self.args
        B, M = self.batch_size, self.minibatch_size
        if self.hip and self.image and self.fused_cnn:
            if self.agent._trunk is None:
                from . import cnn

                self.agent._trunk = cnn.NatureTrunk()
            self._own_trunk_buffers()
        b_inds = np.arange(B)                                             # :312
        b_obs = self.obs.reshape((-1,) + self.obs_shape)                  # :304-309, views
        b_actions = self.actions.reshape((-1,) + self.act_shape)
        b_logprobs, b_advantages = self.logprobs.reshape(-1), self.advantages.reshape(-1)
        b_returns, b_values = self.returns.reshape(-1), self.values.reshape(-1)
        # K3 gathers ONE 32-byte row per minibatch row instead of five 4-byte values out of five arrays: the five behaviour
        # arrays are final here (GAE done; tests may have overwritten them), packed once per iteration
        use_pack = (self.hip and self.discrete and self.mlp is None
                    and type(self).forward_backward_hip is PPOLearner.forward_backward_hip)
        if use_pack:
            if self._pack_buf is None:
                self._pack_buf = torch.empty((B, self.ops.PACK_FLOATS), dtype=torch.float32, device=self.device)
            self._pack = self.ops.batch_pack(b_actions, b_logprobs, b_advantages, b_returns, b_values, out=self._pack_buf)
        k = 0
        folded = 0                       # scalar rows already folded out of the loss slots
        stop = False
        clipfracs = []
        last = None
        replayed = self.hip and self._update_graphs is not None and a.target_kl is None
        if replayed:
            k = self._replay_update(lr, b_inds, b_advantages, use_pack)
        for epoch in range(0 if replayed else int(a.update_epochs)):
            np.random.shuffle(b_inds)                                     # :315 host MT19937, per-rank seed
            if self.hip:
                inds_dev = self.upload_permutation(epoch, b_inds)
                # :337-338 hoisted: (mean, std + 1e-8) of every minibatch of this epoch depend only on the permutation and
                # the GAE output -> one launch per epoch, and K3 runs without its statistics launch
                if not a.norm_adv:
                    adv_md = None
                elif use_pack:
                    adv_md = self.ops.adv_stats_packed(self._pack, inds_dev, M,
                                                       out=self._adv_md_buf[epoch] if self._update_graphs is not None else None)
                else:
                    adv_md = self.ops.adv_stats(b_advantages, inds_dev, M,
                                                out=self._adv_md_buf[epoch] if self._update_graphs is not None else None)
            for start in range(0, B, M):
                end = start + M
                if self.hip and self._update_graphs is not None:
                    if k == 0:
                        self._upload_adam_schedule(lr)                    # every slot's (step size, bias correction) -> device memory
                    # forward + fused loss + backward + clip + Adam of this slot (capture_update)
                    self._update_graphs[epoch][start // M].replay(self)
                    self.flat.step += 1
                elif self.hip:
                    self._mb_adv_md = adv_md[start // M] if adv_md is not None else None
                    self._mb_slot = (self._loss_slots, k) if self._loss_slots is not None else None
                    self._minibatch_hip(inds_dev[start:end], b_obs, b_actions, b_logprobs, b_advantages, b_returns,
                                        b_values, lr, self._scalars[k])
                else:
                    last = self._minibatch_host(b_inds[start:end], b_obs, b_actions, b_logprobs, b_advantages, b_returns,
                                                b_values, lr)
                    clipfracs.append(last[6].item())                      # :328
                k += 1
            if a.target_kl is not None:                                   # :379-380 (local approx_kl, as the reference)
                if self._loss_slots is not None and self.hip:
                    self._loss_slots.fold(k - folded, self._scalars, first=folded)
                    folded = k
                approx_kl = (self._scalars[k - 1, 5] if self.hip else last[5]).item()
                if approx_kl > a.target_kl:
                    stop = True
            if stop:
                break
        self._mb_adv_md = self._mb_slot = self._pack = None               # direct forward_backward_hip calls: arrays, fold at once
        if self.hip and self._update_graphs is not None:
            trunk = getattr(self.agent, "_trunk", None)
            if trunk is not None:                   # the replayed Adam kernels rewrote the parameters: the rollout's packs are stale
                trunk.bufs.weights_version += 1
        if self.hip:
            if self._loss_slots is not None and k > folded:
                self._loss_slots.fold(k - folded, self._scalars, first=folded)   # one launch for every minibatch of the update
            # :382-384 and the logged scalars: async D2H into pinned memory (the caching host allocator keeps a block out of
            # reuse until the copy that used it has executed), one event behind them
            on_gpu = b_values.is_cuda              # (the HIP branches also run on CPU tensors under the tests' stand-in ops)
            host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=on_gpu) for t in (b_values, b_returns, self._scalars[:k])]
            for h, t in zip(host, (b_values, b_returns, self._scalars[:k])):
                h.copy_(t, non_blocking=True)
            ev = None
            if on_gpu:
                ev = torch.cuda.Event()
                ev.record()
            pm = PendingMetrics(ev, host[0], host[1], host[2], None, k)
            if self._peer is not None:
                pm.after_sync = self._peer.check
            return pm
        pm = PendingMetrics(None, b_values, b_returns, None, (last, clipfracs), k)
        pm.result()                     # CPU path: resolved at once (the handle holds views of buffers the next rollout overwrites)
        return pm

    def capture_update_agreed(self, log=None) -> bool:
        """``capture_update`` under ``update_graph_policy``: captures where the policy says so, makes ALL ranks agree on the outcome (a rank
        whose capture -- or self-check -- failed takes every rank back to the eager update), and returns whether the update now replays
        graphs.  ``log``: a callable for one line of diagnosis (default: stderr)."""
        import sys

        say = log or (lambda m: print(m, file=sys.stderr, flush=True))
        policy = update_graph_policy(self.world_size)
        ok = policy != "off"
        if ok:
            try:
                self.capture_update()
            except Exception as exc:      # noqa: BLE001 -- (capture_update restored the parameters, the Adam state and the zeroed gradients)
                say(f"update graphs: capture failed ({type(exc).__name__}: {str(exc).splitlines()[0][:200] if str(exc) else ''}); eager launches")
                ok = False
        # agree on the CAPTURE first: the self-check below issues the update's collectives, so either every rank runs it or none does
        agreed = all_ranks_agree(ok, self.device)
        if ok and not agreed:
            say("update graphs: another rank's capture failed; this rank follows it back to eager launches")
        if agreed and policy == "capture+check":
            try:
                ok = self.self_check_update_graphs()
            except Exception as exc:      # noqa: BLE001
                say(f"update graphs: the self-check raised ({type(exc).__name__}: {str(exc).splitlines()[0][:200] if str(exc) else ''}); eager launches")
                ok = False
            if not ok:
                say("update graphs: the captured update did not reproduce the eager update's parameters on this rank; eager launches")
            agreed = all_ranks_agree(ok, self.device)
            if ok and not agreed:
                say("update graphs: another rank's self-check failed; this rank follows it back to eager launches")
        if not agreed:
            self._update_graphs = None
        return agreed

    def _bump_weights_version(self) -> None:
        """The parameters changed behind the trunk's cached weight packs (raw-pointer writes into the flat buffer)."""
        trunk = getattr(self.agent, "_trunk", None)
        if trunk is not None:
            trunk.bufs.weights_version += 1

    def self_check_update_graphs(self) -> bool:
        """One update from the captured graphs against one eager update from the SAME state and the same permutations (whatever the rollout
        buffers hold -- zeros at start-up: still every kernel and every collective of the update): True iff parameters and Adam moments come
        out bit-identical.  State, gradients, step counter and numpy's stream are restored.  Every rank must call it (it issues the
        update's collectives twice)."""
        assert self._update_graphs is not None, "self_check_update_graphs: capture_update first"
        f = self.flat
        keep = [t.clone() for t in (f.params, f.exp_avg, f.exp_avg_sq, f.grads)]
        step0, rng0, graphs = getattr(f, "step", None), np.random.get_state(), self._update_graphs
        lr = float(self.args.learning_rate)

        def run(use_graphs):
            for t, k in zip((f.params, f.exp_avg, f.exp_avg_sq, f.grads), keep):
                t.copy_(k)
            if step0 is not None:
                f.step = step0
            self._bump_weights_version()
            np.random.set_state(rng0)
            self._update_graphs = graphs if use_graphs else None
            self.update(lr)
            torch.cuda.synchronize(self.device)
            return [t.clone() for t in (f.params, f.exp_avg, f.exp_avg_sq)]

        try:
            a, b = run(True), run(False)
            same = all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(a, b))
        finally:
            for t, k in zip((f.params, f.exp_avg, f.exp_avg_sq, f.grads), keep):
                t.copy_(k)
            if step0 is not None:
                f.step = step0
            self._bump_weights_version()
            np.random.set_state(rng0)
            self._update_graphs = graphs
        return bool(same)

    def capture_update(self) -> None:
        """``bench.py``'s and ``runner.train``'s default (``--no-update-graphs`` / MI355PPO_UPDATE_GRAPHS=0: eager).  World > 1 (round 5): a
        slot is the two or three graphs BETWEEN its collectives, replayed with the eager path's all-reduces issued between them
        (``_SlotGraphs``, ``_capture_slot_segments``; two ranks on one GPU over gloo bit-identical to the eager two-rank update and at the
        reference's distance on config D's golden: tests/test_gpu_multirank.py).  Bit-identical to the eager update over three
        iterations on the MI355X at a small shape, at config B's and on the continuous-action path
        (tests/test_gpu_learner.py::test_captured_update_slots_...); measured in profiles/r04_update_graphs_ab.jsonl.

        One hipGraph per (epoch, minibatch) slot of the update: K5 gather + network forward + fused loss + backward through the
        custom autograd nodes of that slot (:320-358) + the fused clip / Adam step (:376-377), ~55 launches, replayed by
        ``update`` / ``update_async``.  The optimizer's schedule-dependent constants come from a device table (see below).
        Everything a slot reads sits at a fixed address: its rows of the epoch's device permutation (``_inds_dev[epoch]``),
        the packed behaviour rows, the epoch's advantage statistics (``_adv_md_buf``), its loss slot and scalar row, the
        flat parameter / gradient buffers, the weight packs (re-derived in place by launches inside the graph: the capture
        bumps the weights' version before every slot, so the repack launches are part of every graph).  Why: the update then
        costs the host ~3 launches per minibatch instead of ~55, so a busy host no longer drains the GPU's queue in a training
        loop that synchronises every env step (host envs), where ``update_async`` cannot help."""
        assert self.hip and type(self).forward_backward_hip is PPOLearner.forward_backward_hip, "capture_update: the HIP path of the plain PPO learner"
        assert getattr(self.agent, "rpo_alpha", None) is None, "capture_update: RPO draws noise inside the update (not covered)"
        # (the Normal / continuous-action path runs the same code: tests/test_gpu_learner.py::test_captured_update_slots_continuous_path)
        a, dev = self.args, self.device
        B, M = self.batch_size, self.minibatch_size
        assert B % M == 0, "capture_update needs whole minibatches"
        E_, nmb = int(a.update_epochs), B // M
        b_obs = self.obs.reshape((-1,) + self.obs_shape)
        b_actions = self.actions.reshape((-1,) + self.act_shape)
        b_logprobs, b_advantages = self.logprobs.reshape(-1), self.advantages.reshape(-1)
        b_returns, b_values = self.returns.reshape(-1), self.values.reshape(-1)
        if self._pack_buf is None and self.discrete:
            self._pack_buf = torch.empty((B, self.ops.PACK_FLOATS), dtype=torch.float32, device=dev)
        # the warm-up EXECUTES a slot: give it valid operands (indices inside the batch, a non-zero denominator)
        self._inds_dev.copy_(torch.arange(B, dtype=torch.int64, device=dev).expand(E_, B))
        self._adv_md_buf = torch.zeros((E_, nmb, 2), dtype=torch.float32, device=dev)
        self._adv_md_buf[..., 1] = 1.0
        use_pack = self.discrete and self._loss_slots is not None and self.mlp is None
        self._pack = (self.ops.batch_pack(b_actions, b_logprobs, b_advantages, b_returns, b_values, out=self._pack_buf)
                      if use_pack else None)
        if self.image and self.fused_cnn:
            self.warm_rollout_caches()                                    # the trunk exists and caches its packs behind weights_version
        trunk = getattr(self.agent, "_trunk", None)

        def slot(e, j):
            k = e * nmb + j
            if trunk is not None:
                trunk.bufs.weights_version += 1                       # every graph re-derives the packs it reads
            self._mb_adv_md = self._adv_md_buf[e][j] if a.norm_adv else None
            self._mb_slot = (self._loss_slots, k) if self._loss_slots is not None else None
            self.forward_backward_hip(self._inds_dev[e][j * M:(j + 1) * M], b_obs, b_actions, b_logprobs, b_advantages,
                                      b_returns, b_values, self._scalars[k])

        # the optimizer step joins the graph: its two schedule-dependent constants (step size with the annealed learning rate and
        # bias correction 1, sqrt of bias correction 2) are read from row k of a device table that update() refills before the
        # first replay of an iteration (mi355ppo_clip_adam_sched_f32); everything else of the step is fixed at capture time
        n_slots = E_ * nmb
        self._adam_sched = torch.zeros((n_slots, 2), dtype=torch.float32, device=dev)
        self._adam_sched[:, 1] = 1.0                                      # (the warm-up slot executes: step size 0, a finite denominator)
        self._adam_sched_pin = torch.zeros((n_slots, 2), dtype=torch.float32).pin_memory()
        self._adam_sched_ev = None
        slot_fb = slot

        def slot_opt(e, j):
            self.ops.clip_adam_sched_(self.flat.params, self.flat.grads, self.flat.exp_avg, self.flat.exp_avg_sq,
                                      self._adam_sched[e * nmb + j], a.max_grad_norm, grad_scale=1.0 / self.world_size, eps=self.adam_eps,
                                      total_norm_out=self._total_norm)

        def slot(e, j):                                                   # noqa: F811 -- forward/backward, (the peer-memory exchange,) the optimizer step
            slot_fb(e, j)
            if self._peer is not None:
                self._peer.all_reduce_sum_(self.flat.grads)
            slot_opt(e, j)

        # world > 1 over the process group: a slot is two or three graphs with the gradient exchange between them (_SlotGraphs)
        segmented = (self.world_size > 1 and self._peer is None) or self._force_cut

        state0 = [t.clone() for t in (self.flat.params, self.flat.exp_avg, self.flat.exp_avg_sq)]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graphs, pool = [], None
        try:
            with torch.cuda.stream(side):
                slot(0, 0)                                                # warm-up on the capture stream (workspaces, trunk buffers)
                for t, t0 in zip((self.flat.params, self.flat.exp_avg, self.flat.exp_avg_sq), state0):
                    t.copy_(t0)                                           # the warm-up's optimizer step (zero step size) is undone exactly
                self.flat.grads.zero_()
            side.synchronize()
            for e in range(E_):
                row = []
                for j in range(nmb):
                    if segmented:
                        sg, pool = self._capture_slot_segments(side, pool, lambda e=e, j=j: slot_fb(e, j), lambda e=e, j=j: slot_opt(e, j))
                        row.append(sg)
                        continue
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool, stream=side):
                        slot(e, j)
                    pool = pool or g.pool()
                    row.append(_SlotGraphs([g], None))
                graphs.append(row)
        except Exception:
            # a capture that failed (an op the stream capture does not allow, e.g. inside a library GEMM of a head wider than the fused
            # kernel takes) must leave a learner that trains eagerly: nothing the warm-up or a partial capture touched may survive
            graphs = None
            torch.cuda.synchronize(dev)
            for t, t0 in zip((self.flat.params, self.flat.exp_avg, self.flat.exp_avg_sq), state0):
                t.copy_(t0)
            self.flat.grads.zero_()
            self._mb_adv_md = self._mb_slot = self._pack = None
            self._update_graphs = self._adam_sched = None
            self._capture_cut = None
            if trunk is not None:
                trunk.bufs.weights_version += 1
            raise
        torch.cuda.current_stream(dev).wait_stream(side)
        self.flat.grads.zero_()
        self._mb_adv_md = self._mb_slot = self._pack = None
        self._update_graphs = graphs

    def _capture_slot_segments(self, side, pool, forward_backward, optimizer_step):
        """One (epoch, minibatch) slot for world > 1 as the graphs BETWEEN its collectives (ppo_atari_multigpu.py:358-377: backward, all-reduce,
        clip, step): [forward ... the FC weight's gradient] | early bucket | [the conv layers' backward] | rest | [clip + Adam].  The first
        cut falls in the MIDDLE of the autograd backward -- at the callback the eager path starts the early bucket's all-reduce from
        (``_early_all_reduce``) --, i.e. on the autograd engine's device thread: the capture of the first graph ends and the second begins
        there, and ends on the calling thread after the backward returned.  Hence capture mode "relaxed": the other modes tie a capture to the
        thread that began it (and "global" would let the process group's watchdog thread invalidate it, as in ``capture_rollout``).  An agent
        without an early bucket gets two graphs (backward | whole buffer | step).  Returns (_SlotGraphs, pool)."""
        graphs = []

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(**({"pool": pool} if pool is not None else {}), capture_error_mode="relaxed")
            graphs.append(g)

        def end():
            nonlocal pool
            graphs[-1].capture_end()
            pool = pool or graphs[-1].pool()

        def cut():
            self._capture_cut = None          # once per slot (the eager path's `_ar_armed`: the callback and the parameter's hook may both fire)
            end()
            begin()

        torch.cuda.synchronize(self.device)
        with torch.cuda.stream(side):
            capturing = False
            try:
                begin()
                capturing = True
                self._capture_cut = cut if self._ar_early is not None else None
                forward_backward()
                self._capture_cut = None
                early = self._ar_early if len(graphs) == 2 else None      # (the cut fires only on the fused trunk's direct-gradient path)
                cut()
                optimizer_step()
                end()
                capturing = False
            finally:
                self._capture_cut = None
                if capturing:                                             # leave the stream out of capture mode whatever happened
                    try:
                        graphs[-1].capture_end()
                    except Exception:
                        pass
        return _SlotGraphs(graphs, early), pool

    def _replay_update(self, lr: float, b_inds: np.ndarray, b_advantages, use_pack: bool) -> int:
        """The whole update as graph replays (``capture_update``; no early stop): every epoch's host permutation is drawn first --
        the same ``np.random.shuffle`` calls in the same order as the epoch loop (:315) --, all of them travel in ONE pinned H2D
        copy, the advantage statistics of every (epoch, minibatch) come from ONE launch (the flattened permutations are
        epochs x minibatches consecutive segments), then the slots replay back to back.  Why not per epoch: a host-to-device
        copy call returns only once the stream has drained (measured: 0.56 ms on an idle stream, +1 ms behind 32 small graphs;
        tools/gpu/copy_probe.py), which serialised host and GPU at every epoch boundary of the small configurations."""
        a = self.args
        E_, B, M = int(a.update_epochs), self.batch_size, self.minibatch_size
        evs = self.__dict__.setdefault("_inds_ev", {})
        if evs.get("all") is not None:
            evs["all"].synchronize()                                      # the previous iteration's copy has read the pinned rows
        pin_np = self.__dict__.setdefault("_inds_pin_np", self._inds_pin.numpy())
        for e in range(E_):
            np.random.shuffle(b_inds)                                     # :315 host MT19937, per-rank seed
            pin_np[e] = b_inds
        self._inds_dev.copy_(self._inds_pin, non_blocking=True)
        if self._inds_dev.is_cuda:
            if evs.get("all") is None:
                evs["all"] = torch.cuda.Event()
            evs["all"].record()
        if a.norm_adv:                                                    # :337-338 for every slot at once
            flat = self._inds_dev.view(-1)
            if use_pack:
                self.ops.adv_stats_packed(self._pack, flat, M, out=self._adv_md_buf.view(-1, 2))
            else:
                self.ops.adv_stats(b_advantages, flat, M, out=self._adv_md_buf.view(-1, 2))
        self._upload_adam_schedule(lr)                                    # every slot's (step size, bias correction) -> device memory
        for row in self._update_graphs:
            for g in row:
                g.replay(self)                                            # forward + fused loss + backward (+ all-reduce) + clip + Adam of the slot
        n = sum(len(row) for row in self._update_graphs)
        self.flat.step += n
        return n

    def _upload_adam_schedule(self, lr: float) -> None:
        """Rows k = 0 .. slots-1 of the device table the captured optimizer steps read: Adam step ``flat.step + k + 1`` at this
        iteration's learning rate (``ops.adam_schedule``: the library's own arithmetic, so eager and captured steps agree bit for
        bit).  One small pinned H2D copy per iteration; the pinned rows are rewritten only after the previous copy executed."""
        if self._adam_sched_ev is not None:
            self._adam_sched_ev.synchronize()
        pin = self._adam_sched_pin
        pin_np = self.__dict__.setdefault("_adam_sched_pin_np", pin.numpy())
        for k in range(pin.shape[0]):
            pin_np[k] = self.ops.adam_schedule(lr, self.flat.step + k + 1)
        self._adam_sched.copy_(pin, non_blocking=True)
        if self._adam_sched.is_cuda:
            if self._adam_sched_ev is None:
                self._adam_sched_ev = torch.cuda.Event()
            self._adam_sched_ev.record()

    def upload_permutation(self, epoch: int, b_inds: np.ndarray) -> torch.Tensor:
        """Host permutation of this epoch (:315) -> its own pinned row -> its own device row (async H2D).  The pinned row is
        rewritten only after the copy that last read it has executed (an event per row: the host may be an iteration ahead)."""
        pin, dev = self._inds_pin[epoch], self._inds_dev[epoch]
        on_gpu = dev.is_cuda                       # (the HIP branches also run on CPU tensors under the tests' stand-in ops)
        evs = self.__dict__.setdefault("_inds_ev", {})
        ev = evs.get(epoch)
        if ev is not None:
            ev.synchronize()
        pin.copy_(torch.from_numpy(b_inds))
        dev.copy_(pin, non_blocking=True)
        if on_gpu:
            if ev is None:
                ev = evs[epoch] = torch.cuda.Event()
            ev.record()
        return dev

    def _early_all_reduce(self, _param) -> None:
        """Post-accumulate hook of the largest parameter: its slice of the flat gradient is final for this minibatch.  While
        ``capture_update`` records a slot for world > 1 the same moment is where the slot's first graph ends and its second begins."""
        if self._capture_cut is not None:
            self._capture_cut()
            return
        if self._ar_armed:
            off, n = self._ar_early
            self._ar_work = dist.all_reduce(self.flat.grads[off:off + n], op=dist.ReduceOp.SUM, async_op=True)
            self._ar_armed = False

    def _minibatch_hip(self, idx, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, lr, scalars_out):
        self._ar_armed = self._ar_early is not None and self.world_size > 1
        self.forward_backward_hip(idx, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, scalars_out)
        if self._peer is not None:
            self._peer.all_reduce_sum_(self.flat.grads)                   # :360-367 as five small launches on this stream (csrc/dpcomm.hip)
        elif self.world_size > 1:
            g = self.flat.grads
            # Stream-ordering audit for the nccl (RCCL) backend -- every statement below was checked against
            # ProcessGroupNCCL's semantics, which differ from gloo's (gloo's wait() blocks the HOST; nccl's only orders streams):
            #  * a collective is enqueued on RCCL's internal stream AFTER an event recorded on the current (compute) stream
            #    at call time, so it reads gradient values every kernel launched before the call has written -- the early
            #    piece's hook fires after the FC weight-gradient kernel was launched on the compute stream;
            #  * a blocking call (async_op=False) and Work.wait() both make the CURRENT stream wait for the collective: the
            #    fused norm + clip + Adam kernels of optimizer_step_hip() are launched on that same stream afterwards, so
            #    they read the reduced buffer, never a partially reduced one; nothing else reads flat.grads;
            #  * the three pieces are disjoint slices of one buffer, so the in-flight early piece and the two late ones
            #    never touch the same bytes; the buffer is zeroed by the Adam kernel, i.e. after all three completed.
            if self._ar_work is not None:                                 # :367 in three pieces: the early bucket is in flight
                off, n = self._ar_early
                if off > 0:
                    dist.all_reduce(g[:off], op=dist.ReduceOp.SUM)
                if off + n < g.numel():
                    dist.all_reduce(g[off + n:], op=dist.ReduceOp.SUM)
                self._ar_work.wait()                                      # the compute stream waits for RCCL's
                self._ar_work = None
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)                  # :367 on the persistent flat buffer (RCCL)
            self._ar_armed = False
        self.optimizer_step_hip(lr)

    def forward_backward_hip(self, idx, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, scalars_out):
        """K5 gather -> network forward -> K3 fused loss fwd+bwd -> autograd through the network only (:320-358).
        Gradients land in the persistent flat buffer (``.grad`` of every parameter is a view of it)."""
        a, ops = self.args, self.ops
        if self.mlp is not None:
            # K7: b_obs[mb_inds] gather, both networks' forward, the distribution, K3's loss row terms, both backward passes and
            # the weight gradients in two launches; gradients are added into the flat buffer's views
            shift = None
            if getattr(self.agent, "rpo_alpha", None) is not None:        # RPO: loss on mean + U(-alpha, alpha), d/dmean unchanged
                shift = self.agent.perturb_mean(torch.zeros((idx.numel(),) + self.act_shape, device=self.device))
            ls = None if self.discrete else self.agent.actor_logstd
            ops.mlp_ppo_fwd_bwd(b_obs, idx, self.mlp[0], self.mlp[1], b_actions, b_logprobs, b_advantages, b_returns, b_values,
                                a.clip_coef, a.ent_coef, a.vf_coef, a.norm_adv, a.clip_vloss, adv_mean_den=self._mb_adv_md,
                                scalars_out=scalars_out, logstd=None if ls is None else ls.detach(),
                                logstd_grad=None if ls is None else ls.grad, mean_shift=shift)
            return
        if self.image and self.fused_cnn:
            p, value = self.agent.heads_u8(b_obs, idx)                    # :320 gather + /255 fused into conv1
        else:
            if self.image:
                if self._x_mb is None or self._x_mb.shape[0] != idx.numel():
                    self._x_mb = torch.empty((idx.numel(),) + self.obs_shape, device=self.device)
                if self.partial_scale:
                    x = self.agent.scale_frames_(ops.obs_u8_to_f32(b_obs, idx, self._x_mb, False)).permute(0, 3, 1, 2)
                else:
                    x = ops.obs_u8_to_f32(b_obs, idx, self._x_mb).permute(0, 3, 1, 2)   # K5: b_obs[mb_inds] ; x / 255.0
            else:
                x = b_obs.index_select(0, idx)
            p, value = self.agent.heads(x)                                # :320 network forward
        value = value.view(-1)
        if self.discrete and self._pack is not None:                     # inside update(): packed behaviour rows
            _, dp, dvalue = ops.ppo_loss_categorical_packed(p.detach().contiguous(), value.detach().contiguous(), idx, self._pack,
                                                            a.clip_coef, a.ent_coef, a.vf_coef, a.norm_adv, a.clip_vloss,
                                                            scalars_out=scalars_out, adv_mean_den=self._mb_adv_md,
                                                            slot=self._mb_slot)
            torch.autograd.backward([p, value], [dp, dvalue])             # :358
        elif self.discrete:
            _, dp, dvalue = ops.ppo_loss_categorical(p.detach().contiguous(), value.detach().contiguous(), idx, b_actions,
                                                     b_logprobs, b_advantages, b_returns, b_values, a.clip_coef,
                                                     a.ent_coef, a.vf_coef, a.norm_adv, a.clip_vloss,
                                                     scalars_out=scalars_out, adv_mean_den=self._mb_adv_md,
                                                     slot=self._mb_slot)
            torch.autograd.backward([p, value], [dp, dvalue])             # :358
        else:
            if getattr(self.agent, "rpo_alpha", None) is not None:       # RPO: loss on the perturbed mean, d/dmean unchanged
                p_eff = self.agent.perturb_mean(p.detach())
            else:
                p_eff = p.detach()
            _, dmean, dlogstd, dvalue = ops.ppo_loss_normal(p_eff.contiguous(), self.agent.actor_logstd.detach(),
                                                            value.detach().contiguous(), idx, b_actions, b_logprobs,
                                                            b_advantages, b_returns, b_values, a.clip_coef, a.ent_coef,
                                                            a.vf_coef, a.norm_adv, a.clip_vloss, scalars_out=scalars_out,
                                                            adv_mean_den=self._mb_adv_md)
            torch.autograd.backward([p, value], [dmean, dvalue])
            self.agent.actor_logstd.grad.add_(dlogstd.view_as(self.agent.actor_logstd))

    def optimizer_step_hip(self, lr: float) -> None:
        """(sum of grads) / world_size -> clip_grad_norm_ -> Adam (:368-377), one fused kernel pair on the
        flat buffers; also zeroes the gradient buffer for the next backward."""
        a = self.args
        self.flat.step += 1
        self.ops.clip_adam_(self.flat.params, self.flat.grads, self.flat.exp_avg, self.flat.exp_avg_sq, self.flat.step, lr,
                            a.max_grad_norm, grad_scale=1.0 / self.world_size, eps=self.adam_eps,
                            total_norm_out=self._total_norm)
        trunk = getattr(self.agent, "_trunk", None)
        if trunk is not None:                       # the kernel rewrote the parameters through raw pointers
            trunk.bufs.weights_version += 1

    def _minibatch_host(self, mb_inds, b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values, lr):
        """The reference's minibatch body on CPU tensors (ppo.py:250-290; multigpu :360-374 when world_size>1)."""
        a = self.args
        self.optimizer.param_groups[0]["lr"] = lr
        x = b_obs[mb_inds]
        if self.image:
            x = self.agent.scale_frames_(x.clone()) if self.partial_scale else x / 255.0
            if self.hwc_frames:
                x = x.permute((0, 3, 1, 2))                       # "bhwc" -> "bchw" (ppo_procgen.py:150)
        p, newvalue = self.agent.heads(x)
        # K3 through the C ABI's host-pointer twins: the distribution, the three loss terms and their gradients down to the
        # network outputs in one call on the FLAT batch arrays + mb_inds -- the seam the GPU path crosses (forward_backward_hip)
        inds = torch.as_tensor(mb_inds, dtype=torch.int64)
        if self.discrete:
            loss, scalars = host_ops.ppo_loss_categorical(p, newvalue, inds, b_actions, b_logprobs, b_advantages, b_returns,
                                                          b_values, a.clip_coef, a.ent_coef, a.vf_coef, a.norm_adv, a.clip_vloss)
        else:
            p = self.agent.perturb_mean(p) if getattr(self.agent, "rpo_alpha", None) is not None else p
            loss, scalars = host_ops.ppo_loss_normal(p, self.agent.actor_logstd, newvalue, inds, b_actions, b_logprobs,
                                                     b_advantages, b_returns, b_values, a.clip_coef, a.ent_coef, a.vf_coef,
                                                     a.norm_adv, a.clip_vloss)
        self.optimizer.zero_grad()
        loss.backward()
        self._host_allreduce_grads()
        nn.utils.clip_grad_norm_(self.agent.parameters(), a.max_grad_norm)
        self.optimizer.step()
        return scalars

    def _host_allreduce_grads(self) -> None:
        """ppo_atari_multigpu.py:360-374 on the host path: pack, all-reduce(SUM), unpack divided by world_size."""
        if self.world_size <= 1:
            return
        all_grads = torch.cat([p_.grad.view(-1) for p_ in self.agent.parameters() if p_.grad is not None])
        dist.all_reduce(all_grads, op=dist.ReduceOp.SUM)
        offset = 0
        for p_ in self.agent.parameters():
            if p_.grad is not None:
                p_.grad.data.copy_(all_grads[offset:offset + p_.numel()].view_as(p_.grad.data) / self.world_size)
                offset += p_.numel()
