#!/usr/bin/env python
"""bench.py -- env-steps/sec of the MI355X-native PPO actor-learner (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE full PPO iteration of the hot path on synthetic input: a rollout of ``num_steps`` x
``local_num_envs`` (policy forward, Categorical sample, storage writes), bootstrap + GAE, and
``update_epochs x num_minibatches`` minibatch updates (uint8 gather+convert, network forward, fused loss
forward+backward, network backward, gradient all-reduce when N>1, fused clip+Adam).  Workload = BASELINE.json
configs[2]: ppo_atari Breakout, num_envs=1024 per GPU, num_steps=128, synthetic 84x84x4 uint8 observations,
NatureCNN with 4 actions, f32 everywhere (the reference's precision).  Observations come from a
device-resident generator, so inputs are already in HBM when the timed region starts (no PCIe inside it).

Weak scaling: every rank (one process per GPU, RCCL over xGMI) keeps 1024 envs; ``value`` is the whole-job
env-steps/sec = N * 1024 * 128 * K / max-over-ranks(time).

The JSON line also carries
  ``roofline``      for the dominant HIP kernel of the path = the convolution / FC launch with the largest total time in
                    the timed region (KERNEL_INFO below names them; the uint8 gather + /255 of K5, bias, ReLU and
                    ReLU-backward passes are fused into them), timed live with HIP events on the learner's stream around each
                    launch and priced on the pipe it EXECUTES on (``roofline_entry``): a bf16-pipe kernel's ``achieved`` is the
                    algorithmic f32 flops x the MFMA term pairs it issues per f32 product, against the dense bf16 MFMA peak --
                    a bound, ``frac`` <= 1; the f32-equivalent rate (which the split kernels push past the f32 MFMA peak) is the
                    secondary ``frac_of_f32_mfma_peak``; ``traffic`` = HBM bytes per launch from the committed PMC passes
                    (profiles/traffic.json: FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md's gfx950 correction);
  ``hbm_frac``      kernel Q (layer-1 forward, the HBM-bound launch that reads the observation bytes): algorithmic bytes /
                    launch time / 8 TB/s -- north_star's HBM-roofline figure;
  ``iteration_f32_equiv_TFLOPs``  algorithmic f32 flops of the whole iteration / ms_per_step;
  ``kernels``       the same accounting for every conv launch shape, and GB/s for the GAE and fused-loss kernels
                    (latency-bound at this size);
  ``cpu_baseline``  the oracle's CPU port of the reference loop (oracle/cpu_ppo_port.py), rank 0, N=1 only,
                    on a bounded sample (64 envs x 128 steps, >=1 iteration), on the box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); 6290 measured achievable
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (= the f32 vector peak), same guide
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, same guide (the 2:1-sparsity headline figure is never used)
# The hand-written kernels of the conv stack + FC layer, by the letter DESIGN.md gives them.  `pipe`: the matrix pipe a kernel
# multiplies on; `products`: MFMA products executed per f32 product of the convolution / GEMM (bf16 pipe: the term pairs of the
# three-term split -- six by default, MI355PPO_BF16_PAIRS=9 for all nine; kernel P's uint8 taps are exact bf16 operands, so
# only its f32 operand is split: three).
BF16_PAIRS = 9 if os.environ.get("MI355PPO_BF16_PAIRS", "6")[:1] == "9" else 6
KERNEL_INFO = {
    "Q": ("conv1q_fwd_kernel (csrc/conv1q.hip): int8-digit MFMA, exact int32 accumulation; uint8 gather, /255, bias and ReLU fused", "i8", 4),
    "P": ("conv1p_wgrad_kernel (csrc/conv1p.hip): bf16 MFMA, uint8 taps exact, dz in three bf16 terms; uint8 gather fused", "bf16", 3),
    "Z": ("z_kernel (csrc/gemmz.hip): bf16 MFMA on three-term splits, weights pre-split into fragment order, activations loaded "
          "coalesced through wave-private LDS; bias + ReLU / the ReLU-backward mask in the epilogue", "bf16", BF16_PAIRS),
    "V": ("convw_bf16_kernel (csrc/convw.hip): bf16 MFMA on three-term splits, both operands transposed through LDS; bias gradient "
          "fused", "bf16", BF16_PAIRS),
    "W": ("fcw_bf16_kernel (csrc/fcw.hip): bf16 MFMA on three-term splits, both operands transposed through LDS", "bf16", BF16_PAIRS),
    # round 5: the same three kernels on the two-term f16 split (csrc/f16split.h): three v_mfma_f32_32x32x16_f16 per f32 product
    "Zh": ("z_kernel<..., SPLIT = 1> (csrc/gemmz.hip): f16 MFMA on two-term splits under per-tensor power-of-two scales (amax records), "
           "weights pre-split into fragment order, activations loaded coalesced through wave-private LDS; un-scale + bias + ReLU / the "
           "ReLU-backward mask + the result's amax in the epilogue", "f16", 3),
    "Rh": ("r_kernel (csrc/convr.hip): f16 MFMA on two-term splits with the SOURCE of a group of images resident in LDS, split once (kernel Z "
           "splits every element once per tap that reads it); weights from the f16x2 pack through an LDS ring; kernel Z's epilogue; results "
           "bit-identical to kernel Z's", "f16", 3),
    "RBh": ("rb_kernel (csrc/convrb.hip): kernel R's layer-2 data gradient with THREE images per group (padded lines share their border record) and the "
            "group's rows dealt to 32-row tiles by border class: six interior tiles walk all 16 k-steps, four rim tiles only the 8 k-steps of the taps that can "
            "lie inside the image (kernel R multiplies the zero border: 19 % of its products, and leaves 56 of 256 row slots empty) -- two thirds of kernel R's "
            "matrix instructions per image; results bit-identical to kernel R's / Z's", "f16", 3),
    "Uh": ("convu_kernel / convu1_kernel (csrc/convu.hip): f16 MFMA on two-term splits with BOTH operands of a group of images resident in LDS, split once; "
           "MFMA fragments by LDS transpose reads (ds_read_b64_tr_b16), the whole weight gradient in the workgroup's accumulators; layer 1: the uint8 "
           "frame as zero-extended 16-bit = exact f16 subnormals, one plane", "f16", 3),
    "Vh": ("convw_bf16_kernel<..., SPLIT = 1> (csrc/convw.hip): f16 MFMA on two-term splits, both operands transposed through LDS; bias "
           "gradient fused", "f16", 3),
    "Wh": ("fcw_bf16_kernel<3, 1> (csrc/fcw.hip): f16 MFMA on two-term splits, both operands transposed through LDS", "f16", 3),
    # round 6: the FC layer on streaming forms of kernels R / U
    "Gh": ("g_kernel (csrc/gemmg.hip): f16 MFMA on two-term splits with BOTH operands streamed through workgroup-wide LDS rings, A split once per "
           "workgroup (kernel Z: once per wave, inside the k-loop); persistent 128 x 256 blocks in XCD-contiguous supertiles; kernel Z's epilogue; results "
           "bit-identical to kernel Z's", "f16", 3),
    "Hh": ("h_kernel (csrc/gemmh.hip): f16 MFMA on two-term splits, both operands of the batch reduction streamed through a workgroup-wide LDS ring, split "
           "once, MFMA fragments by LDS transpose reads (ds_read_b64_tr_b16); 8 batch slabs = 8 XCDs, partials added in slab order", "f16", 3),
    "F": ("conv_fixed_kernel (csrc/conv.hip): f32-MFMA implicit GEMM; bias + ReLU / the ReLU-backward mask in the epilogue", "f32", 1),
    "T": ("conv_wgrad_taps_kernel (csrc/conv.hip): f32-MFMA implicit GEMM", "f32", 1),
    "Y": ("fcw_kernel (csrc/fcw.hip): f32 MFMA", "f32", 1),
}
OBS_ROW_BYTES = 4 * 84 * 84   # 28,224


def describe_cnn(kernel_of, M):
    """``config.cnn`` and ``matrix_arithmetic`` of the JSON line, DERIVED from the kernels the update's launches ran on (``kernel_of``: launch key
    -> letter of KERNEL_INFO), so that the line cannot describe another arithmetic than the one that was measured."""
    order = ("conv1_fwd", "conv2_fwd", "conv3_fwd", "fc_fwd", "fc_dgrad", "fc_wgrad", "conv3_wgrad", "conv3_dgrad", "conv2_wgrad", "conv2_dgrad", "conv1_wgrad")
    ran = {name: kernel_of.get(f"{name}@{M}") for name in order}
    pipes = {KERNEL_INFO[k][1] for n, k in ran.items() if k and n != "conv1_fwd"}
    if pipes == {"f16"}:
        arith = ("f32-equivalent via a two-term f16 split of both operands under per-tensor power-of-two scales: exact products of the terms (hi hi, hi lo, "
                 "lo hi = 3 v_mfma_f32_32x32x16_f16 per f32 product; layer-1 weight gradient 2: its uint8 operand is one exact term), f32 accumulate; error "
                 "vs float64 at or below the f32 library GEMM's (profiles/r05_err_f16x2.jsonl, tests/test_gpu_f16x2.py)")
    elif pipes == {"bf16"}:
        arith = f"f32-equivalent via a three-term bf16 split of both operands: {BF16_PAIRS} of 9 term pairs on v_mfma_f32_32x32x16_bf16, f32 accumulate"
    elif pipes == {"f32"}:
        arith = "f32 MFMA (v_mfma_f32_32x32x2_f32)"
    else:
        arith = "mixed: " + ", ".join(sorted(p for p in pipes if p))
    fam = ", ".join(f"{n} {k}" for n, k in ran.items() if k)
    text = ("layer-1 forward on the int8 MFMA with exact int32 accumulation over 31-bit fixed-point weights (kernel Q); every other convolution / FC GEMM: "
            + arith + f".  Kernels of one minibatch update at {M} rows, in launch order: {fam} (letters: KERNEL_INFO / DESIGN.md 3.2; a trailing h = the f16 split)")
    return text, arith


def _conv1_fwd_bytes(images):
    """Algorithmic bytes of a layer-1 forward launch: the uint8 frame read once + the f32 activation written once."""
    return images * (OBS_ROW_BYTES + 20 * 20 * 32 * 4)


def _traffic_of(key):
    """HBM bytes per launch of conv launch `key` ("conv1_wgrad@32768") from the committed PMC passes, or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get("hbm_bytes_per_launch", {}).get(key)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--config", choices=sorted(CONFIGS), default="C",
                   help="BASELINE.json workload: C = configs[2] (default, the configuration the metric is quoted on); "
                        "B = configs[1] (128 envs); D = configs[3] at its per-GPU size (256 envs); E = configs[4] "
                        "(ppo_continuous_action, Normal kernels)")
    p.add_argument("--local-num-envs", type=int, default=None, help="override the config's envs per GPU")
    p.add_argument("--num-steps", type=int, default=None, help="override the config's rollout length")
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                   help="collective backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only with --same-device)")
    p.add_argument("--preflight", action="store_true",
                   help="--gpus N > 1: ONLY the process-group preflight -- init_process_group (nccl = RCCL, device_id given), one all-reduce of the flat "
                        "gradient's size (6.75 MB) checked element by element and timed, RCCL's own init / ring / topology lines on stderr "
                        "(NCCL_DEBUG=INFO, subsystems INIT,GRAPH) -- then exit 0.  Every --gpus N > 1 run starts with the same check, without the debug lines")
    p.add_argument("--same-device", action="store_true",
                   help="PLUMBING SMOKE, not a measurement: run all --gpus ranks on cuda:0 (needs --backend gloo: RCCL refuses "
                        "two ranks on one device).  Exercises the launcher, barrier, max-over-ranks timing and rank-0 JSON line")
    p.add_argument("--n-actions", type=int, default=4)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-envs", type=int, default=64)
    p.add_argument("--cpu-baseline-full", choices=["auto", "on", "off"], default="auto",
                   help="how cpu_baseline.value is taken at the METRIC's configuration (1,024 envs x 128 steps, 16 updates of 32,768 rows) on this box's own "
                        "cores.  auto (default, config C): a bounded sample -- every piece of the loop body timed at its full size (16 env steps, GAE, one "
                        "minibatch update after one untimed), times its count: ~1 minute of CPU work.  on: ONE whole iteration (~5 minutes on the GPU box's "
                        "256 threads, ~3 on 8 cores; abandoned for the sample if the rollout alone exceeds 2 minutes).  off: whole iterations at "
                        "--cpu-baseline-envs envs (the rounds 1-5 sample; NOT the metric's configuration)")
    p.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event brackets (pure SPS run)")
    p.add_argument("--no-rollout-graphs", action="store_true", help="issue the rollout kernel by kernel instead of one hipGraph per step")
    p.add_argument("--rollout-steps-per-graph", type=int, default=0,
                   help="consecutive env steps captured into one hipGraph (0 = the whole rollout in one graph, the default: a replay "
                        "boundary costs ~8 us of GPU idle; 1 = a graph per step)")
    p.add_argument("--no-pcie-inclusive", action="store_true", help="skip the host-env (PCIe-inclusive) leg after the timed region")
    p.add_argument("--pcie-env-groups", type=int, default=4)
    p.add_argument("--update-graphs", action="store_true", help="(the default on one GPU; kept for old command lines)")
    p.add_argument("--no-update-graphs", action="store_true",
                   help="issue the update launch by launch instead of replaying one hipGraph per (epoch, minibatch) slot (forward + fused "
                        "loss + backward + clip + Adam; PPOLearner.capture_update, bit-identical to the eager update in the GPU tests; "
                        "profiles/r04_update_graphs_ab.jsonl).  With graphs the per-launch event brackets of `roofline` / `kernels` come "
                        "from an extra eager iteration right after the timed region (the second of two: the first one allocates the eager path's buffers inside "
                        "the brackets; a replay runs no Python to put events around)")
    p.add_argument("--sync-metrics", action="store_true",
                   help="read every iteration's diagnostics before the next one starts (the reference's arrangement: one device "
                        "synchronisation per iteration); default: resolve them one iteration late, so that the host runs ahead "
                        "of the GPU and a host stall does not drain the GPU's queue")
    p.add_argument("--inject-host-stall-ms", type=float, default=0.0,
                   help="DIAGNOSTIC: sleep this long on the host before the 6th minibatch of every update (what a descheduled "
                        "thread / page-in / garbage collection does on a busy box); compare with and without --sync-metrics")
    cli = p.parse_args()
    cfg = CONFIGS[cli.config]
    if cli.local_num_envs is None:
        cli.local_num_envs = cfg["local_num_envs"]
    if cli.num_steps is None:
        cli.num_steps = cfg["num_steps"]
    if cli.same_device and cli.backend != "gloo":
        p.error("--same-device needs --backend gloo (RCCL refuses two ranks on one device)")
    return cli


# BASELINE.json's configs, as bench workloads (per GPU).  Every image config runs the same code path: NatureCNN A = 4,
# 4 epochs x 4 minibatches, synthetic 84x84x4 uint8 frames from the device-resident generator.
CONFIGS = {
    "B": dict(local_num_envs=128, num_steps=128,
              workload="configs[1]: ppo_atari_envpool Breakout-v5 num_envs=128 num_steps=128 (minibatch 4,096 rows)"),
    "C": dict(local_num_envs=1024, num_steps=128,
              workload="configs[2]: ppo_atari Breakout num_envs=1024/GPU num_steps=128 (minibatch 32,768 rows)"),
    "D": dict(local_num_envs=256, num_steps=128,
              workload="configs[3] at its per-GPU size: ppo_atari_multigpu Breakout local_num_envs=256 num_steps=128 "
                       "(minibatch 8,192 rows)"),
    "E": dict(local_num_envs=64, num_steps=2048,
              workload="configs[4]: ppo_continuous_action HalfCheetah-v4-shaped (obs 17, act 6) num_envs=64 num_steps=2048, "
                       "32 minibatches x 10 epochs, Normal sample + log_prob / fused Normal loss kernels"),
}


# ALGORITHMIC bytes per image of the eleven conv / FC launches of a minibatch update: every input read once + every output written once (the
# ReLU masks as bits: 1/32 of an activation's bytes); the FC layer's 6.4 MB weight pack / result are not counted.  (tools/make_traffic_json.py
# holds the same table for profiles/traffic.json)
ALG_BYTES_PER_IMAGE = {
    "conv1_fwd": 28224 + 51200 + 1600, "conv2_fwd": 51200 + 20736 + 648, "conv3_fwd": 20736 + 12544 + 392,
    "conv2_dgrad": 20736 + 1600 + 51200, "conv3_dgrad": 12544 + 648 + 20736,
    "conv1_wgrad": 28224 + 51200, "conv2_wgrad": 51200 + 20736, "conv3_wgrad": 20736 + 12544,
    "fc_fwd": 12544 + 2048, "fc_dgrad": 2048 + 392 + 12544, "fc_wgrad": 2048 + 12544,
}


def hbm_regime_points(device, reps=10):
    """K1 (GAE) and K3 (fused loss) at sizes where the working set is far beyond the 256 MiB Infinity Cache -- the regime north_star's ">= 60 % of the HBM
    roofline" is about (at config C's own size the kernels move 2.6 / 2.2 MB and are launch-latency-bound: `kernels.gae`).  One launch
    shape each, HIP events on the current stream around `reps` launches, after the timed region (never part of `value`):
      gae_hbm_regime:  T = 128, N = 2^20 envs (2.7 GB of algorithmic bytes: 20 T N + 8 N)
      loss_hbm_regime: M = 2^22 minibatch rows, A = 4, packed behaviour rows, advantage statistics hoisted (the learner's mode);
                       `streaming` = mb_inds NULL ((8 A + 28) M bytes), `permuted` = a random permutation of a batch as large ((8 A + 36) M bytes)."""
    from cleanrl_amd import ops, synthetic

    def ev_us(fn):
        fn(); fn()
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    out = {}
    T, N = 128, 1 << 20
    base = {k: v.to(device) for k, v in synthetic.rollout_scalars(T, 4096, 4, seed=1).items()}
    sc = {k: (v.repeat(1, N // 4096) if v.dim() == 2 else v.repeat(N // 4096)).contiguous() for k, v in base.items()}
    adv, ret = torch.empty_like(sc["rewards"]), torch.empty_like(sc["rewards"])
    us = ev_us(lambda: ops.gae(sc["rewards"], sc["dones"], sc["values"], sc["next_done"], sc["next_value"], 0.99, 0.95, adv, ret))
    nb = 20 * T * N + 8 * N
    out["gae_hbm_regime"] = {"T": T, "N": N, "algorithmic_bytes": nb, "avg_us": us, "GBps": nb / us / 1e3, "frac": nb / us / 1e3 / HBM_PEAK_GBPS,
                             "frac_of_measured_achievable_6290": nb / us / 1e3 / 6290.0, "launches_timed": reps, "timing": "HIP events, current stream"}
    del sc, adv, ret, base
    M, A = 1 << 22, 4
    g = torch.Generator(device=device).manual_seed(3)
    logits, value = torch.randn(M, A, device=device, generator=g), torch.randn(M, device=device, generator=g)
    b_actions = torch.randint(0, A, (M,), device=device, generator=g).float()
    b_lp = torch.randn(M, device=device, generator=g) * 0.1 - 1.4
    b_adv, b_ret, b_val = (torch.randn(M, device=device, generator=g) for _ in range(3))
    pack = ops.batch_pack(b_actions, b_lp, b_adv, b_ret, b_val)
    dl, dv, slots = torch.empty_like(logits), torch.empty_like(value), ops.LossSlots(1, device)
    res = {"M": M, "A": A, "launches_timed": reps, "timing": "HIP events, current stream"}
    for mode, inds in (("streaming", None), ("permuted", torch.randperm(M, device=device, generator=g))):
        md = ops.adv_stats(b_adv, inds, M)[0]
        us = ev_us(lambda: ops.ppo_loss_categorical_packed(logits, value, inds, pack, 0.1, 0.01, 0.5, True, True, dlogits_out=dl, dvalue_out=dv,
                                                           adv_mean_den=md, slot=(slots, 0)))
        nb = (8 * A + 28 + (8 if inds is not None else 0)) * M
        res[mode] = {"algorithmic_bytes": nb, "avg_us": us, "GBps": nb / us / 1e3, "frac": nb / us / 1e3 / HBM_PEAK_GBPS,
                     "frac_of_measured_achievable_6290": nb / us / 1e3 / 6290.0}
    res["frac"] = res["streaming"]["frac"]
    out["loss_hbm_regime"] = res
    return out


def hbm_stream_probe():
    """What plain streams reach from HBM on THIS box (tools/hbm_probe quick: grid-stride kernels over 1 - 2 GiB, HIP events, after the timed region):
    the measured counterpart of the 8 TB/s `peak` the roofline fractions are quoted against.  None when the binary is missing or fails."""
    import subprocess

    exe = os.path.join(ROOT, "tools", "hbm_probe")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=60)
        rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        by = {x["kernel"]: x["GBps_avg"] for x in rows}
        return {"read_only_GBps": by["read16 nt U4"], "write_only_GBps": by["write4 plain"], "copy_GBps": by["copy16 ntLS U4"],
                "mix_1r_2w_GBps": by["mix 1r:2w(4B) plain"], "peak_GBps": HBM_PEAK_GBPS,
                "note": "tools/hbm_probe.cpp quick: 16-byte non-temporal reads of 1 GiB; 4-byte-per-lane stores of 1 GiB (the resident kernels' epilogue shape); "
                        "a 1 GiB -> 1 GiB non-temporal copy; one 16-byte read per 32 bytes stored (kernel Q's ratio); mean of 6 launches each, HIP events"}
    except Exception as e:      # a diagnostic leg: never in the way of the line
        print(f"bench.py: hbm_stream_probe failed: {e!r}", file=sys.stderr, flush=True)
        return None


def roofline_entry(key, us, launches_timed, flops, letter, share, traffic):
    """The ``roofline`` object of the JSON line for the dominant conv / FC launch ``key`` ("conv2_dgrad@32768"): the BINDING roof of the
    two-roof model, so ``frac`` <= 1 by construction.  For the launch's algorithmic work

        t_hbm  = algorithmic bytes / 8 TB/s                                   (ALG_BYTES_PER_IMAGE x images)
        t_mfma = algorithmic f32 flops x MFMA products issued per f32 product / the dense peak of the pipe the kernel multiplies on
                 (two-term f16 split, the default since round 5: 3 products, 2,500 TFLOP/s; three-term bf16 split: 6 -- or 9 with
                 MI355PPO_BF16_PAIRS=9; kernel P: 2 / 3, its uint8 operand is exact; f32-pipe kernels F / T / Y: 1, 157.3 TFLOP/s; kernel Q:
                 int8 pipe, always HBM-bound)

    the larger one is the roof the launch cannot beat: ``bound`` names it, ``achieved`` / ``peak`` / ``unit`` are in its terms and
    ``frac`` = t_roof / measured launch time.  Under the f16 split the layer-1 / layer-2 launches (2.4 - 2.9 GB per launch for 0.17 - 0.22
    TFLOP) are HBM-bound by this model; under the bf16 split (twice the matrix instructions) they were matrix-pipe-bound, which is what
    round 4's line priced.  Both views travel in the object (``hbm_*`` / ``mfma_*`` fields), so that lines of different splits compare."""
    text, pipe, products = KERNEL_INFO[letter]
    if letter == "Uh" and key.startswith("conv1_wgrad"):
        products = 2                                   # (the uint8 frame is one exact f16 term: dz hi, dz lo)
    if letter == "P" and os.environ.get("MI355PPO_SPLIT", "f16x2") == "f16x2":
        text, pipe, products = text.replace("bf16 MFMA", "f16 MFMA (round 5: dz in two f16 terms)"), "f16", 2
    tf = flops / us / 1e6
    name, images = key.split("@")[0], int(key.split("@")[1])
    alg = ALG_BYTES_PER_IMAGE[name] * images
    peak_tf = MFMA_BF16_PEAK_TFLOPS if pipe in ("bf16", "f16") else MFMA_F32_PEAK_TFLOPS      # (the dense f16 and bf16 MFMA peaks are equal)
    t_hbm_us = alg / (HBM_PEAK_GBPS * 1e3)
    t_mfma_us = 0.0 if letter == "Q" else flops * products / (peak_tf * 1e6)
    r = {"kernel": f"{key}: kernel {letter} = {text}", "avg_launch_us": us, "launches_timed": launches_timed, "share_of_step_time": share, "traffic": traffic,
         "algorithmic_bytes_per_launch": alg, "algorithmic_flops_per_launch": flops,
         "hbm_roof_us": t_hbm_us, "hbm_GBps": alg / us / 1e3, "hbm_frac": t_hbm_us / us,
         "mfma_roof_us": t_mfma_us, "mfma_pipe": pipe, "mfma_products_per_f32_product": products, "mfma_executed_TFLOPs": tf * products,
         "mfma_frac": t_mfma_us / us, "algorithmic_TFLOPs": tf, "frac_of_f32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS}
    if t_hbm_us >= t_mfma_us:
        r.update({"bound": "hbm", "achieved": alg / us / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": t_hbm_us / us,
                  "frac_of_measured_achievable_6290": alg / us / 1e3 / 6290.0})
    else:
        r.update({"bound": "mfma", "achieved": tf * products, "peak": peak_tf, "unit": "TFLOP/s", "frac": t_mfma_us / us})
    if pipe in ("bf16", "f16"):
        r["mfma"] = "v_mfma_f32_32x32x16_" + pipe
    r["note"] = ("two-roof model: t_hbm = algorithmic bytes / 8 TB/s, t_mfma = algorithmic f32 flops x products issued / the dense peak of the executing "
                 "pipe; `bound` = the larger, frac = t_roof / avg_launch_us.  Padding taps of the data gradients are neither multiplied nor counted "
                 "(t_mfma of a data gradient is therefore an upper estimate of its executed work by 19 - 40 %).  `traffic` = L2-miss bytes of the launch "
                 "from the committed PMC passes (profiles/traffic.json)")
    return r


def iteration_flops(N, T, M, n_actions, epochs):
    """Algorithmic f32 flops of one PPO iteration of the NatureCNN path: (T + 1) x N rollout / bootstrap forwards, and per update
    pass over the batch forward + weight gradients of every layer + data gradients of every layer but the first."""
    conv = [2.0 * hout * hout * cout * cin * k * k for (cin, cout, k, _, _, hout) in
            ((4, 32, 8, 4, 84, 20), (32, 64, 4, 2, 20, 9), (64, 64, 3, 1, 9, 7))]
    fc, heads = 2.0 * 3136 * 512, 2.0 * 512 * (n_actions + 1)
    fwd = sum(conv) + fc + heads
    return (T + 1) * N * fwd + epochs * T * N * (2 * fwd + (fwd - conv[0]))


class KernelTimer:
    """HIP-event brackets around chosen kernel launches on the learner's stream (torch.cuda.Event records on
    the current stream, which is the stream every libmi355ppo launch is enqueued on)."""

    def __init__(self):
        self.pairs = {}

    def wrap(self, name, fn):
        def timed(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.pairs.setdefault(name, []).append((e0, e1))
            return out

        return timed

    def reset(self):
        self.pairs = {}

    def mean_us(self, name):
        ps = self.pairs.get(name, [])
        if not ps:
            return None, 0
        return float(np.mean([a.elapsed_time(b) for a, b in ps])) * 1e3, len(ps)


def preflight(rank: int, world: int, device, backend: str, verbose: bool) -> dict:
    """The first collective of a multi-rank run, made attributable: an all-reduce(SUM) of 1,686,693 f32 (the NatureCNN agent's flat gradient,
    ppo_atari_multigpu.py:360-367) whose result is known exactly -- rank r contributes (i % 251) * (r + 1), so element i must come back as
    (i % 251) * world (world + 1) / 2, exact in f32 -- then five timed repetitions.  Raises on a wrong element; returns the timing."""
    n = 1686693
    base = (torch.arange(n, device=device, dtype=torch.int64) % 251).to(torch.float32)
    x = base * float(rank + 1)
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    want = base * float(world * (world + 1) // 2)
    bad = int((x != want).sum().item())
    if bad:
        raise RuntimeError(f"bench.py preflight: rank {rank}: all-reduce over {backend} returned {bad} wrong elements of {n}")
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize(device)
    us = (time.perf_counter() - t0) / 5 * 1e6
    info = {"backend": backend, "world": world, "bytes": 4 * n, "allreduce_us": us,
            "bus_GBps": 4 * n * 2 * (world - 1) / world / us / 1e3}
    if rank == 0 and verbose:
        print(f"[bench] preflight ok: all-reduce(SUM) of {4 * n} bytes over {world} ranks ({backend}), every element exact; "
              f"{us:.0f} us per call = {info['bus_GBps']:.1f} GB/s bus bandwidth", file=sys.stderr, flush=True)
    return info


def self_launch(cli) -> int:
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU,
    the launch line of the module docstring) and hand their output through.  Refuses -- loudly, non-zero -- when the box
    has fewer than N GPUs: a one-rank run must never pass for an N-GPU measurement."""
    import socket
    import subprocess

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if cli.same_device and have >= 1:
        have = cli.gpus                              # plumbing smoke: every rank on cuda:0
    if have < cli.gpus:
        print(f"bench.py: --gpus {cli.gpus} requested but this box has {have} GPU(s); refusing to run fewer ranks than asked",
              file=sys.stderr)
        return 2
    with socket.socket() as so:                      # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={cli.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    cli = parse()
    if cli.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(cli))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == cli.gpus, f"--gpus {cli.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU"
    assert torch.cuda.is_available(), "bench.py measures the MI355X path and needs a GPU"
    dev_index = 0 if cli.same_device else local_rank
    assert torch.cuda.device_count() > dev_index, f"rank {rank}: no GPU {dev_index} on this box ({torch.cuda.device_count()} visible)"
    torch.cuda.set_device(dev_index)
    device = torch.device(f"cuda:{dev_index}")
    pre = None
    if world > 1:
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cli.preflight and cli.backend == "nccl":
            os.environ.setdefault("NCCL_DEBUG", "INFO")                   # RCCL's init / ring / topology lines, on stderr
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
        # a collective that does not complete ends the job with an error after this long instead of hanging the box (first RCCL runs)
        limit = datetime.timedelta(seconds=int(os.environ.get("MI355PPO_PG_TIMEOUT_S", "300")))
        if cli.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=limit)     # RCCL over xGMI
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)                       # --same-device smoke only
        pre = preflight(rank, world, device, cli.backend, verbose=True)
        if cli.preflight:
            if rank == 0:
                print(json.dumps({"preflight": pre}), flush=True)
            dist.destroy_process_group()
            return
    cli.collective_preflight = pre
    if cli.config == "E":
        return main_continuous(cli, rank, world, device)

    from cleanrl_amd import _lib, learner_smoke, ops
    from cleanrl_amd.agents import AtariAgent
    from cleanrl_amd.envs import DeviceSyntheticAtariVecEnv
    from cleanrl_amd.learner import PPOLearner

    _lib.load(build_if_missing=True)
    N, T = cli.local_num_envs, cli.num_steps
    args = learner_smoke.default_args(num_steps=T, num_minibatches=4, update_epochs=4, learning_rate=2.5e-4, clip_coef=0.1,
                                      ent_coef=0.01, vf_coef=0.5)        # ppo_atari.py defaults
    # seeding protocol of ppo_atari_multigpu.py:206-212,231
    seed = cli.seed + rank
    np.random.seed(seed)
    torch.manual_seed(cli.seed)
    env = DeviceSyntheticAtariVecEnv(N, device, seed=seed, n_actions=cli.n_actions)
    agent = AtariAgent(env).to(device)
    torch.manual_seed(seed)
    learner = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, device, world_size=world,
                         sample_seed=seed)
    learner.observe(0, env.obs_into(learner.stage_obs), learner.dones[0])
    if learner.fused_cnn and not cli.no_rollout_graphs:
        try:
            learner.capture_rollout(env, steps_per_graph=cli.rollout_steps_per_graph or T)      # (before the timing hooks: no event records in a capture)
        except Exception as e:      # noqa: BLE001
            # world > 1: a capture beside RCCL's watchdog thread has only ever run over gloo (one-GPU build boxes).  The rollout holds
            # no collective, so a rank whose capture fails issues its rollout launch by launch instead of ending the whole job.
            if world == 1:
                raise
            print(f"[bench] rank {rank}: rollout-graph capture failed beside the process group ({e!r}); this rank's rollout runs eagerly",
                  file=sys.stderr, flush=True)
            learner._rollout_graphs = None
            env._step_rel = None
            torch.cuda.synchronize()
    update_mode = "eager launches"
    use_update_graphs = not cli.no_update_graphs and learner.fused_cnn
    if use_update_graphs:
        # one policy for every training loop (learner.update_graph_policy / early_bucket_policy): graphs on one GPU and over gloo; over RCCL graphs in
        # the reference's arrangement (one all-reduce behind the backward) behind a captured-vs-eager self-check, ALL ranks agreeing on the outcome
        # through one MIN all-reduce; MI355PPO_UPDATE_GRAPHS=1 adds the early bucket, =0 is the eager update
        from cleanrl_amd.learner import update_graph_policy

        policy = update_graph_policy(world)
        use_update_graphs = learner.capture_update_agreed(log=lambda m: print(f"[bench] rank {rank}: {m}", file=sys.stderr, flush=True))      # (before the timing hooks: no event records in a capture)
        if world == 1 and policy != "off" and not use_update_graphs:
            raise RuntimeError("bench.py: the update-graph capture failed on one GPU (see stderr): that is a defect, not a fallback case")
        if use_update_graphs and world > 1:
            nseg = len(learner._update_graphs[0][0].segs)
            update_mode = (("one hipGraph per (epoch, minibatch) slot with the gradient exchange INSIDE it: forward + fused loss + backward + the peer-memory all-reduce "
                            "(five small launches over HIP IPC segments, csrc/dpcomm.hip; MI355PPO_ALLREDUCE=peer) + clip + Adam "
                            if nseg == 1 else
                            "per (epoch, minibatch) slot three hipGraphs with the gradient exchange between them: [forward + fused loss + backward to the "
                            "FC weight's gradient] | all-reduce of that bucket, asynchronous | [conv backward] | all-reduce of the rest | [clip + Adam] "
                            if nseg == 3 else
                            "per (epoch, minibatch) slot two hipGraphs with the gradient exchange between them (the reference's arrangement, "
                            "ppo_atari_multigpu.py:358-377): [forward + fused loss + backward] | all-reduce of the flat gradient | [clip + Adam] ")
                           + "(PPOLearner.capture_update" + ("; self-checked against one eager update before use" if policy == "capture+check" else "")
                           + "); per-launch event brackets from an extra eager iteration after the timed region")
        elif use_update_graphs:
            update_mode = ("one hipGraph per (epoch, minibatch) slot: forward + fused loss + backward + clip + Adam (PPOLearner.capture_update); "
                           "per-launch event brackets from an extra eager iteration after the timed region")
        else:
            update_mode = ("eager launches (MI355PPO_UPDATE_GRAPHS=0)" if policy == "off"
                           else "eager launches (update-graph capture or its self-check failed on a rank; see stderr)")
    elif not cli.no_update_graphs:
        update_mode = "eager launches (update graphs need the fused CNN kernels: MI355PPO_CNN=miopen)"
    timer = KernelTimer()
    conv_flops = {}      # key "<op>@<rows>" -> algorithmic flops of one launch (the f32 convolution / GEMM: 2 x M x N x K)
    kernel_of = {}       # key -> letter in KERNEL_INFO
    unhook = []          # (module, name, original) of everything the kernel timing wraps

    def install_timing_hooks():
        real_obs, real_gae, real_loss, real_loss_p = ops.obs_u8_to_f32, ops.gae, ops.ppo_loss_categorical, ops.ppo_loss_categorical_packed
        unhook.extend([(ops, "obs_u8_to_f32", real_obs), (ops, "gae", real_gae), (ops, "ppo_loss_categorical", real_loss),
                       (ops, "ppo_loss_categorical_packed", real_loss_p)])
        if learner.fused_cnn:
            from cleanrl_amd import cnn

            lib = _lib.load()
            seen = {}

            def timed_op(name, key_of):
                """Wrap cnn.<name>: key_of(*args) -> (key, flops, kernel letter); rollout-sized launches are bracketed 1 in 16
                (the rollout is host-bound; two event records per launch would slow it)."""
                real = getattr(cnn, name)
                unhook.append((cnn, name, real))

                def hooked(*a, **kw):
                    key, flops, letter = key_of(*a, **kw)
                    conv_flops[key], kernel_of[key] = flops, letter
                    seen[key] = seen.get(key, 0) + 1
                    if key.endswith(f"@{N}") and seen[key] % 16:
                        return real(*a, **kw)
                    return timer.wrap(key, real)(*a, **kw)

                setattr(cnn, name, hooked)

            def conv_flop(layer, images):
                cin, cout, k, _, _, hout = cnn.LAYERS[layer]
                return 2.0 * images * hout * hout * cout * cin * k * k

            def k_fwd(src, Bt, bias, layer, inds=None, out=None, variant=0):
                images = src.shape[0] if inds is None else inds.numel()
                return f"conv{layer}_fwd@{images}", conv_flop(layer, images), ("Q" if variant == cnn.VARIANT_Q else "F")

            def k_trunk(*a, **kw):
                images = a[8].shape[0]
                return f"trunk_fwd(conv1+2+3)@{images}", sum(conv_flop(l, images) for l in cnn.LAYERS), "F"

            def k_wgrad(src, dz, layer, inds=None, out=None, amax=None):
                # (the library's own decision, not a restatement of its switches: kernel U declines tensors beyond the 32-bit buffer range)
                letter = chr(lib.mi355ppo_cnn_conv_wgrad_kernel_f16x2(dz.shape[0], layer) if amax is not None else lib.mi355ppo_cnn_conv_wgrad_kernel(dz.shape[0], layer))
                return f"conv{layer}_wgrad@{dz.shape[0]}", conv_flop(layer, dz.shape[0]), letter + "h" if amax is not None and letter in ("V", "U") else letter

            timed_op("conv_fwd", k_fwd)
            timed_op("trunk_fwd", k_trunk)
            timed_op("conv_dgrad", lambda dz, Bt, act_in, layer, out=None, variant=0: (f"conv{layer}_dgrad@{dz.shape[0]}", conv_flop(layer, dz.shape[0]), "F"))
            timed_op("conv_wgrad", k_wgrad)
            # (amax: the two-term f16 split of round 5 -- the same kernels, letter + "h")
            h = lambda letter, amax: letter + "h" if amax is not None and letter in ("Z", "W", "V") else letter      # (kernels Y / T ignore the records)

            def zr(images, layer, dgrad, bits, amax):       # kernel R takes some of kernel Z's f16x2 launches (csrc/convr.hip): ask the library
                if amax is None or (dgrad and bits is None):
                    return h("Z", amax)
                return {"R": "Rh", "B": "RBh"}.get(chr(lib.mi355ppo_cnn_conv_packed_kernel_f16x2(images, layer, dgrad)), "Zh")

            timed_op("conv1q_fwd_bits", lambda obs, pack, bias, inds, out, bits: (f"conv1_fwd@{out.shape[0]}", conv_flop(1, out.shape[0]), "Q"))
            timed_op("conv1q_fwd_amax", lambda obs, pack, bias, inds, out, bits, dst_amax: (f"conv1_fwd@{out.shape[0]}", conv_flop(1, out.shape[0]), "Q"))
            timed_op("conv_fwd_packed", lambda src, pack, bias, layer, out=None, bits=None, amax=None: (f"conv{layer}_fwd@{src.shape[0]}", conv_flop(layer, src.shape[0]), zr(src.shape[0], layer, 0, bits, amax)))
            timed_op("conv_dgrad_packed", lambda dz, pack, act_in, layer, out=None, bits=None, amax=None: (f"conv{layer}_dgrad@{dz.shape[0]}", conv_flop(layer, dz.shape[0]), zr(dz.shape[0], layer, 1, bits, amax)))
            def zg(M, N, K, dgrad, bits, amax):              # kernel G takes the large f16x2 FC launches (csrc/gemmg.hip): ask the library
                if amax is None or (dgrad and bits is None) or (not dgrad and lib.mi355ppo_fc_fwd_workspace_bytes(M, N, K) > 0):
                    return h("Z", amax)
                return "Gh" if chr(lib.mi355ppo_fc_packed_kernel_f16x2(M, N, K, dgrad)) == "G" else "Zh"

            timed_op("fc_fwd_relu_packed", lambda a, pack, bias, n, out=None, amax=None: (f"fc_fwd@{a.shape[0]}", 2.0 * a.shape[0] * n * a.shape[1], zg(a.shape[0], n, a.shape[1], 0, None, amax)))
            timed_op("fc_dgrad_mask_packed", lambda dz, pack, act_in, out=None, bits=None, amax=None: (f"fc_dgrad@{dz.shape[0]}", 2.0 * dz.shape[0] * dz.shape[1] * act_in.shape[1],
                                                                                                       zg(dz.shape[0], act_in.shape[1], dz.shape[1], 1, bits, amax)))
            timed_op("fc_wgrad", lambda dz, a, hwc_channels=0, out=None, amax=None: (f"fc_wgrad@{dz.shape[0]}", 2.0 * dz.shape[0] * dz.shape[1] * a.shape[1],
                                                                                    (lambda k: k + "h" if k in ("H", "W") else k)(chr(lib.mi355ppo_fc_wgrad_kernel_f16x2(dz.shape[0], dz.shape[1], a.shape[1])))
                                                                                    if amax is not None else chr(lib.mi355ppo_fc_wgrad_kernel(dz.shape[0], dz.shape[1], a.shape[1]))))

        def obs_hook(src, inds=None, out=None, scale_255=True):
            if inds is not None:
                return timer.wrap("obs_gather", real_obs)(src, inds, out, scale_255)
            return real_obs(src, inds, out, scale_255)

        ops.obs_u8_to_f32 = obs_hook
        ops.gae = timer.wrap("gae", real_gae)
        ops.ppo_loss_categorical = timer.wrap("loss", real_loss)
        ops.ppo_loss_categorical_packed = timer.wrap("loss", real_loss_p)      # the learner's call: packed behaviour rows

    # (The event brackets are NEVER installed inside the timed region -- round 6: with the eager update they were, and two event records per launch
    #  made the host the bottleneck: 817 k instead of 1.3 M env-steps/s at config C.  They come from an extra iteration after it, below.)
    total_iters = cli.warmup + cli.steps

    phase_events = []

    def one_step(i):
        lr = (1.0 - i / max(total_iters, 1)) * args.learning_rate           # annealed as in :251-254
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        learner_smoke.rollout(learner, env)
        ev[1].record()
        if cli.sync_metrics:
            m = learner.update(lr)
        else:
            # diagnostics resolved one iteration late (PPOLearner.update_async): the host enqueues iteration i + 1 while the GPU
            # still runs iteration i; everything enqueued is still executed inside the timed region (synchronize at its end)
            pending.append(learner.update_async(lr))
            m = pending.pop(0).result() if len(pending) > 1 else None
        learner.start_iteration()
        ev[2].record()
        phase_events.append(ev)
        return m

    pending = []
    if cli.inject_host_stall_ms > 0:
        real_mb, count = learner._minibatch_hip, [0]

        def stalled_minibatch(*a, **kw):
            count[0] += 1
            if count[0] % 16 == 6:
                time.sleep(cli.inject_host_stall_ms / 1e3)
            return real_mb(*a, **kw)

        learner._minibatch_hip = stalled_minibatch

    for i in range(cli.warmup):
        one_step(i)
    timer.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(cli.steps):
        metrics = one_step(cli.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    while pending:
        metrics = pending.pop(0).result()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    timing_iters = cli.steps
    if not cli.no_kernel_timing:
        # the per-launch durations: one more iteration issued launch by launch with event brackets (outside the timed region; the
        # same kernels on the same buffers as the replayed slots / the eager launches of the timed region)
        graphs, learner._update_graphs = learner._update_graphs, None
        install_timing_hooks()
        cli.sync_metrics, saved_sync = True, cli.sync_metrics
        one_step(total_iters - 1)          # the eager path's own first iteration: its buffers come out of the allocator inside the brackets
        torch.cuda.synchronize()           # (config D's FC forward once read 194 us instead of 130) -- not the one that is reported
        timer.reset()
        one_step(total_iters - 1)
        torch.cuda.synchronize()
        cli.sync_metrics, learner._update_graphs, timing_iters = saved_sync, graphs, 1
    learner.flat.check_views()
    assert np.isfinite(metrics["loss"]), metrics

    if rank == 0:
        env_steps = world * N * T * cli.steps
        M = learner.minibatch_size
        out = {
            "metric": "env-steps/sec (SPS) PPO Breakout num_envs=1024 @1/2/4/8 MI355X; GAE HBM GB/s",
            "value": env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": cli.steps,
            "warmup": cli.warmup,
            "ms_per_step": elapsed / cli.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": CONFIGS[cli.config]["workload"] + ", synthetic 84x84x4 uint8 observations resident in HBM, "
                            "NatureCNN A=4, 4 epochs x 4 minibatches",
                "baseline_config": cli.config,
                "local_num_envs": N, "num_steps": T, "global_num_envs": world * N, "minibatch_rows": M,
                "parallelism": (f"dp{world} (one learner per GPU, " + ("peer-memory all-reduce of the flat f32 gradient over HIP IPC segments (csrc/dpcomm.hip); RCCL for "
                                                                         "the rendezvous and the timing only)" if getattr(learner, "_peer", None) is not None
                                                                         else "RCCL all-reduce of the flat f32 gradient)")) if not cli.same_device
                               else f"dp{world} SAME-DEVICE PLUMBING SMOKE (all ranks on cuda:0, gloo" + ("; gradients over HIP IPC segments" if getattr(learner, "_peer", None) is not None else "")
                                    + "): not a measurement",
                "env": "device-resident synthetic generator (no PCIe in the timed region)",
                "rollout": (f"{len(learner._rollout_graphs)} hipGraph(s) of {-(-T // len(learner._rollout_graphs))} env step(s) each (policy "
                            "forward, sampling, env step, observation store)")
                           if getattr(learner, "_rollout_graphs", None) else "kernel-by-kernel launches",
                "update": update_mode,
                "diagnostics": "synchronous (one device synchronisation per iteration)" if cli.sync_metrics
                               else "read one iteration late (PPOLearner.update_async): the host runs an iteration ahead of the GPU",
                "cnn": "(filled in below from the kernels that ran)" if learner.fused_cnn else "torch Conv2d (MIOpen)",
            },
            "final_loss": metrics["loss"],
        }
        if cli.collective_preflight:
            out["collective_preflight"] = cli.collective_preflight
        if learner.fused_cnn:
            out["config"]["cnn"], out["matrix_arithmetic"] = describe_cnn(kernel_of, M)
        timed = phase_events[cli.warmup:cli.warmup + cli.steps]
        out["phases_ms"] = {"rollout_incl_gae": float(np.mean([e[0].elapsed_time(e[1]) for e in timed])),
                            "update": float(np.mean([e[1].elapsed_time(e[2]) for e in timed])),
                            "note": "GPU-timeline split of ms_per_step (events on the learner's stream)"}
        if not cli.no_kernel_timing and learner.fused_cnn:
            # dominant kernel of the path = the conv launch with the largest total time inside the timed region
            tot = {k: timer.mean_us(k)[0] * timer.mean_us(k)[1] * (16 if k.endswith(f"@{N}") else 1) for k in conv_flops
                   if timer.mean_us(k)[1] > 0}          # (rollout-sized launches inside captured graphs carry no event brackets)
            dom = max(tot, key=tot.get)
            us, n = timer.mean_us(dom)
            out["roofline"] = roofline_entry(dom, us, n, conv_flops[dom], kernel_of[dom], tot[dom] / timing_iters / (elapsed / cli.steps * 1e6),
                                             _traffic_of(dom))
            # north_star's HBM figure: the one HBM-bound launch of the conv stack (kernel Q: uint8 rows in, f32 activations out)
            qk = f"conv1_fwd@{M}"
            if kernel_of.get(qk) == "Q" and timer.mean_us(qk)[1]:
                qus = timer.mean_us(qk)[0]
                out["hbm_frac"] = _conv1_fwd_bytes(M) / qus / 1e3 / HBM_PEAK_GBPS
                out["hbm_frac_note"] = (f"kernel Q ({qk}): {_conv1_fwd_bytes(M)} algorithmic bytes in {qus:.0f} us = "
                                        f"{_conv1_fwd_bytes(M) / qus / 1e3:.0f} GB/s of {HBM_PEAK_GBPS:.0f} GB/s")
            it_flops = iteration_flops(N, T, M, cli.n_actions, int(args.update_epochs))
            out["iteration_f32_equiv_TFLOPs"] = it_flops / (elapsed / cli.steps) / 1e12
            out["iteration_f32_equiv_note"] = (f"{it_flops / 1e12:.2f} algorithmic TFLOP per iteration (f32 convolution / GEMM flops of the rollout "
                                               "forwards and the update's forward, data- and weight-gradient passes; no term pairs, no padding) / "
                                               f"ms_per_step; the dense f32 MFMA peak is {MFMA_F32_PEAK_TFLOPS} TFLOP/s")
            gus, gn = timer.mean_us("gae")
            lus, ln = timer.mean_us("loss")
            gae_bytes = 20 * T * N + 8 * N
            loss_bytes = (8 * cli.n_actions + 28 + 8) * M
            out["kernels"] = {
                "gae": {"algorithmic_bytes": gae_bytes, "avg_us_event_bracket": gus, "GBps": gae_bytes / gus / 1e3,
                        "launches_timed": gn, "note": f"{T}x{N} is launch/latency-bound: see profiles/ for the kernel time"},
                "loss_fwd_bwd": {"algorithmic_bytes": loss_bytes, "avg_us_event_bracket": lus,
                                 "GBps": loss_bytes / lus / 1e3, "launches_timed": ln,
                                 "note": "one launch per minibatch on packed behaviour rows (one 32-byte gather per row; advantage "
                                         "statistics once per epoch, scalar fold once per update)"},
            }
            if world == 1:
                out["kernels"].update(hbm_regime_points(device))            # K1 / K3 where they ARE HBM-bound (north_star's >= 60 % figure)
                probe = hbm_stream_probe()
                if probe is not None:
                    out["hbm_stream_probe"] = probe
            for k in sorted(tot):
                kus, kn = timer.mean_us(k)
                launches = kn * (16 if k.endswith(f"@{N}") else 1)          # rollout-sized launches are sampled 1 in 16
                if k.split("@")[0] not in ALG_BYTES_PER_IMAGE:             # (trunk_fwd: the three forwards in one call of the f32-pipe route)
                    out["kernels"][k] = {"kernel": kernel_of[k], "avg_us": kus, "launches_timed": kn, "TFLOPs": conv_flops[k] / kus / 1e6,
                                         "ms_per_step": kus * launches / timing_iters / 1e3}
                    continue
                e = roofline_entry(k, kus, kn, conv_flops[k], kernel_of[k], 0.0, _traffic_of(k))      # the same two-roof pricing for every launch
                out["kernels"][k] = {"kernel": kernel_of[k], "pipe": e["mfma_pipe"], "avg_us": kus, "launches_timed": kn,
                                     "bound": e["bound"], "frac": e["frac"], "hbm_frac": e["hbm_frac"], "mfma_frac": e["mfma_frac"],
                                     "GBps": e["hbm_GBps"], "TFLOPs": e["algorithmic_TFLOPs"], "frac_of_f32_mfma_peak": e["frac_of_f32_mfma_peak"],
                                     "ms_per_step": kus * launches / timing_iters / 1e3, "hbm_bytes_per_launch_pmc": _traffic_of(k)}
        elif not cli.no_kernel_timing:
            us, n = timer.mean_us("obs_gather")
            alg = OBS_ROW_BYTES * 5 * M
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get("obs_u8_to_f32_kernel", {}).get("hbm_bytes_per_launch")
            out["roofline"] = {
                "kernel": "obs_u8_to_f32_kernel<true> (K5 uint8 gather + /255 convert, minibatch launch)",
                "bound": "hbm", "achieved": alg / us / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": alg / us / 1e3 / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg, "avg_launch_us": us, "launches_timed": n,
                "frac_of_measured_achievable_6290": alg / us / 1e3 / 6290.0,
            }
            gus, gn = timer.mean_us("gae")
            lus, ln = timer.mean_us("loss")
            gae_bytes = 20 * T * N + 8 * N
            loss_bytes = (8 * cli.n_actions + 28 + 8) * M
            out["kernels"] = {
                "gae": {"algorithmic_bytes": gae_bytes, "avg_us_event_bracket": gus, "GBps": gae_bytes / gus / 1e3,
                        "launches_timed": gn, "note": "128x1024 is launch/latency-bound: see profiles/ for the kernel time"},
                "loss_fwd_bwd": {"algorithmic_bytes": loss_bytes, "avg_us_event_bracket_3_launches": lus,
                                 "GBps": loss_bytes / lus / 1e3, "launches_timed": ln},
            }
        if world == 1 and not cli.no_cpu_baseline:
            from oracle import cpu_ppo_port

            # `value` at the METRIC's configuration on THIS box's cores.  A whole iteration there is minutes of CPU work (290 s predicted on the
            # 256-thread GPU box, 170 s on 8 cores), so the default is the BOUNDED sample of cpu_ppo_port.run_metric_sample: every piece of the loop body
            # timed at its full size -- 16 of the 128 env steps at 1,024 envs, the GAE pass, one of the 16 minibatch updates of 32,768 rows (after one
            # untimed) -- times its count.  --cpu-baseline-full on: one WHOLE iteration instead (guarded by the first rollout's time).
            cb = None
            if cli.cpu_baseline_full == "on":
                cb = cpu_ppo_port.run(num_envs=N, num_steps=T, iterations=1, warmup_iterations=0, seed=cli.seed, n_actions=cli.n_actions,
                                      rollout_budget_s=120.0)
                if cb.get("aborted"):
                    print(f"bench.py: the whole-iteration CPU baseline was abandoned (its rollout alone took {cb['rollout_seconds']:.0f} s on "
                          f"{cb['cores']} threads); falling back to the bounded sample", file=sys.stderr, flush=True)
                    cb = None
            if cb is not None:
                out["cpu_baseline"] = {
                    "value": cb["sps_median"], "unit": "env-steps/s", "cores": cb["cores"], "kind": "port", "at_metric_config": True, "extrapolated": False,
                    "sample": f"ONE whole PPO iteration of the reference loop body (ppo_atari_envpool.py:217-341 restated in stock torch CPU ops, f32 "
                              f"observation storage, oracle/cpu_ppo_port.py) AT THE METRIC'S CONFIGURATION: num_envs={cb['num_envs']} x num_steps="
                              f"{cb['num_steps']} = {cb['env_steps']} env-steps, 16 updates of {cb['env_steps'] // 4} rows, in {cb['seconds']:.1f} s on this box's "
                              f"host cores (cpu_count={os.cpu_count()}, torch threads {cb['cores']}), no warm-up iteration; the reference script itself "
                              f"cannot run on this box (no envpool / gym / tyro, and /root/reference does not travel), hence kind=port",
                }
            elif cli.cpu_baseline_full == "auto" and cli.config == "C":
                cb = cpu_ppo_port.run_metric_sample(num_envs=N, num_steps=T, rollout_steps_timed=16, minibatches_timed=1, seed=cli.seed, n_actions=cli.n_actions)
                pc = cb["pieces"]
                out["cpu_baseline"] = {
                    "value": cb["sps"], "unit": "env-steps/s", "cores": cb["cores"], "kind": "port", "at_metric_config": True, "extrapolated": True,
                    "pieces": pc, "cpu_seconds_spent": cb["cpu_seconds_spent"],
                    "sample": f"a bounded sample of ONE PPO iteration of the reference loop body (ppo_atari_envpool.py:217-341 restated in stock torch CPU "
                              f"ops, f32 observation storage, oracle/cpu_ppo_port.py::run_metric_sample) AT THE METRIC'S CONFIGURATION, num_envs={cb['num_envs']} x "
                              f"num_steps={cb['num_steps']}: {pc['rollout_steps_timed']} of the {cb['num_steps']} env steps timed at {cb['num_envs']} envs "
                              f"({pc['rollout_step_s'] * 1e3:.0f} ms per step), the GAE pass over the whole rollout ({pc['gae_s'] * 1e3:.0f} ms), "
                              f"{pc['minibatches_timed']} of the {pc['minibatch_updates_per_iteration']} minibatch updates of {pc['minibatch_rows']} rows gathered "
                              f"from the full f32 observation buffer ({pc['minibatch_s']:.1f} s, after one untimed); iteration = {cb['num_steps']} x step + gae + "
                              f"{pc['minibatch_updates_per_iteration']} x minibatch = {cb['seconds']:.0f} s (a whole iteration is minutes of CPU work: "
                              f"--cpu-baseline-full on runs one); {cb['cpu_seconds_spent']:.0f} s of CPU work on this box's host cores (cpu_count={os.cpu_count()}, "
                              f"torch threads {cb['cores']}); the reference script itself cannot run on this box (no envpool / gym / tyro, and /root/reference "
                              f"does not travel), hence kind=port",
                }
            else:
                # one untimed warm-up iteration AT THE TIMED SHAPE (thread pool, oneDNN primitives and the allocator see the timed
                # shapes), then two timed iterations of the ppo_atari_envpool loop body; the median iteration is reported
                cb = cpu_ppo_port.run(num_envs=cli.cpu_baseline_envs, num_steps=128, iterations=2, warmup_iterations=1,
                                      seed=cli.seed, n_actions=cli.n_actions)
                out["cpu_baseline"] = {
                    "value": cb["sps_median"], "unit": "env-steps/s", "cores": cb["cores"], "kind": "port", "at_metric_config": False,
                    "sample": f"{cb['iterations']} full PPO iterations of the reference loop body (ppo_atari_envpool.py:217-341 restated in "
                              f"stock torch CPU ops, f32 observation storage, oracle/cpu_ppo_port.py) at num_envs={cb['num_envs']} x "
                              f"num_steps={cb['num_steps']} = {cb['env_steps']} env-steps in {cb['seconds']:.1f} s (median iteration "
                              f"reported; per-iteration s: {[round(x, 2) for x in cb['iteration_seconds']]}) after one untimed warm-up "
                              f"iteration of the same shape; host cpu_count={os.cpu_count()}; the reference script itself cannot run "
                              f"on this box (no envpool / gym / tyro, and /root/reference does not travel), hence kind=port; num_envs="
                              f"{cb['num_envs']}, not the metric's 1024 (--cpu-baseline-full on: one whole iteration at the metric's size, ~3 minutes on 8 "
                              f"cores); no extrapolation attempted",
                }
            # the reference's VERBATIM lines at the metric's full size were timed where /root/reference exists (the build container,
            # 8 cores, while minting tests/golden/atari_iteration_cfgC.npz): reported beside the port, never as `value`
            rpath = os.path.join(ROOT, "profiles", "r04_reference_lines_cpu_timing.json")
            if os.path.exists(rpath):
                r = json.load(open(rpath))
                out["cpu_baseline"]["reference_lines_at_metric_config"] = {
                    "value": r["env_steps_per_s"], "unit": "env-steps/s", "cores": r["cores"], "kind": "reference",
                    "sample": r["what"] + "; " + r["host"] + f"; {r['env_steps']} env-steps in {r['iteration_s']} s; " + r["note"]}
            # flat fields (a parser that keeps one level): both baselines at the METRIC's configuration (num_envs = 1024), timed on one
            # host -- the build container, where /root/reference exists -- so that they are comparable with each other
            # (profiles/r05_cpu_port_vs_reference_lines.json); `value` above stays the live measurement on this box's cores
            cpath = os.path.join(ROOT, "profiles", "r05_cpu_port_vs_reference_lines.json")
            if os.path.exists(cpath):
                c = json.load(open(cpath))
                out["cpu_baseline"].update({          # a CROSS-CHECK from another host (8 cores), never `value`
                    "cross_check_8_cores_reference_value": c["reference_lines"]["env_steps_per_s"], "cross_check_8_cores_reference_kind": "reference",
                    "cross_check_8_cores_port_value": c["port"]["env_steps_per_s"], "cross_check_8_cores_port_kind": "port",
                    "cross_check_8_cores_unit": "env-steps/s",
                    "cross_check_8_cores_sample": c["what"] + "; " + c["host"] + "; committed measurement, not taken on this box"})
        if world == 1 and not cli.no_pcie_inclusive:
            # never `value`: the same learner fed by HOST envs (numpy stand-ins on host threads), actions D2H and frames H2D
            # every step as in the reference's loop (:269-272), through the overlapped env-group lanes (cleanrl_amd/pipeline.py)
            del learner, env
            torch.cuda.empty_cache()
            for mod, name, orig in unhook:            # no event brackets around this leg's launches
                setattr(mod, name, orig)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import host_env_bench

            try:      # an auxiliary leg (worker processes, pinned shared memory): its failure must not cost the measured line
                ov = host_env_bench.run(N, T, iters=2, groups=cli.pcie_env_groups, frame_delta=True, device=device)
                se = host_env_bench.run(N, T, iters=1, groups=1, frame_delta=False, device=device)
                # the same lanes with envs that cost (almost) nothing: what the PIPELINE sustains -- D2H of the actions, the workers' hand-off,
                # H2D of the newest frames on the lanes' streams, the captured policy steps -- once the env cost is taken out
                ce = host_env_bench.run(N, T, iters=2, groups=cli.pcie_env_groups, frame_delta=True, device=device, static_frames=True)
                out["pcie_inclusive_sps"] = ov["sps"]
                out["pcie_inclusive"] = {"overlapped": ov, "serial_reference_arrangement": se, "pipeline_ceiling_static_frames": ce,
                                         "stand_in_env_step_us": host_env_bench.stand_in_step_us(N // cli.pcie_env_groups),
                                         "note": "whole PPO iterations with the envs on the host; not comparable with `value`, whose "
                                                 "inputs are resident in HBM.  `overlapped`: numpy stand-in envs (lane_step_us.env_wait_us is mostly the "
                                                 "stand-in's own step, stand_in_env_step_us: materialising four-frame stacks with np.take); "
                                                 "`pipeline_ceiling_static_frames`: the same lanes with envs that cost almost nothing"}
            except Exception as e:      # noqa: BLE001 -- reported in the line, loudly on stderr
                print(f"bench.py: the PCIe-inclusive leg failed: {type(e).__name__}: {e}", file=sys.stderr)
                out["pcie_inclusive_sps"] = None
                out["pcie_inclusive"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        if getattr(learner, "_peer", None) is not None:      # every rank is done with every segment before anybody unmaps / frees
            torch.cuda.synchronize()
            dist.barrier()
            learner._peer.close()
        dist.destroy_process_group()


def mlp_flops_per_row(O, D):
    """Algorithmic f32 flops of the two 64-64 tanh MLPs (actor with D outputs, critic) for one minibatch row of the update: forward,
    weight gradients of every layer, data gradients of layers 2 and 3."""
    fwd = lambda n: 2.0 * (O * 64 + 64 * 64 + 64 * n)          # noqa: E731
    dgrad = lambda n: 2.0 * (64 * 64 + 64 * n)                 # noqa: E731
    return sum(2 * fwd(n) + dgrad(n) for n in (D, 1))


def main_continuous(cli, rank, world, device):
    """``--config E``: BASELINE configs[4], ppo_continuous_action (HalfCheetah-v4-shaped obs 17 / act 6, 64 envs x 2048 steps,
    32 minibatches x 10 epochs; ppo_continuous_action.py:54-76 defaults) on the fused MLP kernel family K7 (csrc/mlp.hip): one
    launch per env step for both networks' forward + the Normal sample / log-prob, K1 GAE over (2048, 64), two launches per
    minibatch for gather + forwards + fused Normal loss + both backward passes + weight gradients, K6 clip + Adam.  The rollout
    replays as captured hipGraphs, the update as one hipGraph per (epoch, minibatch) slot (one GPU).  Env = device-resident
    stand-in stepped by its own kernel (no PCIe in the timed region).  MI355PPO_MLP=torch: the round-3 arrangement (library GEMMs
    behind K2' / K3') for A/B."""
    from cleanrl_amd import _lib, learner_smoke, ops
    from cleanrl_amd.agents import ContinuousAgent
    from cleanrl_amd.envs import DeviceSyntheticContinuousVecEnv
    from cleanrl_amd.learner import PPOLearner

    _lib.load(build_if_missing=True)
    N, T = cli.local_num_envs, cli.num_steps
    args = learner_smoke.default_args(num_steps=T, num_minibatches=32, update_epochs=10, learning_rate=3e-4, clip_coef=0.2,
                                      ent_coef=0.0, vf_coef=0.5)         # ppo_continuous_action.py:54-76
    seed = cli.seed + rank
    np.random.seed(seed)
    torch.manual_seed(cli.seed)
    env = DeviceSyntheticContinuousVecEnv(N, device, seed=seed)
    agent = ContinuousAgent(env).to(device)
    torch.manual_seed(seed)
    learner = PPOLearner(agent, args, env.single_observation_space, env.single_action_space, N, device, world_size=world,
                         sample_seed=seed)
    learner.observe(0, env.obs(), learner.dones[0])
    fused = learner.mlp is not None
    rollout_mode = "kernel-by-kernel launches"
    if fused and not cli.no_rollout_graphs:
        per = cli.rollout_steps_per_graph or 256          # 2048 steps x 2 launches: eight graphs of 512 kernel nodes
        learner.capture_rollout(env, steps_per_graph=per)
        rollout_mode = f"{len(learner._rollout_graphs)} hipGraph(s) of {min(per, T)} env steps each (fused act kernel + env step kernel)"
    update_mode = "eager launches"
    use_update_graphs = not cli.no_update_graphs
    if use_update_graphs:
        from cleanrl_amd.learner import update_graph_policy

        policy = update_graph_policy(world)
        use_update_graphs = learner.capture_update_agreed(log=lambda m: print(f"[bench] rank {rank}: {m}", file=sys.stderr, flush=True))
        if use_update_graphs:
            update_mode = ("one hipGraph per (epoch, minibatch) slot: fused MLP forward + loss + backward, fold, clip + Adam "
                           "(PPOLearner.capture_update); per-launch event brackets from an extra eager iteration after the timed region" if world == 1 else
                           "one hipGraph per (epoch, minibatch) slot with the gradient exchange INSIDE it (five small launches over HIP IPC segments, csrc/dpcomm.hip; "
                           "MI355PPO_ALLREDUCE=peer)" if getattr(learner, "_peer", None) is not None else
                           "two hipGraphs per (epoch, minibatch) slot with the all-reduce of the flat gradient between them: [fused MLP forward + loss + "
                           "backward, fold] | all-reduce | [clip + Adam] (PPOLearner.capture_update)")
        else:
            update_mode = ("eager launches (MI355PPO_UPDATE_GRAPHS=0)" if policy == "off"
                           else "eager launches (update-graph capture or its self-check failed on a rank; see stderr)")
    timer = KernelTimer()

    def install_timing_hooks():
        ops.gae = timer.wrap("gae", ops.gae)
        ops.ppo_loss_normal = timer.wrap("loss_normal", ops.ppo_loss_normal)
        ops.mlp_ppo_fwd_bwd = timer.wrap("mlp_ppo", ops.mlp_ppo_fwd_bwd)
        ops.clip_adam_ = timer.wrap("clip_adam", ops.clip_adam_)

    total_iters = cli.warmup + cli.steps      # (event brackets: never inside the timed region, see main())
    phase_events = []

    def one_step(i):
        lr = (1.0 - i / max(total_iters, 1)) * args.learning_rate
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        if getattr(learner, "_rollout_graphs", None):
            learner.replay_rollout()
        else:
            for step in range(T):                                                 # ppo_continuous_action.py:205-228
                action = learner.act(step)
                obs_dst, done_dst = learner._slot(step + 1)
                env.step_into(action, obs_dst, learner.rewards[step], done_dst)  # next_obs / reward / done straight into their rows
        learner.finish_rollout()
        ev[1].record()
        m = learner.update(lr)
        learner.start_iteration()
        ev[2].record()
        phase_events.append(ev)
        return m

    for i in range(cli.warmup):
        one_step(i)
    timer.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(cli.steps):
        metrics = one_step(cli.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if not cli.no_kernel_timing:
        graphs, learner._update_graphs = learner._update_graphs, None
        install_timing_hooks()
        one_step(total_iters - 1)          # (the eager path's first iteration allocates inside the brackets: not the one reported)
        torch.cuda.synchronize()
        timer.reset()
        one_step(total_iters - 1)
        torch.cuda.synchronize()
        learner._update_graphs = graphs
    learner.flat.check_views()
    assert np.isfinite(metrics["loss"]), metrics
    if rank == 0:
        M, D, O = learner.minibatch_size, env.act_dim, env.obs_dim
        out = {
            "metric": "env-steps/sec (SPS) PPO Breakout num_envs=1024 @1/2/4/8 MI355X; GAE HBM GB/s",
            "value": world * N * T * cli.steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": cli.steps,
            "warmup": cli.warmup, "ms_per_step": elapsed / cli.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": CONFIGS["E"]["workload"] + "; NOT the configuration the metric is quoted on (that is "
                                   "--config C): a parity / coverage workload of BASELINE.json's configs list",
                       "baseline_config": "E", "local_num_envs": N, "num_steps": T, "minibatch_rows": M, "obs_dim": O, "act_dim": D,
                       "parallelism": f"dp{world}", "rollout": rollout_mode, "update": update_mode,
                       "env": "device-resident linear-system stand-in, one kernel per step (no PCIe in the timed region)",
                       "network": "two 64-64 tanh MLPs on the fused MLP kernel family K7 (csrc/mlp.hip: f32 v_fma, lane = hidden unit, "
                                  "activations broadcast through wave-private LDS)" if fused
                                  else "two 64-64 tanh MLPs as torch Linear layers (library GEMMs; MI355PPO_MLP=torch)"},
            "final_loss": metrics["loss"],
        }
        timed = phase_events[cli.warmup:cli.warmup + cli.steps]
        out["phases_ms"] = {"rollout_incl_gae": float(np.mean([e[0].elapsed_time(e[1]) for e in timed])),
                            "update": float(np.mean([e[1].elapsed_time(e[2]) for e in timed])),
                            "note": "GPU-timeline split of ms_per_step (events on the learner's stream)"}
        if not cli.no_kernel_timing:
            alg = {"gae": 20 * T * N + 8 * N,
                   "loss_normal": (12 * D + 28 + 8) * M,                           # mean, action in; dmean out; 5 behaviour scalars, new value; dvalue; indices
                   "mlp_ppo": (4 * O + 4 * D + 16 + 8) * M + 2 * 4 * learner.flat.numel,   # obs row, action row, 4 behaviour scalars, index; parameters in, gradients out
                   "clip_adam": 36 * learner.flat.numel}
            ks = {}
            for k, b in alg.items():
                us, n = timer.mean_us(k)
                if n:
                    ks[k] = {"algorithmic_bytes": b, "avg_us_event_bracket": us, "GBps": b / us / 1e3, "launches_timed": n,
                             "frac_of_hbm_peak": b / us / 1e3 / HBM_PEAK_GBPS}
            out["kernels"] = ks
            if "mlp_ppo" in ks:
                fl = mlp_flops_per_row(O, D) * M
                us = ks["mlp_ppo"]["avg_us_event_bracket"]
                ks["mlp_ppo"].update({"algorithmic_flops": fl, "TFLOPs": fl / us / 1e6})
                out["roofline"] = {
                    "kernel": f"mlp_ppo_kernel + mlp_fold_kernel (K7, csrc/mlp.hip): one minibatch of {M} rows -- gather, both MLPs' forward, Normal "
                              "log-prob, PPO loss row terms, both backward passes, weight gradients (two launches, bracketed together)",
                    "bound": "mfma", "achieved": fl / us / 1e6, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": fl / us / 1e6 / MFMA_F32_PEAK_TFLOPS, "traffic": None, "pipe": "f32 VALU (v_fma_f32; the f32 vector peak equals the "
                    "dense f32 MFMA peak, 157.3 TFLOP/s)", "algorithmic_flops_per_launch": fl, "avg_launch_us": us,
                    "launches_timed": ks["mlp_ppo"]["launches_timed"],
                    "note": f"{fl / 1e6:.0f} MFLOP and {alg['mlp_ppo']} algorithmic bytes per minibatch: 1.6 us at the f32 peak -- the launch pair is "
                            "bound by latency (event brackets include the launch gaps), not by a pipe or by HBM"}
            elif "loss_normal" in ks:
                dom = "loss_normal"
                out["roofline"] = {
                    "kernel": "loss_normal_main (K3', csrc/loss.hip): fused Normal log_prob + entropy + clipped surrogate + value loss, "
                              f"forward + backward, one minibatch of {M} rows x {D} action dims",
                    "bound": "hbm", "achieved": ks[dom]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ks[dom]["frac_of_hbm_peak"], "traffic": None,
                    "algorithmic_bytes_per_launch": alg[dom], "avg_launch_us": ks[dom]["avg_us_event_bracket"],
                    "launches_timed": ks[dom]["launches_timed"],
                    "note": f"{alg[dom]} B per launch is cache-resident: the launch is latency-bound (event brackets include launch overhead)",
                }
        print(json.dumps(out), flush=True)
    if world > 1:
        if getattr(learner, "_peer", None) is not None:      # every rank is done with every segment before anybody unmaps / frees
            torch.cuda.synchronize()
            dist.barrier()
            learner._peer.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
