"""ctypes binding of ``libmi355ppo.so`` (C ABI declared in ``include/mi355ppo.h``).

The HIP library is mandatory for every GPU code path of this package: if it is missing and cannot
be built, loading raises -- there is no silent eager/PyTorch fallback for CUDA tensors.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmi355ppo.so")

_P = c_void_p  # every tensor argument is a raw device pointer

# name -> (restype, argtypes); must list every MI355PPO_API symbol of include/mi355ppo.h
SIGNATURES = {
    "mi355ppo_version": (c_int, []),
    "mi355ppo_last_error": (c_char_p, []),
    "mi355ppo_init": (c_int, [c_int]),
    "mi355ppo_gae_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, _P]),
    "mi355ppo_gae_f32_variant": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, c_int, _P]),
    "mi355ppo_categorical_sample_f32": (c_int, [_P, _P, c_uint64, c_uint64, _P, _P, _P, _P, c_int, c_int, _P]),
    "mi355ppo_categorical_sample_ctr_f32": (c_int, [_P, _P, c_uint64, c_uint64, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "mi355ppo_categorical_logprob_entropy_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    "mi355ppo_categorical_logprob_entropy_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "mi355ppo_normal_logprob_entropy_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "mi355ppo_normal_sample_f32": (c_int, [_P, _P, _P, c_uint64, c_uint64, _P, _P, _P, c_int, c_int, _P]),
    "mi355ppo_normal_logprob_entropy_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    "mi355ppo_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mi355ppo_adv_stats_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "mi355ppo_adv_stats_f32": (c_int, [_P, _P, c_int64, c_int, _P, _P, c_size_t, _P]),
    "mi355ppo_loss_scalars_f32": (c_int, [_P, c_size_t, c_int, _P, _P]),
    "mi355ppo_loss_categorical_fwd_bwd_f32": (
        c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, c_double, c_int, c_int, _P,
                _P, _P, _P, _P, c_size_t, _P]),
    "mi355ppo_batch_pack_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, _P]),
    "mi355ppo_adv_stats_packed_f32": (c_int, [_P, _P, c_int64, c_int, _P, _P, c_size_t, _P]),
    "mi355ppo_loss_categorical_packed_fwd_bwd_f32": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_double, c_double, c_double, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mi355ppo_loss_normal_fwd_bwd_f32": (
        c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, c_double, c_int, c_int, _P,
                _P, _P, _P, _P, _P, c_size_t, _P]),
    "mi355ppo_obs_u8_to_f32": (c_int, [_P, _P, _P, c_int64, c_int64, c_int, _P]),
    "mi355ppo_obs_nchw_to_nhwc_u8": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "mi355ppo_obs_shift_append_u8_c4": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "mi355ppo_clip_adam_workspace_bytes": (c_size_t, [c_int64]),
    "mi355ppo_clip_adam_f32": (
        c_int, [_P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_double, c_double, c_int64,
                _P, _P, c_size_t, _P]),
    "mi355ppo_dp_comm_create": (c_int, [c_int, c_int, c_int64, c_double, _P]),
    "mi355ppo_dp_comm_handle": (c_int, [_P, _P]),
    "mi355ppo_dp_comm_connect": (c_int, [_P, _P]),
    "mi355ppo_dp_allreduce_sum_f32": (c_int, [_P, _P, c_int64, _P]),
    "mi355ppo_dp_comm_status": (c_int, [_P, _P, _P, _P]),
    "mi355ppo_dp_comm_destroy": (c_int, [_P]),
    "mi355ppo_cnn_repack_weights_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "mi355ppo_cnn_conv_fwd_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "mi355ppo_cnn_conv_dgrad_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "mi355ppo_cnn_conv_fwd_f32_variant": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "mi355ppo_cnn_conv_dgrad_f32_variant": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "mi355ppo_cnn_conv_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "mi355ppo_cnn_conv_wgrad_kernel": (c_int, [c_int64, c_int]),
    "mi355ppo_cnn_conv_packed_kernel_f16x2": (c_int, [c_int64, c_int, c_int]),
    "mi355ppo_cnn_conv_wgrad_kernel_f16x2": (c_int, [c_int64, c_int]),
    "mi355ppo_cnn_conv_wgrad_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, c_size_t, _P]),
    "mi355ppo_cnn_trunk_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "mi355ppo_cnn_conv1q_pack_bytes": (c_size_t, []),
    "mi355ppo_cnn_conv1q_pack": (c_int, [_P, _P, _P]),
    "mi355ppo_cnn_conv1q_fwd": (c_int, [_P, _P, _P, _P, _P, c_int64, _P]),
    "mi355ppo_fc_pack_bytes": (c_size_t, [c_int, c_int]),
    "mi355ppo_fc_pack_f32": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "mi355ppo_fc_fwd_relu_packed_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "mi355ppo_fc_dgrad_mask_packed_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "mi355ppo_fc_fwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mi355ppo_fc_fwd_relu_packed_ws_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "mi355ppo_cnn_conv_fwd_packed_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "mi355ppo_cnn_conv_dgrad_packed_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "mi355ppo_cnn_conv1q_fwd_bits": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, _P]),
    "mi355ppo_cnn_conv_fwd_packed_bits_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "mi355ppo_cnn_conv_dgrad_packed_bits_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "mi355ppo_fc_dgrad_maskbits_packed_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "mi355ppo_synth_atari_step_u8": (c_int, [_P, c_int, _P, c_uint64, c_uint64, _P, _P, _P, c_int, c_double, c_int, _P]),
    "mi355ppo_synth_atari_step_ctr_u8": (c_int, [_P, c_int, _P, c_uint64, c_uint64, _P, _P, _P, _P, c_int, c_double, c_int, _P]),
    "mi355ppo_synth_atari_step_hwc_ctr_u8": (c_int, [_P, c_int, _P, c_uint64, c_uint64, _P, _P, _P, _P, c_int, c_double, c_int, _P]),
    "mi355ppo_heads_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "mi355ppo_heads_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mi355ppo_heads_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "mi355ppo_heads_bwd_relu_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "mi355ppo_nature_packs_f32": (c_int, [_P] * 12),
    "mi355ppo_fc_heads_act_categorical_f32": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_uint64, c_uint64, _P,
                                                      _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mi355ppo_fc_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mi355ppo_fc_wgrad_kernel": (c_int, [c_int, c_int, c_int]),
    "mi355ppo_fc_wgrad_kernel_f16x2": (c_int, [c_int, c_int, c_int]),
    "mi355ppo_fc_packed_kernel_f16x2": (c_int, [c_int, c_int, c_int, c_int]),
    "mi355ppo_fc_wgrad_f32": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "mi355ppo_adam_schedule_f32": (c_int, [c_double, c_double, c_double, c_int64, _P]),
    "mi355ppo_clip_adam_sched_f32": (
        c_int, [_P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_double, _P, _P, _P, c_size_t, _P]),
    # K7: the fused MLP family (csrc/mlp.hip); a network = host array of six device pointers
    "mi355ppo_mlp_fwd_f32": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, _P, _P]),
    "mi355ppo_mlp_act_categorical_f32": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, c_uint64, c_uint64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi355ppo_mlp_act_normal_f32": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, _P, c_uint64, c_uint64, _P, _P, _P, _P, _P, _P, _P]),
    "mi355ppo_mlp_ppo_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mi355ppo_mlp_ppo_categorical_fwd_bwd_f32": (
        c_int, [_P, _P, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, c_double, c_double, c_double, c_int, c_int, _P, _P, _P, _P, c_int,
                _P, c_size_t, _P]),
    "mi355ppo_mlp_ppo_normal_fwd_bwd_f32": (
        c_int, [_P, _P, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_double, c_double, c_double, c_int, c_int, _P, _P, _P, _P,
                _P, c_int, _P, c_size_t, _P]),
    "mi355ppo_synth_continuous_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_uint64, _P, _P, c_double, _P, _P, _P, _P, c_int, c_int,
                                                   c_int, _P]),
    # round 5: the NatureCNN GEMMs on the two-term f16 split (csrc/f16split.h): amax records in, amax records out
    "mi355ppo_absmax_f32": (c_int, [_P, c_int64, _P, _P]),
    "mi355ppo_fc_pack_f16x2_bytes": (c_size_t, [c_int, c_int]),
    "mi355ppo_fc_pack_f16x2_f32": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "mi355ppo_nature_packs_f16x2_f32": (c_int, [_P] * 13),
    "mi355ppo_cnn_conv1q_fwd_amax": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, _P, _P]),
    "mi355ppo_cnn_conv_fwd_packed_f16x2_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, _P]),
    "mi355ppo_cnn_conv_dgrad_packed_f16x2_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, _P]),
    "mi355ppo_cnn_conv_wgrad_f16x2_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P, c_size_t, _P, _P, _P]),
    "mi355ppo_cnn_conv1_wgrad_f16x2": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, c_size_t, _P, _P]),
    "mi355ppo_fc_fwd_relu_packed_f16x2_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "mi355ppo_fc_dgrad_packed_f16x2_f32": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "mi355ppo_fc_wgrad_f16x2_f32": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "mi355ppo_heads_bwd_relu_amax_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "mi355ppo_fc_heads_act_categorical_f16x2_f32": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_uint64, c_uint64, _P,
                                                            _P, _P, _P, _P, _P, _P, c_size_t, _P, _P]),
    # host-pointer twins (csrc/host_twins.hip): the device signatures minus stream / workspace
    "mi355ppo_gae_f32_cpu": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double]),
    "mi355ppo_categorical_sample_f32_cpu": (c_int, [_P, _P, c_uint64, c_uint64, _P, _P, _P, _P, c_int, c_int]),
    "mi355ppo_categorical_logprob_entropy_f32_cpu": (c_int, [_P, _P, _P, _P, _P, c_int, c_int]),
    "mi355ppo_categorical_logprob_entropy_bwd_f32_cpu": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int]),
    "mi355ppo_normal_sample_f32_cpu": (c_int, [_P, _P, _P, c_uint64, c_uint64, _P, _P, _P, c_int, c_int]),
    "mi355ppo_normal_logprob_entropy_f32_cpu": (c_int, [_P, _P, _P, _P, _P, c_int, c_int]),
    "mi355ppo_normal_logprob_entropy_bwd_f32_cpu": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int]),
    "mi355ppo_loss_categorical_fwd_bwd_f32_cpu": (
        c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, c_double, c_int, c_int, _P, _P, _P, _P]),
    "mi355ppo_loss_normal_fwd_bwd_f32_cpu": (
        c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, c_double, c_int, c_int, _P, _P, _P, _P, _P]),
    "mi355ppo_clip_adam_f32_cpu": (
        c_int, [_P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_double, c_double, c_int64, _P]),
    "mi355ppo_obs_u8_to_f32_cpu": (c_int, [_P, _P, _P, c_int64, c_int64, c_int]),
}

ABI_VERSION = 210       # == MI355PPO_VERSION of include/mi355ppo.h this binding was written against (major*100 + minor*10 + patch)

_lib = None


class Mi355PpoError(RuntimeError):
    """A libmi355ppo entry point returned a negative status."""


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load (once) and return the library with all prototypes declared."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled libamdhip64.so.7 must be the HIP runtime this library binds to.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH) and build_if_missing:
        try:
            from . import build as _build

            _build.build(verbose=True)
        except Exception as e:  # pragma: no cover - depends on the toolchain
            raise RuntimeError(
                f"libmi355ppo.so is missing at {LIB_PATH} and could not be built ({e}); run "
                "`python -m cleanrl_amd.build` (needs hipcc). The HIP library is mandatory: there is no fallback."
            ) from e
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"libmi355ppo.so is missing at {LIB_PATH}; run `python -m cleanrl_amd.build`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    v = lib.mi355ppo_version()
    if v // 10 != ABI_VERSION // 10:         # major and minor: a signature change bumps the minor (arguments would shift silently)
        raise RuntimeError(f"libmi355ppo.so reports ABI version {v}; this binding expects {ABI_VERSION // 10}x -- rebuild it "
                           "(`python -m cleanrl_amd.build`)")
    _lib = lib
    return lib


_inited: set = set()


def require_device(index: int) -> None:
    """Raise unless HIP device ``index`` can run the library's kernels (``mi355ppo_init``: gfx950 only).  Once per device."""
    if index in _inited:
        return
    lib = load()
    if lib.mi355ppo_init(int(index)) != 0:
        msg = lib.mi355ppo_last_error()
        raise RuntimeError(f"libmi355ppo cannot run on cuda:{index}: {msg.decode() if msg else 'mi355ppo_init failed'}")
    _inited.add(index)


def check(status: int, fn: str) -> None:
    if status != 0:
        msg = load().mi355ppo_last_error()
        raise Mi355PpoError(f"{fn} failed with status {status}: {msg.decode() if msg else ''}")
