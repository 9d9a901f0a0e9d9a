"""C-ABI surface checks that need no GPU: the library loads, exports every symbol include/mi355ppo.h
declares, the ctypes prototypes cover exactly that set, and argument validation fails loudly BEFORE any
HIP call (so these run on a CPU-only box)."""
import ctypes
import os
import re

import pytest
import torch

from cleanrl_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mi355ppo.h")).read()
    return set(re.findall(r"MI355PPO_API\s+[\w\s\*]+?\b(mi355ppo_\w+)\s*\(", src))


def test_header_and_binding_declare_the_same_symbols():
    hs = header_symbols()
    assert len(hs) >= 14
    assert hs == set(_lib.SIGNATURES)


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_symbols():
        assert getattr(lib, name) is not None
    assert lib.mi355ppo_version() == 100


def test_validation_errors_are_loud_and_precede_any_launch():
    lib = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    mis = ctypes.c_void_p(p.value + 2)
    # null pointer
    assert lib.mi355ppo_gae_f32(None, p, p, p, p, p, p, 4, 4, 0.99, 0.95, None) == -1
    assert b"null" in lib.mi355ppo_last_error()
    # bad shape
    assert lib.mi355ppo_gae_f32(p, p, p, p, p, p, p, 0, 4, 0.99, 0.95, None) == -1
    assert b"positive" in lib.mi355ppo_last_error()
    # misaligned
    assert lib.mi355ppo_gae_f32(mis, p, p, p, p, p, p, 4, 4, 0.99, 0.95, None) == -2
    # unsupported action count
    assert lib.mi355ppo_categorical_sample_f32(p, None, 0, 0, p, None, p, None, 4, 65, None) == -1
    # workspace too small / missing
    assert lib.mi355ppo_loss_categorical_fwd_bwd_f32(p, p, None, p, p, p, p, p, 8, 4, 0.1, 0.01, 0.5, 1, 1, p, p, p,
                                                     None, 0, None) == -4
    assert lib.mi355ppo_loss_categorical_fwd_bwd_f32(p, p, None, p, p, p, p, p, 8, 4, 0.1, 0.01, 0.5, 1, 1, p, p, p,
                                                     p, 8, None) == -4
    assert b"workspace" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_loss_workspace_bytes(32768, 0) == (2 * 128 + 128 * 6) * 8
    assert lib.mi355ppo_clip_adam_f32(p, p, p, p, 16, 1.0, 0.5, 1e-3, 0.9, 0.999, 1e-5, 0, None, p, 4096, None) == -1
    assert lib.mi355ppo_obs_u8_to_f32(p, None, p, 4, 6, 1, None) == -1   # row_bytes % 4 != 0


def test_ops_refuse_cpu_tensors_there_is_no_fallback():
    x = torch.zeros(4, 4)
    with pytest.raises(TypeError, match="CUDA/HIP"):
        ops.gae(x, x, x, x[0], x[0], 0.99, 0.95)
    with pytest.raises(TypeError, match="CUDA/HIP"):
        ops.categorical_sample(x)
    with pytest.raises(TypeError, match="CUDA/HIP"):
        ops.obs_u8_to_f32(torch.zeros(2, 8, dtype=torch.uint8))
