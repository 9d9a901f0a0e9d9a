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
HEADER = os.path.join(ROOT, "include", "mi355ppo.h")


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
    assert lib.mi355ppo_version() == _lib.ABI_VERSION == int(re.search(r"#define MI355PPO_VERSION (\d+)", open(HEADER).read()).group(1))


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
    assert lib.mi355ppo_loss_categorical_fwd_bwd_f32(p, p, None, p, p, p, p, p, 8, 4, 0.1, 0.01, 0.5, 1, 1, None, p, p, p,
                                                     None, 0, None) == -4
    assert lib.mi355ppo_loss_categorical_fwd_bwd_f32(p, p, None, p, p, p, p, p, 8, 4, 0.1, 0.01, 0.5, 1, 1, None, p, p, p,
                                                     p, 8, None) == -4
    assert b"workspace" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_loss_workspace_bytes(32768, 0) == 64 + (2 * 1024 + 2048 * 6) * 8   # slot head + statistics partials + 6 sums per workgroup
    assert lib.mi355ppo_loss_workspace_bytes(4096, 6) == 64 + (2 * 1024 + 2048 * 12) * 8
    assert lib.mi355ppo_clip_adam_f32(p, p, p, p, 16, 1.0, 0.5, 1e-3, 0.9, 0.999, 1e-5, 0, None, p, 4096, None) == -1
    assert lib.mi355ppo_obs_u8_to_f32(p, None, p, 4, 6, 1, None) == -1   # row_bytes % 4 != 0


def test_peer_memory_exchange_entry_points_validate_and_fail_loudly_without_a_device(monkeypatch):
    """``mi355ppo_dp_*`` (ABI 2.0, header section a9/e): argument checks come first; without a HIP device the communicator cannot be created -- a
    negative status and a message, no crash, no CPU stand-in; the Python policy accepts exactly 'peer' and 'pg'."""
    import torch

    from cleanrl_amd import dp_comm

    lib = _lib.load()
    comm = ctypes.c_void_p()
    for world, rank, n, tmo in ((0, 0, 16, 1e3), (9, 0, 16, 1e3), (2, 2, 16, 1e3), (2, -1, 16, 1e3), (2, 0, 0, 1e3), (2, 0, 16, 0.0)):
        assert lib.mi355ppo_dp_comm_create(world, rank, n, tmo, ctypes.byref(comm)) == -1 and not comm.value
    assert b"world in 1..8" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_dp_comm_create(2, 0, 16, 1e3, None) == -1
    assert lib.mi355ppo_dp_comm_handle(None, None) == -1 and lib.mi355ppo_dp_comm_connect(None, None) == -1
    assert lib.mi355ppo_dp_allreduce_sum_f32(None, None, 4, None) == -1
    assert lib.mi355ppo_dp_comm_status(None, None, None, None) == -1
    assert lib.mi355ppo_dp_comm_destroy(None) == 0
    if not torch.cuda.is_available():
        assert lib.mi355ppo_dp_comm_create(2, 0, 16, 1e3, ctypes.byref(comm)) == -3 and not comm.value
        assert b"mi355ppo_dp_comm_create" in lib.mi355ppo_last_error()
    assert dp_comm.MAX_WORLD == 8 and dp_comm.HANDLE_BYTES == 64
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mi355ppo.h")).read()
    assert "#define MI355PPO_DP_MAX_WORLD 8" in hdr and "#define MI355PPO_DP_HANDLE_BYTES 64" in hdr
    monkeypatch.delenv("MI355PPO_ALLREDUCE", raising=False)
    assert dp_comm.exchange_policy(8) == "pg"
    monkeypatch.setenv("MI355PPO_ALLREDUCE", "peer")
    assert dp_comm.exchange_policy(8) == "peer" and dp_comm.exchange_policy(1) == "pg"
    monkeypatch.setenv("MI355PPO_ALLREDUCE", "peer:2.5")
    assert dp_comm.exchange_policy(2) == "peer" and dp_comm._setting() == ("peer", 2.5)
    for bad in ("pg:3", "peer:0", "peer:x"):
        monkeypatch.setenv("MI355PPO_ALLREDUCE", bad)
        with pytest.raises(ValueError):
            dp_comm.exchange_policy(2)
    monkeypatch.setenv("MI355PPO_ALLREDUCE", "ring")
    with pytest.raises(ValueError):
        dp_comm.exchange_policy(2)


def test_init_reports_a_box_without_a_device_loudly():
    """mi355ppo_init on this CPU-only box: no HIP device -> EHIP and a message; on the GPU box the same call is part of
    tests/test_gpu_kernels.py (0 for the MI355X, EINVAL for an ordinal out of range)."""
    lib = _lib.load()
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu-marked test")
    assert lib.mi355ppo_init(0) == -3
    assert b"no HIP device" in lib.mi355ppo_last_error()
    with pytest.raises(RuntimeError, match="cannot run on cuda:0"):
        _lib.require_device(0)


def test_ops_refuse_cpu_tensors_there_is_no_fallback():
    x = torch.zeros(4, 4)
    with pytest.raises(TypeError, match="CUDA/HIP"):
        ops.gae(x, x, x, x[0], x[0], 0.99, 0.95)
    with pytest.raises(TypeError, match="CUDA/HIP"):
        ops.categorical_sample(x)
    with pytest.raises(TypeError, match="CUDA/HIP"):
        ops.obs_u8_to_f32(torch.zeros(2, 8, dtype=torch.uint8))


def test_cnn_and_heads_entry_points_validate_before_any_launch():
    """The convolution / heads entry points: bad layer, variant, sizes, pointers and workspaces fail with a code and a
    message, without touching a device."""
    lib = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mi355ppo_cnn_repack_weights_f32(p, p, 4, 0, None) == -1                 # no such layer
    assert lib.mi355ppo_cnn_repack_weights_f32(p, p, 1, 1, None) == -1                 # mode 1 is a layer-3 layout
    assert lib.mi355ppo_cnn_repack_weights_f32(p, p, 2, 3, None) == -1                 # mode 3 likewise
    assert lib.mi355ppo_cnn_conv_fwd_f32(None, None, p, p, p, 8, 1, None) == -1
    assert b"null" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_cnn_conv_fwd_f32(p, p, p, p, p, 8, 2, None) == -1              # row gather only exists for layer 1
    assert b"inds" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_cnn_conv_fwd_f32(p, None, p, p, p, 0, 1, None) == -1
    assert lib.mi355ppo_cnn_conv_fwd_f32_variant(p, None, p, p, p, 8, 1, 9, None) == -1
    assert b"variant" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_cnn_conv_dgrad_f32(p, p, p, p, 8, 1, None) == -1               # conv1's input needs no gradient
    assert lib.mi355ppo_cnn_conv_dgrad_f32_variant(p, p, p, p, 8, 2, 5, None) == -1    # border classes are layer 3 only
    need = lib.mi355ppo_cnn_conv_wgrad_workspace_bytes(32768, 2)
    assert need == (512 + 16) * 64 * 512 * 4 + (512 + 16) * 64 * 4
    assert lib.mi355ppo_cnn_conv_wgrad_workspace_bytes(32768, 1) == (2048 + 64) * 32 * 256 * 4 + (2048 + 64) * 32 * 4
    assert lib.mi355ppo_cnn_conv_wgrad_workspace_bytes(0, 2) == 0
    assert lib.mi355ppo_cnn_conv_wgrad_f32(p, None, p, p, p, 8, 2, p, 16, None) == -4
    assert b"workspace" in lib.mi355ppo_last_error()
    assert lib.mi355ppo_heads_fwd_f32(p, p, p, p, p, p, p, 8, 19, 512, None) == -1     # A must be 1..18 (round 6; 1..7 before)
    assert lib.mi355ppo_heads_fwd_f32(p, p, p, p, p, p, p, 8, 4, 256, None) == -1      # hidden width is 512
    assert lib.mi355ppo_heads_bwd_workspace_bytes(32768, 4) == 512 * 6 * 513 * 4     # A + 1 rows, + 1 for the ReLU variant's bias gradient
    assert lib.mi355ppo_heads_bwd_f32(p, p, p, p, p, p, p, p, p, p, 8, 4, 512, None, 0, None) == -4
    assert lib.mi355ppo_heads_bwd_relu_f32(p, p, p, p, p, p, 514, p, p, p, p, p, 8, 4, 512, p, 1 << 30, None) == -1     # pitch: a multiple of 4
    assert lib.mi355ppo_heads_bwd_relu_f32(p, p, p, p, p, p, 516, p, p, p, p, None, 8, 4, 512, p, 1 << 30, None) == -1  # dbh is required
    assert lib.mi355ppo_fc_wgrad_workspace_bytes(32768, 512, 3136) == 9 * 512 * 3136 * 4       # nine batch slabs
    assert lib.mi355ppo_fc_wgrad_workspace_bytes(700, 512, 3136) == 512 * 3136 * 4
    assert lib.mi355ppo_fc_wgrad_f32(p, 512, p, p, 8, 500, 3136, 64, p, 1 << 30, None) == -1   # N % 32
    assert lib.mi355ppo_fc_wgrad_f32(p, 512, p, p, 8, 512, 3136, 60, p, 1 << 30, None) == -1   # channels must divide K
    assert lib.mi355ppo_fc_wgrad_f32(p, 512, p, p, 8, 512, 3136, 64, p, 16, None) == -4
    assert lib.mi355ppo_synth_atari_step_u8(p, 0, p, 1, 1, p, p, p, 4, 0.01, 1, None) == -1


def test_which_kernel_the_round_6_fc_and_weight_gradient_launches_take(monkeypatch):
    """The host queries bench.py labels its rows with (round 6): kernel G for the FC forward from 16,384 rows / the bit-masked data gradient from
    1,024; kernel H for the FC weight gradient from 4,096 rows; kernel U for every conv weight gradient that fits the 32-bit buffer range -- each
    the launcher's own decision, with its switch read at every call."""
    lib = _lib.load()
    for name in ("MI355PPO_FC_G", "MI355PPO_FC_H", "MI355PPO_CONV_U"):
        monkeypatch.delenv(name, raising=False)
    g = lambda M, dgrad, N=512, K=3136: chr(lib.mi355ppo_fc_packed_kernel_f16x2(M, N, K, dgrad))
    assert [g(M, 0) for M in (1024, 8192, 16383, 16384, 32768)] == ["Z", "Z", "Z", "G", "G"]
    assert [g(M, 1, 3136, 512) for M in (128, 1023, 1024, 8192, 32768)] == ["Z", "Z", "G", "G", "G"]
    assert g(32768, 0, 500, 3136) == "Z" and g(32768, 0, 512, 48) == "Z"                  # N % 32, K % 64: shapes kernel G does not take
    h = lambda M: chr(lib.mi355ppo_fc_wgrad_kernel_f16x2(M, 512, 3136))
    assert [h(M) for M in (1000, 1024, 4095, 4096, 32768)] == ["Y", "W", "Y", "H", "H"] and chr(lib.mi355ppo_fc_wgrad_kernel_f16x2(8192, 500, 3136)) == "Y"
    u = lambda images, layer: chr(lib.mi355ppo_cnn_conv_wgrad_kernel_f16x2(images, layer))
    assert [u(n, l) for n in (1, 256, 32768) for l in (1, 2, 3)] == ["U"] * 9
    assert u(90000, 2) == "T" and u(90000, 1) == "P" and u(90000, 3) == "U" and lib.mi355ppo_cnn_conv_wgrad_kernel_f16x2(64, 4) == 0      # beyond 4 GiB: kernel U declines
    monkeypatch.setenv("MI355PPO_FC_G", "0"); monkeypatch.setenv("MI355PPO_FC_H", "0"); monkeypatch.setenv("MI355PPO_CONV_U", "0")
    assert g(32768, 0) == "Z" and g(32768, 1, 3136, 512) == "Z" and h(32768) == "W" and u(32768, 2) == "V" and u(32768, 1) == "P" and u(240, 3) == "T" and u(304, 3) == "V"
    monkeypatch.setenv("MI355PPO_FC_G", "min:1"); monkeypatch.setenv("MI355PPO_FC_H", "min:1")      # one variable per kernel family: "0", "1" / unset, "min:<n>"
    assert g(5, 0) == "G" and g(5, 1, 3136, 512) == "G" and h(5) == "H"
    monkeypatch.setenv("MI355PPO_FC_G", "1"); monkeypatch.setenv("MI355PPO_FC_H", "1")
    assert g(8192, 0) == "Z" and h(1024) == "W"
    # MI355PPO_CONV_U as a list of layers: "23" keeps layer 1 on kernel P, "13" layer 2 on kernel V
    monkeypatch.setenv("MI355PPO_CONV_U", "23")
    assert [u(32768, l) for l in (1, 2, 3)] == ["P", "U", "U"]
    monkeypatch.setenv("MI355PPO_CONV_U", "13")
    assert [u(32768, l) for l in (1, 2, 3)] == ["U", "V", "U"]
    monkeypatch.setenv("MI355PPO_CONV_U", "1")
    assert [u(32768, l) for l in (1, 2, 3)] == ["U", "U", "U"]
    # kernel V's batch rule with the slab count of the shape that would run (round-5 advisor note): 224 .. 288 images of layer 3 stay on kernel V
    # on the bf16 path (170 slabs); the f16 split's two-tile shape needs 227 slabs' worth
    assert chr(lib.mi355ppo_cnn_conv_wgrad_kernel(240, 3)) == "V" and chr(lib.mi355ppo_cnn_conv_wgrad_kernel(208, 3)) == "T"


def test_which_kernel_a_packed_f16x2_convolution_takes_is_a_host_side_decision(monkeypatch):
    """mi355ppo_cnn_conv_packed_kernel_f16x2 (gemmz.hip::conv_r_takes): kernel R (convr.hip) by default at every size except the
    layer-2 data gradient below 512 images; the switches of DESIGN 3.7 are read at every call."""
    lib = _lib.load()
    monkeypatch.delenv("MI355PPO_CONV_R", raising=False)
    k = lambda images, layer, dgrad: chr(lib.mi355ppo_cnn_conv_packed_kernel_f16x2(images, layer, dgrad))
    assert [k(n, 2, 0) for n in (1, 256, 32768)] == ["R"] * 3 and [k(n, 3, 0) for n in (1, 256, 32768)] == ["R"] * 3
    assert [k(n, 3, 1) for n in (1, 32768)] == ["R", "R"]
    assert [k(n, 2, 1) for n in (1, 511, 512, 3071, 3072, 32768)] == ["Z", "Z", "R", "R", "B", "B"]      # ('B': kernel RB, convrb.hip)
    assert k(0, 2, 0) == "Z" and k(64, 1, 0) == "Z" and k(64, 4, 1) == "Z"          # nothing kernel R could take
    monkeypatch.setenv("MI355PPO_CONV_R", "min:8192")
    assert k(4096, 3, 0) == "Z" and k(8192, 3, 0) == "R" and k(8192, 2, 1) == "B"
    monkeypatch.setenv("MI355PPO_CONV_R", "0")
    assert {k(n, layer, d) for n in (1, 8192, 32768) for layer in (2, 3) for d in (0, 1)} == {"Z"}


def test_weight_matrix_cache_invalidation_logic(monkeypatch):
    """cnn._Buffers: without opting in every use re-derives the matrices; with caching, a new matrix is produced exactly
    when the owner's version, the tensor's own version counter or its storage changes (no GPU needed: the repack call
    is replaced by a counter)."""
    from cleanrl_amd import cnn

    calls = []

    def fake_repack(W, layer, mode=0, out=None):
        calls.append((layer, mode))
        return torch.full((1,), float(len(calls)))

    monkeypatch.setattr(cnn, "repack_weights", fake_repack)
    W = torch.zeros(32, 4, 8, 8)
    b = cnn._Buffers()
    b.weights(W, 1, 0); b.weights(W, 1, 0)
    assert len(calls) == 2                                   # caching is opt-in
    b.cache_weights = True
    m1 = b.weights(W, 1, 0); m2 = b.weights(W, 1, 0)
    assert len(calls) == 3 and m1 is m2
    b.weights(W, 1, 0); b.weights(W, 2, 0)                  # another (layer, mode) key has its own entry
    assert len(calls) == 4
    b.weights_version += 1                                   # the learner's fused optimiser step
    assert b.weights(W, 1, 0).item() == 5.0 and len(calls) == 5
    W.add_(1.0)                                              # an in-place torch update bumps W._version
    assert b.weights(W, 1, 0).item() == 6.0
    W2 = W.clone()                                           # different storage
    assert b.weights(W2, 1, 0).item() == 7.0


def test_torch_free_conv_driver_builds_and_links_the_in_tree_library():
    """tools/conv_traffic (C++ over the C ABI only: timing / dump comparison / PMC passes on a GPU box in seconds) compiles
    for gfx950 and resolves libmi355ppo.so through its $ORIGIN-relative rpath."""
    import subprocess

    from cleanrl_amd import build as b

    _lib.load()
    exe = b.build_tools(verbose=False)
    assert os.access(exe, os.X_OK)
    dyn = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libmi355ppo.so" in dyn and "$ORIGIN/../cleanrl_amd/csrc" in dyn
