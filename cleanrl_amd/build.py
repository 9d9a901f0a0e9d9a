"""Build ``cleanrl_amd/csrc/libmi355ppo.so`` (HIP, gfx950 only) in-tree.

``python -m cleanrl_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
The library links only against the HIP runtime (``libamdhip64.so.7``); when it is loaded into a
process that already imported torch, the loader binds that soname to the copy torch bundles, so the
two share one runtime (streams and device pointers are interchangeable).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmi355ppo.so")
SOURCES = ["api.hip", "gae.hip", "distributions.hip", "loss.hip", "obs.hip", "optim.hip", "conv.hip", "convw.hip", "conv1q.hip", "conv1p.hip", "gemmz.hip", "gemmg.hip", "gemmh.hip", "convr.hip", "convrb.hip", "convu.hip", "fcw.hip", "heads.hip", "mlp.hip", "synth_env.hip", "dpcomm.hip", "host_twins.hip"]
HEADERS = ["common.h", "catrow.h", "ppo_rows.h", "bf16split.h", "f16split.h", "convr_geom.h", "convrb_geom.h", os.path.join("..", "..", "include", "mi355ppo.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: every f32 multiply/add rounds separately, as the reference's un-fused torch ops do.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# conv.hip: the streaming weight-gradient kernel unrolls a whole image (81 steps x 8 MFMAs); past LLVM's default
# `#pragma unroll` size limit the loop is only partly unrolled and its register "arrays" stay in scratch.  The other kernels
# of the file compile to identical code with and without the flag.
# gemmz.hip / fcw.hip / convw.hip: no SLP vectorizer -- it packs pairs of the split's f32 subtractions into v_pk_add_f32 (plus dead
# halves), and packed f32 VALU beside MFMAs is an anti-lever on this chip (MI355X_MICROARCH.md, per-instruction constants).
EXTRA_FLAGS = {"conv.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"], "gemmz.hip": ["-fno-slp-vectorize"], "gemmg.hip": ["-fno-slp-vectorize"], "gemmh.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"], "convr.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"], "convrb.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"], "convu.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"], "fcw.hip": ["-fno-slp-vectorize"], "convw.hip": ["-fno-slp-vectorize"]}


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


def build(force: bool = False, verbose: bool = True) -> str:
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + hdrs):
            jobs.append([HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", src, "-o", obj])
    if jobs:
        if verbose:
            print(f"[cleanrl_amd.build] compiling {len(jobs)} HIP source(s) for gfx950", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or _stale(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs)
        if verbose:
            print(f"[cleanrl_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


def build_tools(verbose: bool = True) -> str:
    """The torch-free conv driver (tools/conv_traffic.cpp: timing, dump comparison and PMC passes over the C ABI;
    contains its own small HIP kernels).  Links against the in-tree library through $ORIGIN-relative rpath."""
    root = os.path.dirname(os.path.dirname(CSRC))
    src = os.path.join(root, "tools", "conv_traffic.cpp")
    exe = os.path.join(root, "tools", "conv_traffic")
    if _stale(exe, [src, LIB, os.path.join(root, "include", "mi355ppo.h")]):
        _run([HIPCC, "--offload-arch=gfx950", "-O2", "-I" + os.path.join(root, "include"), src, "-o", exe, "-L" + CSRC,
              "-lmi355ppo", "-Wl,-rpath,$ORIGIN/../cleanrl_amd/csrc"])
        if verbose:
            print(f"[cleanrl_amd.build] built {exe}", file=sys.stderr)
    # the matrix-pipe floor microbenchmark (tools/mfma_floor.cpp: cycles per MFMA on random operands, with VALU fillers)
    fsrc, fexe = os.path.join(root, "tools", "mfma_floor.cpp"), os.path.join(root, "tools", "mfma_floor")
    if os.path.exists(fsrc) and _stale(fexe, [fsrc]):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", fsrc, "-o", fexe])
        if verbose:
            print(f"[cleanrl_amd.build] built {fexe}", file=sys.stderr)
    # what a stream reaches from HBM on the box (tools/hbm_probe.cpp; bench.py's `hbm_stream_probe` leg runs it with "quick")
    psrc, pexe = os.path.join(root, "tools", "hbm_probe.cpp"), os.path.join(root, "tools", "hbm_probe")
    if os.path.exists(psrc) and _stale(pexe, [psrc]):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", psrc, "-o", pexe])
        if verbose:
            print(f"[cleanrl_amd.build] built {pexe}", file=sys.stderr)
    # does the bf16 MFMA multiply SUBNORMAL inputs exactly? (kernel P's zero-extended uint8 operand relies on it: tools/mfma_denorm.cpp)
    dsrc, dexe = os.path.join(root, "tools", "mfma_denorm.cpp"), os.path.join(root, "tools", "mfma_denorm")
    if os.path.exists(dsrc) and _stale(dexe, [dsrc]):
        _run([HIPCC, "--offload-arch=gfx950", "-O2", "-Wno-unused-result", dsrc, "-o", dexe])
        if verbose:
            print(f"[cleanrl_amd.build] built {dexe}", file=sys.stderr)
    return exe


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_tools()
