"""Timing experiment on kernel Z (csrc/gemmz.hip): builds copies of libmi355ppo.so in which parts of the kernel are compiled OUT, so
that `LD_PRELOAD=tools/oldlib/zdiag_<n>/libmi355ppo.so tools/conv_traffic 32768 4` times what is left (results are garbage).

    bit 0 (1)   no mask loads in the epilogue (every mask value 1.0)
    bit 1 (2)   no stores
    bit 2 (4)   one k-step per tile
    bit 3 (8)   no global loads of A inside the k-loop (the prologue's three stay)
    bit 4 (16)  no loads of B inside the k-loop (with the B ring: no ring fills and no fragment reads; the barriers stay)
    bit 5 (32)  no split (the VALU work between the MFMAs)
    bit 6 (64)  no LDS round trip inside the k-loop

    python tools/z_diag_build.py 1 2 3 4 7 8 16 24 32 64 ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cleanrl_amd", "csrc")
sys.path.insert(0, ROOT)
from cleanrl_amd import build as B  # noqa: E402

PATCHES = [
    ("for (int j = 0; j < NT; ++j) mk[ib][j][e] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_m, at(ro, j), 0, 0);",
     "for (int j = 0; j < NT; ++j) mk[ib][j][e] = (Z_DIAG & 1) ? 0x3f800000u : __builtin_amdgcn_raw_buffer_load_b32(rsrc_m, at(ro, j), 0, 0);"),
    ("                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_c, at(ro, j), 0, 0);\n                    }\n                }\n        }\n    } else {",
     "                        if (!(Z_DIAG & 2) || v == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_c, at(ro, j), 0, 0);\n                    }\n                }\n        }\n    } else {"),
    ("                    v = v > 0.0f ? v : 0.0f;\n                    __builtin_amdgcn_raw_buffer_store_b32(",
     "                    v = v > 0.0f ? v : 0.0f;\n                    if (!(Z_DIAG & 2) || v == 123.456f) __builtin_amdgcn_raw_buffer_store_b32("),
    ("    const int nsteps = RG::CLS ?", "    const int nsteps = (Z_DIAG & 4) ? 1 : RG::CLS ?"),
    ("                    else if constexpr (t == kPieces + 1) load_a(s + 3);", "                    else if constexpr (t == kPieces + 1) { if constexpr (!(Z_DIAG & 8)) load_a(s + 3); }"),
    ("            read_b(q ^ 1, q == 0 ? bcur : bnxt, q ^ 1);", "            if constexpr (!(Z_DIAG & 16)) read_b(q ^ 1, q == 0 ? bcur : bnxt, q ^ 1);"),
    ("                    else if constexpr (t == kPieces + 3) { if constexpr (q == 0) write_bpair(bnxt); }", "                    else if constexpr (t == kPieces + 3) { if constexpr (q == 0 && !(Z_DIAG & 16)) write_bpair(bnxt); }"),
    ("                    else { if constexpr (q == 0) load_bpair(); }", "                    else { if constexpr (q == 0 && !(Z_DIAG & 16)) load_bpair(); }"),
    ("            load_b(q ^ 1, s + 1);\n        }", "            if constexpr (!(Z_DIAG & 16)) load_b(q ^ 1, s + 1);\n        }"),
    ("                    if constexpr (t < kPieces)\n                        split_piece(", "                    if constexpr (t < kPieces && (Z_DIAG & 32)) {\n                    } else if constexpr (t < kPieces)\n                        split_piece("),
    ("                    else if constexpr (t == kPieces) to_lds();", "                    else if constexpr (t == kPieces) { if constexpr (!(Z_DIAG & 64)) to_lds(); }"),
    ("                    else if constexpr (t == kPieces + 2) read_frags();", "                    else if constexpr (t == kPieces + 2) { if constexpr (!(Z_DIAG & 64)) read_frags(); }"),
]


def main():
    src = open(os.path.join(CSRC, "gemmz.hip")).read()
    for old, new in PATCHES:
        assert old in src, old
        src = src.replace(old, new)
    B.build(verbose=False)
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in B.SOURCES if s != "gemmz.hip"]
    for n in sys.argv[1:]:
        d = os.path.join(ROOT, "tools", "oldlib", f"zdiag_{n}")
        os.makedirs(d, exist_ok=True)
        tmp = os.path.join(CSRC, f"_zdiag_{n}.hip")
        open(tmp, "w").write(f"#define Z_DIAG {int(n)}\n" + src)
        try:
            subprocess.run([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get("gemmz.hip", []) + ["-c", tmp, "-o", os.path.join(d, "gemmz.o")], check=True, cwd=CSRC)
            subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-o", os.path.join(d, "libmi355ppo.so"), os.path.join(d, "gemmz.o")] + objs, check=True)
        finally:
            os.remove(tmp)
        print("built", d)


if __name__ == "__main__":
    main()
