"""Build a VARIANT of libmi355ppo.so beside the in-tree one: tools/oldlib/<name>/libmi355ppo.so = the in-tree objects with the named
sources recompiled under extra -D flags.  For same-box A/Bs of build-time knobs on the torch-free driver (tools/gpu/lib_ab.sh).

    python tools/build_variant.py ntst -DMI355_AUX_STREAM_ST=2 -- conv1q.hip convr.hip gemmg.hip
    python tools/build_variant.py q2 -- conv1q.hip=/tmp/conv1q_other.hip          (a source compiled from another file, in csrc/'s include context)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleanrl_amd import build as b  # noqa: E402


def main() -> None:
    name, rest = sys.argv[1], sys.argv[2:]
    cut = rest.index("--")
    defs, srcs = rest[:cut], rest[cut + 1:]
    b.build(force=False, verbose=False)
    out = os.path.join(ROOT, "tools", "oldlib", name)
    os.makedirs(out, exist_ok=True)
    objs = []
    procs = []
    other = dict(x.split("=", 1) for x in srcs if "=" in x)
    srcs = [x.split("=", 1)[0] for x in srcs]
    for src in b.SOURCES:
        obj = os.path.join(b.CSRC, src.replace(".hip", ".o"))
        if src in srcs:
            obj = os.path.join(out, src.replace(".hip", ".o"))
            procs.append(subprocess.Popen([b.HIPCC] + b.FLAGS + b.EXTRA_FLAGS.get(src, []) + defs + ["-I" + b.CSRC, "-c", other.get(src, src), "-o", obj], cwd=b.CSRC))
        objs.append(obj)
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("compile failed")
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-o", os.path.join(out, "libmi355ppo.so")] + objs)
    print(os.path.join(out, "libmi355ppo.so"))


if __name__ == "__main__":
    main()
