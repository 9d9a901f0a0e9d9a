#!/bin/bash
set -u
export TMPDIR=/tmp DET_HEADS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4race4; rm -rf $O; mkdir -p $O
export DET_REF=$O/ref.json
cd $R
N=128
timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep -c "^H"
for rep in $(seq 1 24); do
  timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep "DUMPED" & timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep "DUMPED"; wait
  ls $O/bad_*.npz 2>/dev/null | wc -l | grep -q "^[2-9]" && break
done
ls -la $O
python - <<PY
import glob, numpy as np
ref = np.load("$O/ref.json.npz")
for f in sorted(glob.glob("$O/bad_*.npz")):
    b = np.load(f)
    for k in ("dz", "dWc"):
        a, c = ref[k], b[k]
        d = np.argwhere(a != c)
        print(f, k, "differing elements:", len(d), "of", a.size)
        if len(d):
            rows = np.unique(d[:, 0]); cols = np.unique(d[:, -1])
            print("  rows:", rows[:20], "... n =", len(rows), " cols:", cols[:20], "... n =", len(cols))
            i = tuple(d[0]); print("  first:", i, "ref", a[i], "bad", c[i], " max |diff|", np.abs(a - c).max(), " finite", np.isfinite(c).all())
            if k == "dz":
                r0 = rows[0]; print("  row", r0, "ref[:6]", a[r0, :6], "bad[:6]", c[r0, :6], "dvalue", b["dvalue"][r0])
PY
