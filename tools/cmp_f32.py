"""Compare two float32 dumps of tools/conv_traffic:  python tools/cmp_f32.py a.bin b.bin  -> per-section max |a-b| / max|a|."""
import sys

import numpy as np

SECTIONS = [("dW1", 8192), ("db1", 32), ("dW2", 32768), ("db2", 64), ("dW3", 36864), ("db3", 64)]


def main():
    a, b = (np.fromfile(p, dtype=np.float32) for p in sys.argv[1:3])
    assert a.size == b.size, (a.size, b.size)
    o = 0
    worst = 0.0
    for name, n in SECTIONS + [("samples", a.size - sum(n for _, n in SECTIONS))]:
        x, y = a[o:o + n].astype(np.float64), b[o:o + n].astype(np.float64)
        o += n
        scale = max(np.abs(x).max(), 1e-30)
        err = np.abs(x - y).max() / scale
        worst = max(worst, err)
        print(f"{name}: n={n} max|a|={scale:.6g} max|a-b|/max|a|={err:.3g} identical={bool((x == y).all())} finite={bool(np.isfinite(y).all())}")
    print("WORST", worst)


if __name__ == "__main__":
    main()
