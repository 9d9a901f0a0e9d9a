"""GPU idle time between consecutive kernels of a rocprofv3 rocpd database over the WHOLE trace (no outlier filter):
python tools/rocpd_gaps_all.py <db>.  Prints every minibatch step (clip_adam_kernel to clip_adam_kernel) with wall / busy / idle,
the idle total by the kernel that follows the gap, and the 25 largest single gaps."""
import sqlite3
import sys
from collections import defaultdict


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "clip_adam_kernel" in r[0]]
    by, big = defaultdict(float), []
    print("step wall_us busy_us idle_us launches")
    for k, (a, b) in enumerate(zip(marks[:-1], marks[1:])):
        seg = rows[a + 1:b + 1]
        wall = (seg[-1][2] - rows[a][2]) / 1e3
        busy, prev_end = 0.0, rows[a][2]
        for n, s, e in seg:
            if s > prev_end:
                g = (s - prev_end) / 1e3
                by[n[:80]] += g
                big.append((g, k, n[:80]))
            busy += (e - max(s, prev_end)) / 1e3 if e > prev_end else 0.0
            prev_end = max(prev_end, e)
        print(k, round(wall), round(busy), round(wall - busy), len(seg))
    print("idle by following kernel (us, whole trace):")
    for n, g in sorted(by.items(), key=lambda kv: -kv[1])[:20]:
        print("  %9.0f  %s" % (g, n))
    print("largest gaps (us, step, following kernel):")
    for g, k, n in sorted(big, reverse=True)[:25]:
        print("  %9.0f  %4d  %s" % (g, k, n))


if __name__ == "__main__":
    main()
