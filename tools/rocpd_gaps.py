"""GPU idle time between consecutive kernels of a rocprofv3 rocpd database, per minibatch step of the update:
python tools/rocpd_gaps.py <db>.  A step = from one `clip_adam_kernel` to the next; prints, for the steps whose length is
within 3 % of the median, the mean wall time, kernel-busy time, idle time, launches, and the largest idle gaps by the
kernel that FOLLOWS them."""
import sqlite3
import statistics
import sys
from collections import defaultdict


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "clip_adam_kernel" in r[0]]
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a + 1:b + 1]
        wall = (seg[-1][2] - rows[a][2]) / 1e3
        busy, gaps, prev_end = 0.0, [], rows[a][2]
        for n, s, e in seg:
            if s > prev_end:
                gaps.append(((s - prev_end) / 1e3, n))
            busy += (e - max(s, prev_end)) / 1e3 if e > prev_end else 0.0
            prev_end = max(prev_end, e)
        steps.append((wall, busy, gaps, len(seg)))
    med = statistics.median(w for w, *_ in steps)
    sel = [s for s in steps if abs(s[0] - med) < 0.03 * med]
    print(f"{len(steps)} steps between optimizer kernels, median {med:.0f} us; {len(sel)} within 3 %")
    print("mean wall %.0f us, busy %.0f us, idle %.0f us, launches %.0f" % tuple(statistics.mean(x) for x in zip(*[(s[0], s[1], s[0] - s[1], s[3]) for s in sel])))
    by = defaultdict(list)
    for s in sel:
        for g, n in s[2]:
            by[n[:90]].append(g)
    print("idle before kernel (us per step, mean gap, count per step):")
    for n, g in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:25]:
        print("  %7.1f %6.1f %5.1f  %s" % (sum(g) / len(sel), statistics.mean(g), len(g) / len(sel), n))


if __name__ == "__main__":
    main()
