"""GPU-side gaps of a rocprofv3 kernel trace (``*_results.db``): for every kernel, the idle time between the END of the previous dispatch (by start order)
and its own START, summed per kernel name -- where a launch-bound stretch (the rollout's 8 small launches per env step) loses its time.
    python tools/rocpd_gaps.py <db> [max_gap_us=200]     (gaps above max_gap_us -- host pauses between phases -- are left out)"""
import sqlite3
import sys


def main():
    db, cap = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name}, start, end from kernels order by start").fetchall()
    acc = {}
    prev_end = None
    for n, s, e in rows:
        if prev_end is not None:
            gap = (s - prev_end) / 1e3
            a = acc.setdefault(n[:80], [0, 0.0, 0.0, 0])
            a[0] += 1
            a[2] += (e - s) / 1e3
            if 0.0 <= gap <= cap:
                a[1] += gap
            elif gap < 0.0:
                a[3] += 1
        prev_end = max(prev_end or e, e)
    print("name,calls,gap_before_total_us,gap_before_avg_us,kernel_avg_us,overlapping_starts")
    for n, (c, g, d, o) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
        print('"%s",%d,%.1f,%.2f,%.2f,%d' % (n, c, g, g / c, d / c, o))
    print('"TOTAL",%d,%.1f,,,' % (sum(a[0] for a in acc.values()), sum(a[1] for a in acc.values())))


if __name__ == "__main__":
    main()
