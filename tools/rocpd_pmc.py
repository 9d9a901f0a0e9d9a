"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database:  python tools/rocpd_pmc.py <db> [name-filter]
Prints CSV: kernel, dispatches, avg_us, then one column per counter (mean over dispatches; summed over instances)."""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, dispatch_id, duration, counter_name, sum(counter_value) from pmc_events "
                       "group by name, dispatch_id, counter_name").fetchall()
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for name, disp, d, cname, val in rows:
        if filt and filt not in name:
            continue
        per[name][cname].append(val)
        dur[name][disp] = d
    counters = sorted({c for k in per.values() for c in k})
    print("kernel,dispatches,avg_us," + ",".join(counters))
    for name in sorted(per, key=lambda n: -sum(dur[n].values())):
        n = len(dur[name])
        avg_us = sum(dur[name].values()) / n / 1e3
        vals = [sum(per[name][c]) / max(len(per[name][c]), 1) if c in per[name] else float("nan") for c in counters]
        print('"%s",%d,%.1f,' % (name[:90], n, avg_us) + ",".join("%.4g" % v for v in vals))


if __name__ == "__main__":
    main()
