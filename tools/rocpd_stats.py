"""Summarise a rocprofv3 rocpd database (``*_results.db``) into a per-kernel stats table (CSV on stdout):
name, calls, total_us, avg_us, min_us, max_us, pct.   python tools/rocpd_stats.py <db> [top_n]"""
import sqlite3
import sys


def main():
    db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name}, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                       f"from kernels group by {name} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("name,calls,total_us,avg_us,min_us,max_us,pct")
    for r in rows[:top]:
        print('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (r[0][:150], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    print('"TOTAL (%d kernels)",%d,%.1f,,,,100' % (len(rows), sum(r[1] for r in rows), tot))


if __name__ == "__main__":
    main()
