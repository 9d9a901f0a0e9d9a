"""Dump the kernel-dispatch rows of a rocprofv3 rocpd database as CSV (name, start_ns, end_ns, dur_us, grid, workgroup) in
start order -- for size sweeps, where one kernel name runs at many sizes and a per-name average says nothing.
    python tools/rocpd_dump.py <db> [name_substring]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    pick = lambda *c: next((x for x in c if x in cols), None)
    gx, wx = pick("grid_size_x", "grid_x", "grid_size"), pick("workgroup_size_x", "workgroup_x", "workgroup_size")
    sel = ", ".join([name, "start", "end"] + [c for c in (gx, wx) if c])
    print("name,start_ns,end_ns,dur_us,grid_x,workgroup_x")
    for r in cur.execute(f"select {sel} from kernels order by start"):
        if want and want not in r[0]:
            continue
        extra = list(r[3:]) + [""] * (2 - len(r[3:]))
        print('"%s",%d,%d,%.3f,%s,%s' % (r[0][:120], r[1], r[2], (r[2] - r[1]) / 1e3, extra[0], extra[1]))


if __name__ == "__main__":
    main()
