#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof_cnn
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_cnn -o cnn -- python tools/cnnbench.py 32768 > /dev/null 2>&1
python - <<'PY'
import sqlite3
cur=sqlite3.connect('gpurun_out/prof_cnn/cnn_results.db').cursor()
rows=cur.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1e3 from kernels where name like '%mi355ppo%' group by name, grid_x, grid_y order by 6 desc").fetchall()
for r in rows:
    n=r[0]
    i=n.find('FixedGeom<'); tag=n[i:i+60] if i>=0 else n[:70]
    print("%-64s grid=(%d,%d) wg=%d n=%d avg=%.1f us" % (tag, r[1], r[2], r[3], r[4], r[5]))
PY
