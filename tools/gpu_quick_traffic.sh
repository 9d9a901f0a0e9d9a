#!/bin/bash
# Kernel Z after a locality change: correctness subset, entry-point timings, HBM-side traffic (FETCH_SIZE / WRITE_SIZE passes).
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -q -x -k "dgrad or kernel_z or full_minibatch or trunk") > $O/pytest_cls.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_cls.log | cut -c1-300
for m in 32768 32768 8192 4096 1024; do
  timeout 120 tools/conv_traffic $m 5 2>&1 | head -1 | tee -a $O/loc_timing.jsonl | cut -c1-420
done
for n in fetch write; do
  C=FETCH_SIZE; [ $n = write ] && C=WRITE_SIZE
  rm -rf /tmp/pmc_$n
  timeout 90 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$n -o t -- tools/conv_traffic 32768 3 > /dev/null 2>&1
  db=$(ls /tmp/pmc_$n/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/loc_pmc_$n.csv
done
python - <<'PY'
import csv,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
f={r[0]:r for r in csv.reader(open(O+"/loc_pmc_fetch.csv"))}
w={r[0]:r for r in csv.reader(open(O+"/loc_pmc_write.csv"))}
for k in list(f)[1:12]:
    print(k[18:84].ljust(66), f[k][2].rjust(8), "read %.3f GB write %.3f GB" % (float(f[k][3])*2048/1e9, float(w[k][3])*1024/1e9 if k in w else -1))
PY
