#!/bin/bash
# Same-box A/B of two builds of the library on the torch-free driver: tools/oldlib/base/libmi355ppo.so (the previous commit's) against the
# in-tree one; bit-identity of the dumped results, then timings.   usage: tools/gpu/ab_base.sh [keys...]
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; L=$R/cleanrl_amd/csrc/libmi355ppo.so
cd $R
cp $L /tmp/lib_new.so
for m in 32768 2049 1; do
  cp tools/oldlib/base/libmi355ppo.so $L; timeout 120 tools/conv_traffic $m 1 /tmp/d0_$m.bin > /dev/null 2>&1; echo "base rc=$?"
  cp /tmp/lib_new.so $L; timeout 120 tools/conv_traffic $m 1 /tmp/d1_$m.bin > /dev/null 2>&1; echo "new rc=$?"
  cmp /tmp/d0_$m.bin /tmp/d1_$m.bin && echo "images=$m: bit-identical" || python tools/cmp_f32.py /tmp/d0_$m.bin /tmp/d1_$m.bin | tail -2
done
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then cp tools/oldlib/base/libmi355ppo.so $L; else cp /tmp/lib_new.so $L; fi
  for m in 32768 8192 4096; do
    echo -n "$v "; timeout 120 tools/conv_traffic $m 5 2>&1 | head -1 | tee -a $O/ab_base.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], {k:v for k,v in d.items() if k.endswith('_us') or k=='sum_ms'})"
  done
done
done
cp /tmp/lib_new.so $L
if [ "${AB_FETCH:-0}" = 1 ]; then
for v in base new; do
  if [ $v = base ]; then cp tools/oldlib/base/libmi355ppo.so $L; else cp /tmp/lib_new.so $L; fi
  rm -rf /tmp/pmc_f; timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o t -- tools/conv_traffic 32768 3 > /dev/null 2>&1
  db=$(ls /tmp/pmc_f/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/ab_fetch_$v.csv
  echo "== FETCH_SIZE (read GB = KiB * 2048 / 1e9) $v"; python - <<PY
import csv
for r in list(csv.reader(open("$O/ab_fetch_$v.csv")))[1:12]:
    print(r[0][18:78].ljust(60), r[2].rjust(8), "us  read %.2f GB" % (float(r[3]) * 2048 / 1e9))
PY
done
cp /tmp/lib_new.so $L
fi
