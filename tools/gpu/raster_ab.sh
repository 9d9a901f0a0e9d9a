#!/bin/bash
# Same-box A/B of the workgroup orders on the torch-free driver: kernel Z's supertile order for the FC data gradient
# (MI355PPO_Z_SUPER = row blocks per supertile, 0 = launch order) and kernel V's XCD-contiguous unit order (MI355PPO_V_XCD).
# Bit-identity of the dumped results first, then timings, then the L2-miss read traffic (FETCH_SIZE pass).
# Ran on the commit that introduced the two orders, where both were behind these run-time switches; the switches were removed
# with the result (supertiles of 4 row blocks, XCD-contiguous units): check that commit (f3bbcca) out to repeat the A/B.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/raster; mkdir -p $O
cd $R
for m in 32768 4096 2049; do
  MI355PPO_Z_SUPER=0 MI355PPO_V_XCD=0 timeout 120 tools/conv_traffic $m 1 /tmp/d0_$m.bin > /dev/null 2>&1; echo "base rc=$?"
  timeout 120 tools/conv_traffic $m 1 /tmp/d1_$m.bin > /dev/null 2>&1; echo "new rc=$?"
  cmp /tmp/d0_$m.bin /tmp/d1_$m.bin && echo "images=$m: bit-identical" || python tools/cmp_f32.py /tmp/d0_$m.bin /tmp/d1_$m.bin | tail -2
done
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['images'], {k:v for k,v in d.items() if k in ('fc_dgrad_us','wgrad3_us','wgrad2_us','sum_ms')})"; }
for rep in 1 2; do
  for cfg in "0 0" "4 1" "2 1" "8 1" "16 0"; do
    set -- $cfg
    for m in 32768 8192; do
      echo -n "super=$1 vxcd=$2 "; MI355PPO_Z_SUPER=$1 MI355PPO_V_XCD=$2 timeout 120 tools/conv_traffic $m 5 2>&1 | head -1 | sed "s/^{/{\"super\": $1, \"v_xcd\": $2, /" | tee -a $O/raster_ab.jsonl | show
    done
  done
done
for cfg in "0 0" "4 1"; do
  set -- $cfg
  rm -rf /tmp/pmc_f; MI355PPO_Z_SUPER=$1 MI355PPO_V_XCD=$2 timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o t -- tools/conv_traffic 32768 3 > /dev/null 2>&1
  db=$(ls /tmp/pmc_f/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/fetch_super$1_vxcd$2.csv
  echo "== FETCH_SIZE super=$1 vxcd=$2 (read GB = KiB * 2048 / 1e9)"; python - <<PY
import csv
for r in list(csv.reader(open("$O/fetch_super$1_vxcd$2.csv")))[1:12]:
    print(r[0][18:78].ljust(60), r[2].rjust(8), "us  read %.2f GB" % (float(r[3]) * 2048 / 1e9))
PY
done
timeout 300 python -m pytest tests/test_gpu_cnn.py -q -x -k "fc or wgrad or kernel_z or weight" 2>&1 | tail -3
