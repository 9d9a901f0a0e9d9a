#!/bin/bash
# WARNING: do not use for kernels whose names exist in both builds -- two code objects with the same kernel name in one process: which one a
# launch gets is undefined (profiles/r05_kernel_u_ab.txt).  Use template instances + an environment switch inside ONE library instead.
# A/B of two builds of the library over the torch-free conv driver: gpurun_tmp_ab/libold.so (LD_PRELOAD) against the in-tree build, f16x2,
# three alternating runs each with output hashes (the two builds must agree bit for bit).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ab; rm -rf $O; mkdir -p $O; cd $R
for i in 1 2 3; do
  LD_PRELOAD=$R/gpurun_tmp_ab/libold.so CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic 32768 4 2>&1 | grep '^{' | sed 's/^{/{"build": "old", /' >> $O/ab.jsonl
  CONV_TRAFFIC_HASH=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic 32768 4 2>&1 | grep '^{' | sed 's/^{/{"build": "new", /' >> $O/ab.jsonl
done
python - <<'PY'
import json,os
rows=[json.loads(l) for l in open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r5ab/ab.jsonl")]
for r in rows:
    print(r["build"], {k:(round(v,1) if isinstance(v,float) else v) for k,v in r.items() if k!="build"})
PY
