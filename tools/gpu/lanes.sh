#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for f in "" "--no-pin"; do
timeout 600 python tools/host_env_bench.py --groups 2 4 --iters 2 $f > $O/host_env_async.jsonl 2> $O/host_env_async.err; echo "flags=$f rc=$?"; python - <<PY
import json
for l in open("$O/host_env_async.jsonl"):
    d=json.loads(l); print(d["env_groups"], round(d["sps"]), "rollout", round(d["rollout_ms"],1), {k: round(v) for k, v in d.get("lane_step_us", {}).items()})
PY
done
