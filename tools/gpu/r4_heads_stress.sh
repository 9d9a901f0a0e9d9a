#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== alone"; timeout 300 python tools/gpu/heads_stress.py 3000 4096 2>/dev/null
echo "== two at a time"
for rep in 1 2; do
  timeout 300 python tools/gpu/heads_stress.py 3000 4096 2>/dev/null & timeout 300 python tools/gpu/heads_stress.py 3000 4096 2>/dev/null; wait
done
echo "== two at a time, M = 32768"
timeout 300 python tools/gpu/heads_stress.py 1000 32768 2>/dev/null & timeout 300 python tools/gpu/heads_stress.py 1000 32768 2>/dev/null; wait
