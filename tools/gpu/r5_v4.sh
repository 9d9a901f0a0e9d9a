#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for sp in f16x2 bf16x3; do MI355PPO_SPLIT=$sp timeout 300 python tools/gpu/r5_graph_debug.py 2>&1 | grep -v amdgpu.ids | tail -14; done
