#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m cProfile -s cumtime bench.py --config E --no-cpu-baseline --no-pcie-inclusive --no-kernel-timing --steps 3 --warmup 1 2>/dev/null | grep -v "^{" | head -45 | cut -c1-180
