#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/probe; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/cold_probe.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/cold_probe_first.txt 2> $O/cold_probe_first.err; echo "rc=$?"
cut -c1-220 $O/cold_probe_first.txt | head -60
timeout 300 python tools/cold_probe.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/cold_probe_second.txt 2> $O/cold_probe_second.err; echo "rc=$?"
cut -c1-220 $O/cold_probe_second.txt | head -40
