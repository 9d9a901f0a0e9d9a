#!/bin/bash
# compile one csrc file to /tmp (errors and warnings only): tools/cc1.sh gemmz.hip [extra flags]
f=$1; shift
cd /root/repo/cleanrl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c $f -o /tmp/${f%.hip}.o 2>&1 | grep -E "error|warning" -A3 | head -40
