#!/bin/bash
# Torch-free A/B of two BUILDS of libmi355ppo.so (tools/oldlib/ = the previous build) with tools/conv_traffic.cpp:
# timing JSON per build and a bitwise comparison of the dumps (epilogue-only changes must stay bit-identical).
set -u
export TMPDIR=/tmp
B=tools/conv_traffic
for n in 32768 1000; do
  LD_LIBRARY_PATH=tools/oldlib timeout 40 $B $n 4 /tmp/old.bin | tail -2 | head -1
  timeout 40 $B $n 4 /tmp/new.bin | tail -2 | head -1
  cmp /tmp/old.bin /tmp/new.bin && echo "images=$n: dumps bit-identical"
done
