#!/bin/bash
# round 4, final tree: the trained parameters are bit-identical from process to process -- alone and with a second process on the same GPU
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4detf; rm -rf $O; mkdir -p $O; cd $R
for N in 1024 256; do
  for rep in 1 2 3; do timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/alone N=$N: /" | tee -a $O/det.log; done
done
for rep in 1 2; do
  timeout 200 python tools/gpu/determinism.py 256 2>/dev/null | tail -1 > $O/c1.log & timeout 200 python tools/gpu/determinism.py 256 2>/dev/null | tail -1 > $O/c2.log; wait
  cat $O/c1.log $O/c2.log | cut -c1-200 | sed "s/^/two at a time N=256: /" | tee -a $O/det.log
done
sort $O/det.log | uniq -c
