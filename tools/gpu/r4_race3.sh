#!/bin/bash
set -u
export TMPDIR=/tmp DET_HEADS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4race3; rm -rf $O; mkdir -p $O
cd $R
N=128
timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep "^H" | sed 's/\[.*cuda:0[^]]*\]/[..]/' > $O/solo_$N.txt
wc -l $O/solo_$N.txt
for rep in $(seq 1 14); do
  timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep "^H" > $O/c1_${N}_$rep.txt & timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep "^H" > $O/c2_${N}_$rep.txt; wait
  for f in c1 c2; do
    d=$(diff <(sed 's/\[.*cuda:0[^]]*\]/[..]/' $O/solo_$N.txt) <(sed 's/\[.*cuda:0[^]]*\]/[..]/' $O/${f}_${N}_$rep.txt) | grep "^>" | awk '{print $2 ":" $3}' | head -14 | tr '\n' ' ')
    echo "N=$N rep $rep $f: differs in: ${d:-nothing}"
  done
done
