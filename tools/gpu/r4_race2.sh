#!/bin/bash
# Which TENSOR of the first minibatch differs when a second process competes for the GPU?  (tools/gpu/determinism.py hashes)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4race2; rm -rf $O; mkdir -p $O
cd $R
for N in 128 256; do
  timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep "^H" | sed 's/\[.*cuda:0[^]]*\]/[..]/' > $O/solo_$N.txt
  wc -l $O/solo_$N.txt
  for rep in 1 2 3 4; do
    timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep "^H" > $O/c1_${N}_$rep.txt & timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | grep "^H" > $O/c2_${N}_$rep.txt; wait
    for f in c1 c2; do
      d=$(diff <(sed 's/\[.*cuda:0[^]]*\]/[..]/' $O/solo_$N.txt) <(sed 's/\[.*cuda:0[^]]*\]/[..]/' $O/${f}_${N}_$rep.txt) | grep "^>" | awk '{print $3 ":" $4}' | tr '\n' ' ')
      echo "N=$N concurrent rep $rep $f: differs in: ${d:-nothing}"
    done
  done
done
