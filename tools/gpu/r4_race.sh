#!/bin/bash
# Which launch is not bit-identical when a second process competes for the GPU?  Order-independent hashes of every output tensor of
# the torch-free conv driver: alone (twice), then two instances at a time.
set -u
export TMPDIR=/tmp CONV_TRAFFIC_HASH=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4race; rm -rf $O; mkdir -p $O
cd $R
for M in 8192 4096 32768; do
  timeout 120 tools/conv_traffic $M 2 | grep "^hash" > $O/solo_a_$M.txt
  timeout 120 tools/conv_traffic $M 2 | grep "^hash" > $O/solo_b_$M.txt
  cmp -s $O/solo_a_$M.txt $O/solo_b_$M.txt && echo "M=$M solo: identical" || { echo "M=$M solo: DIFFER"; diff $O/solo_a_$M.txt $O/solo_b_$M.txt; }
  for rep in 1 2 3; do
    timeout 200 tools/conv_traffic $M 3 | grep "^hash" > $O/c1_${M}_$rep.txt & timeout 200 tools/conv_traffic $M 3 | grep "^hash" > $O/c2_${M}_$rep.txt; wait
    for f in c1 c2; do
      d=$(diff $O/solo_a_$M.txt $O/${f}_${M}_$rep.txt | grep "^>" | awk '{print $3}' | tr '\n' ' ')
      echo "M=$M concurrent rep $rep $f: differs in: ${d:-nothing}"
    done
  done
done
