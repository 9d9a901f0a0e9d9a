#!/bin/bash
# s_memtime stamps of kernel R's four instances (workgroup 0: phases of its first four groups, k-steps of the second) at minibatch and rollout size
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5trace; rm -rf $O; mkdir -p $O; cd $R
for n in 32768 1024; do
  MI355PPO_R_TRACE=1 CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic $n 1 > $O/out_$n.txt 2> $O/err_$n.txt
  grep r_trace $O/err_$n.txt | awk '!seen[$0]++' > $O/trace_$n.txt; rm -f $O/err_$n.txt
done
head -c 6000 $O/trace_32768.txt
