#!/bin/bash
# Same-box A/B at the level of the whole training step: the library of commit f3ba6e6 (kernel P with whole-slab staging: tools/oldlib/preP,
# built from a worktree of that commit) against the in-tree one (kernel P staged in pieces), bench.py config C / D / B, alternating.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/benchab; mkdir -p $O; L=$R/cleanrl_amd/csrc/libmi355ppo.so
cd $R
cp $L /tmp/lib_new.so
for rep in 1 2; do
  for v in preP new; do
    if [ $v = preP ]; then cp tools/oldlib/preP/libmi355ppo.so $L; else cp /tmp/lib_new.so $L; fi
    for c in C D B; do
      timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive 2> /dev/null | sed "s/^{/{\"lib\": \"$v\", /" | tee -a $O/bench_ab_kernel_p.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', '$c', round(d['value']), round(d['ms_per_step'],2), d['phases_ms']['update'])"
    done
  done
done
cp /tmp/lib_new.so $L
