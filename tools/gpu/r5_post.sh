#!/bin/bash
# After the closing visit: kernel R's forwards request the next group's source evenly over the k-loop.  The whole GPU suite on that tree, the
# default bench line, the line with the old schedule (MI355PPO_R_SPREAD=0), configs B and D.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5post; rm -rf $O; mkdir -p $O; cd $R
T0=$SECONDS
(time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? t=$((SECONDS-T0))"; tail -4 $O/pytest_gpu.log | cut -c1-200
timeout 400 python bench.py > $O/bench_cfgC.json 2> $O/bench_cfgC.err; echo "bench C rc=$? t=$((SECONDS-T0))"; grep '^{' $O/bench_cfgC.json | cut -c1-300
for mode in 0 1; do
  MI355PPO_R_SPREAD=$mode timeout 300 python bench.py --no-cpu-baseline --no-pcie-inclusive --no-kernel-timing 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(json.dumps({'spread': $mode, 'value': j['value'], 'ms_per_step': j['ms_per_step'], 'phases_ms': {k: v for k, v in j['phases_ms'].items() if k != 'note'}}))" | tee -a $O/bench_spread_ab.jsonl
done
for c in B D; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive > $O/bench_cfg$c.json 2>/dev/null; echo "bench $c rc=$? t=$((SECONDS-T0))"; grep '^{' $O/bench_cfg$c.json | cut -c1-200
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "total t=$((SECONDS-T0))"
