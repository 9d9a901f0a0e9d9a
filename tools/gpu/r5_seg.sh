#!/bin/bash
# Round 5: the world > 1 form of the captured update.  (1) the two-rank tests (gloo, both ranks on the one GPU); (2) what the three-graph slots cost
# against the single-graph slots and the eager update at config D's per-GPU size on one rank (MI355PPO_UPDATE_GRAPHS=cut: collectives skipped).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5seg; rm -rf $O; mkdir -p $O; cd $R
timeout 1400 python -m pytest tests/test_gpu_multirank.py -q -p no:cacheprovider -s -k "config_d or update_graphs" > $O/pytest_multirank.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_multirank.log | cut -c1-300
grep "config D whole iteration" -A3 $O/pytest_multirank.log | cut -c1-400
for i in 1 2; do
  for mode in graphs cut eager; do
    case $mode in
      graphs) env= ; flag= ;;
      cut) env="MI355PPO_UPDATE_GRAPHS=cut"; flag= ;;
      eager) env= ; flag=--no-update-graphs ;;
    esac
    env $env timeout 300 python bench.py --config D --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-inclusive --no-kernel-timing $flag 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(json.dumps({'update': '$mode', 'value': j['value'], 'ms_per_step': j['ms_per_step'], 'phases_ms': j.get('phases_ms'), 'update_mode': j['config']['update'][:90]}))" >> $O/update_graphs_cut_ab.jsonl
  done
done
cat $O/update_graphs_cut_ab.jsonl
