#!/bin/bash
# copy what the closing visit (tools/gpu/r6_final.sh) left under gpurun_out/$1 into profiles/r06_* and re-derive profiles/traffic.json
set -eu
S=gpurun_out/${1:-r6final}
for c in B C D E; do cp $S/bench_cfg$c.json profiles/r06_bench_cfg$c.json; done
cp $S/bench_cfgC_driver_counts.json profiles/r06_bench_cfgC_driver_counts.json
cp $S/kernel_stats_cfgC.csv profiles/r06_kernel_stats_cfgC.csv
for n in fetch write busy mem lds; do cp $S/pmc_$n.csv profiles/r06_pmc_$n.csv; done
cp $S/kernels_gh_ab.txt profiles/r06_kernels_gh_ab.txt
cp $S/sweep_k1k3_events.jsonl profiles/r06_sweep_k1k3_events.jsonl
cp $S/sweep_k1k3_kernel_stats.csv profiles/r06_sweep_k1k3.csv
cp $S/pytest_gpu.log profiles/r06_pytest_gpu.log
python tools/make_traffic_json.py
