#!/bin/bash
# Round-6 closing visit (kernels G / H on the FC layer, heads up to 18 actions, the update-graph policy): the whole GPU suite, the default bench line
# (config C, all legs: the bounded CPU sample at the metric's shapes, the host-env leg with its pipeline ceiling, the HBM-regime K1 / K3 points), the
# line at the driver's step counts, rocprofv3 kernel statistics of C, the five PMC passes over one minibatch update's launches, kernels G / H against
# Z / W (hashes, alternating times), configs B / D / E, the K1 / K3 sweep under rocprofv3, smoke().
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r6final}
rm -rf $O; mkdir -p $O
cd $R
T0=$SECONDS
if [ "${SKIP_TESTS:-0}" != 1 ]; then
(time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? t=$((SECONDS-T0))"; tail -4 $O/pytest_gpu.log | cut -c1-300
fi
(time timeout 1200 python bench.py) > $O/bench_cfgC.json 2> $O/bench_cfgC.err; echo "bench C (default command) rc=$? t=$((SECONDS-T0))"; grep '^{' $O/bench_cfgC.json | cut -c1-300; grep real $O/bench_cfgC.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-inclusive > $O/bench_cfgC_driver_counts.json 2>/dev/null; grep '^{' $O/bench_cfgC_driver_counts.json | cut -c1-200
TAG=${TAG:-r6final}/prof bash tools/gpu/r6_prof.sh > $O/prof.log 2>&1; cp $O/prof/kernel_stats_cfgC.csv $O/kernel_stats_cfgC.csv 2>/dev/null; echo "prof C t=$((SECONDS-T0))"; head -5 $O/kernel_stats_cfgC.csv | cut -c1-60,150-215
TAG=${TAG:-r6final}/pmc bash tools/gpu/r6_pmc.sh > $O/pmc.log 2>&1; for n in fetch write busy mem lds; do cp $O/pmc/pmc_$n.csv $O/pmc_$n.csv 2>/dev/null; done; echo "pmc t=$((SECONDS-T0))"
A="MI355PPO_FC_G=0 MI355PPO_FC_H=0" B="MI355PPO_FC_G=1" TAG=${TAG:-r6final}/gh HSIZES="32768" SIZES="32768 8192 4096" REPS=2 bash tools/gpu/r6_x.sh 2>&1 | tail -16 | cut -c1-260; cp $O/gh/ab.txt $O/kernels_gh_ab.txt 2>/dev/null; echo "G/H A/B t=$((SECONDS-T0))"
for c in B D E; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench $c rc=$? t=$((SECONDS-T0))"; grep '^{' $O/bench_cfg$c.json | cut -c1-200
done
cd /tmp; rm -rf /tmp/prof_sweep; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sweep -o sweep -- python $R/tools/sweep_k1k3.py 10 > $O/sweep_k1k3_events.jsonl 2> $O/sweep.err; echo "sweep rc=$? t=$((SECONDS-T0))"
db=$(find /tmp/prof_sweep -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py "$db" 40 > $O/sweep_k1k3_kernel_stats.csv; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "total t=$((SECONDS-T0))"
