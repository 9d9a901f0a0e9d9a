#!/bin/bash
# Round-4 closing visit: the whole GPU suite, the default bench line (config C, all legs) and its rocprofv3 kernel statistics,
# the HBM-traffic PMC passes over one minibatch update's launches, configs B / D / E (bench lines + kernel statistics of E).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4final
rm -rf $O; mkdir -p $O
cd $R
T0=$SECONDS
if [ "${SKIP_TESTS:-0}" != 1 ]; then
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? t=$((SECONDS-T0))"; tail -4 $O/pytest_gpu.log | cut -c1-200
fi
(time timeout 900 python bench.py) > $O/bench_cfgC.json 2> $O/bench_cfgC.err; echo "bench C (default command) rc=$? t=$((SECONDS-T0))"; grep '^{' $O/bench_cfgC.json | cut -c1-400; grep real $O/bench_cfgC.err
prof() {   # config
  c=$1
  cd /tmp; rm -rf /tmp/prof_$c
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o bench -- python $R/bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_$c.log 2>&1; echo "prof $c rc=$? t=$((SECONDS-T0))"
  db=$(find /tmp/prof_$c -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py "$db" 80 > $O/kernel_stats_cfg$c.csv
  grep '^{' $O/prof_$c.log | tail -1 > $O/bench_cfg${c}_profiled.json
  rm -rf /tmp/prof_$c; cd $R
}
prof C
timeout 120 tools/conv_traffic 32768 4 > $O/conv_traffic.json 2>&1; head -1 $O/conv_traffic.json | cut -c1-500
pmc_pass() {   # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    timeout 90 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o t -- tools/conv_traffic 32768 3 > $O/pmc_$name.log 2>&1
    echo "pmc $name rc=$? t=$((SECONDS-T0))"
    db=$(ls /tmp/pmc_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmc_$name.csv
    rm -rf /tmp/pmc_$name
}
export CONV_TRAFFIC_CALIB=1
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
unset CONV_TRAFFIC_CALIB
rm -f $O/pmc_*.log
for c in B D E; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench $c rc=$? t=$((SECONDS-T0))"; grep '^{' $O/bench_cfg$c.json | cut -c1-240
done
prof E
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "total t=$((SECONDS-T0))"
