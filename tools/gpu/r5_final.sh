#!/bin/bash
# Round-5 closing visit (the f16x2 default): the whole GPU suite, the default bench line (config C, all legs) and its rocprofv3 kernel statistics,
# per-launch A/B of the two splits, kernel R against kernel Z (hashes, times, bench lines), the five PMC passes over one minibatch update's launches, configs B / D / E (+ kernel statistics of E),
# the determinism check (trained-parameter checksums, alone and two processes at a time), smoke().
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5final
rm -rf $O; mkdir -p $O
cd $R
T0=$SECONDS
if [ "${SKIP_TESTS:-0}" != 1 ]; then
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? t=$((SECONDS-T0))"; tail -6 $O/pytest_gpu.log | cut -c1-300
fi
(time timeout 900 python bench.py) > $O/bench_cfgC.json 2> $O/bench_cfgC.err; echo "bench C (default command) rc=$? t=$((SECONDS-T0))"; grep '^{' $O/bench_cfgC.json | cut -c1-400; grep real $O/bench_cfgC.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-inclusive > $O/bench_cfgC_driver_counts.json 2>/dev/null; grep '^{' $O/bench_cfgC_driver_counts.json | cut -c1-200
prof() {   # config
  c=$1
  cd /tmp; rm -rf /tmp/prof_$c
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o bench -- python $R/bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_$c.log 2>&1; echo "prof $c rc=$? t=$((SECONDS-T0))"
  db=$(find /tmp/prof_$c -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py "$db" 80 > $O/kernel_stats_cfg$c.csv
  grep '^{' $O/prof_$c.log | tail -1 > $O/bench_cfg${c}_profiled.json
  rm -rf /tmp/prof_$c; cd $R
}
prof C
for i in 1 2; do
  timeout 120 tools/conv_traffic 32768 4 2>&1 | grep '^{' | sed 's/^{/{"split": "bf16x3", /' >> $O/conv_traffic_ab.jsonl
  CONV_TRAFFIC_F16=1 timeout 120 tools/conv_traffic 32768 4 2>&1 | grep '^{' | sed 's/^{/{"split": "f16x2", /' >> $O/conv_traffic_ab.jsonl
done
cut -c1-400 $O/conv_traffic_ab.jsonl
# kernel R against kernel Z: hashes (bit-identical) and times at three sizes, then alternating timing runs (-> gpurun_out/r5convr/)
MI355PPO_CONV_R=min:1 SIZES="32768 1027 61" bash tools/gpu/r5_convr.sh 2>&1 | cut -c1-260 | tail -14; echo "kernel R A/B t=$((SECONDS-T0))"
cp $R/gpurun_out/r5convr/ab.jsonl $O/kernel_r_ab.jsonl 2>/dev/null; cat $R/gpurun_out/r5convr/hashdiff_*.txt > $O/kernel_r_hashdiff.txt 2>/dev/null
# bench lines with the round's kernel families switched off one at a time (R: input-resident forwards / data gradients; U: weight gradients), alternating
for i in 1 2; do for mode in all noU noR noRU; do
  case $mode in all) e=;; noU) e="MI355PPO_CONV_U=0";; noR) e="MI355PPO_CONV_R=0";; noRU) e="MI355PPO_CONV_R=0 MI355PPO_CONV_U=0";; esac
  env $e timeout 300 python bench.py --no-cpu-baseline --no-pcie-inclusive --no-kernel-timing 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(json.dumps({'kernels': '$mode', 'value': j['value'], 'ms_per_step': j['ms_per_step'], 'phases_ms': j['phases_ms']}))" | tee -a $O/bench_kernel_families_ab.jsonl | cut -c1-200
done; done
pmc_pass() {   # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    CONV_TRAFFIC_F16=1 timeout 120 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o t -- tools/conv_traffic 32768 3 > $O/pmc_$name.log 2>&1
    echo "pmc $name rc=$? t=$((SECONDS-T0))"
    db=$(ls /tmp/pmc_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmc_$name.csv
    rm -rf /tmp/pmc_$name
}
export CONV_TRAFFIC_CALIB=1
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
unset CONV_TRAFFIC_CALIB
pmc_pass busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
pmc_pass mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pmc_pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
rm -f $O/pmc_*.log
for c in B D E; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench $c rc=$? t=$((SECONDS-T0))"; grep '^{' $O/bench_cfg$c.json | cut -c1-240
done
prof E
for N in 1024 256; do
  for rep in 1 2; do timeout 200 python tools/gpu/determinism.py $N 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/alone N=$N: /" | tee -a $O/det.log; done
done
timeout 200 python tools/gpu/determinism.py 256 2>/dev/null | tail -1 > $O/c1.log & timeout 200 python tools/gpu/determinism.py 256 2>/dev/null | tail -1 > $O/c2.log; wait
cat $O/c1.log $O/c2.log | cut -c1-200 | sed "s/^/two at a time N=256: /" | tee -a $O/det.log
sort $O/det.log | uniq -c > $O/determinism.txt; cat $O/determinism.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "total t=$((SECONDS-T0))"
