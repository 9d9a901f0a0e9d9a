#!/bin/bash
# Round-3 closing visit (kernel set after the B ring / mask bits / phased k-order / split-K FC forward / graph lanes): the whole
# GPU suite, bench.py on config C (full line), its rocprofv3 kernel statistics, the PMC passes over one minibatch update's
# launches on the torch-free driver, then configs B, D, E (bench line + kernel statistics).  Most important outputs first:
# the visit may be cut by the GPU-minute budget.  Outputs under gpurun_out/final/ (copied to profiles/r03_* by hand).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
T0=$SECONDS
(time timeout 900 python -m pytest tests -m gpu -q --durations=8) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? t=$((SECONDS-T0))"; tail -4 $O/pytest_gpu.log | cut -c1-200
timeout 600 python bench.py > $O/bench_cfgC.json 2> $O/bench_cfgC.err; echo "bench C rc=$? t=$((SECONDS-T0))"; cut -c1-400 $O/bench_cfgC.json
prof() {   # config
  c=$1
  cd /tmp; rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o bench -- python $R/bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_$c.log 2>&1; echo "prof $c rc=$? t=$((SECONDS-T0))"
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
pmc_pass busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
export CONV_TRAFFIC_CALIB=1
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
unset CONV_TRAFFIC_CALIB
pmc_pass mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pmc_pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
rm -f $O/pmc_*.log
for c in B D E; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-pcie-inclusive > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench $c rc=$? t=$((SECONDS-T0))"; cut -c1-330 $O/bench_cfg$c.json
  prof $c
done
(lscpu | head -20; rocm-smi --showproductname 2>/dev/null | head -12) > $O/gpu_box_host.txt 2>&1
ls -la $O | cut -c1-120
