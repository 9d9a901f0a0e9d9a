#!/bin/bash
# Round-3 visit 2: PMC passes over the torch-free conv driver (where do kernels X / C / F / T spend their cycles), rocprofv3
# kernel stats of bench.py for configs B / D / E / C, the K1 / K3 sweep with packed rows, the GPU suite.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rocprofv3 -L > $O/rocprofv3_counters.txt 2>&1
pmc_pass() {   # name, env assignments..., -- counters
    name=$1; shift
    rm -rf $O/pmc_$name
    timeout 90 rocprofv3 --pmc "$@" --kernel-trace -d $O/pmc_$name -o t -- tools/conv_traffic 32768 3 > $O/pmc_$name.log 2>&1
    echo "pmc $name rc=$?"
    db=$(ls $O/pmc_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmc_$name.csv
    rm -rf $O/pmc_$name
}
pmc_pass busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
pmc_pass mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pmc_pass cache TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
pmc_pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE
CONV_TRAFFIC_FWD_F32=1 pmc_pass busy_f32fwd SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
export CONV_TRAFFIC_CALIB=1
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
unset CONV_TRAFFIC_CALIB
head -20 $O/pmc_busy.csv | cut -c1-230
(time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu.log
cd /tmp
for c in B D E C; do
  rm -rf $O/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o bench -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_$c.log 2>&1; echo "prof $c rc=$?"
  db=$(find $O/prof_$c -name '*.db' | head -1); [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 80 > $O/kernel_stats_cfg$c.csv
  grep '^{' $O/prof_$c.log | tail -1 > $O/bench_cfg${c}_profiled.json
  rm -rf $O/prof_$c
done
rm -rf $O/prof_sweep
timeout 600 rocprofv3 --kernel-trace -d $O/prof_sweep -o sweep -- python $GRAFT_REPO_ROOT/tools/sweep_k1k3.py 20 > $O/sweep_k1k3_events.jsonl 2> $O/sweep.err; echo "sweep rc=$?"
db=$(find $O/prof_sweep -name '*.db' | head -1); [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 80 > $O/sweep_k1k3_kernel_stats.csv
rm -rf $O/prof_sweep
cd $GRAFT_REPO_ROOT
ls -la $O | tail -30
