#!/bin/bash
# round 4: kernel statistics of configs B / D, and counter passes over config E's K7 kernels (eager launches: a replayed graph has no per-kernel rows)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4extra; rm -rf $O; mkdir -p $O
for c in B D; do
  cd /tmp; rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o bench -- python $R/bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-pcie-inclusive > $O/prof_$c.log 2>&1; echo "prof $c rc=$?"
  db=$(find /tmp/prof_$c -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py "$db" 60 > $O/kernel_stats_cfg$c.csv
  rm -rf /tmp/prof_$c
done
cd $R
pmc_e() {   # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o t -- python bench.py --config E --steps 1 --warmup 1 --no-cpu-baseline --no-pcie-inclusive --no-update-graphs --no-rollout-graphs --no-kernel-timing > $O/pmc_$name.log 2>&1
    echo "pmc $name rc=$?"
    db=$(ls /tmp/pmc_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmc_$name.csv
    rm -rf /tmp/pmc_$name
}
pmc_e e_busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
pmc_e e_mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
tail -3 $O/pmc_e_busy.log | cut -c1-200
rm -f $O/pmc_*.log
ls $O; head -5 $O/pmc_e_busy.csv | cut -c1-300
