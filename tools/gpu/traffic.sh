#!/bin/bash
# Torch-free profile of the conv kernels of one minibatch update (driver: tools/conv_traffic.cpp):
#   1. entry-point durations (HIP events)            -> gpurun_out/traffic/timing.json
#   2. rocprofv3 --kernel-trace (per-kernel duration) -> gpurun_out/traffic/trace.csv
#   3. HBM-traffic PMC passes, one counter per pass, --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and
#      WRITE_SIZE do not fit one pass)               -> gpurun_out/traffic/{fetch,write}.csv
set -u
export TMPDIR=/tmp
out=gpurun_out/traffic
rm -rf $out; mkdir -p $out
timeout 40 tools/conv_traffic 32768 4 | tail -2 | head -1 | tee $out/timing.json || { echo "driver failed rc=$?"; exit 1; }
run_pass() {
    name=$1; shift
    timeout 40 rocprofv3 "$@" --kernel-trace -d $out/$name -o t -- tools/conv_traffic 32768 3 > $out/$name.log 2>&1
    echo "$name rc=$?"
    db=$(ls $out/$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python $SUMM "$db" > $out/$name.csv && rm -f "$db"
}
SUMM=tools/rocpd_stats.py run_pass trace
export CONV_TRAFFIC_CALIB=1
SUMM=tools/rocpd_pmc.py run_pass fetch --pmc FETCH_SIZE
SUMM=tools/rocpd_pmc.py run_pass write --pmc WRITE_SIZE
head -12 $out/trace.csv | cut -c1-150
