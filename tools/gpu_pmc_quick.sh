#!/bin/bash
# Quick counter passes (busy / memory / LDS) over one minibatch update's launches on the torch-free driver; CSVs to gpurun_out/pmcq_*.csv
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
pmc_pass() {
    name=$1; shift
    rm -rf /tmp/pmc_$name
    timeout 90 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o t -- tools/conv_traffic 32768 3 > /tmp/pmc_$name.log 2>&1
    echo "pmc $name rc=$?"
    db=$(ls /tmp/pmc_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/rocpd_pmc.py "$db" > $O/pmcq_$name.csv
    rm -rf /tmp/pmc_$name
}
pmc_pass busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
pmc_pass mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pmc_pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
python - <<'PY'
import csv, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
def load(n):
    rows = list(csv.reader(open(f"{O}/pmcq_{n}.csv"))); h = rows[0]
    return {r[0]: dict(zip(h[1:], map(float, r[1:]))) for r in rows[1:]}
b, m, l = load("busy"), load("mem"), load("lds")
print("kernel | us | GHz | mfma busy | wait_any | wait_inst | valu | lds_idx | lds_inst | bankconf | TA")
for k in b:
    if k not in m or k not in l or b[k]["avg_us"] < 100: continue
    cyc = b[k]["GRBM_GUI_ACTIVE"] / 8; cm = m[k]["GRBM_GUI_ACTIVE"] / 8; cl = l[k]["GRBM_GUI_ACTIVE"] / 8
    print(k[18:70], "| %.0f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.3f | %.2f" % (b[k]["avg_us"], cyc / b[k]["avg_us"] / 1e3,
          b[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), b[k]["SQ_WAIT_ANY"] / b[k]["SQ_WAVE_CYCLES"], b[k]["SQ_WAIT_INST_ANY"] / b[k]["SQ_WAVE_CYCLES"],
          4 * m[k]["SQ_ACTIVE_INST_VALU"] / (1024 * cm), l[k]["SQ_LDS_IDX_ACTIVE"] / (256 * cl), 4 * l[k]["SQ_ACTIVE_INST_LDS"] / (1024 * cl),
          l[k]["SQ_LDS_BANK_CONFLICT"] / max(l[k]["SQ_LDS_IDX_ACTIVE"], 1), m[k]["TA_BUSY_avr"] / cm))
PY
