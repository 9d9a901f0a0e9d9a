"""Markdown rows from the PMC passes of tools/gpu/r3_final2.sh (profiles/r03_pmc_{busy,mem,lds,fetch,write}.csv): per kernel of one
minibatch update at 32,768 images -- matrix-pipe busy, effective clock, wave cycles waiting, VALU / LDS / TA busy, L2-miss traffic.

    matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
    clock            = GRBM_GUI_ACTIVE / 8 / duration
    waiting          = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES  (share of resident wave time spent in s_waitcnt)
    VALU / LDS busy  = 4 x SQ_ACTIVE_INST_{VALU,LDS} / (1024 x GRBM_GUI_ACTIVE / 8)   (the SQ counts in units of 4 cycles)
    TA busy          = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8)
    LDS array        = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8): share of the launch the LDS arrays work; of which conflicts =
                       SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (the column that shows a bad lane -> bank mapping at a glance)
    traffic          = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md)
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = [("conv1q_fwd_kernel", "", "Q conv1 fwd"), ("z_kernel", "ZRowsConv<20, 20, 32, 4, 4, 9, 9, 2, 0,", "Z conv2 fwd"), ("r_kernel", "RGeom<20, 20, 0, 4, 4, 9, 9,", "R conv2 fwd"),
         ("z_kernel", "ZRowsConv<9, 9, 64, 3, 3, 7, 7, 1, 0,", "Z conv3 fwd"), ("r_kernel", "RGeom<9, 9, 0, 3, 3, 7, 7,", "R conv3 fwd"), ("z_kernel", "ZRowsLinear, 2, 4, 4, 0, true", "Z FC fwd"), ("g_kernel<0>", "", "G FC fwd"), ("g_kernel<1>", "", "G FC dgrad"), ("h_kernel", "", "H FC wgrad"),
         ("z_kernel", "ZRowsLinear, 2, 4, 4, 3, false", "Z FC dgrad"), ("fcw_bf16_kernel", "", "W FC wgrad"),
         ("convw_bf16_kernel", "VGeom<9, 9, 64,", "V conv3 wgrad"), ("convu_kernel", "UGeom<9, 9, 64,", "U conv3 wgrad"), ("z_kernel", "ZRowsConv<7, 7, 64, 3, 3, 9, 9, 1, -2,", "Z conv3 dgrad"), ("r_kernel", "RGeom<7, 7, 2, 3, 3, 9, 9,", "R conv3 dgrad"),
         ("convw_bf16_kernel", "VGeom<20, 20, 32,", "V conv2 wgrad"), ("convu_kernel", "UGeom<20, 20, 32,", "U conv2 wgrad"), ("z_kernel", "ZRowsConv<9, 9, 64, 2, 2, 10, 10, 1, -1,", "Z conv2 dgrad"), ("r_kernel", "RGeom<9, 9, 1, 2, 2, 10, 10,", "R conv2 dgrad"), ("rb_kernel", "", "RB conv2 dgrad"),
         ("conv1p_wgrad_kernel", "", "P conv1 wgrad"), ("convu1_kernel", "", "U conv1 wgrad")]


def load(name, prefix):
    rows = list(csv.reader(open(os.path.join(ROOT, "profiles", f"{prefix}{name}.csv"))))
    hdr = rows[0]
    return {r[0]: dict(zip(hdr[1:], map(float, r[1:]))) for r in rows[1:]}


def main():
    prefix = sys.argv[1] if len(sys.argv) > 1 else "r03_pmc_"
    busy, mem, lds, fetch, write = (load(n, prefix) for n in ("busy", "mem", "lds", "fetch", "write"))

    def pick(table, kern, geom):
        for k, v in table.items():
            if kern in k and geom in k:
                return v
        return None

    print("| kernel @ 32,768 images | us | clock GHz | matrix pipe busy | waves waiting | VALU busy | LDS busy | LDS array busy (conflict share) | TA busy | L2-miss traffic GB (read + write) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for kern, geom, label in NAMES:
        b, m, l, f, w = (pick(t, kern, geom) for t in (busy, mem, lds, fetch, write))
        if b is None:                                # (a launch runs on kernel Z or on kernel R: one of the two names is in the passes)
            continue
        cyc = b["GRBM_GUI_ACTIVE"] / 8
        cyc_m, cyc_l = m["GRBM_GUI_ACTIVE"] / 8, l["GRBM_GUI_ACTIVE"] / 8
        print(f"| {label} | {b['avg_us']:.0f} | {cyc / b['avg_us'] / 1e3:.2f} | {b['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.2f} | "
              f"{b['SQ_WAIT_INST_ANY'] / b['SQ_WAVE_CYCLES']:.2f} | {4 * m['SQ_ACTIVE_INST_VALU'] / (1024 * cyc_m):.2f} | "
              f"{4 * l['SQ_ACTIVE_INST_LDS'] / (1024 * cyc_l):.2f} | {l['SQ_LDS_IDX_ACTIVE'] / (256 * cyc_l):.2f} ({l['SQ_LDS_BANK_CONFLICT'] / max(l['SQ_LDS_IDX_ACTIVE'], 1.0):.2f}) | {m['TA_BUSY_avr'] / cyc_m:.2f} | "
              f"{f['FETCH_SIZE'] * 2048 / 1e9:.2f} + {w['WRITE_SIZE'] * 1024 / 1e9:.2f} |")


if __name__ == "__main__":
    main()
