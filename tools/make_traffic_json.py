"""profiles/traffic.json from the HBM-traffic PMC passes of tools/gpu/r5_final.sh (profiles/r05_pmc_{fetch,write}.csv: the f16x2 kernel set; round 4:
r04_pmc_*, tools/gpu/r4_final.sh; round 3: r03_pmc_*).

HBM bytes per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB: on gfx950 FETCH_SIZE tallies 128-byte read requests at
64 bytes (MI355X_MICROARCH.md "HBM"; confirmed here by the calibration copies of the same passes: a 1 GiB read reports
512 MiB at both 16 and 4 bytes per lane, a 1 GiB write reports 1 GiB).  Keys are bench.py's launch names.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IMAGES = 32768
# the newest passes over the committed kernel set (tools/gpu/r5_final.sh; CONV_TRAFFIC_F16=1: the default split)
FETCH_CSV, WRITE_CSV = "r06_pmc_fetch.csv", "r06_pmc_write.csv"      # round 6: tools/gpu/r6_final.sh (kernels G / H on the FC layer)
KEYS = {   # bench key -> (kernel-name substring, geometry substring) in rocpd_pmc.py's (truncated) kernel names
    "conv1_fwd": ("conv1q_fwd_kernel", ""),
    "conv2_fwd": [("z_kernel", "ZRowsConv<20, 20, 32, 4, 4, 9, 9, 2, 0,"), ("r_kernel", "RGeom<20, 20, 0, 4, 4, 9, 9,")],
    # (at 32,768 images kernel R runs the layer-3 forward and the layer-2 data gradient: csrc/convr.hip; one of the two names appears in a pass)
    "conv3_fwd": [("z_kernel", "ZRowsConv<9, 9, 64, 3, 3, 7, 7, 1, 0,"), ("r_kernel", "RGeom<9, 9, 0, 3, 3, 7, 7,")],
    "conv2_dgrad": [("z_kernel", "ZRowsConv<9, 9, 64, 2, 2, 10, 10, 1, -1,"), ("r_kernel", "RGeom<9, 9, 1, 2, 2, 10, 10,"), ("rb_kernel", "")],
    "conv3_dgrad": [("z_kernel", "ZRowsConv<7, 7, 64, 3, 3, 9, 9, 1, -2,"), ("r_kernel", "RGeom<7, 7, 2, 3, 3, 9, 9,")],
    "conv1_wgrad": [("conv1p_wgrad_kernel", ""), ("convu1_kernel", "")],
    "conv2_wgrad": [("convw_bf16_kernel", "VGeom<20, 20, 32,"), ("convu_kernel", "UGeom<20, 20, 32,")],      # (kernel U, csrc/convu.hip, replaces V / P under the f16 split)
    "conv3_wgrad": [("convw_bf16_kernel", "VGeom<9, 9, 64,"), ("convu_kernel", "UGeom<9, 9, 64,")],
    "fc_fwd": [("z_kernel", "ZRowsLinear, 2, 4, 4, 0, true"), ("g_kernel<0>", "")],      # (round 6: kernel G, csrc/gemmg.hip)
    "fc_dgrad": [("z_kernel", "ZRowsLinear, 2, 4, 4, 3, false"), ("g_kernel<1>", "")],     # epilogue 3 / <1> = ReLU-backward from mask bits
    "fc_wgrad": [("fcw_bf16", ""), ("h_kernel", "")],      # kernel W / kernel H (csrc/gemmh.hip) -- without the partial reduce
}
ALGORITHMIC = {   # bytes per image the algorithm must move (inputs read once + outputs written once); the ReLU masks travel as bits
    # (1/32 of the activation's bytes): written by the forward that produces the activation, read by the data gradient above it
    "conv1_fwd": 28224 + 51200 + 1600, "conv2_fwd": 51200 + 20736 + 648, "conv3_fwd": 20736 + 12544 + 392,
    "conv2_dgrad": 20736 + 1600 + 51200, "conv3_dgrad": 12544 + 648 + 20736,
    "conv1_wgrad": 28224 + 51200, "conv2_wgrad": 51200 + 20736, "conv3_wgrad": 20736 + 12544,
    "fc_fwd": 12544 + 2048, "fc_dgrad": 2048 + 392 + 12544,          # per row: a3 + h;  dz + mask bits + da3 (the 6.4 MB weight not counted)
    "fc_wgrad": 2048 + 12544,                                         # per row: dz + a3 (the 6.4 MB result and its partials not counted)
}


def load(name):
    rows = list(csv.reader(open(os.path.join(ROOT, "profiles", name))))
    return {r[0]: float(r[3]) for r in rows[1:]}


def main():
    fetch, write = load(FETCH_CSV), load(WRITE_CSV)
    out = {"source": f"profiles/{FETCH_CSV}, profiles/{WRITE_CSV} (tools/gpu/r6_final.sh -> tools/gpu/r6_pmc.sh): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over CONV_TRAFFIC_F16=1 tools/conv_traffic 32768 3",
           "correction": "bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950: 128-byte read requests tallied at 64 bytes)",
           "calibration_fetch_KiB_for_1GiB_read": {k[:40]: v for k, v in fetch.items() if "calib" in k},
           "calibration_write_KiB_for_1GiB_write": {k[:40]: v for k, v in write.items() if "calib" in k},
           "hbm_bytes_per_launch": {}, "read_bytes": {}, "write_bytes": {}, "algorithmic_bytes": {}, "traffic_over_algorithmic": {}}
    for key, alts in KEYS.items():
        alts = alts if isinstance(alts, list) else [alts]
        hit = lambda k: any(kern in k and geom in k for kern, geom in alts)      # noqa: E731
        rd = sum(v for k, v in fetch.items() if hit(k)) * 1024 * 2
        wr = sum(v for k, v in write.items() if hit(k)) * 1024
        name = f"{key}@{IMAGES}"
        out["read_bytes"][name], out["write_bytes"][name] = rd, wr
        out["hbm_bytes_per_launch"][name] = rd + wr
        out["algorithmic_bytes"][name] = ALGORITHMIC[key] * IMAGES
        out["traffic_over_algorithmic"][name] = round((rd + wr) / (ALGORITHMIC[key] * IMAGES), 3)
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    for k, v in out["traffic_over_algorithmic"].items():
        print(k, f"{out['hbm_bytes_per_launch'][k] / 1e9:.3f} GB", v)


if __name__ == "__main__":
    sys.exit(main())
