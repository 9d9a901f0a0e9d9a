"""LDS bank conflicts of kernel U's transpose reads (csrc/convu.hip), computed from its lane addresses -- the host-side count that found kernel R's
conflicts (tests/host/convr_geom_check.cpp) applied to `ds_read_b64_tr_b16`: a wave instruction is served in two passes of 32 lanes (MI355X_MICROARCH.md,
LDS), 8 bytes per lane = two of the 64 four-byte banks; a pass takes as many LDS cycles as the busiest bank has DISTINCT addresses.

    python tools/lds_conflicts_u.py      ->  cycles per pass (1.0 = conflict-free) for the dz and the source fragments of layers 3 and 2

The measured counterpart: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of profiles/r05_pmc_lds.csv (tools/pmc_table.py: 0.38 layer 3, 0.43 layer 2)."""


def pass_cycles(addrs):
    per_bank = {}
    for a in addrs:
        for k in (0, 1):
            per_bank.setdefault(((a >> 2) + k) & 63, set()).add(a)
    return max(len(v) for v in per_bank.values())


def lanes():
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        yield lane, g, i >> 2, i & 3, g >> 1          # lane, sixteen-lane group, block row r, column chunk c4, block half


ROTATE = True      # round 6: dz's odd lines stored rotated by one record (convu.hip::unit_dst); False: round 5's layout (2.00 cycles per source pass)


def layer3():
    SW, OH, PIX, SRC_REC, KSTEPS = 9, 7, 272, 81, 14      # UGeom3: a2 (9, 9, 64), dz3 (7, 7, 64) in lines padded to 8 records, 4 images
    dz = src = n_dz = n_src = 0
    for s in range(KSTEPS):
        for t in (0, 1):
            for plane in (0, 128):
                for ct in (0, 1):
                    a = [(16 * s + 4 * r + 2 * h + t) * PIX + (32 * ct + 16 * (g & 1) + 4 * c4) * 2 + plane for _, g, r, c4, h in lanes()]
                    dz += pass_cycles(a[:32]) + pass_cycles(a[32:]); n_dz += 2
                for tap in range(9):
                    for cpart in (0, 1):
                        ty, tx = divmod(tap, 3)
                        a = []
                        for _, g, r, c4, h in lanes():
                            L = 2 * s + (r >> 1)
                            rec = (L // OH) * SRC_REC + (L % OH) * SW + 4 * (r & 1) + 2 * h + t + ty * SW + tx - (r >> 1 if ROTATE else 0)      # (round 6: odd lines one record to the left)
                            a.append(rec * PIX + (16 * (g & 1) + 4 * c4) * 2 + 64 * cpart + plane)
                        src += pass_cycles(a[:32]) + pass_cycles(a[32:]); n_src += 2
    return dz / n_dz, src / n_src


def layer2():
    OW, OP, G, PIXS, PIXD, KSTEPS = 9, 81, 2, 144, 272, 11      # UGeom2: a1 (20, 20, 32) at stride 2 (even columns stored first), dz2 (9, 9, 64), 2 images

    def pidx(y, x):
        return y * 20 + (x & 1) * 10 + (x >> 1)

    dz = src = n_dz = n_src = 0
    for s in range(KSTEPS):
        for t in (0, 1):
            for plane_d, plane_s in ((0, 0), (128, 64)):
                for ct in (0, 1):
                    a = [(16 * s + 4 * r + 2 * h + t) * PIXD + (32 * ct + 16 * (g & 1) + 4 * c4) * 2 + plane_d for _, g, r, c4, h in lanes()]
                    dz += pass_cycles(a[:32]) + pass_cycles(a[32:]); n_dz += 2
                for tap in range(16):
                    ty, tx = divmod(tap, 4)
                    a = []
                    for _, g, r, c4, h in lanes():
                        pix = 16 * s + 4 * r + 2 * h + t
                        gi, p = divmod(pix, OP)
                        y, x = divmod(p, OW)
                        rec = gi * 400 + pidx(2 * y, 2 * x) if pix < G * OP else 0
                        a.append((rec + pidx(ty, tx)) * PIXS + (16 * (g & 1) + 4 * c4) * 2 + plane_s)
                    src += pass_cycles(a[:32]) + pass_cycles(a[32:]); n_src += 2
    return dz / n_dz, src / n_src


if __name__ == "__main__":
    for name, (d, s_), reads in (("layer 3 (UGeom3)", layer3(), (1, 3)), ("layer 2 (UGeom2)", layer2(), (1, 4))):
        # per k-step a wave reads its dz fragment once and a source fragment per tile it owns (3 / 4 tiles)
        share = (reads[0] * (d - 1) + reads[1] * (s_ - 1)) / (reads[0] * d + reads[1] * s_)
        print(f"{name}: dz fragments {d:.2f} cycles per pass, source fragments {s_:.2f}; conflict share of the transpose reads' LDS cycles {share:.2f}")
