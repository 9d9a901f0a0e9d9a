// Kernel R's compile-time geometry (convr.hip): the shapes of its four instances, the table that deals a group's rows to the lanes, the
// order of the k-steps and the LDS offsets of the taps.  Plain constexpr C++ -- no device code -- so that the host can check it:
// tests/test_kernel_r_geometry.py compiles tests/host/convr_geom_check.cpp against this header with g++ (every row placed exactly once,
// conflict-free 16-lane sets, the k-step orders are permutations, (window origin) + (tap) is additive in the record index, LDS budgets).
#pragma once

#if defined(__HIPCC__)
#define R_GEOM_FN __device__ __forceinline__ constexpr
#else
#define R_GEOM_FN inline constexpr
#endif

namespace mi355ppo {

// Source (images, IH, IW, 64) f32 channels-last, zero border of HL pixels, window KH x KW at stride 1 over the padded grid, output grid
// OH x OW per image; NT 32-column tiles of the pack; G images per group on NW waves of MT 32-row tiles each; destination pixel of grid
// pixel (gy, gx) and column tile j: R_MASKB_CLS4 -> (2 gy + (j >> 1), 2 gx + (j & 1)) of a (2 OH, 2 OW, 32) image, else pixel (gy, gx) of
// an (OH, OW, 32 NT) image.  ORDER 1: the layer-3 forward's phase order of kernel Z (z_kstep), 0: ascending.  RP: records per padded row in
// LDS (0 = IWP, no padding) -- extra records at the end of every row shift the rows' window origins against one another mod 16 (RRowTable).
template <int IH_, int IW_, int HL_, int KH_, int KW_, int OH_, int OW_, int NT_, int G_, int NW_, int MT_, int ORDER_, int SS_, int WGS_, int NS_ = 1, int IC_ = 64, int S_ = 1, int RP_ = 0>
struct RGeom {
    static constexpr int IH = IH_, IW = IW_, HL = HL_, KH = KH_, KW = KW_, OH = OH_, OW = OW_, NT = NT_, G = G_, NW = NW_, MT = MT_, ORDER = ORDER_;
    static constexpr int IC = IC_, S = S_;                                          // source channels; stride of the window over the padded grid
    static constexpr int PIX = 4 * IC + 16, LO = 2 * IC, C16 = IC / 16, UPP = IC / 4;      // bytes per pixel record: 2 IC hi | 2 IC lo | 16 pad; chunks / 16-byte units per pixel
    static constexpr int SS = SS_;                                                  // k-steps per ring slot (one barrier per slot)
    static constexpr int WGS = WGS_;                                                // workgroups per CU the kernel is sized for (LDS, registers)
    // NS: the waves split the column tiles -- wave w multiplies row tiles of row-wave w % (NW / NS) against column tiles (w / (NW / NS)) NT / NS ..
    static constexpr int NS = NS_, NTW = NT / NS, RW = NW / NS;
    static constexpr int IHP = IH + 2 * HL, IWP = IW + 2 * HL, RP = RP_ > 0 ? RP_ : IWP, IPIX = IHP * RP, OP = OH * OW, ROWS = G * OP, SLOTS = 32 * MT * (NW / NS_);
    static constexpr int KSTEPS = KH * KW * C16, SPR = KW * C16;
    static constexpr int IMGB = IPIX * PIX, ABYTES = G * IMGB;
    static constexpr int STEPB = NT * 2048, SLOTB = SS * STEPB;                     // one k-step of the pack; ring slot
    static constexpr int UNITS = G * IH * IW * UPP;                                 // 16-byte units of a group's source
    // Record index of padded pixel (y, x).  Stride 2: the columns are stored even ones first, then the odd ones, so that the windows of
    // consecutive outputs start at consecutive records (tap column tx picks the half); additive in (window origin) + (tap).
    static constexpr int pidx(int y, int x) { return S == 2 ? y * RP + (x & 1) * (IWP / 2) + (x >> 1) : y * RP + x; }
    static_assert(RP >= IWP && (S == 1 || (S == 2 && IWP % 2 == 0)) && IC % 16 == 0 && (PIX / 16) % 2 == 1, "strides; whole chunks; an odd record pitch in 16-byte slots");
    static constexpr int THREADS = 64 * NW, NI = (UNITS + THREADS - 1) / THREADS;
    static_assert(ROWS <= SLOTS && KSTEPS % SS == 0 && NT % NS == 0 && NW % NS == 0, "a group's rows fit the waves' tiles; whole ring slots");
    static_assert((ABYTES + 2 * SLOTB) * WGS <= 160 * 1024, "LDS");
};
// One persistent workgroup per CU.  Layer 3 (NT = 2): eight waves of 32 rows, two per SIMD -- 384 / 527 us where four waves of 64 rows (one per
// SIMD, half the weight-fragment reads per MFMA) took 440 / 585; the layer-2 data gradient (NT = 4: 64 accumulators per 32-row tile): eight
// waves of 64 rows, splitting the class tiles two and two (NS = 2), 750 us against 850 (four waves, all tiles) and 915 (eight waves of 32 rows);
// the layer-2 forward: twelve waves of 32 rows (three per SIMD), splitting the two column tiles -- 6 x 32 row slots for its 162 rows: 775 us
// against 860 with eight waves and 256 slots.  All measured shapes: profiles/r05_tile_shape_experiments.txt.
// (RConv2's rows of 21 records: with 20 the window origins 40 gy + gx fall on two residues mod 16 per column -- 18 rows on residues 0 and 8 for
//  12 sixteen-lane sets: at least 12 two-way conflicts per fragment read of the six tiles, 23 with the empty slots; measured 39 % of the
//  kernel's LDS cycles as conflicts (profiles/r05_pmc_lds.csv).  With 21: at most 12 rows per residue, a conflict-free table.  +5.8 KB.)
using RConv2 = RGeom<20, 20, 0, 4, 4, 9, 9, 2, 2, 12, 1, 2, 4, 1, 2, 32, 2, 21>;  // a1 (20, 20, 32) -> a2 (9, 9, 64), stride 2: 162 rows of 192 (a1 is 51 KB per image: two images per 160 KB)
using RConv3 = RGeom<9, 9, 0, 3, 3, 7, 7, 2, 5, 8, 1, 1, 6, 1>;          // a2 (9, 9, 64) -> a3 (7, 7, 64): 245 rows of 256
using RDgrad3 = RGeom<7, 7, 2, 3, 3, 9, 9, 2, 3, 8, 1, 0, 6, 1>;         // dz3 (7, 7, 64) -> da2 (9, 9, 64): 243 rows of 256
using RDgrad2 = RGeom<9, 9, 1, 2, 2, 10, 10, 4, 2, 8, 2, 0, 4, 1, 2>;       // dz2 (9, 9, 64) -> da1 (20, 20, 32), four stride-parity classes = four column tiles: 200 rows of 256

// Which row of the group sits in which lane.  A fragment read (ds_read_b128, lane = row) is served in groups of 16 lanes -- {0-3, 12-15,
// 20-27} and {4-11, 16-19, 28-31} of each wave half -- and is conflict-free when the 16 records start in 16 different sixteen-byte slots of
// the 256-byte bank row, i.e. (record pitch = 17 slots) when the 16 window origins differ mod 16.  Raster order does not give that (a
// line of 7 or 9 outputs, then a jump); any order of the rows is as good as any other for everything else, so the table deals the rows
// to the 16-lane sets by the residue of their window origin: set s never gets a residue twice while another set can still take it.
// Slot = (wave MT + tile) 32 + lane % 32.  `row`: the group's row in the slot, -1 = none; `src`: the row a slot computes -- its own, or
// (empty slots) a row of the SAME sixteen-lane set: the same address as that lane's, a broadcast, where row 0 would be one more record on
// some bank; an empty slot computes and stores that row's values a second time.
template <class RG>
struct RRowTable {
    short row[RG::SLOTS], src[RG::SLOTS];
    constexpr RRowTable() : row{}, src{} {
        constexpr int NSET = RG::SLOTS / 16;
        int fill[NSET] = {};
        bool has[NSET][16] = {};
        int slot_of[NSET][16] = {};                        // k-th lane (0..15) of set s -> slot
        for (int s = 0; s < NSET; ++s) {
            int k = 0;
            for (int l = 0; l < 32; ++l) {
                const bool first = l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28);
                if (first == ((s & 1) == 0)) slot_of[s][k++] = (s >> 1) * 32 + l;
            }
        }
        for (int i = 0; i < RG::SLOTS; ++i) row[i] = -1;
        int next = 0;
        for (int pass = 0; pass < 2; ++pass)               // pass 0: conflict-free placements only; pass 1: whatever is left, anywhere
            for (int r = 0; r < RG::ROWS; ++r) {
                const int gi = r / RG::OP, p = r - gi * RG::OP, gy = p / RG::OW, gx = p - gy * RG::OW;
                const int c = (gi * RG::IPIX + RG::pidx(RG::S * gy, RG::S * gx)) & 15;
                bool placed = false;
                for (int i = 0; i < RG::SLOTS && !placed; ++i) placed = row[i] == r;
                for (int t = 0; t < NSET && !placed; ++t) {
                    const int s = (next + t) % NSET;
                    if (fill[s] < 16 && (pass == 1 || !has[s][c])) {
                        row[slot_of[s][fill[s]++]] = (short)r;
                        has[s][c] = true;
                        placed = true;
                        next = s + 1;
                    }
                }
            }
        for (int s = 0; s < NSET; ++s) {
            int have = 0;                                   // (a set without any row: all sixteen lanes on row 0, one address)
            for (int k = 0; k < 16; ++k)
                if (row[slot_of[s][k]] >= 0) { have = row[slot_of[s][k]]; break; }
            for (int k = 0; k < 16; ++k) src[slot_of[s][k]] = row[slot_of[s][k]] >= 0 ? row[slot_of[s][k]] : (short)have;
        }
    }
};

// visited index v -> (k-step of the pack, byte offset of the step's hi fragment from the lane's window origin)
template <class RG>
R_GEOM_FN int r_kstep(int v) {
    if constexpr (RG::ORDER == 1) {                       // kernel Z's phase order of the 3 x 3 / 64-channel forward (z_kstep)
        const int lp = v >= 18 ? 1 : 0, w = v - 18 * lp, combo = w >> 1;
        const int ty = combo / 3, tx = combo - 3 * ty;
        return ty * RG::SPR + 4 * tx + 2 * lp + (w & 1);
    } else if constexpr (RG::ORDER == 2) {                // ... of the 4 x 4 stride-2 / 32-channel forward: (row parity, column parity) phases
        const int g = v >> 3, w = v & 7;
        const int ty = (g >> 1) + 2 * (w >> 2), tx = (g & 1) + 2 * ((w >> 1) & 1);
        return ty * RG::SPR + 2 * tx + (w & 1);
    } else {
        return v;
    }
}
template <class RG>
R_GEOM_FN int r_tapoff(int ks) {
    const int ty = ks / RG::SPR, us = ks - ty * RG::SPR, tx = us / RG::C16, chunk = us - tx * RG::C16;
    return RG::pidx(ty, tx) * RG::PIX + chunk * 32;
}

}  // namespace mi355ppo
