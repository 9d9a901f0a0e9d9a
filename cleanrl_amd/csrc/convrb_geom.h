// Kernel RB's compile-time geometry (convrb.hip): the layer-2 data gradient with the rows of a group dealt to tiles BY BORDER CLASS.
// Plain constexpr C++ -- no device code -- so that the host can check it (tests/test_kernel_r_geometry.py compiles
// tests/host/convrb_geom_check.cpp against this header with g++).
//
// The GEMM of a group (kernel R's RDgrad2, convr_geom.h): rows = the 10 x 10 grid of window origins over the zero-bordered 11 x 11 dz2 image,
// k = (tap row, tap column, 64 channels) of a 2 x 2 window = 16 k-steps, columns = 4 stride-parity classes x 32 channels.  A tap of a row on
// the grid's rim reads the zero border: 19 % of kernel R's products are exact zeros, and 2 images' 200 rows leave 56 of its 256 row slots empty.
// Here a group is THREE images (the right border record of a padded line IS the left border record of the next: 10 records per line, 117 per
// image instead of 121 -- three images + the weights' ring fit the 160 KB), and its 300 rows go to ten 32-row tiles by class:
//   * 6 interior tiles (3 x 64 rows with all four taps inside the image): 16 k-steps each;
//   * 4 rim tiles -- bottom line, top line, right column, left column (3 x 8 rows each + the three images' corner that shares the tile's
//     valid taps: 27 rows) -- that visit ONLY the two taps that can be inside the image: 8 k-steps each.
// 128 tile-k-steps per 3 images against kernel R's 128 per 2: two thirds of the matrix instructions, the same products in the same order
// minus exact zeros (bit-identical results).  Waves: 8 = 4 classes (one column tile each) x 2 row-waves; a row-wave holds 3 interior tiles +
// 2 rim tiles and in every ring slot (= one tap, 4 k-steps) multiplies its 3 interior tiles and the ONE rim tile that tap is valid for:
//   row-wave 0: bottom (taps of tap row 0: slots 0, 1), top (tap row 1: slots 2, 3);   row-wave 1: right (tap column 0: slots 0, 2), left (slots 1, 3)
// -- four tiles x one column tile x three products = the 12 matrix instructions per k-step kernel R issues, every wave, every step.
#pragma once

#if defined(__HIPCC__)
#define RB_GEOM_FN __device__ __forceinline__ constexpr
#else
#define RB_GEOM_FN inline constexpr
#endif

namespace mi355ppo {

struct RBGeom {
    static constexpr int IH = 9, IW = 9, IC = 64, OH = 10, OW = 10, OP = OH * OW;       // dz2 image; grid of window origins
    static constexpr int G = 3, NW = 8, NT = 4, RW = 2, MT = 5, NI3 = 3;                // images per group; waves; column tiles (classes); row-waves; tiles per row-wave (NI3 interior + 2 rim)
    static constexpr int RP = 10, IPIX = 117;                                           // records per padded line (shared border record); per image (111 used: the
                                                                                        // stride 117 = 5 mod 16 keeps the rim tiles' window origins apart, see RBRowTable)
    static constexpr int PIX = 4 * IC + 16, LO = 2 * IC;                                // record: 128 B hi | 128 B lo | 16 B pad (17 sixteen-byte slots: odd)
    static constexpr int KSTEPS = 16, SS = 4, NSLOT = 4, SPR = 8, C16 = 4;              // k-steps; per ring slot (= one tap); k-steps per tap row
    static constexpr int IMGB = IPIX * PIX, ABYTES = G * IMGB;
    static constexpr int STEPB = NT * 2048, SLOTB = SS * STEPB;
    static constexpr int ROWS = G * OP, SLOTS = 32 * MT * RW;
    static constexpr int UPP = IC / 4, UNITS = G * IH * IW * UPP, THREADS = 64 * NW, NI = (UNITS + THREADS - 1) / THREADS;
    static constexpr int ROFFB = RW * MT * 2 * 16 * 4;                                  // the epilogue's row offsets: [row-wave][tile][lane half][value] words
    static constexpr int LDSB = ABYTES + 2 * SLOTB + ROFFB;
    // record of padded pixel (y, x), 0 <= y, x <= 10: (y, 10) and (y + 1, 0) are the same (zero) record
    static constexpr int pidx(int y, int x) { return y * RP + x; }
    // the rim tile (index NI3 + e, e = 0 / 1) a row-wave multiplies in ring slot s = 2 (tap row) + (tap column)
    static constexpr int rim_of(int rw, int s) { return rw == 0 ? (s >> 1) : (s & 1); }
    // is tap (ty, tx) of window origin (gy, gx) inside the image?  (padded pixel (gy + ty, gx + tx), the image at 1 .. 9)
    static constexpr bool tap_inside(int gy, int gx, int ty, int tx) { return gy + ty >= 1 && gy + ty <= IH && gx + tx >= 1 && gx + tx <= IW; }
};

static_assert(RBGeom::pidx(RBGeom::IH + 1, RBGeom::IW + 1) < RBGeom::IPIX && RBGeom::LDSB <= 160 * 1024 && (RBGeom::PIX / 16) % 2 == 1 && RBGeom::ROWS <= RBGeom::SLOTS,
              "records, LDS, odd pitch, row slots");

RB_GEOM_FN int rb_tapoff(int ks) {                         // byte offset of k-step ks' hi fragment from the lane's window origin
    const int ty = ks / RBGeom::SPR, us = ks - ty * RBGeom::SPR, tx = us / RBGeom::C16, chunk = us - tx * RBGeom::C16;
    return RBGeom::pidx(ty, tx) * RBGeom::PIX + chunk * 32;
}

// Which row of the group sits in which lane: slot (row-wave MT + tile) 32 + lane % 32 -> row id = image 100 + gy 10 + gx (`row`: -1 = an empty
// slot; `src`: the row the slot computes -- its own, or for an empty slot a row of the same sixteen-lane set, i.e. the same LDS address: a
// broadcast; its values are stored a second time).  A fragment read (ds_read_b128, lane = row) is served in sets of 16 lanes and is
// conflict-free when the 16 window origins differ mod 16 (record pitch 17 slots).  Interior rows: the residue of image 117 + 10 gy + gx over
// gy, gx in 1 .. 8 takes every value 12 times per group = once per set of the six interior tiles: the k-th row of a residue goes to set k.
// Rim tiles: 27 rows in two sets; a residue occurs at most twice in the line tiles (conflict-free) and three times for two residues of
// each column tile (two two-way conflicts per column tile: 4 extra LDS cycles per 40 fragment reads of a rim tile).
struct RBRowTable {
    short row[RBGeom::SLOTS], src[RBGeom::SLOTS];
    static constexpr bool first_set(int l) { return l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28); }
    static constexpr int residue(int gi, int gy, int gx) { return (gi * RBGeom::IPIX + RBGeom::pidx(gy, gx)) & 15; }
    constexpr RBRowTable() : row{}, src{} {
        using RG = RBGeom;
        for (int i = 0; i < RG::SLOTS; ++i) row[i] = -1;
        // lane positions of a tile's two sets, in lane order
        int lane_of[2][16] = {};
        for (int h = 0; h < 2; ++h) {
            int k = 0;
            for (int l = 0; l < 32; ++l)
                if (first_set(l) == (h == 0)) lane_of[h][k++] = l;
        }
        // ---- interior: global interior tile t = 0 .. 5 -> row-wave t / 3, tile t % 3; set 2 t + half
        int seen[16] = {}, fill[12] = {};
        for (int gi = 0; gi < RG::G; ++gi)
            for (int gy = 1; gy <= 8; ++gy)
                for (int gx = 1; gx <= 8; ++gx) {
                    const int c = residue(gi, gy, gx), set = seen[c]++;      // (12 rows per residue: set < 12)
                    const int t = set >> 1, h = set & 1;
                    row[((t / 3) * RG::MT + t % 3) * 32 + lane_of[h][fill[set]++]] = (short)(gi * 100 + gy * 10 + gx);
                }
        // ---- rim tiles: (row-wave, tile) = (0, 3) bottom + corner (9, 9); (0, 4) top + (0, 0); (1, 3) right + (0, 9); (1, 4) left + (9, 0)
        for (int e = 0; e < 4; ++e) {
            const int rw = e >> 1, tile = RG::NI3 + (e & 1), base = (rw * RG::MT + tile) * 32;
            bool has[2][16] = {};
            int n[2] = {};
            for (int pass = 0; pass < 2; ++pass)               // pass 0: conflict-free placements only; pass 1: whatever is left
                for (int gi = 0; gi < RG::G; ++gi)
                    for (int k = 0; k < 9; ++k) {
                        int gy = 0, gx = 0;
                        if (e == 0) { gy = 9; gx = 1 + k; }            // bottom line 1 .. 8, then the corner (9, 9)
                        else if (e == 1) { gy = 0; gx = k; }           // top: the corner (0, 0), then 1 .. 8
                        else if (e == 2) { gy = k; gx = 9; }           // right: the corner (0, 9), then lines 1 .. 8
                        else { gy = 1 + k; gx = 0; }                   // left: lines 1 .. 8, then the corner (9, 0)
                        const short id = (short)(gi * 100 + gy * 10 + gx);
                        bool placed = false;
                        for (int l = 0; l < 32 && !placed; ++l) placed = row[base + l] == id;
                        const int c = residue(gi, gy, gx);
                        for (int h = 0; h < 2 && !placed; ++h)
                            if (n[h] < 16 && (pass == 1 || !has[h][c])) {
                                row[base + lane_of[h][n[h]++]] = id;
                                has[h][c] = true;
                                placed = true;
                            }
                    }
        }
        // ---- empty slots compute a row of their own set
        for (int t = 0; t < RG::SLOTS / 32; ++t)
            for (int h = 0; h < 2; ++h) {
                int have = -1;
                for (int k = 0; k < 16; ++k)
                    if (row[t * 32 + lane_of[h][k]] >= 0) { have = row[t * 32 + lane_of[h][k]]; break; }
                for (int k = 0; k < 16; ++k) {
                    const int s = t * 32 + lane_of[h][k];
                    src[s] = row[s] >= 0 ? row[s] : (short)have;
                }
            }
    }
};

}  // namespace mi355ppo
