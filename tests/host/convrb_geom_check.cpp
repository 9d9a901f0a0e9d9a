// Host-side check of kernel RB's compile-time geometry (cleanrl_amd/csrc/convrb_geom.h), compiled with g++ by tests/test_kernel_r_geometry.py.
// Prints one JSON object; exits non-zero on the first violated property.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "convrb_geom.h"

using namespace mi355ppo;

#define REQUIRE(cond, ...)                                   \
    do {                                                     \
        if (!(cond)) {                                       \
            std::fprintf(stderr, "RBGeom: ");                \
            std::fprintf(stderr, __VA_ARGS__);               \
            std::fprintf(stderr, "\n");                      \
            std::exit(1);                                    \
        }                                                    \
    } while (0)

int main() {
    using RG = RBGeom;
    static constexpr RBRowTable table{};
    // ---- every (image, gy, gx) of the group sits in exactly one slot
    std::vector<int> seen(RG::G * 100, 0);
    int empty = 0;
    for (int i = 0; i < RG::SLOTS; ++i) {
        const int r = table.row[i];
        if (r < 0) { ++empty; continue; }
        REQUIRE(r < RG::G * 100, "slot %d holds row id %d", i, r);
        ++seen[r];
    }
    for (int r = 0; r < RG::G * 100; ++r) REQUIRE(seen[r] == 1, "row id %d sits in %d slots", r, seen[r]);
    REQUIRE(empty == RG::SLOTS - RG::ROWS, "%d empty slots, %d expected", empty, RG::SLOTS - RG::ROWS);
    // ---- a tile visits every tap of its rows that lies inside the image (the taps it skips read the zero border for ALL its rows), and every
    // visited read stays inside the image's records
    int visited = 0, skipped_zero = 0, visited_zero = 0;
    for (int rw = 0; rw < RG::RW; ++rw)
        for (int t = 0; t < RG::MT; ++t)
            for (int l = 0; l < 32; ++l) {
                const int slot = (rw * RG::MT + t) * 32 + l, r = table.src[slot];
                REQUIRE(r >= 0 && r < RG::G * 100 && (table.row[slot] < 0 || table.row[slot] == r), "slot %d computes row id %d", slot, r);
                const int gy = (r % 100) / 10, gx = r % 10;
                for (int s = 0; s < RG::NSLOT; ++s) {
                    const int ty = s >> 1, tx = s & 1;
                    const bool visits = t < RG::NI3 || RG::rim_of(rw, s) == t - RG::NI3;
                    const bool inside = RG::tap_inside(gy, gx, ty, tx);
                    REQUIRE(visits || !inside, "slot %d (row-wave %d, tile %d) row (%d, %d): tap (%d, %d) is inside the image but the tile skips ring slot %d", slot, rw, t, gy, gx, ty, tx, s);
                    if (visits) {
                        ++visited;
                        visited_zero += inside ? 0 : 1;
                        REQUIRE(RG::pidx(gy + ty, gx + tx) < RG::IPIX, "row (%d, %d) tap (%d, %d) reads past the image's records", gy, gx, ty, tx);
                        // the record read is the pixel's own -- or a border record: no image pixel maps onto a border record
                        for (int k = 0; k < 4; ++k) {
                            const int off = RG::pidx(gy, gx) * RG::PIX + rb_tapoff(s * 4 + k);
                            REQUIRE(off == RG::pidx(gy + ty, gx + tx) * RG::PIX + k * 32, "tap offsets are not additive");
                        }
                    } else {
                        ++skipped_zero;
                    }
                }
            }
    std::set<int> image_recs;
    for (int qy = 0; qy < RG::IH; ++qy)
        for (int qx = 0; qx < RG::IW; ++qx) image_recs.insert(RG::pidx(qy + 1, qx + 1));
    REQUIRE((int)image_recs.size() == RG::IH * RG::IW, "two image pixels share a record");
    for (int y = 0; y <= RG::IH + 1; ++y)
        for (int x = 0; x <= RG::IW + 1; ++x)
            if (y == 0 || y == RG::IH + 1 || x == 0 || x == RG::IW + 1) REQUIRE(!image_recs.count(RG::pidx(y, x)), "border pixel (%d, %d) shares a record with an image pixel", y, x);
    // ---- fragment reads: distinct records per sixteen-byte slot residue within each 16-lane set
    int conflicts = 0, interior_conflicts = 0;
    for (int tile = 0; tile < RG::SLOTS / 32; ++tile)
        for (int half = 0; half < 2; ++half) {
            std::set<long> at[16];
            for (int l = 0; l < 32; ++l) {
                if (RBRowTable::first_set(l) != (half == 0)) continue;
                const int r = table.src[tile * 32 + l];
                const long rec = (long)(r / 100) * RG::IPIX + RG::pidx((r % 100) / 10, r % 10);
                at[(rec * (RG::PIX / 16)) & 15].insert(rec);
            }
            for (int c = 0; c < 16; ++c) {
                const int extra = at[c].size() > 1 ? (int)at[c].size() - 1 : 0;
                conflicts += extra;
                if (tile % RG::MT < RG::NI3) interior_conflicts += extra;
            }
        }
    REQUIRE(interior_conflicts == 0, "%d bank conflicts in the interior tiles' fragment reads", interior_conflicts);
    REQUIRE(conflicts <= 4, "%d bank conflicts in the rim tiles' fragment reads", conflicts);
    std::printf("{\"instance\": \"RBGeom\", \"rows\": %d, \"slots\": %d, \"tile_slot_visits_per_group\": %d, \"visited_zero_taps\": %d, \"skipped_taps\": %d, \"conflicts\": %d, \"lds_bytes\": %d, \"rounds_per_thread\": %d}\n",
                RG::ROWS, RG::SLOTS, visited / 32, visited_zero, skipped_zero, conflicts, RG::LDSB, RG::NI);
    return 0;
}
