// Kernel U -- the weight (+ bias) gradients of the NatureCNN's convolutions on the two-term f16 split with BOTH operands of a group of images
// resident in LDS, split once.  Described for layer 3; layer 2 = the LW = 0 form of UGeom, layer 1 = convu1_kernel further down (kernel R's idea applied to kernel V's problem; cleanrl/ppo_atari_multigpu.py:141 and its backward, :358):
//     dW[co][(ty, tx, ci)] = sum over output pixels p of dz[p][co] * src[pixel of p shifted by tap (ty, tx)][ci]
// The reduction index is the PIXEL, the slow index of both tensors in memory.  Kernel V (convw.hip) transposes both operands through
// wave-private LDS with 4-byte reads and splits every fragment in registers -- 32 LDS reads and ~100 VALU instructions per 12 MFMAs, every
// source pixel fetched and split once per tap (profiles/r05_pmc_*.csv: VALU 0.54 busy, matrix pipe 0.31).  Here a workgroup
//   * loads the f32 source (9 x 9 x 64) and dz (7 x 7 x 64) of G = 4 images once, splits every element once (each tensor's scale from its amax
//     record) and stores hi / lo halves as pixel records (128 B hi | 128 B lo | 16 B pad, as kernel R);
//   * reads MFMA fragments with `ds_read_b64_tr_b16` -- the LDS transpose read of gfx950: a 16-lane group reads a [4 pixels][16 channels]
//     block, four contiguous halves per lane, and every lane receives one channel's four pixels (tools/tr_probe.cpp) -- so "8 consecutive k of
//     one channel" is two such reads, no VALU, no address arithmetic (compile-time offsets; one select per k-step for the image the step's
//     two pixel lines lie in);
//   * keeps the whole dW (64 x 576 = 36 tiles of 32 x 32) in the accumulators of its 12 waves (3 tiles each, 3 waves per SIMD) across ALL its
//     groups and writes one partial per workgroup at the end; conv.hip's two-stage reduce adds the partials in a fixed order (deterministic).
// k order: the reduction runs over pixel LINES padded to 8 (7 outputs + one zero dz record), so that a block of four k is four consecutive
// pixels of one line (consecutive records for both operands: dz's line, and the source line shifted by the tap); the padded position
// multiplies a zero dz by a finite source value.  16 k = two lines per k-step, 14 k-steps per group of 4 images.
// The bias gradient: every thread adds the raw f32 dz values of the 16-byte units it loads (its four channels are the same for every unit:
// 768 threads, 16 units per pixel), folded per channel in a fixed order at the end.
// Arithmetic: exact products of the f16 terms (hi hi, hi lo, lo hi), f32 accumulation in another order than kernel V's (per-workgroup sums
// over whole images instead of pixel slabs): held to float64 with kernel V's bars, not bit-compared (tests/test_gpu_f16x2.py).
#include "common.h"
#include "f16split.h"

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float u_f32x16 __attribute__((ext_vector_type(16)));
typedef short u_s16x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kUOob = 0xFFFFF000u;
constexpr int kURsrcWord3 = 0x00020000;

// Source (images, SH, SW, C), window KH x KW at stride S; dz (images, OH, OW, 64); G images per group on NW waves.
// LW > 0 (layer 3): the reduction runs over dz LINES padded to LW records, a k-step = 16 / LW lines -- the source record of a block row is
//   (compile-time record of its line) + x + tap.
// LW = 0 (layer 2): the reduction runs over the group's pixels in raster order, padded at the END of the group to whole k-steps (9-pixel
//   lines would pad to 12); the window origin of every pixel a lane ever touches sits in a per-lane table of 16-bit record numbers
//   (2 KSTEPS entries, filled once per launch: the same for every group).
template <int SH_, int SW_, int C_, int KH_, int KW_, int S_, int OH_, int OW_, int LW_, int G_, int NW_, int OCC_>
struct UGeom {
    static constexpr int SH = SH_, SW = SW_, C = C_, KH = KH_, KW = KW_, S = S_, OH = OH_, OW = OW_, LW = LW_, G = G_, NW = NW_, OCC = OCC_, CO = 64;
    static constexpr int THREADS = 64 * NW, OP = OH * OW;
    static constexpr int PIXS = 4 * C + 16, LOS = 2 * C, PIXD = 4 * CO + 16, LOD = 2 * CO;       // pixel records: 2 C bytes hi | 2 C bytes lo | 16 B pad (an odd number of 16-byte slots)
    static constexpr int SRC_REC = SH * SW;                                                     // source records per image
    static constexpr int KPIX = LW > 0 ? G * OH * LW : (G * OP + 15) / 16 * 16, KSTEPS = KPIX / 16;      // dz records per group = k per group
    static constexpr int SRC_SLACK = LW > 0 ? 3 : 0;               // LW > 0: the padded position of the last line reads up to 2 records past the group: zeroed slack
    static constexpr int SRCB = (G * SRC_REC + SRC_SLACK) * PIXS, DZB = KPIX * PIXD;
    static constexpr int UPS = C / 4, UPD = CO / 4;                // 16-byte units per source / dz pixel
    static constexpr int SRC_UNITS = G * SRC_REC * UPS, DZ_UNITS = G * OP * UPD;
    static constexpr int NIS = (SRC_UNITS + THREADS - 1) / THREADS, NID = (DZ_UNITS + THREADS - 1) / THREADS, NI = NIS + NID;      // rounds of loads per thread: a round is all source or all dz
    static constexpr int PRE_PER = (NI + KSTEPS - 2) / (KSTEPS - 1);                            // loads of the next group per k-step
    static constexpr int CT = C / 32, NTILES = KH * KW * CT, TPW = 2 * NTILES / NW;              // column tiles (tap, 32-channel part); tiles per wave (one of the two co tiles each)
    static constexpr int pidx(int y, int x) { return S == 2 ? y * SW + (x & 1) * (SW / 2) + (x >> 1) : y * SW + x; }      // (stride 2: even columns first, as kernel R)
    static_assert(KPIX % 16 == 0 && 2 * NTILES % NW == 0 && THREADS % 16 == 0 && THREADS % UPS == 0 && SRCB + DZB <= 160 * 1024 && (S == 1 || SW % 2 == 0) && (LW == 0 || 16 % LW == 0), "shape");
};
using UGeom3 = UGeom<9, 9, 64, 3, 3, 1, 7, 7, 8, 4, 12, 3>;         // a2 / dz3: 4 images, 14 k-steps, 12 waves x 3 tiles, 149 KB
using UGeom2 = UGeom<20, 20, 32, 4, 4, 2, 9, 9, 0, 2, 8, 2>;        // a1 / dz2: 2 images = 162 pixels in 11 k-steps, 8 waves x 4 tiles, 159 KB

template <class UG>
__global__ __launch_bounds__(UG::THREADS) __attribute__((amdgpu_waves_per_eu(UG::OCC, UG::OCC))) void convu_kernel(
    const float* __restrict__ src, const float* __restrict__ dz, float* __restrict__ part_w, float* __restrict__ part_b, long long images,
    int groups, unsigned src_bytes, unsigned dz_bytes, const unsigned* __restrict__ dz_amax, const unsigned* __restrict__ src_amax) {
    constexpr int NI = UG::NI, TPW = UG::TPW, PIXS = UG::PIXS, PIXD = UG::PIXD;
    __shared__ __attribute__((aligned(16))) unsigned char lds[UG::SRCB + UG::DZB];
    unsigned char* const lsrc = lds;
    unsigned char* const ldz = lds + UG::SRCB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15, r = i >> 2, c4 = i & 3;             // 16-lane group; its lane's block row / column chunk
    const int ct = wave & 1, ng = wave >> 1;                                    // co tile; first column tile / TPW

    const int es = f16_scale_exp(amax_load(src_amax, lane)), ed = f16_scale_exp(amax_load(dz_amax, lane));
    const float ss = f16_pow2(es), sd = f16_pow2(ed), un = f16_unscale(es, ed);

    // zero everything once: dz's padded records and the source slack stay zero (the fill writes real pixels only)
    for (int o = tid * 16; o < UG::SRCB + UG::DZB; o += UG::THREADS * 16) *reinterpret_cast<s_u32x4*>(lds + o) = (s_u32x4){0u, 0u, 0u, 0u};

    // ---- fragment addresses.  Lane (g, r, c4) passes the address of 4 contiguous halves: row r of its group's block, channels 16 (g & 1) +
    // 4 c4 .. + 3 of the tile.  Which four pixels form a block is free (any bijection of the step's 16 k, the same for both operands): block
    // b = 2 (g >> 1) + t (t: first / second read of a fragment) takes pixels b, b + 4, b + 8, b + 12 of the step -- rows FOUR records apart:
    // 4 x 272 B = 8 eight-byte chunks (mod 32), so the 32 lanes of a pass (4 rows x 2 groups x 4 chunks) cover all 64 banks once.  (Four
    // CONSECUTIVE pixels per block put the rows 2 chunks apart: ~3-way conflicts, the B reads were 43 % of the kernel's time.)
    const int half = g >> 1;
    const unsigned char* const dz_lane = ldz + (4 * r + 2 * half) * PIXD + (32 * ct + 16 * (g & 1) + 4 * c4) * 2;       // + 16 s records, + t records (+ LOD)
    // LW > 0: pixel 4 r + b of the step = line (16 / LW) s + (4 r + b) / LW, x = (4 r + b) % LW; with LW = 8: line 2 s + (r >> 1), x = 4 (r & 1) + b
    const int rline = r >> 1;
    const unsigned char* const src_lane = lsrc + (UG::LW > 0 ? (4 * (r & 1) + 2 * half) * PIXS : 0) + (16 * (g & 1) + 4 * c4) * 2;      // + record of (pixel | line, tap), + t records, + 64 cpart (+ LOS)
    // LW = 0: the window origins of pixels 16 s + 4 r + 2 half + t, s < KSTEPS, t < 2, two 16-bit record numbers per register (pixels past the
    // group: record 0 -- their dz is zero)
    unsigned origin[UG::LW > 0 ? 1 : UG::KSTEPS];
    if constexpr (UG::LW == 0) {
#pragma unroll
        for (int s = 0; s < UG::KSTEPS; ++s) {
            unsigned pk = 0u;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int pix = 16 * s + 4 * r + 2 * half + t, gi = pix / UG::OP, p = pix - gi * UG::OP, y = p / UG::OW, x = p - y * UG::OW;
                const unsigned rec = pix < UG::G * UG::OP ? (unsigned)(gi * UG::SRC_REC + UG::pidx(UG::S * y, UG::S * x)) : 0u;
                pk |= rec << (16 * t);
            }
            origin[s] = pk;
        }
    }

    // ---- the group's source and dz: unit = 16 bytes = 4 channels of a pixel; in round `it` thread tid takes unit it * THREADS + tid of the source
    // (rounds 0 .. NIS - 1) or of dz (the rest); a thread's dz units all carry the channel quad tid % 16
    const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)src_bytes, kURsrcWord3);
    const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, (int)dz_bytes, kURsrcWord3);
    s_u32x4 pre[NI];
    float db4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto unit_dst = [&](int it) __attribute__((always_inline)) -> unsigned {      // LDS byte offset of the unit's hi half; ~0u: no such unit
        if (it < UG::NIS) {
            const int u = it * UG::THREADS + tid, pix = u / UG::UPS, q4 = u - pix * UG::UPS, gi = pix / UG::SRC_REC, q = pix - gi * UG::SRC_REC, qy = q / UG::SW, qx = q - qy * UG::SW;
            return u < UG::SRC_UNITS ? (unsigned)((gi * UG::SRC_REC + UG::pidx(qy, qx)) * PIXS + q4 * 8) : ~0u;
        }
        const int v = (it - UG::NIS) * UG::THREADS + tid, q = v >> 4, gi = q / UG::OP, p = q - gi * UG::OP, y = p / UG::OW, x = p - y * UG::OW;
        // LW > 0: line L = gi OH + y of the group; ODD lines are stored rotated by one record (pixel x in slot x + 1, the zero pad in slot 0), so that
        // the source rows of a transpose-read block -- two of line L, two of line L + 1, read one record to the LEFT on odd lines -- sit 0, 4, 8, 12
        // records apart (9 - 1 = 8 between the lines) = four different quarters of the 64 banks; unrotated they sat 0, 4, 9, 13 apart and the last
        // row wrapped onto the first one's banks (tools/lds_conflicts_u.py: 2.00 -> 1.36 LDS cycles per pass; what is left: the image boundaries)
        const int rec = UG::LW > 0 ? (gi * UG::OH + y) * UG::LW + (((gi * UG::OH + y) & 1) ? x + 1 : x) : q;
        return v < UG::DZ_UNITS ? (unsigned)(UG::SRCB + rec * PIXD + (v & 15) * 8) : ~0u;
    };
    auto prefetch = [&](int grp, int it) __attribute__((always_inline)) {        // units past the tensors (last group, groups past the end) load zeros
        const bool any = grp < groups;
        if (it < UG::NIS) {
            const int u = it * UG::THREADS + tid;
            pre[it] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_s, (any && u < UG::SRC_UNITS) ? (unsigned)grp * (unsigned)(UG::SRC_UNITS * 16) + (unsigned)u * 16u : kUOob, 0, MI355_AUX_WGRAD_LD));
        } else {
            const int v = (it - UG::NIS) * UG::THREADS + tid;
            pre[it] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_d, (any && v < UG::DZ_UNITS) ? (unsigned)grp * (unsigned)(UG::DZ_UNITS * 16) + (unsigned)v * 16u : kUOob, 0, MI355_AUX_WGRAD_LD));
        }
    };
    auto fill = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const bool is_dz = it >= UG::NIS;
            unsigned hi[2], lo[2];
            f16_split4(pre[it], is_dz ? sd : ss, hi, lo);
            const unsigned d = unit_dst(it);
            if (d != ~0u) {
                *reinterpret_cast<uint2*>(lds + d) = make_uint2(hi[0], hi[1]);
                *reinterpret_cast<uint2*>(lds + d + (is_dz ? UG::LOD : UG::LOS)) = make_uint2(lo[0], lo[1]);
            }
            if (is_dz) {                                  // (units past the end loaded zeros)
#pragma unroll
                for (int c = 0; c < 4; ++c) db4[c] += __uint_as_float(pre[it][c]);
            }
        }
    };

    u_f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    typedef u_s16x4 __attribute__((address_space(3))) * lds_v4;
    auto tr2 = [&](const unsigned char* p0, const unsigned char* p1) __attribute__((always_inline)) -> s_u32x4 {      // 8 k of this lane's channel: two transpose reads (blocks b, b + 1)
        const u_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0));
        const u_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p1));
        const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
        return (s_u32x4){ua.x, ua.y, ub.x, ub.y};
    };

    int grp = blockIdx.x;
#pragma unroll
    for (int it = 0; it < NI; ++it) prefetch(grp, it);
    for (; grp < groups; grp += gridDim.x) {
        __syncthreads();                                  // every wave is done with the previous group's records (first pass: the zero fill)
        fill();
        __syncthreads();
        // (step, tile) pairs in one software pipeline: the transpose reads of pair q + 1 go out before the matrix instructions of pair q (a wave
        // cannot run ahead of the matrix pipe; the compiler's own order was "all reads of a step, wait, its MFMAs")
        s_u32x4 afr[2][2], bfr[2][2];                     // [parity][hi, lo]: dz fragment of a step; source fragment of a pair
        auto load_a = [&](int par, int s) __attribute__((always_inline)) {
            const unsigned char* const dline = dz_lane + 16 * s * PIXD;
            afr[par][0] = tr2(dline, dline + PIXD);
            afr[par][1] = tr2(dline + UG::LOD, dline + PIXD + UG::LOD);
        };
        auto load_b = [&](int par, int s, int t) __attribute__((always_inline)) {
            // this lane's two source block rows of the step (first / second read): window origins without the tap
            const unsigned char *s0, *s1;
            if constexpr (UG::LW > 0) {                   // the step's pixel lines L = 2 s + (0 | 1) -> image L / OH, row L % OH; the block row lies in line r >> 1
                constexpr int OH = UG::OH;
                const int L0 = 2 * s, L1 = 2 * s + 1;
                const int rec0 = (L0 / OH) * UG::SRC_REC + (L0 % OH) * UG::SW, rec1 = (L1 / OH) * UG::SRC_REC + (L1 % OH) * UG::SW;
                s0 = src_lane + (rline ? rec1 - 1 : rec0) * PIXS;      // (line 2 s + 1 is odd: its dz slots hold pixel x - 1 -- see unit_dst)
                s1 = s0 + PIXS;
            } else {
                s0 = src_lane + (origin[s] & 0xffffu) * PIXS;
                s1 = src_lane + (origin[s] >> 16) * PIXS;
            }
            const int tile = ng * TPW + t, tap = tile / UG::CT, cpart = tile - tap * UG::CT;      // (wave-uniform)
            const int ty = tap / UG::KW, tx = tap - ty * UG::KW;
            const int off = UG::pidx(ty, tx) * PIXS + 64 * cpart;
            bfr[par][0] = tr2(s0 + off, s1 + off);
            bfr[par][1] = tr2(s0 + off + UG::LOS, s1 + off + UG::LOS);
        };
        load_a(0, 0);
        load_b(0, 0, 0);
#pragma unroll
        for (int q = 0; q < UG::KSTEPS * TPW; ++q) {
            const int s = q / TPW, t = q - s * TPW;
            if (t == 0 && s >= 1) {                       // the next group's source and dz, PRE_PER loads per k-step
#pragma unroll
                for (int u = 0; u < UG::PRE_PER; ++u)
                    if ((s - 1) * UG::PRE_PER + u < NI) prefetch(grp + gridDim.x, (s - 1) * UG::PRE_PER + u);
            }
            if (q + 1 < UG::KSTEPS * TPW) {
                const int s1 = (q + 1) / TPW, t1 = (q + 1) - s1 * TPW;
                if (t1 == 0) load_a(s1 & 1, s1);
                load_b((q + 1) & 1, s1, t1);
            }
            __builtin_amdgcn_sched_barrier(0);
            const s_u32x4 a_hi = afr[s & 1][0], a_lo = afr[s & 1][1], b_hi = bfr[q & 1][0], b_lo = bfr[q & 1][1];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, a_hi), __builtin_bit_cast(s_f16x8, b_hi), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, a_hi), __builtin_bit_cast(s_f16x8, b_lo), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, a_lo), __builtin_bit_cast(s_f16x8, b_hi), acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- this workgroup's partial dW: accumulator e of tile t = row co = 32 ct + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column (tap, ci = 32 cpart + lane % 32)
    constexpr int K = UG::KH * UG::KW * UG::C;
    float* const pw = part_w + (size_t)blockIdx.x * (UG::CO * K);
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = ng * TPW + t, tap = tile / UG::CT, cpart = tile - tap * UG::CT;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = 32 * ct + (e & 3) + 8 * (e >> 2) + 4 * lh;
            pw[co * K + tap * UG::C + 32 * cpart + li] = acc[t][e] * un;
        }
    }
    // ---- and partial db: the threads' sums, channel quad tid % 16, folded in the order of the thread index
    __syncthreads();
    float* const red = reinterpret_cast<float*>(lds);     // [THREADS / 16][64]
#pragma unroll
    for (int c = 0; c < 4; ++c) red[(tid >> 4) * 64 + 4 * (tid & 15) + c] = db4[c];
    __syncthreads();
    if (tid < UG::CO) {
        float sum = 0.0f;
        for (int k = 0; k < UG::THREADS / 16; ++k) sum += red[k * 64 + tid];
        part_b[(size_t)blockIdx.x * UG::CO + tid] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------- layer 1
// The layer-1 weight gradient the same way (kernel P's problem, conv1p.hip: dW1[co][(ty, tx, ci)] = sum over the 20 x 20 output pixels of
// dz1[p][co] * frame[4 y + ty][4 x + tx][ci] / 255, the uint8 frames gathered through `inds`).  One IMAGE per pass:
//   * the frame sits in LDS as 16-bit integers, 84 rows of 90 four-channel slots (8 bytes; 6 slots of padding per row): 0x00vv IS the f16
//     subnormal v * 2^-24, which the f16 MFMA multiplies exactly (tools/mfma_denorm.cpp) -- one plane, no split, no conversion;
//   * dz1 as hi / lo records of 144 bytes (32 channels), 400 of them in pixel order;
//   * a k-step is a 4 x 4 PATCH of output pixels (25 patches): block b = column x0 + b, its four rows = the patch's four lines.  Both operands'
//     block rows are then one output line apart -- 20 dz records = 360 eight-byte chunks, and 4 frame rows = 4 * 90 slots = 360 chunks: 8 (mod
//     32) each, the transpose reads are conflict-free -- and every address is (lane part) + (compile-time part of the step, tap row, read);
//   * a column tile = one tap row ty: its 32 columns (tx, ci) are 32 contiguous halves of frame row 4 y + ty starting at slot 4 x.
// Eight waves, one tap row each, two per SIMD (four waves with two tap rows each, one per SIMD: 682 us against 601);
// dW1 (32 x 256) stays in the accumulators across all images of the workgroup; 2 matrix instructions (dz hi, dz lo) per tile and k-step.  Partial sums carry 2^(e_dz - 24): removed on the way out; conv.hip's reduce applies 1 / 255.
template <int NW_>
struct UGeom1 {
    static constexpr int FH = 84, FW = 84, FC = 4, FWP = 90, OH = 20, OW = 20, CO = 32, NW = NW_, THREADS = 64 * NW, TPW = 8 / NW;
    static constexpr int SLOT = 8, FROW = FWP * SLOT, FB = FH * FROW;           // frame: 8-byte slots, 720-byte rows, 60,480 bytes
    static constexpr int PIXD = 4 * CO + 16, LOD = 2 * CO, DZB = OH * OW * PIXD; // dz records: 64 B hi | 64 B lo | 16 B pad
    static constexpr int F_UNITS = FH * FW * FC / 16, D_UNITS = OH * OW * CO / 4; // 16-byte units: 1,764 of the frame (4 pixels each), 3,200 of dz
    static constexpr int NIF = (F_UNITS + THREADS - 1) / THREADS, NID = (D_UNITS + THREADS - 1) / THREADS, NI = NIF + NID;
    static constexpr int KSTEPS = (OH / 4) * (OW / 4);                          // 25 patches
    static_assert(OH % 4 == 0 && OW % 4 == 0 && THREADS % 8 == 0 && FW % 4 == 0 && NI <= KSTEPS - 1 && FB + DZB <= 160 * 1024, "shape");
};

template <class UG>
__global__ __launch_bounds__(UG::THREADS) __attribute__((amdgpu_waves_per_eu(UG::NW / 4, UG::NW / 4))) void convu1_kernel(
    const unsigned char* __restrict__ frames, const long long* __restrict__ inds, const float* __restrict__ dz, float* __restrict__ part_w,
    float* __restrict__ part_b, int images, unsigned dz_bytes, const unsigned* __restrict__ dz_amax) {
    constexpr int NI = UG::NI, PIXD = UG::PIXD;
    __shared__ __attribute__((aligned(16))) unsigned char lds[UG::FB + UG::DZB];
    unsigned char* const lfr = lds;
    unsigned char* const ldz = lds + UG::FB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = tap row ty of its column tile
    const int g = lane >> 4, i = lane & 15, r = i >> 2, c4 = i & 3, half = g >> 1;

    const int ed = f16_scale_exp(amax_load(dz_amax, lane));
    const float sd = f16_pow2(ed), un = f16_pow2(24 - ed);                      // (ed in [-100, 60]: 2^(24 - ed) is a normal f32)

    for (int o = tid * 16; o < UG::FB + UG::DZB; o += UG::THREADS * 16) *reinterpret_cast<s_u32x4*>(lds + o) = (s_u32x4){0u, 0u, 0u, 0u};

    // block b = 2 half + t = patch column, block row r = patch line: dz record (y0 + r) * 20 + x0 + b; frame slot (4 (y0 + r) + ty, 4 (x0 + b))
    const unsigned char* const dz_lane = ldz + (r * UG::OW + 2 * half) * PIXD + (16 * (g & 1) + 4 * c4) * 2;
    const unsigned char* const fr_lane = lfr + (4 * r + wave * UG::TPW) * UG::FROW + 4 * (2 * half) * UG::SLOT + (16 * (g & 1) + 4 * c4) * 2;      // + tile's tap row

    const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, (int)dz_bytes, kURsrcWord3);
    s_u32x4 pre[NI];
    float db4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto prefetch = [&](int img, int it) __attribute__((always_inline)) {        // images past the batch: zeros (the frame through a null range, dz out of range)
        const bool any = img < images;
        if (it < UG::NIF) {
            const int u = it * UG::THREADS + tid;
            const long long row = any ? (inds ? inds[img] : (long long)img) : 0;
            const __amdgpu_buffer_rsrc_t rsrc_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(frames) + row * (UG::FH * UG::FW * UG::FC), 0,
                                                                                 any ? UG::FH * UG::FW * UG::FC : 0, kURsrcWord3);
            pre[it] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_f, u < UG::F_UNITS ? (unsigned)u * 16u : kUOob, 0, MI355_AUX_WGRAD_LD));
        } else {
            const int v = (it - UG::NIF) * UG::THREADS + tid;
            pre[it] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_d, (any && v < UG::D_UNITS) ? (unsigned)img * (unsigned)(UG::D_UNITS * 16) + (unsigned)v * 16u : kUOob, 0, MI355_AUX_WGRAD_LD));
        }
    };
    auto fill = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            if (it < UG::NIF) {                           // 4 pixels x 4 bytes -> 4 slots of 4 x 16 bits (zero-extended)
                const int u = it * UG::THREADS + tid, row = u / (UG::FW / 4), x4 = u - row * (UG::FW / 4);
                if (u < UG::F_UNITS) {
                    s_u32x4 lo4, hi4;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const unsigned w0 = pre[it][2 * p], w1 = pre[it][2 * p + 1];
                        const s_u32x4 e = {(w0 & 0xffu) | ((w0 & 0xff00u) << 8), ((w0 >> 16) & 0xffu) | ((w0 >> 24) << 16),
                                           (w1 & 0xffu) | ((w1 & 0xff00u) << 8), ((w1 >> 16) & 0xffu) | ((w1 >> 24) << 16)};
                        if (p == 0) lo4 = e; else hi4 = e;
                    }
                    unsigned char* const d = lfr + row * UG::FROW + 4 * x4 * UG::SLOT;
                    *reinterpret_cast<s_u32x4*>(d) = lo4;
                    *reinterpret_cast<s_u32x4*>(d + 16) = hi4;
                }
            } else {
                const int v = (it - UG::NIF) * UG::THREADS + tid;
                unsigned hi[2], lo[2];
                f16_split4(pre[it], sd, hi, lo);
                if (v < UG::D_UNITS) {
                    unsigned char* const d = ldz + (v >> 3) * PIXD + (v & 7) * 8;
                    *reinterpret_cast<uint2*>(d) = make_uint2(hi[0], hi[1]);
                    *reinterpret_cast<uint2*>(d + UG::LOD) = make_uint2(lo[0], lo[1]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) db4[c] += __uint_as_float(pre[it][c]);      // (units past the end loaded zeros)
            }
        }
    };

    u_f32x16 acc[UG::TPW];
#pragma unroll
    for (int t = 0; t < UG::TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    typedef u_s16x4 __attribute__((address_space(3))) * lds_v4;
    auto tr2 = [&](const unsigned char* p0, const unsigned char* p1) __attribute__((always_inline)) -> s_u32x4 {
        const u_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0));
        const u_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p1));
        const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
        return (s_u32x4){ua.x, ua.y, ub.x, ub.y};
    };

    int img = blockIdx.x;
#pragma unroll
    for (int it = 0; it < NI; ++it) prefetch(img, it);
    for (; img < images; img += gridDim.x) {
        __syncthreads();
        fill();
        __syncthreads();
        s_u32x4 fr[2][2 + UG::TPW];                       // [parity]: dz hi, dz lo, the frame fragment of each of the wave's tap rows
        auto load_step = [&](int par, int s) __attribute__((always_inline)) {
            const int y0 = 4 * (s / (UG::OW / 4)), x0 = 4 * (s % (UG::OW / 4));
            const unsigned char* const d = dz_lane + (y0 * UG::OW + x0) * PIXD;
            const unsigned char* const f = fr_lane + 4 * y0 * UG::FROW + 4 * x0 * UG::SLOT;
            fr[par][0] = tr2(d, d + PIXD);
            fr[par][1] = tr2(d + UG::LOD, d + PIXD + UG::LOD);
#pragma unroll
            for (int t = 0; t < UG::TPW; ++t) fr[par][2 + t] = tr2(f + t * UG::FROW, f + t * UG::FROW + 4 * UG::SLOT);
        };
        load_step(0, 0);
#pragma unroll
        for (int s = 0; s < UG::KSTEPS; ++s) {
            if (s >= 1 && s <= NI) prefetch(img + gridDim.x, s - 1);          // the next image, one load per k-step
            if (s + 1 < UG::KSTEPS) load_step((s + 1) & 1, s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < UG::TPW; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, fr[s & 1][0]), __builtin_bit_cast(s_f16x8, fr[s & 1][2 + t]), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, fr[s & 1][1]), __builtin_bit_cast(s_f16x8, fr[s & 1][2 + t]), acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- partial dW1: accumulator e = row co = (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column ty * 32 + lane % 32 = (ty, tx, ci)
    float* const pw = part_w + (size_t)blockIdx.x * (UG::CO * 256);
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < UG::TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) pw[((e & 3) + 8 * (e >> 2) + 4 * lh) * 256 + (wave * UG::TPW + t) * 32 + li] = acc[t][e] * un;
    // ---- partial db1: the threads' sums, channel quad tid % 8, folded in the order of the thread index
    __syncthreads();
    float* const red = reinterpret_cast<float*>(lds);     // [THREADS / 8][32]
#pragma unroll
    for (int c = 0; c < 4; ++c) red[(tid >> 3) * 32 + 4 * (tid & 7) + c] = db4[c];
    __syncthreads();
    if (tid < UG::CO) {
        float sum = 0.0f;
        for (int k = 0; k < UG::THREADS / 8; ++k) sum += red[k * 32 + tid];
        part_b[(size_t)blockIdx.x * UG::CO + tid] = sum;
    }
}

// MI355PPO_CONV_U: unset / "1": the weight gradients of all three layers on kernel U; "0": none (kernels V / P: A/B runs); a list of layer digits -- "3",
// "23", "13" -- those layers only.  Read at every call.
bool convu_on(int layer) {
    const char* e = getenv("MI355PPO_CONV_U");
    if (!e || !e[0] || (e[0] == '1' && !e[1])) return true;
    if (e[0] == '0' && !e[1]) return false;
    for (; *e; ++e)
        if (*e == '0' + layer) return true;
    return false;
}

int convu_max_parts() { return 256; }

// Which f16x2 weight-gradient launches kernel U takes -- the ONE place that decides (the launchers below and the host query
// mi355ppo_cnn_conv_wgrad_kernel_f16x2 ask it): layers 2 / 3 while the source tensor stays inside the 32-bit buffer range, layer 1 while dz does.
bool convu_takes(int64_t images, int layer) {
    if (images <= 0) return false;
    if (layer == 1) return convu_on(1) && (long long)images * 20 * 20 * 32 * 4 < (1LL << 32) - 8192;
    if ((layer != 2 && layer != 3) || !convu_on(layer)) return false;
    return (long long)images * (layer == 3 ? 9 * 9 * 64 : 20 * 20 * 32) * 4 < (1LL << 32) - 8192;
}

template <class UG>
static int convu_launch_t(const float* src, const float* dz, float* part_w, float* part_b, int64_t images, int* nparts, hipStream_t s,
                          const unsigned* dz_amax, const unsigned* src_amax) {
    const long long srcb = (long long)images * UG::SRC_REC * UG::C * 4, dzb = (long long)images * UG::OP * UG::CO * 4;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        cus = n < convu_max_parts() ? n : convu_max_parts();
    }
    const int groups = (int)((images + UG::G - 1) / UG::G);
    const int grid = groups < cus ? groups : cus;
    hipLaunchKernelGGL((convu_kernel<UG>), dim3((unsigned)grid), dim3(UG::THREADS), 0, s, src, dz, part_w, part_b, (long long)images, groups,
                       (unsigned)srcb, (unsigned)dzb, dz_amax, src_amax);
    *nparts = grid;
    return 0;
}

// -> 0 launched (nparts partials written), 1 not applicable
int convu_launch(const float* src, const float* dz, float* part_w, float* part_b, int64_t images, int layer, int* nparts, hipStream_t s,
                 const unsigned* dz_amax, const unsigned* src_amax) {
    if (!dz_amax || !src_amax || !convu_takes(images, layer)) return 1;
    return layer == 3 ? convu_launch_t<UGeom3>(src, dz, part_w, part_b, images, nparts, s, dz_amax, src_amax)
                      : convu_launch_t<UGeom2>(src, dz, part_w, part_b, images, nparts, s, dz_amax, src_amax);
}

// Layer 1 (MI355PPO_CONV_U=23: kernel P).  -> 0 launched (nparts partials; the reduce applies 1 / 255 only), 1 not applicable
int convu1_launch(const unsigned char* frames, const int64_t* inds, const float* dz, float* part_w, float* part_b, int64_t images, int* nparts,
                  hipStream_t s, const unsigned* dz_amax) {
    if (!dz_amax || !convu_takes(images, 1)) return 1;
    const long long dzb = (long long)images * 20 * 20 * 32 * 4;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        cus = n < convu_max_parts() ? n : convu_max_parts();
    }
    const int grid = images < cus ? (int)images : cus;
    hipLaunchKernelGGL((convu1_kernel<UGeom1<8>>), dim3((unsigned)grid), dim3(512), 0, s, frames, reinterpret_cast<const long long*>(inds), dz, part_w, part_b,
                       (int)images, (unsigned)dzb, dz_amax);
    *nparts = grid;
    return 0;
}

}  // namespace mi355ppo
