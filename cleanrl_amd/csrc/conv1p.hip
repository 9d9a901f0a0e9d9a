// Kernel P -- weight + bias gradient of the NatureCNN's first convolution on the bf16 matrix pipe, exact products (gfx950).
//
// dW1[n][k] = sum over images and output pixels p of dz1[p][n] * frame[p, k]   (cleanrl/ppo_atari_multigpu.py:136, the
// backward of Conv2d(4, 32, 8, stride=4) on uint8 frames; the /255 is applied once to the reduced sum).  The frame bytes
// are integers 0..255 -- EXACT in bf16 -- and an f32 gradient splits into three bf16 terms that sum to it exactly
// (hi = top 8 significand bits, mid = the next 8, lo = the last 8: x = hi + mid + lo, every term a bf16).  So
//     dz * v = hi * v + mid * v + lo * v
// with every product exact in f32 (8 x 8 significand bits), accumulated in f32 by `v_mfma_f32_32x32x16_bf16`: the same
// arithmetic class as kernel R's f32 MFMA (exact products, f32 accumulation), at 3 x 32 cycles per 16 pixels instead of
// 8 x 64 -- 5.3x less matrix-pipe time, which turns the launch from pipe-bound (1.6 ms at 32,768 images) to HBM-bound.
//
// Roles in one MFMA (32 x 32 x 16): rows = the 32 output channels, columns = the 32 taps (kw, c) of ONE tap row r, the
// 16 reduction slots = two groups of 8 horizontally adjacent output pixels (slots 0-7 from the lower half-wave's group,
// 8-15 from the upper one's).  8 tap rows x 3 terms = 24 MFMAs per 16 pixels; the whole dW (8 tiles x 16 VGPRs) stays in
// the wave's accumulators for the life of the kernel.
//
// A wave is autonomous (kernel R's scheme): wave w owns output rows 5w .. 5w+4 of every image of its workgroup, i.e. source
// rows 20w .. 20w+23.  Its 15 pixel groups (3 per output row: pixels 0-7, 8-15, 16-19 + 4 padding slots whose dz is
// forced to zero) make 8 steps.
//   * dz operand: a lane (channel li, half lh) loads its group's 8 pixels with 8 coalesced dword loads (128 bytes per
//     pixel and half-wave), one step ahead, and splits them in registers (4 VALU + 1.5 v_perm per value);
//   * frame operand: a lane (tap column kw = li/4, channel c = li%4) needs the bytes of 8 CONSECUTIVE output pixels of one
//     tap -- 16 bytes apart in the pixel-interleaved frame.  The wave therefore stages its slab TRANSPOSED in its own LDS
//     region: line (source row R, t = x mod 4, c) holds the bytes of x = 4q + t, q = 0..23, so that pixels ox0..ox0+7
//     of tap kw are 8 consecutive bytes starting at q = ox0 + kw/4.  One ds_read_b64 + one ds_read_b32 + two
//     v_alignbyte per tap row; 8 v_cvt_f32_ubyte + 4 v_perm make the bf16 operand.  The transposition itself happens in
//     registers (4x4 byte transposes, 8 v_perm per four dwords) between the 16-byte global loads of the NEXT image's slab
//     (issued at the start of an image) and the ds_write_b32s (after the last step's reads: LDS instructions of one wave
//     execute in order, so one buffer suffices and there is no workgroup barrier anywhere).
// Round 4 (ZEXT): the frame operand enters the pipe by ZERO-EXTENSION.  The 16-bit pattern 0x00vv read as a bf16 is v * 2^-133 for every
// byte v (subnormal below 128, exponent field 1 from 128 on -- the subnormals continue the normal range), and gfx950's bf16 MFMA multiplies
// subnormal inputs exactly (tools/mfma_denorm.cpp, profiles/r04_mfma_denorm.json; MI200's flushed them).  One v_perm with a per-lane selector
// builds two operand elements straight from the LDS words -- 4 VALU per tap row instead of 2 v_alignbyte + 8 v_cvt_f32_ubyte + 4 v_perm, 26 %
// fewer VALU instructions in the kernel -- and dz is multiplied by 2^80 before its split (exact for |dz| < 2^48), so the accumulators hold
// 2^-53 x the sums and stay in the normal range (sums of magnitude above 2^-73 = 1e-22; tests/test_gpu_cnn.py runs gradients of scale 1e-15
// and 1e+10); the reduction multiplies by 2^53 / 255 instead of 1 / 255.
// Every product is still exact and every sum an f32 accumulation; what changes is the pipe's internal alignment of an instruction's 16
// products (all frame values now carry ONE exponent), so 6 % of dW1's elements differ from the converted-operand route in the last bit or
// two (<= 8.8e-8 of the scale), with the same error against float64 (max 4.47e-7 both, mean 6.37e-8 vs 6.34e-8 of the scale; torch's own
// f32 weight gradient: 1.9e-7; profiles/r04_kernel_p_zext_error.jsonl); db1 is bit-identical.  923 -> 833 us at 32,768 images, 310 -> 284 at
// 8,192, 201 -> 188 at 4,096 (profiles/r04_kernel_p_zext_ab.txt).  MI355PPO_P_ZEXT=0: the conversion route (A/B runs).
// Round 5 (F16): on the f16 pipe with dz in TWO f16 terms under its tensor's power-of-two scale (the amax record conv2's data gradient
// fills; csrc/f16split.h) -- 16 instead of 24 matrix instructions per step, 24 instead of 44 split instructions per 8 values.  The frame
// operand is the same zero-extended pattern: 0x00vv read as an f16 is v * 2^-24 (a subnormal for every byte; the f16 MFMA multiplies
// subnormals exactly: tools/mfma_denorm.cpp, profiles/r05_mfma_denorm.json).  The accumulators hold s * 2^-24 x the sums; the wave removes
// the factor (a power of two) when it stores its partial, so the reduction's factor is 1 / 255 alone.
// Partials: one (32 x 256) matrix + 32 bias sums per wave, layout and fixed-order two-stage reduction of kernel R
// (conv_wgrad_reduce1/2 in conv.hip) -- deterministic.
#include "common.h"
#include "f16split.h"

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float p_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int p_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 p_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kPPitch = 336, kPImg = 84 * 336;            // bytes per source row / frame
constexpr int kPRowsPerWave = 5, kPSlabRows = 24;         // output rows per wave; source rows they touch
constexpr int kPLine = 32;                                // bytes per (R, t, c) line: q = 0..23 used (+ slack for the shifted read)
constexpr int kPRowLds = 16 * kPLine;                     // 512 bytes of LDS per source row
constexpr int kPSlabLds = kPSlabRows * kPRowLds;          // 12,288 bytes per wave
constexpr int kPSteps = 8, kPGroups = 15;
constexpr int kPPieceQuads = 8 * 5;                       // staging piece = 8 source rows: 40 four-chunk quads (q = 0..19), one per lane
                                                          // (+ the 8 chunks q = 20, one each for lanes 0..7)

// group g of a wave's 5 output rows -> (output row, first pixel, valid pixels); g = 15 is the empty 16th slot
__host__ __device__ constexpr int p_row(int g) { return g < kPGroups ? g / 3 : kPRowsPerWave - 1; }
__host__ __device__ constexpr int p_ox0(int g) { return g < kPGroups ? 8 * (g % 3) : 16; }
__host__ __device__ constexpr int p_nv(int g) { return g < kPGroups ? ((g % 3) < 2 ? 8 : 4) : 0; }

__device__ __forceinline__ unsigned p_perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// two f32 whose low 16 bits are irrelevant/zero -> packed bf16 pair (element 0 in the low half)
__device__ __forceinline__ unsigned p_pack_hi16(float e1, float e0) { return p_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u); }

constexpr float kPZextDzScale = 0x1p80f, kPZextOutScale = 0x1p53f;      // 2^80 * 2^-133 = 2^-53

// MODE 0: bf16, converted frame operand; 1: bf16, zero-extended (ZEXT); 2: f16 split of dz, zero-extended frame operand (F16)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1p_wgrad_kernel(
    const unsigned char* __restrict__ src, const int64_t* __restrict__ inds, const float* __restrict__ dz,
    float* __restrict__ part_w,      // [grid * 4][32][256]
    float* __restrict__ part_b,      // [grid * 4][32]
    int images, const unsigned* __restrict__ dz_amax) {
    constexpr bool ZEXT = MODE >= 1, F16 = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    float sdz = 1.0f, un = 1.0f;           // F16: dz's scale; the factor that takes s * 2^-24 off the partial
    if constexpr (F16) {
        const int ez = f16_scale_exp(amax_load(dz_amax, lane));
        sdz = f16_pow2(ez);
        un = f16_pow2(24 - ez);            // ez in [-100, 60]: 2^-36 .. 2^124
    }
    unsigned char* const tt = p_smem + wave * kPSlabLds;
    const int kw = li >> 2, c = li & 3;
    const unsigned sh = (unsigned)(kw >> 2);                              // tap columns 4..7 read one q further
    // ZEXT: v_perm selectors (0x0c = the constant byte 0): elements (sh, sh + 1) and (sh + 2, sh + 3) of an 8-byte window, zero-extended
    const unsigned sel0 = 0x0c000c00u | ((sh + 1u) << 16) | sh, sel1 = 0x0c000c00u | ((sh + 3u) << 16) | (sh + 2u);
    const int lds_lane = ((kw & 3) * 4 + c) * kPLine;                     // this lane's line within a source row
    const int dz_lane = (wave * (kPRowsPerWave * 20)) * 32 + li;          // float offset of (wave's first pixel, channel li)
    const int step_img = gridDim.x;

    p_f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    float bsum = 0.0f;

    auto row_of = [&](int img) { return inds ? inds[img] : (long long)img; };     // scalar load
    auto slab = [&](long long simg) { return src + simg * (long long)kPImg + wave * (4 * kPRowsPerWave * kPPitch); };

    // ---- staging: global (16-byte chunks = 4 pixels) -> registers -> 4x4 byte transposes -> LDS lines ----
    // In three PIECES of 8 source rows (40 four-chunk quads + the 8 chunks q = 20: one quad and at most one chunk per lane,
    // 20 registers in flight instead of the whole slab's 36 -- the kernel spilled 32 VGPRs at its two-waves-per-SIMD budget).
    // Piece P of the NEXT image replaces rows 8P .. 8P+7 as soon as the current image's last reader of those rows has issued
    // (output row r reads source rows 4r .. 4r+7; LDS instructions of one wave execute in order): piece 0 after step 2
    // (groups 0-5 = output rows 0, 1), piece 1 after step 5 (rows 2, 3), piece 2 after step 7.
    p_u32x4 st[4], st20;
    auto stage_load = [&](auto pc, const unsigned char* g0) {
        constexpr int P = decltype(pc)::value;
        const int id = lane < kPPieceQuads ? lane : kPPieceQuads - 1;
        const int R = 8 * P + id / 5, Q = id - 5 * (id / 5);
#pragma unroll
        for (int k = 0; k < 4; ++k) st[k] = *reinterpret_cast<const p_u32x4*>(g0 + R * kPPitch + (4 * Q + k) * 16);
        st20 = *reinterpret_cast<const p_u32x4*>(g0 + (8 * P + (lane < 8 ? lane : 7)) * kPPitch + 20 * 16);
    };
    auto stage_store = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if (lane < kPPieceQuads) {
            const int R = 8 * P + lane / 5, Q = lane - 5 * (lane / 5);
            unsigned char* const base = tt + R * kPRowLds + 4 * Q;
#pragma unroll
            for (int t = 0; t < 4; ++t) {                                  // pixel 4q + t of chunks q = 4Q .. 4Q+3
                const unsigned d0 = st[0][t], d1 = st[1][t], d2 = st[2][t], d3 = st[3][t];
                const unsigned lo01 = p_perm(d1, d0, 0x05010400u), hi01 = p_perm(d1, d0, 0x07030602u);
                const unsigned lo23 = p_perm(d3, d2, 0x05010400u), hi23 = p_perm(d3, d2, 0x07030602u);
                *reinterpret_cast<unsigned*>(base + (t * 4 + 0) * kPLine) = p_perm(lo23, lo01, 0x05040100u);
                *reinterpret_cast<unsigned*>(base + (t * 4 + 1) * kPLine) = p_perm(lo23, lo01, 0x07060302u);
                *reinterpret_cast<unsigned*>(base + (t * 4 + 2) * kPLine) = p_perm(hi23, hi01, 0x05040100u);
                *reinterpret_cast<unsigned*>(base + (t * 4 + 3) * kPLine) = p_perm(hi23, hi01, 0x07060302u);
            }
        }
        if (lane < 8) {                                                    // q = 20 (x = 80..83): bytes 21..23 of the dword are padding
            unsigned char* const base = tt + (8 * P + lane) * kPRowLds + 20;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    *reinterpret_cast<unsigned*>(base + (t * 4 + ch) * kPLine) = (st20[t] >> (8 * ch)) & 0xffu;
        }
    };

    // dz of step s (compile-time) of the image at gd: 8 consecutive pixels of the row of this lane's group, all loads at
    // (one per-lane base) + immediates.  A full group (pixels 0-7 / 8-15) loads its own pixels; the short group (16-19)
    // loads pixels 12-19 and uses the upper four (slot j <- loaded[j + 4], slots 4-7 <- 0: see dz_slot); the empty 16th
    // slot loads the same window and uses nothing.  Nothing is read outside the image.
    float ring[8];           // (a second ring -- dz two steps ahead -- round 3: 877 -> 901 us, 7 spilled VGPRs, profiles/r03_kernel_p_ring2_ab.jsonl;
                             //  round 4, zero-extended operand: 812 -> 857 us, 13 spilled, profiles/r04_kernel_p_ahead_ab.txt;
                             //  dz global -> LDS directly (global_load_lds in inline asm, hand-counted vmcnt, 2 / 3 steps ahead in a ring): 800 -> 840 us
                             //  and the hand-counted waits were not sufficient (results differed from run to run): profiles/r04_kernel_p_dz_lds_direct_ab.txt)
    int lhx = lh;            // re-materialised per image (see the loop): keeps the per-step selects from being hoisted into 30 live VGPRs
    auto dz_fetch = [&](const float* gd, auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int g0 = 2 * s, g1 = 2 * s + 1;
        constexpr int b0 = (p_row(g0) * 20 + (p_nv(g0) == 8 ? p_ox0(g0) : 12)) * 32, b1 = (p_row(g1) * 20 + (p_nv(g1) == 8 ? p_ox0(g1) : 12)) * 32;
        const float* const p = gd + (lhx ? b1 : b0);
#pragma unroll
        for (int j = 0; j < 8; ++j) ring[j] = p[j * 32];
    };
    // value of reduction slot j of step s for this lane, from the ring loaded by dz_fetch(s)
    auto dz_slot = [&](auto sc, auto jc) -> float {
        constexpr int s = decltype(sc)::value, j = decltype(jc)::value;
        constexpr int n0 = p_nv(2 * s), n1 = p_nv(2 * s + 1);
        const float v0 = n0 == 8 ? ring[j] : (n0 == 4 && j < 4) ? ring[j + 4] : 0.0f;
        const float v1 = n1 == 8 ? ring[j] : (n1 == 4 && j < 4) ? ring[j + 4] : 0.0f;
        if constexpr (n0 == 8 && n1 == 8) return ring[j];
        else return lhx ? v1 : v0;
    };

    int img = blockIdx.x;
    if (img >= images) return;                                           // (grid <= images: never taken)
    {
        const unsigned char* const g0 = slab(row_of(img));
        stage_load(std::integral_constant<int, 0>{}, g0); stage_store(std::integral_constant<int, 0>{});
        stage_load(std::integral_constant<int, 1>{}, g0); stage_store(std::integral_constant<int, 1>{});
        stage_load(std::integral_constant<int, 2>{}, g0); stage_store(std::integral_constant<int, 2>{});
    }
    dz_fetch(dz + (long long)img * (400 * 32) + dz_lane, std::integral_constant<int, 0>{});
    int nimg = img + step_img < images ? img + step_img : img;           // clamped: past the end the prefetches are never consumed
    long long snext = row_of(nimg);

    for (; img < images; img += step_img) {
        const float* const gd = dz + (long long)img * (400 * 32) + dz_lane;
        const float* const gdn = dz + (long long)nimg * (400 * 32) + dz_lane;
        asm volatile("" : "+v"(lhx));
        const unsigned char* const gnext = slab(snext);                  // the next image's slab, staged piece by piece during the 8 steps
        stage_load(std::integral_constant<int, 0>{}, gnext);
        nimg = nimg + step_img < images ? nimg + step_img : nimg;
        snext = row_of(nimg);

        auto step = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int g0 = 2 * s, g1 = 2 * s + 1;
            // ---- dz operand: three exact bf16 terms of the 8 prefetched values; bias sum
            unsigned a_hi[4], a_mid[4], a_lo[4];
            [&]<int... J>(std::integer_sequence<int, J...>) {
                ([&] {
                    constexpr int j = 2 * J;
                    float x0 = dz_slot(sc, std::integral_constant<int, j>{}), x1 = dz_slot(sc, std::integral_constant<int, j + 1>{});
                    bsum += x0;
                    bsum += x1;
                    if constexpr (F16) {                                          // two f16 terms of s dz (f16split.h's steps for one pair)
                        const s_f32x2 v = s_f32x2{x0, x1} * sdz;
                        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, s_f16x2));
                        const s_f32x2 r = {f16_resid_lo(v[0], h), f16_resid_hi(v[1], h)};
                        a_hi[J] = h;
                        a_lo[J] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, s_f16x2));
                        a_mid[J] = 0u;
                        return;
                    }
                    if constexpr (ZEXT) {                                         // exact (a power of two; |dz| < 2^48)
                        x0 *= kPZextDzScale;
                        x1 *= kPZextDzScale;
                    }
                    const float h0 = __uint_as_float(__float_as_uint(x0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
                    const float r0 = x0 - h0, r1 = x1 - h1;                       // exact
                    const float m0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), m1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                    const float l0 = r0 - m0, l1 = r1 - m1;                       // exact, <= 8 significand bits
                    a_hi[J] = p_pack_hi16(x1, x0);
                    a_mid[J] = p_pack_hi16(r1, r0);
                    a_lo[J] = p_pack_hi16(l1, l0);
                }(), ...);
            }(std::make_integer_sequence<int, 4>{});
            const p_u32x4 ah = {a_hi[0], a_hi[1], a_hi[2], a_hi[3]}, am = {a_mid[0], a_mid[1], a_mid[2], a_mid[3]},
                          al = {a_lo[0], a_lo[1], a_lo[2], a_lo[3]};
            // prefetch the next step's dz (the next image's first step at the end)
            if constexpr (s + 1 < kPSteps) dz_fetch(gd, std::integral_constant<int, s + 1>{});
            else dz_fetch(gdn, std::integral_constant<int, 0>{});
            // ---- frame operand per tap row + 3 MFMAs
            constexpr int o0 = 4 * p_row(g0) * kPRowLds + p_ox0(g0), o1 = 4 * p_row(g1) * kPRowLds + p_ox0(g1);
            const unsigned char* const lp = tt + lds_lane + (lhx ? o1 : o0);
            // tap rows in PAIRS: the six MFMAs of a pair alternate between two accumulators (a dependent MFMA chain would wait
            // out the pipe's latency three times per row), and the LDS words of the next pair are requested first
            auto lds_row = [&](int r, unsigned (&w)[3]) {
                const uint2 w01 = *reinterpret_cast<const uint2*>(lp + r * kPRowLds);
                w[0] = w01.x;
                w[1] = w01.y;
                w[2] = *reinterpret_cast<const unsigned*>(lp + r * kPRowLds + 8);
            };
            auto operand = [&](const unsigned (&w)[3]) -> p_bf16x8 {
                if constexpr (ZEXT) {
                    const p_u32x4 bz = {p_perm(w[1], w[0], sel0), p_perm(w[1], w[0], sel1), p_perm(w[2], w[1], sel0), p_perm(w[2], w[1], sel1)};
                    return __builtin_bit_cast(p_bf16x8, bz);
                }
                const unsigned lo = __builtin_amdgcn_alignbyte(w[1], w[0], sh), hi = __builtin_amdgcn_alignbyte(w[2], w[1], sh);
                const float f0 = (float)(lo & 0xffu), f1 = (float)((lo >> 8) & 0xffu), f2 = (float)((lo >> 16) & 0xffu), f3 = (float)(lo >> 24);
                const float f4 = (float)(hi & 0xffu), f5 = (float)((hi >> 8) & 0xffu), f6 = (float)((hi >> 16) & 0xffu), f7 = (float)(hi >> 24);
                const p_u32x4 bv = {p_pack_hi16(f1, f0), p_pack_hi16(f3, f2), p_pack_hi16(f5, f4), p_pack_hi16(f7, f6)};
                return __builtin_bit_cast(p_bf16x8, bv);
            };
            const p_bf16x8 Ah = __builtin_bit_cast(p_bf16x8, ah), Am = __builtin_bit_cast(p_bf16x8, am), Al = __builtin_bit_cast(p_bf16x8, al);
            unsigned raw[2][2][3];
            lds_row(0, raw[0][0]);
            lds_row(1, raw[0][1]);
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                if (rp + 1 < 4) {
                    lds_row(2 * rp + 2, raw[(rp + 1) & 1][0]);
                    lds_row(2 * rp + 3, raw[(rp + 1) & 1][1]);
                }
                const p_bf16x8 b0 = operand(raw[rp & 1][0]), b1 = operand(raw[rp & 1][1]);
                if constexpr (F16) {
                    const s_f16x8 h0 = __builtin_bit_cast(s_f16x8, b0), h1 = __builtin_bit_cast(s_f16x8, b1);
                    acc[2 * rp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, ah), h0, acc[2 * rp], 0, 0, 0);
                    acc[2 * rp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, ah), h1, acc[2 * rp + 1], 0, 0, 0);
                    acc[2 * rp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, al), h0, acc[2 * rp], 0, 0, 0);
                    acc[2 * rp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, al), h1, acc[2 * rp + 1], 0, 0, 0);
                } else {
                acc[2 * rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, b0, acc[2 * rp], 0, 0, 0);
                acc[2 * rp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, b1, acc[2 * rp + 1], 0, 0, 0);
                acc[2 * rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, b0, acc[2 * rp], 0, 0, 0);
                acc[2 * rp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, b1, acc[2 * rp + 1], 0, 0, 0);
                acc[2 * rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, b0, acc[2 * rp], 0, 0, 0);
                acc[2 * rp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, b1, acc[2 * rp + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);                        // one pair's conversions at a time (register pressure)
            }
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        stage_store(std::integral_constant<int, 0>{});                   // rows 0-7: output rows 0, 1 have issued their reads
        stage_load(std::integral_constant<int, 1>{}, gnext);
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        stage_store(std::integral_constant<int, 1>{});                   // rows 8-15: output rows 2, 3 done
        stage_load(std::integral_constant<int, 2>{}, gnext);
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
        stage_store(std::integral_constant<int, 2>{});                   // rows 16-23: after the last step's reads
    }

    float* pw = part_w + (size_t)(blockIdx.x * 4 + wave) * 32 * 256;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) pw[(size_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * 256 + r * 32 + li] = F16 ? acc[r][e] * un : acc[r][e];
    const float both = bsum + __shfl_xor(bsum, 32, 64);
    if (lh == 0) part_b[(size_t)(blockIdx.x * 4 + wave) * 32 + li] = both;
}

static bool conv1p_zext() { return true; }      // (the conversion route behind MI355PPO_P_ZEXT=0 was an A/B switch of round 4: profiles/r04_*)

// Launches kernel P; *partial_scale = the factor (beside 1 / 255) the reduction applies to its partial sums: 2^53 with the zero-extended
// bf16 frame operand, 1 otherwise.  dz_amax (dz's amax record): the f16 variant.
int conv1p_launch(const unsigned char* src, const int64_t* inds, const float* dz, float* part_w, float* part_b, int images, int grid,
                  hipStream_t s, const unsigned* dz_amax, float* partial_scale) {
    // (one wave per SIMD -- 372 registers, no spill at all -- measured 1,135 us against 868 at 32,768 images: profiles/r03_kernel_p_pieces_ab.jsonl)
    const int mode = dz_amax ? 2 : conv1p_zext() ? 1 : 0;
    *partial_scale = mode == 1 ? kPZextOutScale : 1.0f;
    const size_t sm = 4 * (size_t)kPSlabLds;
    static bool attr_done = false;           // 48 KiB: within the default dynamic-LDS limit, but set it explicitly once
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1p_wgrad_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv1p_wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv1p_wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
            (void)hipGetLastError();
            set_error("conv1p_launch: hipFuncSetAttribute(%zu bytes of LDS) failed", sm);
            return MI355PPO_EHIP;
        }
        attr_done = true;
    }
    if (mode == 2) hipLaunchKernelGGL(conv1p_wgrad_kernel<2>, dim3(grid), dim3(256), sm, s, src, inds, dz, part_w, part_b, images, dz_amax);
    else if (mode == 1) hipLaunchKernelGGL(conv1p_wgrad_kernel<1>, dim3(grid), dim3(256), sm, s, src, inds, dz, part_w, part_b, images, dz_amax);
    else hipLaunchKernelGGL(conv1p_wgrad_kernel<0>, dim3(grid), dim3(256), sm, s, src, inds, dz, part_w, part_b, images, dz_amax);
    return check_launch("conv1p_wgrad_kernel");
}

}  // namespace mi355ppo
