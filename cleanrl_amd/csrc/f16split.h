// Two-term f16 split of f32 operands for the matrix-pipe kernels (Z: gemmz.hip, W: fcw.hip, V: convw.hip), round 5:
//     s x = hi + lo + r,   hi = f16(s x) (round to nearest),   lo = f16(s x - hi),   |r| <= 2^-23 |s x|
// with s a power of two per TENSOR that puts the tensor's largest magnitude in [2^14, 2^15) (f16: 11 significant bits, largest
// finite value 65,504, normal numbers down to 2^-14 -- without the scale a gradient tensor's `lo` terms are f16 subnormals and the
// error is 100 x larger: tools/err_f16x2.py, profiles/r05_err_f16x2.jsonl).  Products of terms have 22 significant bits: exact in
// f32, so `v_mfma_f32_32x32x16_f16` on the term pairs hi hi, hi lo, lo hi computes the f32 GEMM with exact products and f32
// accumulation -- THREE matrix instructions per f32 product where the three-term bf16 split (bf16split.h) needs six, and 10 VALU
// instructions per four elements instead of 22.  The dropped lo lo pair and the split residue r are each <= 2^-22 of a product;
// measured against float64 on the full-size operands the result is at or below the six-pair bf16 split's error (fewer f32
// accumulator roundings per k-step) and at the f32 library GEMM's.
//
// Where the scale comes from: every tensor that feeds a split carries an "amax record" -- kAmaxSlots uint32 slots 64 bytes apart
// holding the bit pattern of max |x| (non-negative floats order like their bit patterns).  The kernel that PRODUCES the tensor
// folds its epilogue values into the record (one atomic max per wave, spread over the slots by workgroup index; max is
// order-independent, so the result is deterministic); the consumer reads the 16 slots and derives s.  The caller zeroes the ACTIVATION /
// GRADIENT records before the producers run (one fill kernel per forward / backward pass -- not a memset node: DESIGN 3.3); the WEIGHT tensors'
// records are not accumulated at all: zabsmax_store_kernel (gemmz.hip) STORES every slot, 16 blocks per tensor, no zeroing, no atomics.  A record that UNDERSTATES the tensor's maximum
// makes the f16 conversion overflow to infinity -- loud, never silently wrong.
#pragma once
#include "common.h"
#include "bf16split.h"

namespace mi355ppo {

typedef _Float16 s_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s_f16x2 __attribute__((ext_vector_type(2)));
typedef float s_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kAmaxSlots = MI355PPO_AMAX_WORDS / 16, kAmaxStride = 16;      // 16 slots, one per 64 bytes
constexpr int kF16PackHeader = 64;                                         // bytes in front of an f16x2 pack: word 0 = bits of max |B|

// wave-uniform maximum of a per-lane unsigned without LDS permutes: four DPP steps inside each row of 16 lanes (quad swaps, half-row
// mirror, row mirror -- every lane reads a live lane), then the four row results through readlane.  All 64 lanes must be active.
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); v = v > t ? v : t;       // quad_perm [1,0,3,2]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); v = v > t ? v : t;       // quad_perm [2,3,0,1]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); v = v > t ? v : t;      // row_half_mirror
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); v = v > t ? v : t;      // row_mirror
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}

// the record's value (bits of max |x|), wave-uniform; all 64 lanes active
__device__ __forceinline__ unsigned amax_load(const unsigned* __restrict__ rec, int lane) {
    return wave_umax(rec[(lane & (kAmaxSlots - 1)) * kAmaxStride]);
}

// fold this wave's per-lane maxima (bit patterns of non-negative floats) into the record: one atomic per wave
__device__ __forceinline__ void amax_commit(unsigned* __restrict__ rec, unsigned lane_max, unsigned spread, int lane) {
    const unsigned m = wave_umax(lane_max);
    if (lane == 0) atomicMax(rec + (spread & (kAmaxSlots - 1)) * kAmaxStride, m);
}

// e with 2^e * amax in [2^14, 2^15) (amax = 0 or tiny: 2^60; huge: 2^-100) -- MI355_HD: the pack kernels and the host tests share it
MI355_HD int f16_scale_exp(unsigned amax_bits) {
    const int e = 141 - (int)((amax_bits >> 23) & 0xffu);
    return e > 60 ? 60 : (e < -100 ? -100 : e);
}
MI355_HD float f16_pow2(int e) {      // 2^e, -126 <= e <= 127
    union { unsigned u; float f; } c;
    c.u = (unsigned)(127 + e) << 23;
    return c.f;
}
// the factor that undoes both operands' scales in the epilogue: 2^-(ea + eb), ea + eb in [-200, 120]
MI355_HD float f16_unscale(int ea, int eb) {
    const int u = -(ea + eb);
    return f16_pow2(u > 127 ? 127 : u);
}

// x - (float)(lower / upper half of h as f16): ONE v_fma_mix_f32 (the f16 operand is widened inside the instruction, times -1.0 is
// exact, one rounding of an exactly representable result)
__device__ __forceinline__ float f16_resid_lo(float x, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}
__device__ __forceinline__ float f16_resid_hi(float x, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}
// four f32 (as loaded) -> hi[2], lo[2] (2 x f16 each): 4 v_mul_f32, 2 v_cvt_pk_f16_f32, 4 v_fma_mix_f32, 2 v_cvt_pk_f16_f32
__device__ __forceinline__ void f16_split4(const s_u32x4 x, float s, unsigned (&hi)[2], unsigned (&lo)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        // (two scalar multiplies on purpose: a v_pk_mul_f32 beside MFMAs costs more than the two plain VALU it replaces -- MI355X_MICROARCH.md,
        //  per-instruction constants)
        const float v0 = __uint_as_float(x[2 * p]) * s, v1 = __uint_as_float(x[2 * p + 1]) * s;
        const s_f32x2 v = {v0, v1};
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, s_f16x2));
        const s_f32x2 r = {f16_resid_lo(v[0], h), f16_resid_hi(v[1], h)};
        hi[p] = h;
        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, s_f16x2));
    }
}
// one element for the pack kernels (off the hot path): {hi, lo} as f16 bit patterns
__device__ __forceinline__ void f16_split1(float x, float s, unsigned short& hi, unsigned short& lo) {
    const float v = x * s;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

}  // namespace mi355ppo
