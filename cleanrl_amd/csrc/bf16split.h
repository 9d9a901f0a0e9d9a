// Three-term bf16 split of f32 operands for the exact-product kernels (X: fcx.hip, C: convx.hip): x = hi + mid + lo with
// every term exactly representable in bf16, so that products of terms are exact in f32 on the bf16 matrix pipe.
#pragma once
#include "common.h"

namespace mi355ppo {

typedef unsigned int s_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));

struct SplitTerms {
    s_bf16x8 t[3];       // hi, mid, lo of 8 consecutive k
};

__device__ __forceinline__ unsigned split_pack(float e1, float e0) {      // {bf16(e1), bf16(e0)} by truncation: the two high halves
    return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}

// MODE 0 ("adaptive", rounds 1-2): hi = x truncated to 8 significand bits, mid = (x - hi) truncated to ITS top 8 bits,
//   lo = the rest.  A five-deep dependency chain per element (and, sub, and, sub, pack).
// MODE 1 ("fixed", round 3): hi / mid / lo = significand bits 23..16 / 15..8 / 7..0 of x: t8 = x & 0xffff0000,
//   t16 = x & 0xffffff00, mid = t16 - t8, lo = x - t16 -- both subtractions exact (the operands share sign and exponent), each
//   result has at most 8 significant bits.  The same 11 VALU instructions per pair of elements, but three deep instead of
//   five and two independent instructions per element at every depth: with ONE wave per SIMD nothing else hides a dependent
//   VALU instruction's latency (profiles/r03_pmc_*: 56 % of kernel C's wave cycles were issue stalls with the pipe 37 % busy).
// `m8`, `m16`: the two masks, passed in from kernel arguments so that they live in SGPRs (as literals every v_and is an
// 8-byte instruction).
template <int MODE>
__device__ __forceinline__ SplitTerms split8(const s_u32x4& lo4, const s_u32x4& hi4, unsigned m8, unsigned m16) {
    const unsigned xb[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    unsigned h[4], m[4], l[4];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float x0 = __uint_as_float(xb[j]), x1 = __uint_as_float(xb[j + 1]);
            const float r0 = x0 - __uint_as_float(xb[j] & m8), r1 = x1 - __uint_as_float(xb[j + 1] & m8);
            const float l0 = r0 - __uint_as_float(__float_as_uint(r0) & m8), l1 = r1 - __uint_as_float(__float_as_uint(r1) & m8);
            h[j >> 1] = split_pack(x1, x0);
            m[j >> 1] = split_pack(r1, r0);
            l[j >> 1] = split_pack(l1, l0);
        }
    } else {
        // four elements at a time, stage by stage (8 ands, then 8 subs, then 6 packs): an instruction's operands were written
        // at least four instructions earlier.  sched_barriers pin the order (the scheduler otherwise re-serialises element by
        // element to save four registers); the MFMAs the caller interleaves are placed by its sched_group_barriers.
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
            unsigned t8[4], t16[4];
            float mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t8[j] = xb[g + j] & m8;
                t16[j] = xb[g + j] & m16;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mid[j] = __uint_as_float(t16[j]) - __uint_as_float(t8[j]);
                lo[j] = __uint_as_float(xb[g + j]) - __uint_as_float(t16[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                h[(g + j) >> 1] = __builtin_amdgcn_perm(xb[g + j + 1], xb[g + j], 0x07060302u);
                m[(g + j) >> 1] = split_pack(mid[j + 1], mid[j]);
                l[(g + j) >> 1] = split_pack(lo[j + 1], lo[j]);
            }
        }
    }
    SplitTerms o;
    o.t[0] = __builtin_bit_cast(s_bf16x8, (s_u32x4){h[0], h[1], h[2], h[3]});
    o.t[1] = __builtin_bit_cast(s_bf16x8, (s_u32x4){m[0], m[1], m[2], m[3]});
    o.t[2] = __builtin_bit_cast(s_bf16x8, (s_u32x4){l[0], l[1], l[2], l[3]});
    return o;
}

// MI355PPO_BF16_SPLIT = adaptive | fixed (default), read once: which split the kernels use (A/B switch)
inline int bf16_split_mode() {
    static const int m = [] {
        const char* e = getenv("MI355PPO_BF16_SPLIT");
        return (e && e[0] == 'a') ? 0 : 1;
    }();
    return m;
}

}  // namespace mi355ppo
