// Three-term bf16 split of f32 operands for the exact-product kernels (Z: gemmz.hip, W: fcw.hip, V: convw.hip): x = hi + mid + lo
// with every term exactly representable in bf16, so that products of terms are exact in f32 on the bf16 matrix pipe.
//
// hi / mid / lo = significand bits 23..16 / 15..8 / 7..0 of x: t8 = x & 0xffff0000, t16 = x & 0xffffff00, mid = t16 - t8,
// lo = x - t16 -- both subtractions exact (the operands share sign and exponent), each result has at most 8 significant bits.
// 11 VALU instructions per pair of elements, three deep, two independent instructions per element at every depth: with ONE
// wave per SIMD nothing else hides a dependent VALU instruction's latency.  The kernels inline these steps into their
// hand-ordered instruction streams (one MFMA, then its share of split / load / LDS work); the masks come in as kernel
// arguments so that they live in SGPRs (as literals every v_and is an 8-byte instruction).
#pragma once
#include "common.h"

namespace mi355ppo {

typedef unsigned int s_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned split_pack(float e1, float e0) {      // {bf16(e1), bf16(e0)} by truncation: the two high halves
    return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}

}  // namespace mi355ppo
