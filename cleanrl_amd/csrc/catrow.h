// Row math of torch.distributions.Categorical(logits=...) shared by the sampling and loss kernels.
#pragma once
#include "common.h"
#include <float.h>

#pragma clang fp contract(off)

namespace mi355ppo {

// ---- row math shared with the loss kernel ------------------------------------------------------
// torch: logits_n = logits - logsumexp(logits);  probs = softmax(logits_n);  (categorical.py:78-80)
template <int AMAX>
struct CatRow {
    float lp[AMAX];   // normalised logits (log-probabilities)
    float p[AMAX];    // probabilities
    float H;          // entropy = -sum clamp(lp, min=FLT_lowest) * p
};

template <int AMAX>
__device__ __forceinline__ void load_row(float (&x)[AMAX], const float* __restrict__ row, int A) {
    if (AMAX == 4 && A == 4) {   // (B,4) rows are 16-byte aligned whenever the base pointer is
        if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(row);
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < AMAX; ++j) x[j] = (j < A) ? row[j] : -INFINITY;
}

template <int AMAX>
MI355_HD void categorical_row(const float (&x)[AMAX], int A, CatRow<AMAX>& out) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < AMAX; ++j) if (j < A) m = fmaxf(m, x[j]);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < AMAX; ++j) if (j < A) s += expf(x[j] - m);
    const float lse = m + logf(s);
    float m2 = -INFINITY;
#pragma unroll
    for (int j = 0; j < AMAX; ++j) {
        out.lp[j] = (j < A) ? x[j] - lse : -INFINITY;
        if (j < A) m2 = fmaxf(m2, out.lp[j]);
    }
    float s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < AMAX; ++j) {
        out.p[j] = (j < A) ? expf(out.lp[j] - m2) : 0.0f;
        s2 += out.p[j];
    }
    float h = 0.0f;
#pragma unroll
    for (int j = 0; j < AMAX; ++j) {
        out.p[j] = out.p[j] / s2;
        if (j < A) h += fmaxf(out.lp[j], -FLT_MAX) * out.p[j];
    }
    out.H = -h;
}

// One draw of torch.multinomial(probs, 1) for a row whose CatRow is `c`: argmax_j probs_j / q_j with q ~ Exp(1) (ATen's
// multinomial; first maximum wins).  The draws are the caller's (`noise_row`, parity mode) or come from the Philox stream keyed by
// (seed; counter = row * ceil(A / 4) + j / 4, offset): the ONE definition behind K2 (distributions.hip), K7's rollout step (mlp.hip)
// and the fused FC + heads + sampling kernel of the NatureCNN rollout (heads.hip) -- same row, same words, same action.
template <int AMAX>
__device__ __forceinline__ int categorical_sample_row(const CatRow<AMAX>& c, int A, const float* __restrict__ noise_row, uint64_t seed,
                                                      uint64_t offset, uint64_t row, float* best_lp_out) {
    float q[AMAX];
    if (noise_row) {
#pragma unroll
        for (int j = 0; j < AMAX; ++j) q[j] = j < A ? noise_row[j] : 1.0f;
    } else {
        const Philox rng(seed);
        const int nblk = (A + 3) / 4;
#pragma unroll
        for (int g = 0; g < (AMAX + 3) / 4; ++g) {
            if (g * 4 < A) {
                const uint4 r = rng(row * nblk + g, offset);
                const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (g * 4 + k < AMAX) q[g * 4 + k] = -logf(u32_to_unit_open(rr[k]));
            }
        }
    }
    int best = 0;
    float bestv = -INFINITY, best_lp = 0.0f;
#pragma unroll
    for (int j = 0; j < AMAX; ++j) {
        if (j < A) {
            const float v = c.p[j] / q[j];
            if (v > bestv) { bestv = v; best = j; best_lp = c.lp[j]; }
        }
    }
    *best_lp_out = best_lp;
    return best;
}

}  // namespace mi355ppo
