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


}  // namespace mi355ppo
