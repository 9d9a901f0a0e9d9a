// Row / element math of the PPO kernels (K1 GAE step, K3 loss row terms and advantage statistics, the Adam element update),
// defined ONCE for both sides: the device kernels (gae.hip, loss.hip, optim.hip) and their host-pointer twins
// (host_twins.hip, the *_cpu entry points of include/mi355ppo.h) call the same functions, compiled without FMA contraction.
#pragma once
#include "common.h"
#include <float.h>
#include <math.h>

#pragma clang fp contract(off)

namespace mi355ppo {

// ---- K1: one step of the GAE recurrence, the reference's op order (cleanrl/ppo_atari_multigpu.py:290-301) ----------------
MI355_HD float gae_step(float r, float v, float nextv, float nextd, float last,
                                          float gamma, float gl, float* ret_out) {
    const float nnt = 1.0f - nextd;
    float x = gamma * nextv;
    x = x * nnt;
    x = r + x;
    const float delta = x - v;
    float c = gl * nnt;
    c = c * last;
    const float adv = delta + c;
    *ret_out = adv + v;
    return adv;
}

// ---- K3: minibatch loss (cleanrl/ppo_atari_multigpu.py:320-355) -------------------------------------------------------
constexpr int kNumSums = 6;             // pg, v, entropy, -logratio, (ratio-1)-logratio, clip indicator

struct LossParams {
    float lo, hi;        // (float)(1 - clip), (float)(1 + clip): torch.clamp(ratio, 1 - c, 1 + c) scalar args
    float clip;          // (float)clip
    float ent_coef, vf_coef;
    int norm_adv, clip_vloss;
    int M;
    int stats_blocks;    // partial pairs written by loss_adv_stats; 0 = the caller supplied (mean, std + 1e-8)
};

// torch: (adv - adv.mean()) / (adv.std() + 1e-8), std unbiased.
MI355_HD void mean_den_from_sums(double s, double ss, double n, float* mean, float* den) {
    const double mu = s / n;
    double var = (ss - s * mu) / (n - 1.0);
    if (var < 0.0) var = 0.0;
    *mean = (float)mu;
    *den = (float)sqrt(var) + 1e-8f;
}

struct RowTerms {
    float g_lp;      // d loss / d newlogprob for this row (already / M)
    float dvalue;    // d loss / d newvalue
    float sums[kNumSums];
};

// Everything that does not depend on the distribution family.
MI355_HD RowTerms ppo_row_terms(float newlp, float H, float v, float old_lp, float adv, float ret,
                                                  float old_v, float mean, float den, const LossParams& P) {
    RowTerms o;
    const float inv_m = 1.0f / (float)P.M;
    const float logratio = newlp - old_lp;
    const float ratio = expf(logratio);
    float A = adv;
    if (P.norm_adv) A = (adv - mean) / den;
    const float nA = -A;
    const float pg1 = nA * ratio;
    const float clamped = fminf(fmaxf(ratio, P.lo), P.hi);
    const float pg2 = nA * clamped;
    const float inr = (ratio >= P.lo && ratio <= P.hi) ? 1.0f : 0.0f;
    float w;   // d max(pg1,pg2) / d ratio, in units of nA
    if (pg1 > pg2) w = 1.0f;
    else if (pg2 > pg1) w = inr;
    else w = 0.5f + 0.5f * inr;
    o.g_lp = (inv_m * (nA * w)) * ratio;

    const float du = v - ret;
    const float u = du * du;
    float vterm, gv;
    if (P.clip_vloss) {
        const float dv = v - old_v;
        const float cl = fminf(fmaxf(dv, -P.clip), P.clip);
        const float vc = old_v + cl;
        const float dc = vc - ret;
        const float c = dc * dc;
        const float inv = (dv >= -P.clip && dv <= P.clip) ? 1.0f : 0.0f;
        vterm = fmaxf(u, c);
        if (u > c) gv = 2.0f * du;
        else if (c > u) gv = (2.0f * dc) * inv;
        else gv = 0.5f * (2.0f * du) + 0.5f * ((2.0f * dc) * inv);
    } else {
        vterm = u;
        gv = 2.0f * du;
    }
    o.dvalue = ((P.vf_coef * 0.5f) * inv_m) * gv;
    o.sums[0] = fmaxf(pg1, pg2);
    o.sums[1] = vterm;
    o.sums[2] = H;
    o.sums[3] = -logratio;
    o.sums[4] = (ratio - 1.0f) - logratio;
    o.sums[5] = (fabsf(ratio - 1.0f) > P.clip) ? 1.0f : 0.0f;
    return o;
}

// ---- a8: clip + Adam element update (torch adam.py / clip_grad.py op order) -------------------------------------------
struct AdamParams {
    float scale, max_norm;
    float w1;         // (float)(1 - beta1)              lerp weight
    float beta2;      // (float)beta2
    float w2;         // (float)(1 - beta2)
    float bc2_sqrt;   // (float)sqrt(1 - beta2^step)
    float eps;
    float neg_step;   // (float)(-(lr / (1 - beta1^step)))
    int nblocks;
    int zero_grads;
};

MI355_HD void adam_elem(float& p, float& g, float& m, float& v, float coef, const AdamParams& A) {
    float gg = g * A.scale;
    gg = gg * coef;
    m = m + A.w1 * (gg - m);                 // exp_avg.lerp_(grad, 1 - beta1), small-weight form
    v = v * A.beta2;                         // exp_avg_sq.mul_(beta2)
    v = v + (A.w2 * gg) * gg;                //            .addcmul_(grad, grad, value=1 - beta2)
    float denom = sqrtf(v) / A.bc2_sqrt;     // (exp_avg_sq.sqrt() / bias_correction2_sqrt)
    denom = denom + A.eps;                   //            .add_(eps)
    p = p + A.neg_step * (m / denom);        // param.addcdiv_(exp_avg, denom, value=-step_size)
    g = A.zero_grads ? 0.0f : gg;
}

}  // namespace mi355ppo
