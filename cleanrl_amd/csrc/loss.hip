// K3 -- PPO minibatch loss: clipped surrogate + (clipped) value loss + entropy bonus, forward AND
// backward in one pass over the minibatch (gfx950).
//
// Replaces cleanrl/ppo_atari_multigpu.py:320-355 (5 index-gathers, exp, 3 KL statistics, advantage
// mean/std normalisation, clamp/max/mean chains: ~45 small launches) plus the autograd backward of
// all of them down to the network outputs (~as many again) and the `.item()` sync of :328.
//
// Three launches on the caller's stream, no host sync, deterministic (fixed-order) reductions:
//   1. loss_adv_stats   : sum / sum-of-squares (f64) of b_advantages[mb_inds] -> <=128 block partials
//   2. loss_*_main      : one lane per minibatch row.  Every workgroup first folds the <=128 stats
//                         partials itself (2 KiB from L2 -- cheaper than a grid barrier or a 4th
//                         launch), then gathers the row's old logprob / advantage / return / value /
//                         action through mb_inds, evaluates the distribution, the three loss terms and
//                         their closed-form gradients, writes dlogits (or dmean) and dvalue, and emits
//                         6 (+D) f64 block partials.
//   3. loss_finalize    : one workgroup folds the block partials into the 7 scalars (+ dlogstd).
// A kernel boundary (~1.5 us) is cheaper on MI355X than a software grid barrier (>=4 us), which is
// why the phases are launches rather than one persistent kernel.
//
// HBM traffic == algorithmic bytes: logits are read once and dlogits written once ((8A+28)*M bytes,
// +8*M for mb_inds); the five (Bflat) arrays are gathered 4 bytes at a time and stay L2-resident.
//
// Gradient of torch.max(a, b) at a == b is split 1/2 + 1/2 (derivatives.yaml `maximum`), and
// torch.clamp passes gradient on the closed interval; both are reproduced.
#include "common.h"
#include "catrow.h"

#pragma clang fp contract(off)

namespace mi355ppo {

constexpr int kStatsMaxBlocks = 128;
constexpr int kNumSums = 6;   // pg, v, entropy, -logratio, (ratio-1)-logratio, clip indicator
constexpr int kMaxD = 64;

struct LossParams {
    float lo, hi;        // (float)(1 - clip), (float)(1 + clip): torch.clamp(ratio, 1 - c, 1 + c) scalar args
    float clip;          // (float)clip
    float ent_coef, vf_coef;
    int norm_adv, clip_vloss;
    int M;
    int stats_blocks;
};

// ---- 1. advantage statistics ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_adv_stats(const int64_t* __restrict__ inds,
                                                      const float* __restrict__ b_adv, int M,
                                                      double* __restrict__ partials) {
    __shared__ double red[4];
    double s = 0.0, ss = 0.0;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const int64_t i = inds ? inds[m] : m;
        const double a = (double)b_adv[i];
        s += a;
        ss += a * a;
    }
    const double bs = block_sum<4>(s, red);
    const double bss = block_sum<4>(ss, red);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = bs;
        partials[2 * blockIdx.x + 1] = bss;
    }
}

// mean and (std + 1e-8) of the minibatch advantages, identical in every workgroup.
// torch: (adv - adv.mean()) / (adv.std() + 1e-8), std unbiased.
__device__ __forceinline__ void fold_adv_stats(const double* __restrict__ partials, const LossParams& P,
                                               float* s_mean, float* s_den) {
    if (threadIdx.x < 64) {
        double s = 0.0, ss = 0.0;
        for (int b = threadIdx.x; b < P.stats_blocks; b += 64) {
            s += partials[2 * b];
            ss += partials[2 * b + 1];
        }
        s = wave_sum(s);
        ss = wave_sum(ss);
        if (threadIdx.x == 0) {
            const double n = (double)P.M;
            const double mean = s / n;
            double var = (ss - s * mean) / (n - 1.0);
            if (var < 0.0) var = 0.0;
            const float stdf = (float)sqrt(var);
            *s_mean = (float)mean;
            *s_den = stdf + 1e-8f;
        }
    }
    __syncthreads();
}

struct RowTerms {
    float g_lp;      // d loss / d newlogprob for this row (already / M)
    float dvalue;    // d loss / d newvalue
    float sums[kNumSums];
};

// Everything that does not depend on the distribution family.
__device__ __forceinline__ RowTerms ppo_row_terms(float newlp, float H, float v, float old_lp, float adv, float ret,
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

__device__ __forceinline__ void emit_block_sums(const float (&sums)[kNumSums], double* __restrict__ out, int stride,
                                                double (*red)[kNumSums + kMaxD]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) {
        const double w = wave_sum((double)sums[k]);
        if (lane == 0) red[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < kNumSums) {
        const int k = threadIdx.x;
        out[(int64_t)blockIdx.x * stride + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

// ---- 2a. categorical main ------------------------------------------------------------------------
template <int AMAX>
__global__ __launch_bounds__(256) void loss_categorical_main(
    const float* __restrict__ logits, const float* __restrict__ value, const int64_t* __restrict__ inds,
    const float* __restrict__ b_actions, const float* __restrict__ b_logprobs, const float* __restrict__ b_adv,
    const float* __restrict__ b_ret, const float* __restrict__ b_val, int A, LossParams P,
    const double* __restrict__ stats_partials, double* __restrict__ block_partials, float* __restrict__ dlogits,
    float* __restrict__ dvalue) {
    __shared__ double red[4][kNumSums + kMaxD];
    __shared__ float s_mean, s_den;
    if (P.norm_adv) fold_adv_stats(stats_partials, P, &s_mean, &s_den);
    const float mean = P.norm_adv ? s_mean : 0.0f, den = P.norm_adv ? s_den : 1.0f;

    const int m = blockIdx.x * 256 + threadIdx.x;
    float sums[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) sums[k] = 0.0f;
    if (m < P.M) {
        const int64_t i = inds ? inds[m] : m;
        const int a = (int)b_actions[i];              // b_actions.long()[mb_inds]  (:320)
        const float old_lp = b_logprobs[i], adv = b_adv[i], ret = b_ret[i], old_v = b_val[i];
        const float v = value[m];
        float x[AMAX];
        load_row<AMAX>(x, logits + (int64_t)m * A, A);
        CatRow<AMAX> c;
        categorical_row<AMAX>(x, A, c);
        float newlp = 0.0f;
#pragma unroll
        for (int j = 0; j < AMAX; ++j) if (j == a) newlp = c.lp[j];
        const RowTerms t = ppo_row_terms(newlp, c.H, v, old_lp, adv, ret, old_v, mean, den, P);
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) sums[k] = t.sums[k];
        dvalue[m] = t.dvalue;
        // d loss/d logits_j = g_lp*(1[j==a] - p_j) + (ent_coef/M) * p_j * (lp_j + H)
        const float ge = P.ent_coef / (float)P.M;
        float g[AMAX];
#pragma unroll
        for (int j = 0; j < AMAX; ++j) {
            const float onehot = (j == a) ? 1.0f : 0.0f;
            const float lpj = fmaxf(c.lp[j], -FLT_MAX);
            g[j] = t.g_lp * (onehot - c.p[j]) + ge * (c.p[j] * (lpj + c.H));
        }
        float* out = dlogits + (int64_t)m * A;
        if (AMAX == 4 && A == 4 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            *reinterpret_cast<float4*>(out) = make_float4(g[0], g[1], g[2], g[3]);
        } else {
#pragma unroll
            for (int j = 0; j < AMAX; ++j) if (j < A) out[j] = g[j];
        }
    }
    emit_block_sums(sums, block_partials, kNumSums, red);
}

// ---- 2b. normal main -----------------------------------------------------------------------------
#define MI355_LOG_SQRT_2PI 0.91893853320467274178f
#define MI355_HALF_LOG_2PIE 1.4189385332046727418f

__global__ __launch_bounds__(256) void loss_normal_main(
    const float* __restrict__ mean_in, const float* __restrict__ logstd, const float* __restrict__ value,
    const int64_t* __restrict__ inds, const float* __restrict__ b_actions, const float* __restrict__ b_logprobs,
    const float* __restrict__ b_adv, const float* __restrict__ b_ret, const float* __restrict__ b_val, int D,
    LossParams P, const double* __restrict__ stats_partials, double* __restrict__ block_partials,
    float* __restrict__ dmean, float* __restrict__ dvalue) {
    __shared__ double red[4][kNumSums + kMaxD];
    __shared__ float s_mean, s_den;
    if (P.norm_adv) fold_adv_stats(stats_partials, P, &s_mean, &s_den);
    const float amean = P.norm_adv ? s_mean : 0.0f, den = P.norm_adv ? s_den : 1.0f;

    const int m = blockIdx.x * 256 + threadIdx.x;
    const bool active = m < P.M;
    const int stride = kNumSums + D;
    float sums[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) sums[k] = 0.0f;
    int64_t i = 0;
    float g_lp = 0.0f;
    if (active) {
        i = inds ? inds[m] : m;
        float lp = 0.0f, ent = 0.0f;
        for (int d = 0; d < D; ++d) {       // ppo_continuous_action.py:134-141 via torch normal.py
            const float mu = mean_in[(int64_t)m * D + d];
            const float sd = expf(logstd[d]);
            const float a = b_actions[i * D + d];
            const float diff = a - mu;
            const float var = sd * sd;
            const float log_scale = logf(sd);
            float t = -(diff * diff);
            t = t / (2.0f * var);
            t = t - log_scale;
            t = t - MI355_LOG_SQRT_2PI;
            lp += t;
            ent += MI355_HALF_LOG_2PIE + log_scale;
        }
        const RowTerms t = ppo_row_terms(lp, ent, value[m], b_logprobs[i], b_adv[i], b_ret[i], b_val[i], amean, den, P);
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) sums[k] = t.sums[k];
        dvalue[m] = t.dvalue;
        g_lp = t.g_lp;
    }
    const float g_ent = -(P.ent_coef / (float)P.M);   // d loss / d entropy_row ; d entropy_row / d logstd_d = 1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 0; d < D; ++d) {                      // uniform trip count: wave_sum needs every lane
        float contrib = 0.0f;
        if (active) {
            const float mu = mean_in[(int64_t)m * D + d];
            const float sd = expf(logstd[d]);
            const float a = b_actions[i * D + d];
            const float diff = a - mu;
            const float var = sd * sd;
            dmean[(int64_t)m * D + d] = g_lp * (diff / var);
            contrib = g_lp * ((diff * diff) / var - 1.0f) + g_ent;
        }
        const double w = wave_sum((double)contrib);
        if (lane == 0) red[wave][kNumSums + d] = w;
    }
    emit_block_sums(sums, block_partials, stride, red);   // barrier inside also publishes red[*][6+d]
    if (threadIdx.x < D) {
        const int k = kNumSums + threadIdx.x;
        block_partials[(int64_t)blockIdx.x * stride + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

// ---- 3. finalize ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_finalize(const double* __restrict__ block_partials, int nblocks, int D,
                                                     LossParams P, float* __restrict__ scalars7,
                                                     float* __restrict__ dlogstd) {
    __shared__ double red[4];
    __shared__ double tot[kNumSums + kMaxD];
    const int stride = kNumSums + D;
    for (int k = 0; k < stride; ++k) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 256) s += block_partials[(int64_t)b * stride + k];
        const double r = block_sum<4>(s, red);
        if (threadIdx.x == 0) tot[k] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = (double)P.M;
        const float pg_loss = (float)(tot[0] / n);
        const float v_loss = 0.5f * (float)(tot[1] / n);
        const float entropy = (float)(tot[2] / n);
        float loss = pg_loss - P.ent_coef * entropy;     // :355  pg_loss - ent_coef*entropy + v_loss*vf_coef
        loss = loss + v_loss * P.vf_coef;
        scalars7[0] = loss;
        scalars7[1] = pg_loss;
        scalars7[2] = v_loss;
        scalars7[3] = entropy;
        scalars7[4] = (float)(tot[3] / n);
        scalars7[5] = (float)(tot[4] / n);
        scalars7[6] = (float)(tot[5] / n);
    }
    if (dlogstd && threadIdx.x < D) dlogstd[threadIdx.x] = (float)tot[kNumSums + threadIdx.x];
}

static inline int stats_blocks_for(int M) {
    const int b = (M + 255) / 256;
    return b < kStatsMaxBlocks ? b : kStatsMaxBlocks;
}
static inline size_t ws_bytes(int M, int D) {
    const size_t main_blocks = ((size_t)M + 255) / 256;
    return (2 * (size_t)kStatsMaxBlocks + main_blocks * (size_t)(kNumSums + D)) * sizeof(double);
}
static LossParams make_params(int M, double clip, double ent, double vf, int norm_adv, int clip_vloss) {
    LossParams P;
    P.lo = (float)(1.0 - clip);
    P.hi = (float)(1.0 + clip);
    P.clip = (float)clip;
    P.ent_coef = (float)ent;
    P.vf_coef = (float)vf;
    P.norm_adv = norm_adv ? 1 : 0;
    P.clip_vloss = clip_vloss ? 1 : 0;
    P.M = M;
    P.stats_blocks = stats_blocks_for(M);
    return P;
}

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API size_t mi355ppo_loss_workspace_bytes(int M, int D) {
    if (M <= 0 || D < 0) return 0;
    return ws_bytes(M, D);
}

static int check_common(const char* fn, const void* a, const void* b, const float* b_actions, const float* b_logprobs,
                        const float* b_adv, const float* b_ret, const float* b_val, int M, const void* scalars,
                        const void* g1, const void* g2, void* ws, size_t ws_bytes_given, int D) {
    MI355_REQUIRE(a && b && b_actions && b_logprobs && b_adv && b_ret && b_val && scalars && g1 && g2, MI355PPO_EINVAL,
                  "%s: null pointer", fn);
    MI355_REQUIRE(M > 0, MI355PPO_EINVAL, "%s: M=%d must be positive", fn, M);
    MI355_REQUIRE(ws && ws_bytes_given >= ws_bytes(M, D), MI355PPO_EWORKSPACE,
                  "%s: workspace %zu bytes < required %zu", fn, ws ? ws_bytes_given : (size_t)0, ws_bytes(M, D));
    MI355_REQUIRE(aligned(ws, 8), MI355PPO_EALIGN, "%s: workspace must be 8-byte aligned", fn);
    MI355_REQUIRE(aligned(a, 4) && aligned(b, 4) && aligned(b_actions, 4) && aligned(b_logprobs, 4) && aligned(b_adv, 4) &&
                      aligned(b_ret, 4) && aligned(b_val, 4) && aligned(scalars, 4) && aligned(g1, 4) && aligned(g2, 4),
                  MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_loss_categorical_fwd_bwd_f32(const float* new_logits, const float* new_value,
                                                     const int64_t* mb_inds, const float* b_actions_f32,
                                                     const float* b_logprobs, const float* b_advantages,
                                                     const float* b_returns, const float* b_values, int M, int A,
                                                     double clip_coef, double ent_coef, double vf_coef, int norm_adv,
                                                     int clip_vloss, float* scalars7, float* dlogits, float* dvalue,
                                                     void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "mi355ppo_loss_categorical_fwd_bwd_f32";
    int rc = check_common(fn, new_logits, new_value, b_actions_f32, b_logprobs, b_advantages, b_returns, b_values, M,
                          scalars7, dlogits, dvalue, workspace, workspace_bytes, 0);
    if (rc) return rc;
    MI355_REQUIRE(A > 0 && A <= 64, MI355PPO_EINVAL, "%s: A=%d must be in 1..64", fn, A);
    MI355_REQUIRE(aligned(mb_inds, 8), MI355PPO_EALIGN, "%s: mb_inds must be 8-byte aligned", fn);
    hipStream_t s = as_stream(stream);
    const LossParams P = make_params(M, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss);
    double* stats = static_cast<double*>(workspace);
    double* partials = stats + 2 * kStatsMaxBlocks;
    const int blocks = (M + 255) / 256;
    if (P.norm_adv) {
        hipLaunchKernelGGL(loss_adv_stats, dim3(P.stats_blocks), dim3(256), 0, s, mb_inds, b_advantages, M, stats);
        rc = check_launch("loss_adv_stats");
        if (rc) return rc;
    }
#define LAUNCH(AMAX, ...)                                                                                          \
    hipLaunchKernelGGL((loss_categorical_main<AMAX>), dim3(blocks), dim3(256), 0, s, new_logits, new_value, mb_inds, \
                       b_actions_f32, b_logprobs, b_advantages, b_returns, b_values, A, P, stats, partials, dlogits,  \
                       dvalue)
    if (A <= 4) { LAUNCH(4); } else if (A <= 8) { LAUNCH(8); } else if (A <= 18) { LAUNCH(18); } else { LAUNCH(64); }
#undef LAUNCH
    rc = check_launch("loss_categorical_main");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, s, partials, blocks, 0, P, scalars7, (float*)nullptr);
    return check_launch("loss_finalize");
}

extern "C" MI355PPO_API int mi355ppo_loss_normal_fwd_bwd_f32(const float* new_mean, const float* logstd, const float* new_value,
                                                const int64_t* mb_inds, const float* b_actions,
                                                const float* b_logprobs, const float* b_advantages,
                                                const float* b_returns, const float* b_values, int M, int D,
                                                double clip_coef, double ent_coef, double vf_coef, int norm_adv,
                                                int clip_vloss, float* scalars7, float* dmean, float* dlogstd,
                                                float* dvalue, void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "mi355ppo_loss_normal_fwd_bwd_f32";
    MI355_REQUIRE(D > 0 && D <= kMaxD, MI355PPO_EINVAL, "%s: D=%d must be in 1..%d", fn, D, kMaxD);
    MI355_REQUIRE(logstd && dlogstd, MI355PPO_EINVAL, "%s: null pointer", fn);
    int rc = check_common(fn, new_mean, new_value, b_actions, b_logprobs, b_advantages, b_returns, b_values, M, scalars7,
                          dmean, dvalue, workspace, workspace_bytes, D);
    if (rc) return rc;
    MI355_REQUIRE(aligned(mb_inds, 8) && aligned(logstd, 4) && aligned(dlogstd, 4), MI355PPO_EALIGN,
                  "%s: misaligned pointer", fn);
    hipStream_t s = as_stream(stream);
    const LossParams P = make_params(M, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss);
    double* stats = static_cast<double*>(workspace);
    double* partials = stats + 2 * kStatsMaxBlocks;
    const int blocks = (M + 255) / 256;
    if (P.norm_adv) {
        hipLaunchKernelGGL(loss_adv_stats, dim3(P.stats_blocks), dim3(256), 0, s, mb_inds, b_advantages, M, stats);
        rc = check_launch("loss_adv_stats");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(loss_normal_main, dim3(blocks), dim3(256), 0, s, new_mean, logstd, new_value, mb_inds, b_actions,
                       b_logprobs, b_advantages, b_returns, b_values, D, P, stats, partials, dmean, dvalue);
    rc = check_launch("loss_normal_main");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, s, partials, blocks, D, P, scalars7, dlogstd);
    return check_launch("loss_finalize");
}
