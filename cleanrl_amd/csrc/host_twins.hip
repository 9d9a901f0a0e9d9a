// Host-pointer twins of the PPO-path entry points ("every entry point has a *_cpu twin with host pointers,
// same math, plain C++, for config A and for tests").
//
// What they are for: BASELINE config A (`cleanrl/ppo.py` CartPole, num_envs = 4, on CPU -- plumbing, no GPU) and the
// world_size-2 gloo tests of the data-parallel logic run the reference's loop on CPU tensors; with the twins that loop calls
// the SAME seams of this library as the GPU path (GAE, sampling, fused loss forward + backward, clip + Adam) instead of a
// second restatement in torch ops.  What they are NOT: a fallback.  Nothing in the library or in cleanrl_amd/ routes a device
// pointer here; a CUDA device always runs the HIP kernels and raises when they are missing (cleanrl_amd/_lib.py).
//
// Same math by construction: the row / element functions (gae_step, categorical_row, ppo_row_terms, mean_den_from_sums,
// adam_elem, the Philox stream) are the device kernels' own, compiled for the host from the same headers (ppo_rows.h,
// catrow.h, common.h), without FMA contraction.  Differences to the device results can come only from libm vs the device
// math library (expf / logf / sincosf: a few ulp) and from the order of the f64 reductions (row order here, fixed tree there).
// Serial, single-threaded: sizes of config A are a few hundred rows.
#include "common.h"
#include "catrow.h"
#include "ppo_rows.h"

#include <math.h>
#include <stddef.h>
#include <string.h>

#pragma clang fp contract(off)

using namespace mi355ppo;

#define TWIN_LOG_SQRT_2PI 0.91893853320467274178f     /* as in distributions.hip / loss.hip */
#define TWIN_HALF_LOG_2PIE 1.4189385332046727418f

namespace {

constexpr int kAMax = 64;

inline void load_host_row(float (&x)[kAMax], const float* row, int A) {
    for (int j = 0; j < kAMax; ++j) x[j] = (j < A) ? row[j] : -INFINITY;
}

inline int action_of(const int64_t* a_i64, const float* a_f32, int64_t row) {
    return a_i64 ? (int)a_i64[row] : (int)a_f32[row];
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------ K1
extern "C" MI355PPO_API int mi355ppo_gae_f32_cpu(const float* rewards, const float* dones, const float* values,
                                                 const float* next_done, const float* next_value, float* advantages,
                                                 float* returns, int T, int N, double gamma, double gae_lambda) {
    const char* fn = "mi355ppo_gae_f32_cpu";
    MI355_REQUIRE(rewards && dones && values && next_done && next_value && advantages && returns, MI355PPO_EINVAL,
                  "%s: null pointer", fn);
    MI355_REQUIRE(T > 0 && N > 0, MI355PPO_EINVAL, "%s: T=%d N=%d must be positive", fn, T, N);
    const float g = (float)gamma, gl = (float)(gamma * gae_lambda);      // g*l formed in double, rounded once (gae.hip)
    for (int n = 0; n < N; ++n) {
        float last = 0.0f, nextv = next_value[n], nextd = next_done[n];
        for (int t = T - 1; t >= 0; --t) {
            const size_t i = (size_t)t * N + n;
            float ret;
            last = gae_step(rewards[i], values[i], nextv, nextd, last, g, gl, &ret);
            advantages[i] = last;
            returns[i] = ret;
            nextv = values[i];
            nextd = dones[i];
        }
    }
    return MI355PPO_OK;
}

// ------------------------------------------------------------------------------------------------------------------ K2
extern "C" MI355PPO_API int mi355ppo_categorical_sample_f32_cpu(const float* logits, const float* noise_exp1, uint64_t seed,
                                                                uint64_t offset, int64_t* action_i64, float* action_f32,
                                                                float* logprob, float* entropy, int B, int A) {
    const char* fn = "mi355ppo_categorical_sample_f32_cpu";
    MI355_REQUIRE(logits && logprob, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(action_i64 || action_f32, MI355PPO_EINVAL, "%s: no action output", fn);
    MI355_REQUIRE(B > 0 && A > 0 && A <= kAMax, MI355PPO_EINVAL, "%s: B=%d must be >0 and A=%d in 1..64", fn, B, A);
    const Philox rng(seed);
    const int nblk = (A + 3) / 4;
    for (int row = 0; row < B; ++row) {
        float x[kAMax], q[kAMax];
        load_host_row(x, logits + (size_t)row * A, A);
        CatRow<kAMax> c;
        categorical_row<kAMax>(x, A, c);
        if (noise_exp1) {
            for (int j = 0; j < A; ++j) q[j] = noise_exp1[(size_t)row * A + j];
        } else {                                   // the device kernel's stream: counter = row * nblk + group, key = seed
            for (int g = 0; g * 4 < A; ++g) {
                const uint4 r = rng((uint64_t)row * nblk + g, offset);
                const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
                for (int k = 0; k < 4 && g * 4 + k < kAMax; ++k) q[g * 4 + k] = -logf(u32_to_unit_open(rr[k]));
            }
        }
        int best = 0;                              // multinomial(probs, 1) == argmax_j probs_j / q_j (first maximum wins)
        float bestv = -INFINITY, best_lp = 0.0f;
        for (int j = 0; j < A; ++j) {
            const float v = c.p[j] / q[j];
            if (v > bestv) { bestv = v; best = j; best_lp = c.lp[j]; }
        }
        if (action_i64) action_i64[row] = best;
        if (action_f32) action_f32[row] = (float)best;
        logprob[row] = best_lp;
        if (entropy) entropy[row] = c.H;
    }
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_categorical_logprob_entropy_f32_cpu(const float* logits, const int64_t* action_i64,
                                                                         const float* action_f32, float* logprob,
                                                                         float* entropy, int B, int A) {
    const char* fn = "mi355ppo_categorical_logprob_entropy_f32_cpu";
    MI355_REQUIRE(logits && logprob, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE((action_i64 != nullptr) != (action_f32 != nullptr), MI355PPO_EINVAL,
                  "%s: exactly one of action_i64/action_f32 must be given", fn);
    MI355_REQUIRE(B > 0 && A > 0 && A <= kAMax, MI355PPO_EINVAL, "%s: B=%d must be >0 and A=%d in 1..64", fn, B, A);
    for (int row = 0; row < B; ++row) {
        float x[kAMax];
        load_host_row(x, logits + (size_t)row * A, A);
        CatRow<kAMax> c;
        categorical_row<kAMax>(x, A, c);
        const int a = action_of(action_i64, action_f32, row);
        logprob[row] = (a >= 0 && a < A) ? c.lp[a] : 0.0f;
        if (entropy) entropy[row] = c.H;
    }
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_categorical_logprob_entropy_bwd_f32_cpu(const float* logits, const int64_t* action_i64,
                                                                             const float* action_f32, const float* g_logprob,
                                                                             const float* g_entropy, float* dlogits, int B,
                                                                             int A) {
    const char* fn = "mi355ppo_categorical_logprob_entropy_bwd_f32_cpu";
    MI355_REQUIRE(logits && dlogits, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE((action_i64 != nullptr) != (action_f32 != nullptr), MI355PPO_EINVAL,
                  "%s: exactly one of action_i64/action_f32 must be given", fn);
    MI355_REQUIRE(B > 0 && A > 0 && A <= kAMax, MI355PPO_EINVAL, "%s: B=%d must be >0 and A=%d in 1..64", fn, B, A);
    for (int row = 0; row < B; ++row) {
        float x[kAMax];
        load_host_row(x, logits + (size_t)row * A, A);
        CatRow<kAMax> c;
        categorical_row<kAMax>(x, A, c);
        const int a = action_of(action_i64, action_f32, row);
        const float gl = g_logprob ? g_logprob[row] : 0.0f, ge = g_entropy ? g_entropy[row] : 0.0f;
        for (int j = 0; j < A; ++j) {
            const float onehot = (j == a) ? 1.0f : 0.0f;
            dlogits[(size_t)row * A + j] = gl * (onehot - c.p[j]) - ge * (c.p[j] * (fmaxf(c.lp[j], -FLT_MAX) + c.H));
        }
    }
    return MI355PPO_OK;
}

// ------------------------------------------------------------------------------------------------------------------ K2'
namespace {
// One row of Normal(mean, exp(logstd)): log_prob and entropy summed over D (torch normal.py op order, distributions.hip).
template <bool SAMPLE>
inline void normal_row(const float* mean, const float* logstd, const float* noise, const Philox& rng, uint64_t offset, int64_t row,
                       float* action_out, const float* action_in, float* lp_out, float* ent_out, int D) {
    const int nblk = (D + 3) / 4;
    float lp = 0.0f, ent = 0.0f;
    float z4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d) {
        const float mu = mean[(size_t)row * D + d];
        const float sd = expf(logstd[d]);
        float a;
        if (SAMPLE) {
            float z;
            if (noise) {
                z = noise[(size_t)row * D + d];
            } else {
                if ((d & 3) == 0) {   // Box-Muller: 4 uint32 -> 2 (u1,u2) pairs -> 4 standard normals
                    const uint4 r = rng((uint64_t)row * nblk + (d >> 2), offset);
                    const float r0 = sqrtf(-2.0f * logf(u32_to_unit_open(r.x)));
                    const float r1 = sqrtf(-2.0f * logf(u32_to_unit_open(r.z)));
                    float s0, c0, s1, c1;
                    sincosf(6.283185307179586f * u32_to_unit_open(r.y), &s0, &c0);
                    sincosf(6.283185307179586f * u32_to_unit_open(r.w), &s1, &c1);
                    z4[0] = r0 * c0; z4[1] = r0 * s0; z4[2] = r1 * c1; z4[3] = r1 * s1;
                }
                z = z4[d & 3];
            }
            a = z * sd;          // torch.normal(mean, std): normal_(0,1).mul_(std).add_(mean)
            a = a + mu;
            action_out[(size_t)row * D + d] = a;
        } else {
            a = action_in[(size_t)row * D + d];
        }
        const float diff = a - mu;
        const float var = sd * sd;
        const float log_scale = logf(sd);
        float t = -(diff * diff);
        t = t / (2.0f * var);
        t = t - log_scale;
        t = t - TWIN_LOG_SQRT_2PI;
        lp += t;
        ent += TWIN_HALF_LOG_2PIE + log_scale;
    }
    *lp_out = lp;
    *ent_out = ent;
}
}  // namespace

extern "C" MI355PPO_API int mi355ppo_normal_sample_f32_cpu(const float* mean, const float* logstd, const float* noise_std_normal,
                                                           uint64_t seed, uint64_t offset, float* action, float* logprob_sum,
                                                           float* entropy_sum, int B, int D) {
    const char* fn = "mi355ppo_normal_sample_f32_cpu";
    MI355_REQUIRE(mean && logstd && action && logprob_sum, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(B > 0 && D > 0, MI355PPO_EINVAL, "%s: B=%d D=%d must be positive", fn, B, D);
    const Philox rng(seed);
    for (int row = 0; row < B; ++row) {
        float lp, ent;
        normal_row<true>(mean, logstd, noise_std_normal, rng, offset, row, action, nullptr, &lp, &ent, D);
        logprob_sum[row] = lp;
        if (entropy_sum) entropy_sum[row] = ent;
    }
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_normal_logprob_entropy_f32_cpu(const float* mean, const float* logstd, const float* action,
                                                                    float* logprob_sum, float* entropy_sum, int B, int D) {
    const char* fn = "mi355ppo_normal_logprob_entropy_f32_cpu";
    MI355_REQUIRE(mean && logstd && action && logprob_sum, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(B > 0 && D > 0, MI355PPO_EINVAL, "%s: B=%d D=%d must be positive", fn, B, D);
    const Philox rng(0);
    for (int row = 0; row < B; ++row) {
        float lp, ent;
        normal_row<false>(mean, logstd, nullptr, rng, 0, row, nullptr, action, &lp, &ent, D);
        logprob_sum[row] = lp;
        if (entropy_sum) entropy_sum[row] = ent;
    }
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_normal_logprob_entropy_bwd_f32_cpu(const float* mean, const float* logstd,
                                                                        const float* action, const float* g_logprob,
                                                                        const float* g_entropy, float* dmean,
                                                                        float* dlogstd_rows, int B, int D) {
    const char* fn = "mi355ppo_normal_logprob_entropy_bwd_f32_cpu";
    MI355_REQUIRE(mean && logstd && action && dmean && dlogstd_rows, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(B > 0 && D > 0, MI355PPO_EINVAL, "%s: B=%d D=%d must be positive", fn, B, D);
    for (int row = 0; row < B; ++row) {
        const float gl = g_logprob ? g_logprob[row] : 0.0f, ge = g_entropy ? g_entropy[row] : 0.0f;
        for (int d = 0; d < D; ++d) {
            const float sd = expf(logstd[d]);
            const float var = sd * sd;
            const float diff = action[(size_t)row * D + d] - mean[(size_t)row * D + d];
            dmean[(size_t)row * D + d] = gl * (diff / var);
            dlogstd_rows[(size_t)row * D + d] = gl * ((diff * diff) / var - 1.0f) + ge;
        }
    }
    return MI355PPO_OK;
}

// ------------------------------------------------------------------------------------------------------------------ K3
namespace {

int loss_params(const char* fn, int M, double clip_coef, double ent_coef, double vf_coef, int norm_adv, int clip_vloss,
                LossParams* P) {
    MI355_REQUIRE(M > 0, MI355PPO_EINVAL, "%s: M=%d must be positive", fn, M);
    MI355_REQUIRE(!norm_adv || M > 1, MI355PPO_EINVAL, "%s: norm_adv needs M > 1 (unbiased std)", fn);
    P->lo = (float)(1.0 - clip_coef);
    P->hi = (float)(1.0 + clip_coef);
    P->clip = (float)clip_coef;
    P->ent_coef = (float)ent_coef;
    P->vf_coef = (float)vf_coef;
    P->norm_adv = norm_adv;
    P->clip_vloss = clip_vloss;
    P->M = M;
    P->stats_blocks = 0;
    return MI355PPO_OK;
}

// (mean, unbiased std + 1e-8) of b_adv[mb_inds]: f64 sums in row order, then the device kernels' own fold.
void adv_mean_den_host(const float* b_adv, const int64_t* inds, int M, const float* given, float* mean, float* den) {
    if (given) { *mean = given[0]; *den = given[1]; return; }
    double s = 0.0, ss = 0.0;
    for (int m = 0; m < M; ++m) {
        const double a = (double)b_adv[inds ? inds[m] : m];
        s += a;
        ss += a * a;
    }
    mean_den_from_sums(s, ss, (double)M, mean, den);
}

// the seven scalars from the six f64 sums: loss_finalize's arithmetic (loss.hip)
void fold_scalars7(const double (&tot)[kNumSums], const LossParams& P, float* scalars7) {
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

}  // namespace

extern "C" MI355PPO_API int mi355ppo_loss_categorical_fwd_bwd_f32_cpu(
    const float* new_logits, const float* new_value, const int64_t* mb_inds, const float* b_actions_f32, const float* b_logprobs,
    const float* b_advantages, const float* b_returns, const float* b_values, int M, int A, double clip_coef, double ent_coef,
    double vf_coef, int norm_adv, int clip_vloss, const float* adv_mean_den, float* scalars7, float* dlogits, float* dvalue) {
    const char* fn = "mi355ppo_loss_categorical_fwd_bwd_f32_cpu";
    MI355_REQUIRE(new_logits && new_value && b_actions_f32 && b_logprobs && b_advantages && b_returns && b_values && scalars7 &&
                      dlogits && dvalue,
                  MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(A > 0 && A <= kAMax, MI355PPO_EINVAL, "%s: A=%d must be in 1..64", fn, A);
    LossParams P;
    if (int rc = loss_params(fn, M, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss, &P)) return rc;
    float mean = 0.0f, den = 1.0f;
    if (norm_adv) adv_mean_den_host(b_advantages, mb_inds, M, adv_mean_den, &mean, &den);
    double tot[kNumSums] = {0, 0, 0, 0, 0, 0};
    const float ge = P.ent_coef / (float)P.M;
    for (int m = 0; m < M; ++m) {
        const int64_t i = mb_inds ? mb_inds[m] : m;
        float x[kAMax];
        load_host_row(x, new_logits + (size_t)m * A, A);
        CatRow<kAMax> c;
        categorical_row<kAMax>(x, A, c);
        const int a = (int)b_actions_f32[i];
        const float newlp = (a >= 0 && a < A) ? c.lp[a] : 0.0f;
        const RowTerms t = ppo_row_terms(newlp, c.H, new_value[m], b_logprobs[i], b_advantages[i], b_returns[i], b_values[i], mean,
                                         den, P);
        for (int k = 0; k < kNumSums; ++k) tot[k] += (double)t.sums[k];
        dvalue[m] = t.dvalue;
        for (int j = 0; j < A; ++j) {       // d loss/d logits_j = g_lp*(1[j==a] - p_j) + (ent_coef/M) * p_j * (lp_j + H)
            const float onehot = (j == a) ? 1.0f : 0.0f;
            const float lpj = fmaxf(c.lp[j], -FLT_MAX);
            dlogits[(size_t)m * A + j] = t.g_lp * (onehot - c.p[j]) + ge * (c.p[j] * (lpj + c.H));
        }
    }
    fold_scalars7(tot, P, scalars7);
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_loss_normal_fwd_bwd_f32_cpu(
    const float* new_mean, const float* logstd, const float* new_value, const int64_t* mb_inds, const float* b_actions,
    const float* b_logprobs, const float* b_advantages, const float* b_returns, const float* b_values, int M, int D,
    double clip_coef, double ent_coef, double vf_coef, int norm_adv, int clip_vloss, const float* adv_mean_den, float* scalars7,
    float* dmean, float* dlogstd, float* dvalue) {
    const char* fn = "mi355ppo_loss_normal_fwd_bwd_f32_cpu";
    MI355_REQUIRE(new_mean && logstd && new_value && b_actions && b_logprobs && b_advantages && b_returns && b_values && scalars7 &&
                      dmean && dlogstd && dvalue,
                  MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(D > 0 && D <= 64, MI355PPO_EINVAL, "%s: D=%d must be in 1..64", fn, D);
    LossParams P;
    if (int rc = loss_params(fn, M, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss, &P)) return rc;
    float amean = 0.0f, den = 1.0f;
    if (norm_adv) adv_mean_den_host(b_advantages, mb_inds, M, adv_mean_den, &amean, &den);
    double tot[kNumSums] = {0, 0, 0, 0, 0, 0};
    double dls[64];
    for (int d = 0; d < D; ++d) dls[d] = 0.0;
    const float g_ent = -(P.ent_coef / (float)P.M);   // d loss / d entropy_row ; d entropy_row / d logstd_d = 1
    const Philox unused(0);
    for (int m = 0; m < M; ++m) {
        const int64_t i = mb_inds ? mb_inds[m] : m;
        float lp, ent;
        normal_row<false>(new_mean, logstd, nullptr, unused, 0, m, nullptr, b_actions + ((ptrdiff_t)i - (ptrdiff_t)m) * D, &lp, &ent, D);   // (row m of new_mean, row i of b_actions)
        const RowTerms t = ppo_row_terms(lp, ent, new_value[m], b_logprobs[i], b_advantages[i], b_returns[i], b_values[i], amean,
                                         den, P);
        for (int k = 0; k < kNumSums; ++k) tot[k] += (double)t.sums[k];
        dvalue[m] = t.dvalue;
        for (int d = 0; d < D; ++d) {
            const float mu = new_mean[(size_t)m * D + d];
            const float sd = expf(logstd[d]);
            const float diff = b_actions[(size_t)i * D + d] - mu;
            const float var = sd * sd;
            dmean[(size_t)m * D + d] = t.g_lp * (diff / var);
            dls[d] += (double)(t.g_lp * ((diff * diff) / var - 1.0f) + g_ent);
        }
    }
    fold_scalars7(tot, P, scalars7);
    for (int d = 0; d < D; ++d) dlogstd[d] = (float)dls[d];
    return MI355PPO_OK;
}

// ------------------------------------------------------------------------------------------------------------------ a8
extern "C" MI355PPO_API int mi355ppo_clip_adam_f32_cpu(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                                       double grad_scale, double max_grad_norm, double lr, double beta1,
                                                       double beta2, double eps, int64_t step, float* total_norm_out) {
    const char* fn = "mi355ppo_clip_adam_f32_cpu";
    MI355_REQUIRE(params && grads && exp_avg && exp_avg_sq, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(n > 0 && step >= 1, MI355PPO_EINVAL, "%s: n=%lld must be >0 and step=%lld >= 1", fn, (long long)n,
                  (long long)step);
    AdamParams A;                                   // as mi355ppo_clip_adam_f32 forms them (optim.hip)
    A.scale = (float)grad_scale;
    A.max_norm = (float)max_grad_norm;
    A.w1 = (float)(1.0 - beta1);
    A.beta2 = (float)beta2;
    A.w2 = (float)(1.0 - beta2);
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    A.bc2_sqrt = (float)sqrt(bc2);
    A.eps = (float)eps;
    A.neg_step = (float)(-(lr / bc1));
    A.nblocks = 1;
    A.zero_grads = 1;
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const float a = grads[i] * A.scale;
        s += (double)a * a;
    }
    const float total = (float)sqrt(s);
    float coef = A.max_norm / (total + 1e-6f);      // clip_grad.py: max_norm / (total_norm + 1e-6)
    coef = fminf(coef, 1.0f);                        //               clamp(max=1.0)
    if (total_norm_out) *total_norm_out = total;
    for (int64_t i = 0; i < n; ++i) adam_elem(params[i], grads[i], exp_avg[i], exp_avg_sq[i], coef, A);
    return MI355PPO_OK;
}

// ------------------------------------------------------------------------------------------------------------------ K5
extern "C" MI355PPO_API int mi355ppo_obs_u8_to_f32_cpu(const uint8_t* src_u8, const int64_t* inds, float* dst_f32, int64_t rows,
                                                       int64_t row_bytes, int scale_255) {
    const char* fn = "mi355ppo_obs_u8_to_f32_cpu";
    MI355_REQUIRE(src_u8 && dst_f32, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(rows > 0 && row_bytes > 0, MI355PPO_EINVAL, "%s: rows=%lld row_bytes=%lld must be positive", fn, (long long)rows,
                  (long long)row_bytes);
    for (int64_t r = 0; r < rows; ++r) {
        const uint8_t* s = src_u8 + (size_t)(inds ? inds[r] : r) * row_bytes;
        float* d = dst_f32 + (size_t)r * row_bytes;
        for (int64_t k = 0; k < row_bytes; ++k) d[k] = scale_255 ? (float)s[k] / 255.0f : (float)s[k];
    }
    return MI355PPO_OK;
}
