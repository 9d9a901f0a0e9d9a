/*
 * ppo_oracle.c -- scalar C restatement of the PPO hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).
 *
 * Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  Each function follows the reference lines it cites (cleanrl/ppo_atari_multigpu.py unless
 * stated) in plain f32 arithmetic, compiled with -ffp-contract=off so each op rounds as torch's
 * un-fused CPU ops do.  Pinned against the tests/golden fixtures (outputs of the reference's own lines) by
 * tests/test_oracle_golden.py.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* K1 -- :290-301.  rewards/dones/values (T,N) row-major; gamma*lambda formed in double (Python). */
void oracle_gae_f32(const float* rewards, const float* dones, const float* values, const float* next_done,
                    const float* next_value, float* advantages, float* returns, int T, int N, double gamma,
                    double gae_lambda) {
    const float g = (float)gamma, gl = (float)(gamma * gae_lambda);
    for (int n = 0; n < N; ++n) {
        float last = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const float nd = (t == T - 1) ? next_done[n] : dones[(size_t)(t + 1) * N + n];
            const float nv = (t == T - 1) ? next_value[n] : values[(size_t)(t + 1) * N + n];
            const float nnt = 1.0f - nd;
            float x = g * nv;
            x = x * nnt;
            x = rewards[(size_t)t * N + n] + x;
            const float delta = x - values[(size_t)t * N + n];
            float c = gl * nnt;
            c = c * last;
            last = delta + c;
            advantages[(size_t)t * N + n] = last;
            returns[(size_t)t * N + n] = last + values[(size_t)t * N + n];
        }
    }
}

/* Categorical(logits=.) row: :156 -> torch categorical.py (logits - logsumexp; probs = softmax). */
static void cat_row(const float* x, int A, float* lp, float* p, float* H) {
    float m = -INFINITY, s = 0.0f, m2 = -INFINITY, s2 = 0.0f, h = 0.0f;
    for (int j = 0; j < A; ++j) m = fmaxf(m, x[j]);
    for (int j = 0; j < A; ++j) s += expf(x[j] - m);
    const float lse = m + logf(s);
    for (int j = 0; j < A; ++j) { lp[j] = x[j] - lse; m2 = fmaxf(m2, lp[j]); }
    for (int j = 0; j < A; ++j) { p[j] = expf(lp[j] - m2); s2 += p[j]; }
    for (int j = 0; j < A; ++j) { p[j] = p[j] / s2; h += fmaxf(lp[j], -FLT_MAX) * p[j]; }
    *H = -h;
}

/* K2 -- :156-159; sample = argmax(probs / q), q ~ Exp(1) supplied (multinomial's one-draw fast path). */
void oracle_categorical_sample_f32(const float* logits, const float* noise_exp1, int64_t* action, float* logprob,
                                   float* entropy, int B, int A) {
    float lp[64], p[64];
    for (int b = 0; b < B; ++b) {
        float H;
        cat_row(logits + (size_t)b * A, A, lp, p, &H);
        int best = 0;
        float bv = -INFINITY;
        for (int j = 0; j < A; ++j) {
            const float v = p[j] / noise_exp1[(size_t)b * A + j];
            if (v > bv) { bv = v; best = j; }
        }
        action[b] = best;
        logprob[b] = lp[best];
        entropy[b] = H;
    }
}

/* K3 -- :320-355 forward, with the closed-form backward to the network outputs.
 * scalars7 = loss, pg_loss, v_loss, entropy, old_approx_kl, approx_kl, clipfrac. */
void oracle_loss_categorical_f32(const float* logits, const float* value, const int64_t* mb_inds,
                                 const float* b_actions, const float* b_logprobs, const float* b_adv,
                                 const float* b_ret, const float* b_val, int M, int A, double clip_coef,
                                 double ent_coef, double vf_coef, int norm_adv, int clip_vloss, float* scalars7,
                                 float* dlogits, float* dvalue) {
    const float lo = (float)(1.0 - clip_coef), hi = (float)(1.0 + clip_coef), clip = (float)clip_coef;
    const float entc = (float)ent_coef, vfc = (float)vf_coef;
    double s = 0.0, ss = 0.0;
    for (int m = 0; m < M; ++m) {
        const double a = b_adv[mb_inds ? mb_inds[m] : m];
        s += a;
    }
    const double mean_d = s / M;
    for (int m = 0; m < M; ++m) {
        const double a = b_adv[mb_inds ? mb_inds[m] : m] - mean_d;
        ss += a * a;
    }
    const float mean = (float)mean_d;
    const float den = (float)sqrt(ss / (M - 1.0)) + 1e-8f;   /* unbiased std + 1e-8 (:332) */
    const float inv_m = 1.0f / (float)M;
    double S[6] = {0, 0, 0, 0, 0, 0};
    float lp[64], p[64];
    for (int m = 0; m < M; ++m) {
        const int64_t i = mb_inds ? mb_inds[m] : m;
        const int a = (int)b_actions[i];
        float H;
        cat_row(logits + (size_t)m * A, A, lp, p, &H);
        const float logratio = lp[a] - b_logprobs[i];
        const float ratio = expf(logratio);
        float Adv = b_adv[i];
        if (norm_adv) Adv = (Adv - mean) / den;
        const float nA = -Adv;
        const float pg1 = nA * ratio;
        const float pg2 = nA * fminf(fmaxf(ratio, lo), hi);
        const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
        const float w = pg1 > pg2 ? 1.0f : (pg2 > pg1 ? inr : 0.5f + 0.5f * inr);
        const float g_lp = (inv_m * (nA * w)) * ratio;
        const float v = value[m], R = b_ret[i], vo = b_val[i];
        const float du = v - R, u = du * du;
        float vterm = u, gv = 2.0f * du;
        if (clip_vloss) {
            const float dv = v - vo;
            const float vc = vo + fminf(fmaxf(dv, -clip), clip);
            const float dc = vc - R, c = dc * dc;
            const float in2 = (dv >= -clip && dv <= clip) ? 1.0f : 0.0f;
            vterm = fmaxf(u, c);
            if (u > c) gv = 2.0f * du;
            else if (c > u) gv = (2.0f * dc) * in2;
            else gv = 0.5f * (2.0f * du) + 0.5f * ((2.0f * dc) * in2);
        }
        dvalue[m] = ((vfc * 0.5f) * inv_m) * gv;
        const float ge = entc / (float)M;
        for (int j = 0; j < A; ++j) {
            const float onehot = (j == a) ? 1.0f : 0.0f;
            dlogits[(size_t)m * A + j] = g_lp * (onehot - p[j]) + ge * (p[j] * (fmaxf(lp[j], -FLT_MAX) + H));
        }
        S[0] += fmaxf(pg1, pg2);
        S[1] += vterm;
        S[2] += H;
        S[3] += -logratio;
        S[4] += (ratio - 1.0f) - logratio;
        S[5] += (fabsf(ratio - 1.0f) > clip) ? 1.0 : 0.0;
    }
    const float pg_loss = (float)(S[0] / M), v_loss = 0.5f * (float)(S[1] / M), ent = (float)(S[2] / M);
    float loss = pg_loss - entc * ent;
    loss = loss + v_loss * vfc;
    scalars7[0] = loss;
    scalars7[1] = pg_loss;
    scalars7[2] = v_loss;
    scalars7[3] = ent;
    scalars7[4] = (float)(S[3] / M);
    scalars7[5] = (float)(S[4] / M);
    scalars7[6] = (float)(S[5] / M);
}

/* K5 -- b_obs[mb_inds] then x / 255.0 (:320,154), from uint8 storage. */
void oracle_obs_u8_to_f32(const uint8_t* src, const int64_t* inds, float* dst, int64_t rows, int64_t row_bytes,
                          int scale_255) {
    for (int64_t r = 0; r < rows; ++r) {
        const uint8_t* s = src + (inds ? inds[r] : r) * row_bytes;
        float* d = dst + r * row_bytes;
        for (int64_t k = 0; k < row_bytes; ++k) d[k] = scale_255 ? (float)s[k] / 255.0f : (float)s[k];
    }
}
