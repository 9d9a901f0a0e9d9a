// K2 / K2' -- Categorical and Normal: sample + log_prob + entropy fused into one launch (gfx950).
//
// Replaces the distribution op chains of Agent.get_action_and_value:
//   cleanrl/ppo_atari_multigpu.py:156-159  (Categorical: logits - logsumexp, softmax, multinomial,
//                                           gather, clamp*probs, sum)            ~8 launches
//   cleanrl/ppo_continuous_action.py:134-141 (Normal: exp, normal, log_prob, sum, entropy, sum) ~10
// One lane owns one row.  Rows are tiny (A <= 18 actions, D <= ~20 action dims), so a row lives in
// registers (compile-time capacity AMAX, runtime A predicated); the (B,A) array is read once and
// nothing but the per-row results is written: traffic == algorithmic bytes.  At rollout sizes
// (B = num_envs = 1024) the launch is latency-bound; the point of the fusion is 1 launch instead of 8.
#include "common.h"
#include "catrow.h"

#pragma clang fp contract(off)

namespace mi355ppo {

// ---- Categorical: sample ------------------------------------------------------------------------
template <int AMAX>
__global__ __launch_bounds__(256) void categorical_sample_kernel(const float* __restrict__ logits,
                                                                 const float* __restrict__ noise, uint64_t seed,
                                                                 uint64_t offset, const uint64_t* __restrict__ offset_base,
                                                                 int64_t* __restrict__ action_i64,
                                                                 float* __restrict__ action_f32,
                                                                 float* __restrict__ logprob,
                                                                 float* __restrict__ entropy, int B, int A) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    float x[AMAX];
    load_row<AMAX>(x, logits + (int64_t)row * A, A);
    CatRow<AMAX> c;
    categorical_row<AMAX>(x, A, c);

    if (offset_base) offset += *offset_base;           // the stream position lives in device memory: a captured launch can be replayed
    float best_lp;
    const int best = categorical_sample_row<AMAX>(c, A, noise ? noise + (int64_t)row * A : nullptr, seed, offset, (uint64_t)row, &best_lp);
    if (action_i64) action_i64[row] = best;
    if (action_f32) action_f32[row] = (float)best;
    logprob[row] = best_lp;
    if (entropy) entropy[row] = c.H;
}

template <int AMAX>
__global__ __launch_bounds__(256) void categorical_eval_kernel(const float* __restrict__ logits,
                                                               const int64_t* __restrict__ action_i64,
                                                               const float* __restrict__ action_f32,
                                                               float* __restrict__ logprob, float* __restrict__ entropy,
                                                               int B, int A) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    float x[AMAX];
    load_row<AMAX>(x, logits + (int64_t)row * A, A);
    CatRow<AMAX> c;
    categorical_row<AMAX>(x, A, c);
    const int a = action_i64 ? (int)action_i64[row] : (int)action_f32[row];
    float lp = 0.0f;
#pragma unroll
    for (int j = 0; j < AMAX; ++j) if (j == a) lp = c.lp[j];
    logprob[row] = lp;
    if (entropy) entropy[row] = c.H;
}

// ---- Normal -------------------------------------------------------------------------------------
// constants as torch rounds them: math.log(math.sqrt(2*pi)) and 0.5 + 0.5*math.log(2*pi), double -> f32
#define MI355_LOG_SQRT_2PI 0.91893853320467274178f
#define MI355_HALF_LOG_2PIE 1.4189385332046727418f

// One lane per row; D is small (<= ~20): the row is walked once, element by element.
// SAMPLE: action = z*std + mean is written;  !SAMPLE: action is read.
template <bool SAMPLE>
__global__ __launch_bounds__(256) void normal_kernel(const float* __restrict__ mean, const float* __restrict__ logstd,
                                                     const float* __restrict__ noise, uint64_t seed, uint64_t offset,
                                                     float* __restrict__ action_out,
                                                     const float* __restrict__ action_in,
                                                     float* __restrict__ logprob_sum, float* __restrict__ entropy_sum,
                                                     int B, int D) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    const Philox rng(seed);
    const int nblk = (D + 3) / 4;
    float lp = 0.0f, ent = 0.0f;
    float z4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d) {
        const float mu = mean[(int64_t)row * D + d];
        const float sd = expf(logstd[d]);
        float a;
        if (SAMPLE) {
            float z;
            if (noise) {
                z = noise[(int64_t)row * D + d];
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
                const int k = d & 3;
                z = k == 0 ? z4[0] : k == 1 ? z4[1] : k == 2 ? z4[2] : z4[3];
            }
            a = z * sd;          // torch.normal(mean, std): normal_(0,1).mul_(std).add_(mean)
            a = a + mu;
            action_out[(int64_t)row * D + d] = a;
        } else {
            a = action_in[(int64_t)row * D + d];
        }
        // normal.py log_prob: -((v - loc)**2) / (2*var) - log_scale - log(sqrt(2 pi)),  var = scale**2
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
    logprob_sum[row] = lp;
    if (entropy_sum) entropy_sum[row] = ent;
}


// ---- backward of (log_prob, entropy) w.r.t. the distribution parameters --------------------------
// For user code that differentiates Agent.get_action_and_value(x, action) itself (the learner's update
// uses the fused loss kernel instead).  dlogits_j = g_lp*(1[j==a] - p_j) - g_ent * p_j * (lp_j + H).
template <int AMAX>
__global__ __launch_bounds__(256) void categorical_bwd_kernel(const float* __restrict__ logits,
                                                              const int64_t* __restrict__ action_i64,
                                                              const float* __restrict__ action_f32,
                                                              const float* __restrict__ g_lp,
                                                              const float* __restrict__ g_ent,
                                                              float* __restrict__ dlogits, int B, int A) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    float x[AMAX];
    load_row<AMAX>(x, logits + (int64_t)row * A, A);
    CatRow<AMAX> c;
    categorical_row<AMAX>(x, A, c);
    const int a = action_i64 ? (int)action_i64[row] : (int)action_f32[row];
    const float gl = g_lp ? g_lp[row] : 0.0f, ge = g_ent ? g_ent[row] : 0.0f;
#pragma unroll
    for (int j = 0; j < AMAX; ++j) {
        if (j < A) {
            const float onehot = (j == a) ? 1.0f : 0.0f;
            dlogits[(int64_t)row * A + j] = gl * (onehot - c.p[j]) - ge * (c.p[j] * (fmaxf(c.lp[j], -FLT_MAX) + c.H));
        }
    }
}

// dmean = g_lp*(a-mu)/var ; per-row dlogstd contribution = g_lp*((a-mu)^2/var - 1) + g_ent (summed by the caller)
__global__ __launch_bounds__(256) void normal_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ logstd,
                                                         const float* __restrict__ action,
                                                         const float* __restrict__ g_lp, const float* __restrict__ g_ent,
                                                         float* __restrict__ dmean, float* __restrict__ dlogstd_rows,
                                                         int B, int D) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    const float gl = g_lp ? g_lp[row] : 0.0f, ge = g_ent ? g_ent[row] : 0.0f;
    for (int d = 0; d < D; ++d) {
        const float sd = expf(logstd[d]);
        const float var = sd * sd;
        const float diff = action[(int64_t)row * D + d] - mean[(int64_t)row * D + d];
        dmean[(int64_t)row * D + d] = gl * (diff / var);
        dlogstd_rows[(int64_t)row * D + d] = gl * ((diff * diff) / var - 1.0f) + ge;
    }
}

static inline int grid_for(int B, int block) { return (B + block - 1) / block; }
static inline int block_for(int B) { return B >= 65536 ? 256 : 64; }   // small B: spread rows over more CUs

}  // namespace mi355ppo

using namespace mi355ppo;

#define MI355_DISPATCH_AMAX(A, FN, ...)                                     \
    do {                                                                    \
        if ((A) <= 4) { FN(4, __VA_ARGS__); }                               \
        else if ((A) <= 8) { FN(8, __VA_ARGS__); }                          \
        else if ((A) <= 18) { FN(18, __VA_ARGS__); }                        \
        else { FN(64, __VA_ARGS__); }                                       \
    } while (0)

extern "C" MI355PPO_API int mi355ppo_categorical_sample_ctr_f32(const float* logits, const float* noise_exp1, uint64_t seed,
                                               uint64_t offset, const uint64_t* offset_base, int64_t* action_i64,
                                               float* action_f32, float* logprob, float* entropy, int B, int A, void* stream) {
    MI355_REQUIRE(logits && logprob, MI355PPO_EINVAL, "mi355ppo_categorical_sample_f32: null pointer");
    MI355_REQUIRE(action_i64 || action_f32, MI355PPO_EINVAL, "mi355ppo_categorical_sample_f32: no action output");
    MI355_REQUIRE(B > 0 && A > 0 && A <= 64, MI355PPO_EINVAL,
                  "mi355ppo_categorical_sample_f32: B=%d must be >0 and A=%d in 1..64", B, A);
    MI355_REQUIRE(aligned(logits, 4) && aligned(logprob, 4) && aligned(action_i64, 8) && aligned(action_f32, 4) &&
                      aligned(noise_exp1, 4) && aligned(entropy, 4) && aligned(offset_base, 8),
                  MI355PPO_EALIGN, "mi355ppo_categorical_sample_f32: misaligned pointer");
    const int block = block_for(B);
#define LAUNCH(AMAX, ...)                                                                                          \
    hipLaunchKernelGGL((categorical_sample_kernel<AMAX>), dim3(grid_for(B, block)), dim3(block), 0,                \
                       as_stream(stream), logits, noise_exp1, seed, offset, offset_base, action_i64, action_f32, logprob, \
                       entropy, B, A)
    MI355_DISPATCH_AMAX(A, LAUNCH, 0);
#undef LAUNCH
    return check_launch("categorical_sample_kernel");
}

extern "C" MI355PPO_API int mi355ppo_categorical_sample_f32(const float* logits, const float* noise_exp1, uint64_t seed,
                                               uint64_t offset, int64_t* action_i64, float* action_f32, float* logprob,
                                               float* entropy, int B, int A, void* stream) {
    return mi355ppo_categorical_sample_ctr_f32(logits, noise_exp1, seed, offset, nullptr, action_i64, action_f32, logprob, entropy, B, A,
                                               stream);
}

extern "C" MI355PPO_API int mi355ppo_categorical_logprob_entropy_f32(const float* logits, const int64_t* action_i64,
                                                        const float* action_f32, float* logprob, float* entropy,
                                                        int B, int A, void* stream) {
    MI355_REQUIRE(logits && logprob, MI355PPO_EINVAL, "mi355ppo_categorical_logprob_entropy_f32: null pointer");
    MI355_REQUIRE((action_i64 != nullptr) != (action_f32 != nullptr), MI355PPO_EINVAL,
                  "mi355ppo_categorical_logprob_entropy_f32: exactly one of action_i64/action_f32 must be given");
    MI355_REQUIRE(B > 0 && A > 0 && A <= 64, MI355PPO_EINVAL,
                  "mi355ppo_categorical_logprob_entropy_f32: B=%d must be >0 and A=%d in 1..64", B, A);
    MI355_REQUIRE(aligned(logits, 4) && aligned(logprob, 4) && aligned(action_i64, 8) && aligned(action_f32, 4) &&
                      aligned(entropy, 4),
                  MI355PPO_EALIGN, "mi355ppo_categorical_logprob_entropy_f32: misaligned pointer");
    const int block = block_for(B);
#define LAUNCH(AMAX, ...)                                                                                     \
    hipLaunchKernelGGL((categorical_eval_kernel<AMAX>), dim3(grid_for(B, block)), dim3(block), 0,             \
                       as_stream(stream), logits, action_i64, action_f32, logprob, entropy, B, A)
    MI355_DISPATCH_AMAX(A, LAUNCH, 0);
#undef LAUNCH
    return check_launch("categorical_eval_kernel");
}

extern "C" MI355PPO_API int mi355ppo_normal_sample_f32(const float* mean, const float* logstd, const float* noise_std_normal,
                                          uint64_t seed, uint64_t offset, float* action, float* logprob_sum,
                                          float* entropy_sum, int B, int D, void* stream) {
    MI355_REQUIRE(mean && logstd && action && logprob_sum, MI355PPO_EINVAL, "mi355ppo_normal_sample_f32: null pointer");
    MI355_REQUIRE(B > 0 && D > 0, MI355PPO_EINVAL, "mi355ppo_normal_sample_f32: B=%d D=%d must be positive", B, D);
    MI355_REQUIRE(aligned(mean, 4) && aligned(logstd, 4) && aligned(noise_std_normal, 4) && aligned(action, 4) &&
                      aligned(logprob_sum, 4) && aligned(entropy_sum, 4),
                  MI355PPO_EALIGN, "mi355ppo_normal_sample_f32: misaligned pointer");
    const int block = block_for(B);
    hipLaunchKernelGGL((normal_kernel<true>), dim3(grid_for(B, block)), dim3(block), 0, as_stream(stream), mean, logstd,
                       noise_std_normal, seed, offset, action, (const float*)nullptr, logprob_sum, entropy_sum, B, D);
    return check_launch("normal_kernel<sample>");
}

extern "C" MI355PPO_API int mi355ppo_normal_logprob_entropy_f32(const float* mean, const float* logstd, const float* action,
                                                   float* logprob_sum, float* entropy_sum, int B, int D, void* stream) {
    MI355_REQUIRE(mean && logstd && action && logprob_sum, MI355PPO_EINVAL,
                  "mi355ppo_normal_logprob_entropy_f32: null pointer");
    MI355_REQUIRE(B > 0 && D > 0, MI355PPO_EINVAL, "mi355ppo_normal_logprob_entropy_f32: B=%d D=%d must be positive", B, D);
    MI355_REQUIRE(aligned(mean, 4) && aligned(logstd, 4) && aligned(action, 4) && aligned(logprob_sum, 4) &&
                      aligned(entropy_sum, 4),
                  MI355PPO_EALIGN, "mi355ppo_normal_logprob_entropy_f32: misaligned pointer");
    const int block = block_for(B);
    hipLaunchKernelGGL((normal_kernel<false>), dim3(grid_for(B, block)), dim3(block), 0, as_stream(stream), mean, logstd,
                       (const float*)nullptr, 0ull, 0ull, (float*)nullptr, action, logprob_sum, entropy_sum, B, D);
    return check_launch("normal_kernel<eval>");
}

extern "C" MI355PPO_API int mi355ppo_categorical_logprob_entropy_bwd_f32(const float* logits, const int64_t* action_i64,
                                                                         const float* action_f32, const float* g_logprob,
                                                                         const float* g_entropy, float* dlogits, int B,
                                                                         int A, void* stream) {
    const char* fn = "mi355ppo_categorical_logprob_entropy_bwd_f32";
    MI355_REQUIRE(logits && dlogits, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE((action_i64 != nullptr) != (action_f32 != nullptr), MI355PPO_EINVAL,
                  "%s: exactly one of action_i64/action_f32 must be given", fn);
    MI355_REQUIRE(B > 0 && A > 0 && A <= 64, MI355PPO_EINVAL, "%s: B=%d must be >0 and A=%d in 1..64", fn, B, A);
    MI355_REQUIRE(aligned(logits, 4) && aligned(dlogits, 4) && aligned(action_i64, 8) && aligned(action_f32, 4) &&
                      aligned(g_logprob, 4) && aligned(g_entropy, 4),
                  MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    const int block = block_for(B);
#define LAUNCH(AMAX, ...)                                                                                    \
    hipLaunchKernelGGL((categorical_bwd_kernel<AMAX>), dim3(grid_for(B, block)), dim3(block), 0,             \
                       as_stream(stream), logits, action_i64, action_f32, g_logprob, g_entropy, dlogits, B, A)
    MI355_DISPATCH_AMAX(A, LAUNCH, 0);
#undef LAUNCH
    return check_launch("categorical_bwd_kernel");
}

extern "C" MI355PPO_API int mi355ppo_normal_logprob_entropy_bwd_f32(const float* mean, const float* logstd,
                                                                    const float* action, const float* g_logprob,
                                                                    const float* g_entropy, float* dmean,
                                                                    float* dlogstd_rows, int B, int D, void* stream) {
    const char* fn = "mi355ppo_normal_logprob_entropy_bwd_f32";
    MI355_REQUIRE(mean && logstd && action && dmean && dlogstd_rows, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(B > 0 && D > 0, MI355PPO_EINVAL, "%s: B=%d D=%d must be positive", fn, B, D);
    MI355_REQUIRE(aligned(mean, 4) && aligned(logstd, 4) && aligned(action, 4) && aligned(dmean, 4) &&
                      aligned(dlogstd_rows, 4) && aligned(g_logprob, 4) && aligned(g_entropy, 4),
                  MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    const int block = block_for(B);
    hipLaunchKernelGGL(normal_bwd_kernel, dim3(grid_for(B, block)), dim3(block), 0, as_stream(stream), mean, logstd,
                       action, g_logprob, g_entropy, dmean, dlogstd_rows, B, D);
    return check_launch("normal_bwd_kernel");
}
