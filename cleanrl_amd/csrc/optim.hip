// a8/a9 -- flat-buffer optimiser step: grad scale (1/world_size) -> global-norm clip -> Adam, fused
// (gfx950).
//
// Replaces, per minibatch, cleanrl/ppo_atari_multigpu.py:368-377: 10 `copy_(... / world_size)`,
// nn.utils.clip_grad_norm_ (per-tensor norms, stack, norm, clamp, 10 muls) and the ~8 foreach kernels
// of optim.Adam.step -- about 40 launches -- with two launches over one persistent flat buffer:
//   1. sumsq   : f64 block partials of sum((g*scale)^2)                     (reads 4 B/param)
//   2. update  : every workgroup folds the <=256 partials (fixed order), forms
//                coef = min(1, max_norm / (norm + 1e-6)) exactly as clip_grad.py does, then per element
//                m,v,p update with torch's Adam formulas (lerp / addcmul / sqrt / addcdiv order) and
//                zeroes the gradient for the next backward           (reads 16, writes 16 B/param)
// Algorithmic bytes = 36 B/param (1.69 M params -> 60.7 MB per step, HBM/L2-streaming, float4).
#include "common.h"
#include "ppo_rows.h"

#pragma clang fp contract(off)

namespace mi355ppo {

constexpr int kNormMaxBlocks = 256;

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, int64_t n, float scale,
                                                         double* __restrict__ partials) {
    __shared__ double red[4];
    double s = 0.0;
    const int64_t n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = g4[i];
        const float a = v.x * scale, b = v.y * scale, c = v.z * scale, d = v.w * scale;
        s += (double)a * a + (double)b * b + (double)c * c + (double)d * d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float a = g[n4 * 4 + threadIdx.x] * scale;
        s += (double)a * a;
    }
    const double r = block_sum<4>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        AdamParams A, const double* __restrict__ partials,
                                                        float* __restrict__ total_norm_out, const float* __restrict__ sched2) {
    __shared__ float s_coef;
    if (sched2) {                  // the step's two schedule-dependent constants live in device memory (captured launches)
        A.neg_step = sched2[0];
        A.bc2_sqrt = sched2[1];
    }
    if (threadIdx.x < 64) {
        double s = 0.0;
        for (int b = threadIdx.x; b < A.nblocks; b += 64) s += partials[b];
        s = wave_sum(s);
        if (threadIdx.x == 0) {
            const float total = (float)sqrt(s);
            float coef = A.max_norm / (total + 1e-6f);      // clip_grad.py: max_norm / (total_norm + 1e-6)
            coef = fminf(coef, 1.0f);                        //               clamp(max=1.0)
            s_coef = coef;
            if (blockIdx.x == 0 && total_norm_out) *total_norm_out = total;
        }
    }
    __syncthreads();
    const float coef = s_coef;
    const int64_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam_elem(pp.x, gg.x, mm.x, vv.x, coef, A);
        adam_elem(pp.y, gg.y, mm.y, vv.y, coef, A);
        adam_elem(pp.z, gg.z, mm.z, vv.z, coef, A);
        adam_elem(pp.w, gg.w, mm.w, vv.w, coef, A);
        p4[i] = pp; g4[i] = gg; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        adam_elem(p[i], g[i], m[i], v[i], coef, A);
    }
}

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API size_t mi355ppo_clip_adam_workspace_bytes(int64_t n) {
    (void)n;
    return (size_t)kNormMaxBlocks * sizeof(double);
}

// The two constants of an Adam step that depend on the schedule (learning rate, step count), as the kernel consumes them:
// out2 = {(float)(-(lr / (1 - beta1^step))), (float)sqrt(1 - beta2^step)} (torch adam.py: step_size, bias_correction2_sqrt).
extern "C" MI355PPO_API int mi355ppo_adam_schedule_f32(double lr, double beta1, double beta2, int64_t step, float* out2_host) {
    MI355_REQUIRE(out2_host && step >= 1, MI355PPO_EINVAL, "mi355ppo_adam_schedule_f32: null pointer or step=%lld < 1", (long long)step);
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    out2_host[0] = (float)(-(lr / bc1));
    out2_host[1] = (float)sqrt(bc2);
    return MI355PPO_OK;
}

static int clip_adam_impl(const char* fn, float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double grad_scale,
                          double max_grad_norm, double lr, double beta1, double beta2, double eps, int64_t step, const float* sched2,
                          float* total_norm_out, void* workspace, size_t workspace_bytes, void* stream) {
    MI355_REQUIRE(params && grads && exp_avg && exp_avg_sq, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(n > 0 && step >= 1, MI355PPO_EINVAL, "%s: n=%lld must be >0 and step=%lld >= 1", fn, (long long)n,
                  (long long)step);
    MI355_REQUIRE(workspace && workspace_bytes >= (size_t)kNormMaxBlocks * sizeof(double), MI355PPO_EWORKSPACE,
                  "%s: workspace %zu bytes < required %zu", fn, workspace ? workspace_bytes : (size_t)0,
                  (size_t)kNormMaxBlocks * sizeof(double));
    MI355_REQUIRE(aligned(params, 16) && aligned(grads, 16) && aligned(exp_avg, 16) && aligned(exp_avg_sq, 16) &&
                      aligned(workspace, 8) && aligned(total_norm_out, 4) && aligned(sched2, 4),
                  MI355PPO_EALIGN, "%s: flat buffers must be 16-byte aligned", fn);
    hipStream_t s = as_stream(stream);
    const int64_t n4 = (n + 3) / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > kNormMaxBlocks) blocks = kNormMaxBlocks;
    double* partials = static_cast<double*>(workspace);
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grads, n, (float)grad_scale, partials);
    int rc = check_launch("grad_sumsq_kernel");
    if (rc) return rc;
    AdamParams A;
    A.scale = (float)grad_scale;
    A.max_norm = (float)max_grad_norm;
    A.w1 = (float)(1.0 - beta1);
    A.beta2 = (float)beta2;
    A.w2 = (float)(1.0 - beta2);
    float sc[2];
    mi355ppo_adam_schedule_f32(lr, beta1, beta2, step, sc);
    A.neg_step = sc[0];
    A.bc2_sqrt = sc[1];
    A.eps = (float)eps;
    A.nblocks = (int)blocks;
    A.zero_grads = 1;
    int64_t ublocks = (n4 + 255) / 256;
    if (ublocks > 2048) ublocks = 2048;
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)ublocks), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                       A, partials, total_norm_out, sched2);
    return check_launch("clip_adam_kernel");
}

extern "C" MI355PPO_API int mi355ppo_clip_adam_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                      double grad_scale, double max_grad_norm, double lr, double beta1, double beta2,
                                      double eps, int64_t step, float* total_norm_out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    return clip_adam_impl("mi355ppo_clip_adam_f32", params, grads, exp_avg, exp_avg_sq, n, grad_scale, max_grad_norm, lr, beta1, beta2, eps,
                          step, nullptr, total_norm_out, workspace, workspace_bytes, stream);
}

// The same step with its two schedule-dependent constants read from DEVICE memory (`sched2` = what mi355ppo_adam_schedule_f32
// writes, copied to the device by the caller): a launch captured into a hipGraph is replayed with the next step's learning
// rate and bias corrections without re-capturing.  Bit-identical to mi355ppo_clip_adam_f32 for the same (lr, step).
extern "C" MI355PPO_API int mi355ppo_clip_adam_sched_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                                        double grad_scale, double max_grad_norm, double beta1, double beta2, double eps,
                                                        const float* sched2, float* total_norm_out, void* workspace,
                                                        size_t workspace_bytes, void* stream) {
    MI355_REQUIRE(sched2, MI355PPO_EINVAL, "mi355ppo_clip_adam_sched_f32: null pointer");
    return clip_adam_impl("mi355ppo_clip_adam_sched_f32", params, grads, exp_avg, exp_avg_sq, n, grad_scale, max_grad_norm, 0.0, beta1, beta2,
                          eps, 1, sched2, total_norm_out, workspace, workspace_bytes, stream);
}
