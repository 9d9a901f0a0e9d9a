// Test / benchmark support (NOT part of the reference path): one step of the device-resident synthetic Atari
// vector env that bench.py and the GPU tests use in place of envpool / ALE (absent from this image).  Same byte
// streams as cleanrl_amd/envs.py::SyntheticAtariVecEnv: observation n = planes[cursor_n .. cursor_n + 3] of a fixed
// random plane pool (consecutive observations share 3 of 4 channels, like FrameStack(4)), reward in {-1, 0, +1} with
// P = (.05, .9, .05), episode end Bernoulli(done_p) after which the cursor jumps.  As torch ops this was ~14 launches
// per env step; here it is two: the per-env scalars, and the frame gather (28,224 B per env, 16 B per lane).
#include "common.h"

namespace mi355ppo {

__global__ __launch_bounds__(256) void synth_env_scalars_kernel(long long* __restrict__ cursor, float* __restrict__ reward,
                                                                float* __restrict__ done, int N, int pool, float done_p,
                                                                uint64_t seed, uint64_t step, const uint64_t* __restrict__ step_base) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    if (step_base) step += *step_base;               // replayable: the env's step count lives in device memory
    const uint4 r = Philox(seed)((uint64_t)n, step);
    const float u = u32_to_unit_open(r.x), ud = u32_to_unit_open(r.y);
    reward[n] = (u > 0.95f ? 1.0f : 0.0f) - (u < 0.05f ? 1.0f : 0.0f);
    const bool d = ud < done_p;
    done[n] = d ? 1.0f : 0.0f;
    cursor[n] = d ? (long long)(r.z % (uint32_t)pool) : cursor[n] + 1;
}

// obs[n][c] = planes[(cursor[n] + c) % pool]; a plane is 84*84 = 7056 B = 441 x 16 B
__global__ __launch_bounds__(256) void synth_env_frames_kernel(const uint8_t* __restrict__ planes,
                                                               const long long* __restrict__ cursor,
                                                               uint8_t* __restrict__ obs, int pool) {
    const int nc = blockIdx.x;                       // n*4 + c
    const long long src = (cursor[nc >> 2] + (nc & 3)) % pool;
    const uint4* s = reinterpret_cast<const uint4*>(planes + src * 7056LL);
    uint4* d = reinterpret_cast<uint4*>(obs + (long long)nc * 7056LL);
    for (int e = threadIdx.x; e < 441; e += 256) d[e] = s[e];
}

// The same gather written pixel-interleaved -- obs[n][y][x][c] = planes[(cursor[n] + c) % pool][y][x], the learner's rollout-row
// layout: a lane reads one dword (4 pixels) of each of the env's 4 planes, transposes the 4 x 4 byte block in registers
// (obs.hip's relayout) and stores 16 contiguous bytes.  One launch where the channel-planar gather + the relayout kernel were two.
__global__ __launch_bounds__(256) void synth_env_frames_hwc_kernel(const uint8_t* __restrict__ planes, const long long* __restrict__ cursor,
                                                                   uint8_t* __restrict__ obs, int pool) {
    const int n = blockIdx.x;
    const int q = blockIdx.y * 256 + threadIdx.x;                // pixel quad, 1,764 per plane
    if (q >= 1764) return;
    const long long c0 = cursor[n];
    uint32_t w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = reinterpret_cast<const uint32_t*>(planes + ((c0 + c) % pool) * 7056LL)[q];
    uint4 o;
    o.x = (w[0] & 0xffu) | ((w[1] & 0xffu) << 8) | ((w[2] & 0xffu) << 16) | (w[3] << 24);
    o.y = ((w[0] >> 8) & 0xffu) | (w[1] & 0xff00u) | ((w[2] & 0xff00u) << 8) | ((w[3] & 0xff00u) << 16);
    o.z = ((w[0] >> 16) & 0xffu) | ((w[1] >> 8) & 0xff00u) | (w[2] & 0xff0000u) | ((w[3] & 0xff0000u) << 8);
    o.w = (w[0] >> 24) | ((w[1] >> 16) & 0xff00u) | ((w[2] >> 8) & 0xff0000u) | (w[3] & 0xff000000u);
    reinterpret_cast<uint4*>(obs + (long long)n * 28224LL)[q] = o;
}

// ---- the continuous-control stand-in (cleanrl_amd/envs.py::DeviceSyntheticContinuousVecEnv, bench.py --config E) --------------------
// a = clip(action, -1, 1); next = noise[k % bank][n] + state[n] @ At + a @ Bm; reward = next . w - 0.1 |a|^2; 1000-step truncation;
// the state row is the observation.  As torch ops this was ~12 launches per env step.  Thread (env, j) of a half-wave computes
// element j of its env's next state (O <= 32 lanes of the half-wave active); the reward is a half-wave reduction.
__global__ __launch_bounds__(256) void synth_continuous_step_kernel(float* __restrict__ state, const float* __restrict__ reset_state,
                                                                    const float* __restrict__ At, const float* __restrict__ Bm,
                                                                    const float* __restrict__ w, const float* __restrict__ noise, int bank,
                                                                    uint64_t k, const uint64_t* __restrict__ k_base, float* __restrict__ steps,
                                                                    float horizon, const float* __restrict__ action, float* __restrict__ obs_out,
                                                                    float* __restrict__ reward, float* __restrict__ done, int N, int O, int D) {
    const int j = threadIdx.x & 31;
    const int n = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (n >= N) return;                                   // (whole half-waves leave together)
    if (k_base) k += *k_base;
    const bool act = j < O;
    const int jc = act ? j : 0;
    float v = noise[((size_t)(k % (uint64_t)bank) * N + n) * O + jc];
    for (int i = 0; i < O; ++i) v = fmaf(state[(size_t)n * O + i], At[i * O + jc], v);
    float a2 = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float a = fminf(fmaxf(action[(size_t)n * D + d], -1.0f), 1.0f);
        v = fmaf(a, Bm[d * O + jc], v);
        a2 = fmaf(a, a, a2);
    }
    float r = act ? v * w[jc] : 0.0f;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) r += __shfl_xor(r, off, 32);
    const float t = steps[n] + 1.0f;
    const bool dn = t >= horizon;
    const float nx = dn ? reset_state[(size_t)n * O + jc] : v;
    // every lane of the half-wave has read its inputs (the state row above all) before any lane overwrites the row
    __builtin_amdgcn_wave_barrier();
    if (act) {
        state[(size_t)n * O + j] = nx;
        if (obs_out != state) obs_out[(size_t)n * O + j] = nx;
    }
    if (j == 0) {
        reward[n] = r - 0.1f * a2;
        done[n] = dn ? 1.0f : 0.0f;
        steps[n] = dn ? 0.0f : t;
    }
}

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API int mi355ppo_synth_continuous_step_f32(float* state, const float* reset_state, const float* At, const float* Bm,
                                                              const float* w, const float* noise, int bank, uint64_t k,
                                                              const uint64_t* k_base, float* steps, double horizon, const float* action,
                                                              float* obs_out, float* reward, float* done, int N, int O, int D,
                                                              void* stream) {
    const char* fn = "mi355ppo_synth_continuous_step_f32";
    MI355_REQUIRE(state && reset_state && At && Bm && w && noise && steps && action && obs_out && reward && done, MI355PPO_EINVAL,
                  "%s: null pointer", fn);
    MI355_REQUIRE(N > 0 && bank > 0 && O > 0 && O <= 32 && D > 0 && D <= 8, MI355PPO_EINVAL, "%s: N=%d bank=%d O=%d (1..32) D=%d (1..8)", fn, N,
                  bank, O, D);
    MI355_REQUIRE(aligned(state, 4) && aligned(reset_state, 4) && aligned(At, 4) && aligned(Bm, 4) && aligned(w, 4) && aligned(noise, 4) &&
                      aligned(k_base, 8) && aligned(steps, 4) && aligned(action, 4) && aligned(obs_out, 4) && aligned(reward, 4) && aligned(done, 4),
                  MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    hipLaunchKernelGGL(synth_continuous_step_kernel, dim3((N + 7) / 8), dim3(256), 0, as_stream(stream), state, reset_state, At, Bm, w,
                       noise, bank, k, k_base, steps, (float)horizon, action, obs_out, reward, done, N, O, D);
    return check_launch(fn);
}

static int synth_atari_step(const char* fn, bool hwc, const uint8_t* planes, int pool, int64_t* cursor, uint64_t seed, uint64_t step,
                            const uint64_t* step_base, uint8_t* obs, float* reward, float* done, int N, double done_p, int advance,
                            void* stream) {
    MI355_REQUIRE(planes && cursor && obs, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(N > 0 && pool > 0, MI355PPO_EINVAL, "%s: N=%d pool=%d must be positive", fn, N, pool);
    MI355_REQUIRE(!advance || (reward && done), MI355PPO_EINVAL, "%s: reward/done are required when advancing", fn);
    MI355_REQUIRE(aligned(planes, 16) && aligned(obs, 16) && aligned(cursor, 8) && aligned(step_base, 8), MI355PPO_EALIGN,
                  "%s: planes/obs must be 16-byte aligned", fn);
    hipStream_t s = as_stream(stream);
    if (advance) {
        hipLaunchKernelGGL(synth_env_scalars_kernel, dim3((N + 255) / 256), dim3(256), 0, s, reinterpret_cast<long long*>(cursor),
                           reward, done, N, pool, (float)done_p, seed, step, step_base);
        int rc = check_launch("synth_env_scalars_kernel");
        if (rc) return rc;
    }
    if (hwc) {
        hipLaunchKernelGGL(synth_env_frames_hwc_kernel, dim3(N, 7), dim3(256), 0, s, planes, reinterpret_cast<const long long*>(cursor), obs, pool);
        return check_launch("synth_env_frames_hwc_kernel");
    }
    hipLaunchKernelGGL(synth_env_frames_kernel, dim3(N * 4), dim3(256), 0, s, planes, reinterpret_cast<const long long*>(cursor), obs, pool);
    return check_launch("synth_env_frames_kernel");
}

extern "C" MI355PPO_API int mi355ppo_synth_atari_step_ctr_u8(const uint8_t* planes, int pool, int64_t* cursor, uint64_t seed,
                                                             uint64_t step, const uint64_t* step_base, uint8_t* obs, float* reward,
                                                             float* done, int N, double done_p, int advance, void* stream) {
    return synth_atari_step("mi355ppo_synth_atari_step_u8", false, planes, pool, cursor, seed, step, step_base, obs, reward, done, N, done_p,
                            advance, stream);
}

// The same step with the observation written pixel-interleaved, (N, 84, 84, 4): the layout of the learner's rollout rows.
extern "C" MI355PPO_API int mi355ppo_synth_atari_step_hwc_ctr_u8(const uint8_t* planes, int pool, int64_t* cursor, uint64_t seed,
                                                                 uint64_t step, const uint64_t* step_base, uint8_t* obs, float* reward,
                                                                 float* done, int N, double done_p, int advance, void* stream) {
    return synth_atari_step("mi355ppo_synth_atari_step_hwc_ctr_u8", true, planes, pool, cursor, seed, step, step_base, obs, reward, done, N,
                            done_p, advance, stream);
}

extern "C" MI355PPO_API int mi355ppo_synth_atari_step_u8(const uint8_t* planes, int pool, int64_t* cursor, uint64_t seed,
                                                         uint64_t step, uint8_t* obs, float* reward, float* done, int N,
                                                         double done_p, int advance, void* stream) {
    return mi355ppo_synth_atari_step_ctr_u8(planes, pool, cursor, seed, step, nullptr, obs, reward, done, N, done_p, advance, stream);
}
