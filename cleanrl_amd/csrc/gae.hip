// K1 -- Generalised Advantage Estimation as one fused reverse scan (gfx950).
//
// Replaces cleanrl/ppo_atari_multigpu.py:290-301 (T x ~10 tiny elementwise launches + 1).
// The recurrence  adv[t] = delta[t] + (g*l)*nnt[t]*adv[t+1]  is sequential in t and independent
// across env columns, so a column is one lane's dependency chain and ALL loads are off that chain:
//
//   * gae_cols<V,U>  -- large N (>= a few hundred waves): one lane owns V adjacent columns and walks
//     t = T-1..0, loading U rows ahead into registers.  Loads/stores are row-contiguous across lanes
//     (4*V bytes per lane, 16 B at V=4); every input byte is read once and every output byte written
//     once, so HBM traffic == the algorithmic 20*T*N + 8*N bytes.  HBM-bound.
//   * gae_staged<CW> -- small N (BASELINE.json sizes, e.g. 128 x 1024): too few columns to hide memory latency with wave
//     parallelism; everything off the recurrence's chain is done by all 256 threads, the chain runs out of LDS (see the kernel).
//     (Round 1's single-scanner tile kernels gae_tile<CW> -- variants 2 / 4 / 5 -- were retired in round 6: `auto` never chose them.)
//
// Bit-exactness: contraction is disabled so every multiply/add rounds separately, in the
// reference's order ((g*nv)*nnt), ((r+.)-v), (((g*l)*nnt)*last); g*l is formed in double on the
// host and rounded to f32 once, as Python/torch do for `args.gamma * args.gae_lambda * tensor`.
#include "common.h"
#include "ppo_rows.h"

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float gae_f32x2 __attribute__((ext_vector_type(2)));
typedef float gae_f32x4 __attribute__((ext_vector_type(4)));
template <int V> struct Vec;
template <> struct Vec<1> { using type = float; };
template <> struct Vec<2> { using type = gae_f32x2; };
template <> struct Vec<4> { using type = gae_f32x4; };

// NT: non-temporal (`nt`) accesses -- the column kernel in its HBM regime touches every byte once (a read-only stream reaches 6.1 instead of
// 5.5 TB/s with them on the same box: profiles/r06_hbm_probe.jsonl)
template <int V, bool NT = false>
__device__ __forceinline__ void vload(float (&dst)[V], const float* p) {
    using vec = typename Vec<V>::type;
    const vec x = NT ? __builtin_nontemporal_load(reinterpret_cast<const vec*>(p)) : *reinterpret_cast<const vec*>(p);
    const float* xs = reinterpret_cast<const float*>(&x);
#pragma unroll
    for (int i = 0; i < V; ++i) dst[i] = xs[i];
}
template <int V, bool NT = false>
__device__ __forceinline__ void vstore(float* p, const float (&src)[V]) {
    using vec = typename Vec<V>::type;
    vec x;
    float* xs = reinterpret_cast<float*>(&x);
#pragma unroll
    for (int i = 0; i < V; ++i) xs[i] = src[i];
    if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<vec*>(p));
    else *reinterpret_cast<vec*>(p) = x;
}

// ---------------------------------------------------------------------------------- column kernel
template <int V, int U, bool NT>
__global__ __launch_bounds__(256) void gae_cols(const float* __restrict__ rewards, const float* __restrict__ dones,
                                                const float* __restrict__ values, const float* __restrict__ next_done,
                                                const float* __restrict__ next_value, float* __restrict__ advantages,
                                                float* __restrict__ returns, int T, int N, float gamma, float gl) {
    const int64_t col = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (col >= N) return;
    float last[V], nextv[V], nextd[V];
    vload<V>(nextv, next_value + col);
    vload<V>(nextd, next_done + col);
#pragma unroll
    for (int i = 0; i < V; ++i) last[i] = 0.0f;

    int t = T - 1;
    for (; t >= U - 1; t -= U) {
        float rb[U][V], vb[U][V], db[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (int64_t)(t - u) * N + col;
            vload<V, NT>(rb[u], rewards + off);
            vload<V, NT>(vb[u], values + off);
            vload<V, NT>(db[u], dones + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (int64_t)(t - u) * N + col;
            float a[V], rt[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                a[i] = gae_step(rb[u][i], vb[u][i], nextv[i], nextd[i], last[i], gamma, gl, &rt[i]);
                last[i] = a[i];
                nextv[i] = vb[u][i];
                nextd[i] = db[u][i];
            }
            vstore<V, NT>(advantages + off, a);
            vstore<V, NT>(returns + off, rt);
        }
    }
    for (; t >= 0; --t) {
        const int64_t off = (int64_t)t * N + col;
        float r[V], v[V], d[V], a[V], rt[V];
        vload<V, NT>(r, rewards + off);
        vload<V, NT>(v, values + off);
        vload<V, NT>(d, dones + off);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            a[i] = gae_step(r[i], v[i], nextv[i], nextd[i], last[i], gamma, gl, &rt[i]);
            last[i] = a[i];
            nextv[i] = v[i];
            nextd[i] = d[i];
        }
        vstore<V, NT>(advantages + off, a);
        vstore<V, NT>(returns + off, rt);
    }
}

// ------------------------------------------------------------------------------- staged-scan kernel
// Small N, revisited (BASELINE sizes: 128 x 1024 = 2.6 MB, L2-resident): everything that is NOT on the
// recurrence's dependency chain is done by all 256 threads, coalesced; the chain itself is two dependent f32 ops
// per step fed from LDS.
//   phase 1a  all threads: rewards / values / dones of a (TC rows x CW columns) chunk -> LDS (+ the row above it:
//             values[t+1], dones[t+1], or next_value / next_done for the last row)
//   phase 1b  all threads: delta[t] = (r + (g*nv)*nnt) - v  and  coef[t] = gl*nnt   (the reference's op order)
//   phase 2   CW lanes:    adv[t] = delta[t] + coef[t]*adv[t+1], t descending, carry kept across chunks
//   phase 3   all threads: returns = adv + values; both outputs stored coalesced
// CW columns per workgroup (4 / 16 / 64 by N) so that even N = 64 x T = 2048 spreads over 16 CUs.
constexpr int kStageFloats = 2048;      // elements of one array per chunk

template <int CW>
__global__ __launch_bounds__(256) void gae_staged(const float* __restrict__ rewards, const float* __restrict__ dones,
                                                  const float* __restrict__ values, const float* __restrict__ next_done,
                                                  const float* __restrict__ next_value, float* __restrict__ advantages,
                                                  float* __restrict__ returns, int T, int N, float gamma, float gl) {
    constexpr int TC = kStageFloats / CW;                       // rows per chunk
    constexpr int PER = kStageFloats / 256;                     // elements per thread per array (8)
    __shared__ float sR[TC * CW];                               // rewards -> delta -> advantages
    __shared__ float sV[(TC + 1) * CW];                         // values, row TC = the row above the chunk
    __shared__ float sD[(TC + 1) * CW];                         // dones, same
    __shared__ float sC[TC * CW];                               // coef = gl * nnt
    const int tid = threadIdx.x;
    const int col0 = blockIdx.x * CW;
    const int ncols = min(CW, N - col0);
    float last = 0.0f;                                          // lastgaelam = 0 (Python int in the reference)
    for (int hi = T; hi > 0; hi -= TC) {
        const int lo = max(0, hi - TC), rows = hi - lo;
        // ---- 1a
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q, j = e / CW, c = e % CW;
            if (j < rows && c < ncols) {
                const int64_t off = (int64_t)(lo + j) * N + col0 + c;
                sR[e] = rewards[off];
                sV[e] = values[off];
                sD[e] = dones[off];
            }
        }
        if (tid < ncols) {                                      // the row above the chunk
            const int c = tid;
            sV[rows * CW + c] = hi == T ? next_value[col0 + c] : values[(int64_t)hi * N + col0 + c];
            sD[rows * CW + c] = hi == T ? next_done[col0 + c] : dones[(int64_t)hi * N + col0 + c];
        }
        __syncthreads();
        // ---- 1b
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q, j = e / CW, c = e % CW;
            if (j < rows && c < ncols) {
                const float nnt = 1.0f - sD[e + CW];
                float x = gamma * sV[e + CW];
                x = x * nnt;
                x = sR[e] + x;
                sR[e] = x - sV[e];                               // delta
                sC[e] = gl * nnt;
            }
        }
        __syncthreads();
        // ---- 2
        if (tid < ncols) {
            // groups of 8 steps; the LDS reads of the next group are issued before the current group's chain runs
            int j = rows - 1;
            float dl[8], cf[8], dn[8], cn[8];
            if (j >= 7) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    dl[u] = sR[(j - u) * CW + tid];
                    cf[u] = sC[(j - u) * CW + tid];
                }
            }
            for (; j >= 7; j -= 8) {
                const bool more = j - 8 >= 7;
                if (more) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        dn[u] = sR[(j - 8 - u) * CW + tid];
                        cn[u] = sC[(j - 8 - u) * CW + tid];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float c = cf[u] * last;
                    last = dl[u] + c;
                    sR[(j - u) * CW + tid] = last;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    dl[u] = dn[u];
                    cf[u] = cn[u];
                }
            }
            for (; j >= 0; --j) {
                const float c = sC[j * CW + tid] * last;
                last = sR[j * CW + tid] + c;
                sR[j * CW + tid] = last;
            }
        }
        __syncthreads();
        // ---- 3
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q, j = e / CW, c = e % CW;
            if (j < rows && c < ncols) {
                const int64_t off = (int64_t)(lo + j) * N + col0 + c;
                const float a = sR[e];
                advantages[off] = a;
                returns[off] = a + sV[e];
            }
        }
        __syncthreads();                                        // LDS is rewritten by the next chunk
    }
}

template <int CW>
static int launch_staged(const float* r, const float* d, const float* v, const float* nd, const float* nv, float* adv,
                         float* ret, int T, int N, float gamma, float gl, hipStream_t s) {
    const int grid = (N + CW - 1) / CW;
    hipLaunchKernelGGL((gae_staged<CW>), dim3(grid), dim3(256), 0, s, r, d, v, nd, nv, adv, ret, T, N, gamma, gl);
    return check_launch("gae_staged");
}

template <int V, int U>
static int launch_cols(const float* r, const float* d, const float* v, const float* nd, const float* nv, float* adv,
                       float* ret, int T, int N, float gamma, float gl, hipStream_t s) {
    const int64_t threads = ((int64_t)N + V - 1) / V;
    const int block = threads >= 256 * 256 ? 256 : 64;   // small problems: one wave per workgroup -> more CUs
    const int grid = (int)((threads + block - 1) / block);
    // non-temporal accesses once the five arrays are past the 256-MiB Infinity Cache (below it the consumers of `adv` / `ret` find them cached)
#ifndef MI355_GAE_NT_BYTES
#define MI355_GAE_NT_BYTES (256ll << 20)
#endif
    if (20ll * T * N > MI355_GAE_NT_BYTES) hipLaunchKernelGGL((gae_cols<V, U, true>), dim3(grid), dim3(block), 0, s, r, d, v, nd, nv, adv, ret, T, N, gamma, gl);
    else hipLaunchKernelGGL((gae_cols<V, U, false>), dim3(grid), dim3(block), 0, s, r, d, v, nd, nv, adv, ret, T, N, gamma, gl);
    return check_launch("gae_cols");
}

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API int mi355ppo_gae_f32_variant(const float* rewards, const float* dones, const float* values,
                                        const float* next_done, const float* next_value, float* advantages,
                                        float* returns, int T, int N, double gamma, double gae_lambda, int variant,
                                        void* stream) {
    MI355_REQUIRE(rewards && dones && values && next_done && next_value && advantages && returns, MI355PPO_EINVAL,
                  "mi355ppo_gae_f32: null pointer");
    MI355_REQUIRE(T > 0 && N > 0, MI355PPO_EINVAL, "mi355ppo_gae_f32: T=%d N=%d must be positive", T, N);
    MI355_REQUIRE(aligned(rewards, 4) && aligned(dones, 4) && aligned(values, 4) && aligned(next_done, 4) &&
                      aligned(next_value, 4) && aligned(advantages, 4) && aligned(returns, 4),
                  MI355PPO_EALIGN, "mi355ppo_gae_f32: pointers must be 4-byte aligned");
    MI355_REQUIRE(variant == 0 || variant == 1 || variant == 3 || variant == 6, MI355PPO_EINVAL,
                  "mi355ppo_gae_f32: unknown variant %d (0, 1, 3, 6; the single-scanner tile kernels 2 / 4 / 5 were retired in round 6: auto never chose them)", variant);
    const float g = (float)gamma;
    const float gl = (float)(gamma * gae_lambda);   // double product, one rounding (see header)
    hipStream_t s = as_stream(stream);
    const bool vec4 = (N % 4 == 0) && aligned(rewards, 16) && aligned(dones, 16) && aligned(values, 16) &&
                      aligned(next_done, 16) && aligned(next_value, 16) && aligned(advantages, 16) &&
                      aligned(returns, 16);
    if (variant == 0) {
        // auto: the staged kernel while the column kernels could not even put one wave on every SIMD
        if (N < 65536) variant = 6;     // measured (profiles/r01_kbench_gae_staged_sweep.jsonl): staged wins up to ~32K columns
        else if (vec4 && N >= 262144) variant = 3;
        else variant = 1;
    }
    switch (variant) {
        case 1: return launch_cols<1, 8>(rewards, dones, values, next_done, next_value, advantages, returns, T, N, g, gl, s);
        case 3:
            MI355_REQUIRE(vec4, MI355PPO_EALIGN, "mi355ppo_gae_f32: variant 3 needs N%%4==0 and 16-byte aligned pointers");
            return launch_cols<4, 4>(rewards, dones, values, next_done, next_value, advantages, returns, T, N, g, gl, s);
        case 6:
            if (N <= 256) return launch_staged<4>(rewards, dones, values, next_done, next_value, advantages, returns, T, N, g, gl, s);
            if (N <= 4096) return launch_staged<16>(rewards, dones, values, next_done, next_value, advantages, returns, T, N, g, gl, s);
            return launch_staged<64>(rewards, dones, values, next_done, next_value, advantages, returns, T, N, g, gl, s);
    }
    return MI355PPO_EINVAL;
}

extern "C" MI355PPO_API int mi355ppo_gae_f32(const float* rewards, const float* dones, const float* values, const float* next_done,
                                const float* next_value, float* advantages, float* returns, int T, int N, double gamma,
                                double gae_lambda, void* stream) {
    return mi355ppo_gae_f32_variant(rewards, dones, values, next_done, next_value, advantages, returns, T, N, gamma,
                                    gae_lambda, 0, stream);
}
