// K5 -- observation path: uint8 rollout storage -> f32 network input, gather + convert fused (gfx950).
//
// Replaces `b_obs[mb_inds]` (an f32 index-gather of (M,4,84,84)) followed by `x / 255.0`
// (cleanrl/ppo_atari_multigpu.py:320,154): the reference reads 4 B + writes 4 B per pixel for the
// gather and again for the division (16 B/pixel of HBM traffic per minibatch row).  Here a pixel costs
// 1 B read + 4 B written: rows are gathered straight from the uint8 rollout buffer and converted in
// registers.  Algorithmic bytes per row = row_bytes * 5; at M=32768 rows of 28,224 B that is 4.62 GB
// per minibatch -- by far the largest byte mover of the PPO update and purely HBM-bound.
//
// Mapping: grid = (rows, ceil(dwords_per_row / 1024)); a workgroup owns 1024 consecutive source dwords
// of ONE row, so the row index (and mb_inds[row]) is wave-uniform (SGPR) and there is no per-lane
// division.  A lane loads 4 independent dwords (256 B per wave-instruction, contiguous) and issues 4
// float4 stores (1 KiB per wave-instruction, contiguous): every global instruction is fully coalesced.
//
// x/255.0f must be the correctly-rounded IEEE quotient to stay bit-equal to torch.  A reciprocal
// multiply is wrong for 126 of the 256 byte values; one Newton residual step fixes all of them
// (q = x*r; e = fma(-q, 255, x); q' = fma(e, r, q)), verified exhaustively in tests/test_gpu_obs.py.
#include "common.h"

namespace mi355ppo {

__device__ __forceinline__ float u8_div255(float x) {
    const float r = 1.0f / 255.0f;
    const float q = x * r;
    const float e = fmaf(-q, 255.0f, x);
    return fmaf(e, r, q);
}

template <bool SCALE>
__device__ __forceinline__ float4 convert4(uint32_t w) {
    float4 o;
    o.x = (float)(w & 0xffu);
    o.y = (float)((w >> 8) & 0xffu);
    o.z = (float)((w >> 16) & 0xffu);
    o.w = (float)(w >> 24);
    if (SCALE) {
        o.x = u8_div255(o.x); o.y = u8_div255(o.y); o.z = u8_div255(o.z); o.w = u8_div255(o.w);
    }
    return o;
}

constexpr int kObsUnroll = 4;

template <bool SCALE>
__global__ __launch_bounds__(256) void obs_u8_to_f32_kernel(const uint8_t* __restrict__ src,
                                                            const int64_t* __restrict__ inds,
                                                            float* __restrict__ dst, int64_t row_bytes,
                                                            int dwords_per_row) {
    const int64_t r = blockIdx.x;
    const int64_t sr = inds ? inds[r] : r;
    const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(src + sr * row_bytes);
    float4* __restrict__ d = reinterpret_cast<float4*>(dst + r * row_bytes);
    const int base = blockIdx.y * (256 * kObsUnroll) + threadIdx.x;
    uint32_t w[kObsUnroll];
#pragma unroll
    for (int u = 0; u < kObsUnroll; ++u) {
        const int k = base + u * 256;
        w[u] = (k < dwords_per_row) ? s[k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kObsUnroll; ++u) {
        const int k = base + u * 256;
        if (k < dwords_per_row) d[k] = convert4<SCALE>(w[u]);
    }
}


// ---- planar (C,H*W) uint8 frames -> interleaved (H*W,C): the store-time relayout of the rollout buffer ----
// Env frames arrive channel-planar (FrameStack: (4,84,84)).  Keeping the rollout buffer pixel-interleaved
// ((84,84,4), "NHWC") lets the gather+convert kernel above stay a pure streaming copy AND hands the conv
// stack channels-last activations, which removes every layout transpose MIOpen otherwise inserts around its
// NHWC implicit-GEMM kernels (18% of an iteration when measured).  The relayout touches uint8 data once per
// env step (29 MB at N=1024), 4x cheaper than doing it on f32.
// C == 4: a lane reads one dword (4 pixels) from each of the 4 planes, transposes the 4x4 byte block in
// registers and writes 16 contiguous bytes: all loads and the store are fully coalesced, no LDS.
__global__ __launch_bounds__(256) void nchw_to_nhwc_u8_c4_kernel(const uint8_t* __restrict__ src,
                                                                 uint8_t* __restrict__ dst, int quads_per_row) {
    const int64_t r = blockIdx.x;
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= quads_per_row) return;
    const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(src + r * (int64_t)quads_per_row * 16);
    const uint32_t w0 = s[q], w1 = s[quads_per_row + q], w2 = s[2 * quads_per_row + q], w3 = s[3 * quads_per_row + q];
    uint4 o;
    o.x = (w0 & 0xffu) | ((w1 & 0xffu) << 8) | ((w2 & 0xffu) << 16) | (w3 << 24);
    o.y = ((w0 >> 8) & 0xffu) | (w1 & 0xff00u) | ((w2 & 0xff00u) << 8) | ((w3 & 0xff00u) << 16);
    o.z = ((w0 >> 16) & 0xffu) | ((w1 >> 8) & 0xff00u) | (w2 & 0xff0000u) | ((w3 & 0xff0000u) << 8);
    o.w = (w0 >> 24) | ((w1 >> 16) & 0xff00u) | ((w2 >> 8) & 0xff0000u) | (w3 & 0xff000000u);
    reinterpret_cast<uint4*>(dst + r * (int64_t)quads_per_row * 16)[q] = o;
}

// ---- FrameStack(4) delta: next row = previous row with its channels shifted down by one + the newest frame ----
// A frame-stacked Atari env returns obs[t+1][0:3] == obs[t][1:4] for every env that was not reset, so only the newest
// 84x84 plane has to cross PCIe (7 KB per env and step instead of 28 KB; the reference sends 113 KB of f32,
// ppo_atari_multigpu.py:272).  Pixel-interleaved rows make the shift one `>> 8` per pixel dword; a lane handles four
// pixels: 16 bytes of the previous row + 4 bytes of the new plane -> 16 bytes out, all coalesced.
__global__ __launch_bounds__(256) void shift_append_u8_c4_kernel(const uint8_t* __restrict__ prev, const uint8_t* __restrict__ newest,
                                                                 uint8_t* __restrict__ dst, int quads_per_row) {
    const int64_t r = blockIdx.x;
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= quads_per_row) return;
    const uint4 p = reinterpret_cast<const uint4*>(prev + r * (int64_t)quads_per_row * 16)[q];
    const uint32_t nw = reinterpret_cast<const uint32_t*>(newest + r * (int64_t)quads_per_row * 4)[q];
    uint4 o;
    o.x = (p.x >> 8) | ((nw & 0xffu) << 24);
    o.y = (p.y >> 8) | ((nw & 0xff00u) << 16);
    o.z = (p.z >> 8) | ((nw & 0xff0000u) << 8);
    o.w = (p.w >> 8) | (nw & 0xff000000u);
    reinterpret_cast<uint4*>(dst + r * (int64_t)quads_per_row * 16)[q] = o;
}

// generic C (slow path, byte granularity)
__global__ __launch_bounds__(256) void nchw_to_nhwc_u8_generic_kernel(const uint8_t* __restrict__ src,
                                                                      uint8_t* __restrict__ dst, int C, int HW) {
    const int64_t r = blockIdx.x;
    const int i = blockIdx.y * 256 + threadIdx.x;     // output byte index within the row: pixel*C + c
    if (i >= C * HW) return;
    const int pix = i / C, c = i % C;
    dst[r * (int64_t)C * HW + i] = src[r * (int64_t)C * HW + (int64_t)c * HW + pix];
}

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API int mi355ppo_obs_u8_to_f32(const uint8_t* src_u8, const int64_t* inds, float* dst_f32, int64_t rows,
                                      int64_t row_bytes, int scale_255, void* stream) {
    const char* fn = "mi355ppo_obs_u8_to_f32";
    MI355_REQUIRE(src_u8 && dst_f32, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(rows > 0 && rows <= 2147483647LL, MI355PPO_EINVAL, "%s: rows=%lld out of range", fn, (long long)rows);
    MI355_REQUIRE(row_bytes > 0 && row_bytes % 4 == 0 && row_bytes / 4 <= (int64_t)65535 * 1024, MI355PPO_EINVAL,
                  "%s: row_bytes=%lld must be a positive multiple of 4 (<= 256 MiB)", fn, (long long)row_bytes);
    MI355_REQUIRE(aligned(src_u8, 4) && aligned(dst_f32, 16) && aligned(inds, 8), MI355PPO_EALIGN,
                  "%s: src must be 4-byte, dst 16-byte, inds 8-byte aligned", fn);
    const int dpr = (int)(row_bytes / 4);
    const dim3 grid((unsigned)rows, (unsigned)((dpr + 256 * kObsUnroll - 1) / (256 * kObsUnroll)));
    if (scale_255)
        hipLaunchKernelGGL((obs_u8_to_f32_kernel<true>), grid, dim3(256), 0, as_stream(stream), src_u8, inds, dst_f32,
                           row_bytes, dpr);
    else
        hipLaunchKernelGGL((obs_u8_to_f32_kernel<false>), grid, dim3(256), 0, as_stream(stream), src_u8, inds, dst_f32,
                           row_bytes, dpr);
    return check_launch("obs_u8_to_f32_kernel");
}

extern "C" MI355PPO_API int mi355ppo_obs_nchw_to_nhwc_u8(const uint8_t* src, uint8_t* dst, int64_t rows, int C, int HW,
                                                         void* stream) {
    const char* fn = "mi355ppo_obs_nchw_to_nhwc_u8";
    MI355_REQUIRE(src && dst && src != dst, MI355PPO_EINVAL, "%s: null or aliased pointers", fn);
    MI355_REQUIRE(rows > 0 && rows <= 2147483647LL && C > 0 && HW > 0 && (int64_t)C * HW <= (int64_t)65535 * 256,
                  MI355PPO_EINVAL, "%s: rows=%lld C=%d HW=%d out of range", fn, (long long)rows, C, HW);
    if (C == 4 && HW % 4 == 0 && aligned(src, 4) && aligned(dst, 16)) {
        const int quads = HW / 4;
        hipLaunchKernelGGL(nchw_to_nhwc_u8_c4_kernel, dim3((unsigned)rows, (unsigned)((quads + 255) / 256)), dim3(256), 0,
                           as_stream(stream), src, dst, quads);
        return check_launch("nchw_to_nhwc_u8_c4_kernel");
    }
    hipLaunchKernelGGL(nchw_to_nhwc_u8_generic_kernel, dim3((unsigned)rows, (unsigned)((C * HW + 255) / 256)), dim3(256),
                       0, as_stream(stream), src, dst, C, HW);
    return check_launch("nchw_to_nhwc_u8_generic_kernel");
}

extern "C" MI355PPO_API int mi355ppo_obs_shift_append_u8_c4(const uint8_t* prev_rows, const uint8_t* newest_planes, uint8_t* dst_rows,
                                                            int64_t rows, int HW, void* stream) {
    const char* fn = "mi355ppo_obs_shift_append_u8_c4";
    MI355_REQUIRE(prev_rows && newest_planes && dst_rows, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(rows > 0 && rows <= 2147483647LL && HW > 0 && HW % 4 == 0 && HW / 4 <= 65535 * 256, MI355PPO_EINVAL,
                  "%s: rows=%lld HW=%d out of range (HW must be a multiple of 4)", fn, (long long)rows, HW);
    MI355_REQUIRE(aligned(prev_rows, 16) && aligned(dst_rows, 16) && aligned(newest_planes, 4), MI355PPO_EALIGN,
                  "%s: rows must be 16-byte, planes 4-byte aligned", fn);
    const int quads = HW / 4;
    hipLaunchKernelGGL(shift_append_u8_c4_kernel, dim3((unsigned)rows, (unsigned)((quads + 255) / 256)), dim3(256), 0,
                       as_stream(stream), prev_rows, newest_planes, dst_rows, quads);
    return check_launch("shift_append_u8_c4_kernel");
}
