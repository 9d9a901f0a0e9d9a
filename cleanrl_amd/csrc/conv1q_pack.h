// Kernel Q's integer-digit pack of Conv2d(4,32,8,s4).weight (conv1q.hip), as a workgroup-wide device function: its own launch
// (mi355ppo_cnn_conv1q_pack) and one block of the all-packs launch after an optimizer step (gemmz.hip) run the same code.
#pragma once
#include "common.h"

namespace mi355ppo {

constexpr int kQRows = 8, kQDigits = 4;
constexpr int kQDigBytes = kQRows * kQDigits * 64 * 16;          // int8 digits in operand layout: [row][digit][lane][16]
constexpr int kQAccOff = kQDigBytes;                             // int32 [digit][channel]: 128 * sum_k digit
constexpr int kQScaleOff = kQAccOff + kQDigits * 32 * 4;         // f32 [channel]: 2^(E_n - 6) / 255
constexpr int kQPackBytes = kQScaleOff + 32 * 4;                 // 33,408
constexpr int kQH = 84, kQW = 84, kQC = 4, kQG = 20, kQPitch = kQW * kQC, kQPerImg = kQG * kQG;
constexpr unsigned kQOob = 0xFFFFF000u;
constexpr int kQRsrcWord3 = 0x00020000;

// One workgroup: W (32,4,8,8) f32 as torch stores it -> the pack.
__device__ __forceinline__ void conv1q_pack_body(const float* __restrict__ W, unsigned char* __restrict__ pack) {
    __shared__ int s_E[32];
    __shared__ float s_max[32][8];
    __shared__ int s_sum[kQDigits][32][8];
    const int tid = threadIdx.x;
    {   // per-channel largest magnitude: thread (n, r) scans its 32 weights, 8 partial maxima per channel
        const int n = tid >> 3, r = tid & 7;
        float m = 0.0f;
        for (int e32 = 0; e32 < 32; ++e32) m = fmaxf(m, fabsf(W[((n * kQC + (e32 & 3)) * 8 + r) * 8 + (e32 >> 2)]));
        s_max[n][r] = m;
    }
    __syncthreads();
    if (tid < 32) {
        float m = 0.0f;
        for (int r = 0; r < 8; ++r) m = fmaxf(m, s_max[tid][r]);
        int E = 0;
        if (m > 0.0f) {
            (void)frexpf(m, &E);                 // m = f * 2^E, f in [0.5, 1)
        }
        s_E[tid] = E;
    }
    __syncthreads();
    // thread -> (channel n, tap row r): the 32 weights W[n][c][r][kw] of one operand row pair
    const int n = tid >> 3, r = tid & 7;
    const int E = s_E[n];
    int sums[kQDigits] = {0, 0, 0, 0};
    for (int e32 = 0; e32 < 32; ++e32) {         // byte e32 of the pixel's tap row: column kw = e32 / 4, channel c = e32 % 4
        const int kw = e32 >> 2, c = e32 & 3;
        const float w = W[((n * kQC + c) * 8 + r) * 8 + kw];
        long long q = llrint(ldexp((double)w, 30 - E));
        int dig[kQDigits];
        for (int d = kQDigits - 1; d >= 0; --d) {
            const int lo = (int)(((q + 128) & 255) - 128);
            dig[d] = lo;
            q = (q - lo) >> 8;
        }
        const int lh = e32 >> 4, e = e32 & 15;
        for (int d = 0; d < kQDigits; ++d) {
            pack[((r * kQDigits + d) * 64 + lh * 32 + n) * 16 + e] = (unsigned char)(signed char)dig[d];
            sums[d] += dig[d];
        }
    }
    for (int d = 0; d < kQDigits; ++d) s_sum[d][n][r] = sums[d];
    __syncthreads();
    if (tid < kQDigits * 32) {
        const int d = tid >> 5, nn = tid & 31;
        int s = 0;
        for (int rr = 0; rr < 8; ++rr) s += s_sum[d][nn][rr];
        reinterpret_cast<int*>(pack + kQAccOff)[d * 32 + nn] = 128 * s;
    }
    if (tid < 32) reinterpret_cast<float*>(pack + kQScaleOff)[tid] = (float)(ldexp(1.0, s_E[tid] - 6) / 255.0);
}

}  // namespace mi355ppo
