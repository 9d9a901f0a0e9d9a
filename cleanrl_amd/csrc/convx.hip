// Kernel C -- forward of Conv2d(32,64,4,stride 2) and Conv2d(64,64,3,stride 1) of the NatureCNN
// (cleanrl/ppo_atari_multigpu.py:139-142) at minibatch size on the bf16 matrix pipe with EXACT products, gfx950.
//
// The implicit GEMM of kernel F -- out[p][co] = relu(bias[co] + sum_k A[p][k] * B[co][k]) with p an output pixel,
// k = (tap row, tap column, input channel) -- computed the way kernel X (fcx.hip) computes the FC layer: every f32 operand
// is the exact sum of three bf16 terms, the 3 x 3 term products are exact in f32, nine `v_mfma_f32_32x32x16_bf16` (f32
// accumulation) do the work of eight `v_mfma_f32_32x32x2_f32` in 9 x 32 instead of 8 x 64 matrix-pipe cycles per 16 k.
// Same arithmetic class as the f32-MFMA kernel it replaces (exact products, f32 accumulation; sums in another order).
//
// A (the activations of the layer below, channels-last) needs no im2col and no LDS: a lane = an output pixel, and the
// KW * C values of one tap row of its window are contiguous in memory (conv2: 128 floats, conv3: 192), so 8 consecutive k
// are 32 contiguous bytes -- the operand layout of the bf16 MFMA -- at compile-time offsets from the pixel's window origin.
// A fragments are split in registers between the MFMAs (kernel X's pipeline).  B (the weights, rewritten once per optimizer
// step) is split AHEAD by `convx_pack_kernel` into fragment order -- [k-step][co tile][term][lane][8 bf16], one contiguous
// KiB per fragment load -- 196 / 221 KB per layer: too large for the LDS kernel F keeps its f32 matrix in, small enough to
// live in L2, streamed by every wave.
//
// One wave per SIMD (512 registers) owns 128 pixels x 64 channels (4 x 2 tiles = 128 accumulator registers).  Per k-step:
// 8 + 6 loads, 176 VALU (four A fragments split), 72 MFMAs.
#include "common.h"
#include "bf16split.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float c_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int c_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 c_bf16x8 __attribute__((ext_vector_type(8)));

template <int H_, int W_, int C_, int KH_, int KW_, int OH_, int OW_, int S_>
struct CGeom {
    static constexpr int H = H_, W = W_, C = C_, KH = KH_, KW = KW_, OH = OH_, OW = OW_, S = S_;
    static constexpr int K = KH * KW * C, RUN = KW * C, PITCH = W * C, PER_IMG = OH * OW;
    static constexpr int SPR = RUN / 16, NSTEPS = K / 16;          // k-steps per tap row, in all
    static_assert(RUN % 32 == 0, "an even number of k-steps per tap row (the register double buffer alternates per step)");
};
using CGeom2 = CGeom<20, 20, 32, 4, 4, 9, 9, 2>;
using CGeom3 = CGeom<9, 9, 64, 3, 3, 7, 7, 1>;
constexpr int kCCout = 64, kCMT = 4, kCNT = 2;
constexpr int kCStepBytes = kCNT * 3 * 64 * 16;                    // 6 KiB of B terms per k-step

using CTerms = SplitTerms;

// pack[s][j][t][lane][e] (bf16) = term t of W[co = 32 j + (lane & 31)][ci][ty][tx] with k = 16 s + 8 (lane >> 5) + e =
// (ty * KW + tx) * C + ci; W is the Conv2d weight (Cout, C, KH, KW).
template <class G>
__global__ __launch_bounds__(256) void convx_pack_kernel(const float* __restrict__ W, unsigned short* __restrict__ pack) {
    const int idx = blockIdx.x * 256 + threadIdx.x;                 // (s, j, lane, e)
    if (idx >= G::NSTEPS * kCNT * 64 * 8) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, j = (idx >> 9) % kCNT, s = idx / (512 * kCNT);
    const int co = 32 * j + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
    const int tap = k / G::C, ci = k - tap * G::C, ty = tap / G::KW, tx = tap - ty * G::KW;
    const float x = W[((co * G::C + ci) * G::KH + ty) * G::KW + tx];
    const float r = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float l = r - __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    const int o = ((s * kCNT + j) * 3 * 64 + lane) * 8 + e;         // term 0; terms 1, 2 follow at + 512, + 1024
    pack[o] = (unsigned short)(__float_as_uint(x) >> 16);
    pack[o + 512] = (unsigned short)(__float_as_uint(r) >> 16);
    pack[o + 1024] = (unsigned short)(__float_as_uint(l) >> 16);
}

// NP = 9: all 3 x 3 term pairs (exact products); NP = 6: the pairs with x + y <= 2 (common.h: bf16_term_pairs).
template <class G, int NP, int SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void convx_fwd_kernel(
    const float* __restrict__ src, const unsigned char* __restrict__ pack, const float* __restrict__ bias, float* __restrict__ dst,
    long long P, unsigned m8, unsigned m16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const long long p0 = ((long long)blockIdx.x * 4 + wave) * (32 * kCMT);
    if (p0 >= P) return;                                  // (whole wave; no barriers in this kernel)
    // window origins of this lane's four pixels (pixels past the end: the last one, results dropped at the store)
    const float* pa[kCMT];
#pragma unroll
    for (int i = 0; i < kCMT; ++i) {
        long long p = p0 + 32 * i + li;
        p = p < P ? p : P - 1;
        const long long img = p / G::PER_IMG;
        const int rem = (int)(p - img * G::PER_IMG), oy = rem / G::OW, ox = rem - oy * G::OW;
        pa[i] = src + ((img * G::H + oy * G::S) * G::W + ox * G::S) * (long long)G::C + 8 * lh;
    }
    const unsigned char* const pb = pack + 16 * lane;

    c_f32x16 acc[kCMT][kCNT];
#pragma unroll
    for (int i = 0; i < kCMT; ++i)
#pragma unroll
        for (int j = 0; j < kCNT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    c_u32x4 raw[2][kCMT][2];                              // [k-step parity][fragment][16-byte half]: A as f32
    CTerms ta[2][kCMT], tb[2][kCNT];                      // operands of k-step s live in t*[s & 1]
    // k-step (row, u): tap row `row`, floats 16 u .. 16 u + 15 of its KW * C run.  row_off / next_off = element offsets of
    // tap rows `row` and min(row + 1, KH - 1) (past the last row: re-read, never multiplied).
    auto fetchA = [&](int par, int off) {
#pragma unroll
        for (int i = 0; i < kCMT; ++i) {
            raw[par][i][0] = *reinterpret_cast<const c_u32x4*>(pa[i] + off);
            raw[par][i][1] = *reinterpret_cast<const c_u32x4*>(pa[i] + off + 4);
        }
    };
    auto fetchB = [&](int par, int step) {
        const unsigned char* p = pb + (size_t)step * kCStepBytes;
#pragma unroll
        for (int j = 0; j < kCNT; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t) tb[par][j].t[t] = __builtin_bit_cast(c_bf16x8, *reinterpret_cast<const c_u32x4*>(p + (j * 3 + t) * 1024));
    };
    // one k-step u of tap row `row` (q = u & 1 compile-time): MFMAs of (ta[q], tb[q]); between them the split of raw[q ^ 1]
    // (k-step + 1) into ta[q ^ 1]; at its start the loads of A for k-step + 2 into raw[q] and of B for k-step + 1 into tb[q ^ 1].
    auto step = [&](auto uc, int row, int row_off, int next_off) {
        constexpr int u = decltype(uc)::value, q = u & 1;
        constexpr int u2 = (u + 2) % G::SPR, c2 = (u + 2) / G::SPR;
        const int s = row * G::SPR + u;
        // B first: the in-order counter then lets the next step wait for B alone and leaves the A loads in flight for
        // another step
        fetchB(q ^ 1, s + 1 < G::NSTEPS ? s + 1 : G::NSTEPS - 1);
        __builtin_amdgcn_sched_barrier(0);
        fetchA(q, (c2 ? next_off : row_off) + 16 * u2);
#pragma unroll
        for (int i = 0; i < kCMT; ++i) ta[q ^ 1][i] = split8<SPLIT>(raw[q ^ 1][i][0], raw[q ^ 1][i][1], m8, m16);
        // term pairs outermost, the eight independent tiles innermost: no MFMA waits for the one before it
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                if (NP == 6 && x + y > 2) continue;
#pragma unroll
                for (int i = 0; i < kCMT; ++i)
#pragma unroll
                    for (int j = 0; j < kCNT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[q][i].t[x], tb[q][j].t[y], acc[i][j], 0, 0, 0);
            }
        // issue order: the 14 loads first, then NP x 8 x (1 MFMA, 3 or 4 VALU): the 176 VALU of the splits spread over the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * kCMT + 3 * kCNT, 0);
#pragma unroll
        for (int g = 0; g < NP * kCMT * kCNT; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NP == 9 ? 3 : 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: k-step 0 split into ta[0] with its B terms in tb[0]; A of k-step 1 in flight in raw[1]
    fetchA(0, 0);
    fetchB(0, 0);
#pragma unroll
    for (int i = 0; i < kCMT; ++i) ta[0][i] = split8<SPLIT>(raw[0][i][0], raw[0][i][1], m8, m16);
    fetchA(1, 16);
    __builtin_amdgcn_sched_barrier(0);
    for (int row = 0; row < G::KH; ++row) {
        const int row_off = row * G::PITCH, next_off = (row + 1 < G::KH ? row + 1 : row) * G::PITCH;
        [&]<int... U>(std::integer_sequence<int, U...>) {
            (step(std::integral_constant<int, U>{}, row, row_off, next_off), ...);
        }(std::make_integer_sequence<int, G::SPR>{});
    }

    // epilogue: accumulator element e of tile (i, j) is out[p0 + 32 i + (e & 3) + 8 (e >> 2) + 4 lh][32 j + li]
    float bj[kCNT];
#pragma unroll
    for (int j = 0; j < kCNT; ++j) bj[j] = bias[32 * j + li];
#pragma unroll
    for (int i = 0; i < kCMT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long long p = p0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (p < P) {
                float* row = dst + p * kCCout + li;
#pragma unroll
                for (int j = 0; j < kCNT; ++j) {
                    const float v = acc[i][j][e] + bj[j];
                    row[32 * j] = v > 0.0f ? v : 0.0f;
                }
            }
        }
}

template <class G>
static int convx_launch(const float* src, const void* pack, const float* bias, float* dst, long long images, hipStream_t s) {
    const long long P = images * G::PER_IMG, tiles = (P + 32 * kCMT - 1) / (32 * kCMT);
#define CONVX_FWD(NP, SP)                                                                                                   \
    hipLaunchKernelGGL((convx_fwd_kernel<G, NP, SP>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, src,                 \
                       static_cast<const unsigned char*>(pack), bias, dst, P, 0xffff0000u, 0xffffff00u)
    const int np = bf16_term_pairs(), sp = bf16_split_mode();
    if (np == 9 && sp == 0) CONVX_FWD(9, 0);
    else if (np == 9) CONVX_FWD(9, 1);
    else if (sp == 0) CONVX_FWD(6, 0);
    else CONVX_FWD(6, 1);
#undef CONVX_FWD
    return check_launch("convx_fwd_kernel");
}

// entry points used by conv.hip's dispatch (repack mode 6, forward variant 7)
size_t convx_pack_bytes(int layer) {
    return layer == 2 ? (size_t)CGeom2::NSTEPS * kCStepBytes : layer == 3 ? (size_t)CGeom3::NSTEPS * kCStepBytes : 0;
}

int convx_pack(const float* W, void* pack, int layer, hipStream_t s) {
    if (layer == 2)
        hipLaunchKernelGGL((convx_pack_kernel<CGeom2>), dim3((CGeom2::NSTEPS * kCNT * 512 + 255) / 256), dim3(256), 0, s, W, static_cast<unsigned short*>(pack));
    else
        hipLaunchKernelGGL((convx_pack_kernel<CGeom3>), dim3((CGeom3::NSTEPS * kCNT * 512 + 255) / 256), dim3(256), 0, s, W, static_cast<unsigned short*>(pack));
    return check_launch("convx_pack_kernel");
}

int convx_fwd(const float* src, const void* pack, const float* bias, float* dst, long long images, int layer, hipStream_t s) {
    return layer == 2 ? convx_launch<CGeom2>(src, pack, bias, dst, images, s) : convx_launch<CGeom3>(src, pack, bias, dst, images, s);
}

}  // namespace mi355ppo
