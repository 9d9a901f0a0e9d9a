// Kernel X -- Linear(3136, 512) of the NatureCNN (cleanrl/ppo_atari_multigpu.py:144-145) on the bf16 matrix pipe with EXACT
// products: C[m][n] = sum_k A[m][k] * B[n][k], f32 operands, both K-contiguous.
//
// An f32 number is the exact sum of three bf16 terms (hi = its top 8 significand bits, mid = the next 8, lo = the last 8),
// so a[m][k] * b[n][k] = sum over the 3 x 3 term pairs of products that are each exact in f32 (8 x 8 significand bits).
// Nine `v_mfma_f32_32x32x16_bf16` (f32 accumulation) therefore compute what eight `v_mfma_f32_32x32x2_f32` compute --
// exact products, f32 accumulation -- in 9 x 32 instead of 8 x 64 matrix-pipe cycles per 16 k.  No reduced precision.
//
// No LDS: both operands are already in MFMA operand layout as they lie in memory (a lane = one row, 8 consecutive k = 32
// contiguous bytes = two 16-byte loads), the scheme of kernel F's A operand; the fragments are split in registers
// (4 VALU + 1.5 v_perm per value).  Tiling and pipeline: see the kernel.
//
// Epilogues: EPI_BIAS_RELU (forward: h = relu(a @ Wp^T + b)) and EPI_MASK (data gradient: da = (dz @ Wp) * (a > 0) -- the
// ReLU backward of the layer BELOW, conv3, applied where the gradient is produced, as the conv data-gradient kernels do;
// the reference runs it as a separate pass over the 411 MB tensor).
#include "common.h"
#include "bf16split.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float x_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int x_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 x_bf16x8 __attribute__((ext_vector_type(8)));

enum { X_BIAS_RELU = 0, X_MASK = 1 };

using XTerms = SplitTerms;

// A (M, K) row-major with leading dimension lda, B (N, K) with ldb, C (M, N) with ldc; K % 16 == 0.
// bias (N) for X_BIAS_RELU; cmask (M, N) with ldc for X_MASK.
// One wave per SIMD (512 registers): a wave owns 64 x 128 of C (2 x 4 tiles, 128 accumulator registers); the four waves of a
// workgroup sit on top of each other (256 x 128) and share the B rows through L1.  Per 16 k: 6 fragments (12 loads), 264 VALU
// for the splits, 72 MFMAs.  The loop is software-pipelined by hand -- loads two stages ahead, the split of stage s+1 issued
// BETWEEN the MFMAs of stage s (one MFMA : four VALU, `sched_group_barrier`), because a wave that first splits and then
// multiplies leaves the matrix pipe idle during the split (measured: 50 % of the pipe at two such waves per SIMD).
constexpr int kXMT = 2, kXNT = 4, kXF = kXMT + kXNT;

// WAVES_N: the four waves of a workgroup sit side by side (64 x 512: they share the A rows, each A row leaves HBM once --
// the forward, where A is the 411 MB activation and B the 6.4 MB weight; measured 3.6 x the algorithmic bytes with the
// waves stacked) instead of on top of each other (256 x 128: they share the B rows).
// NP = 9: all 3 x 3 term pairs (exact products); NP = 6: the pairs with ta + tb <= 2 (common.h: bf16_term_pairs).
template <int EPI, bool WAVES_N, int NP, int SPLIT, bool COAL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fcx_gemm_nt_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, const float* __restrict__ bias,
    const float* __restrict__ cmask, float* __restrict__ C, int ldc, int M, int N, int K, unsigned m8, unsigned m16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = WAVES_N ? blockIdx.y * 64 : blockIdx.y * 256 + wave * 64;
    const int n0 = WAVES_N ? blockIdx.x * 512 + wave * 128 : blockIdx.x * 128;
    if (m0 >= M || n0 >= N) return;                       // (whole wave; no barriers in this kernel)
    // row pointers of this lane's six fragments (rows past the edge re-read the last row: results dropped at the store)
    const float* pf[kXF];
#pragma unroll
    for (int i = 0; i < kXMT; ++i) {
        const int r = m0 + 32 * i + li;
        pf[i] = A + (size_t)(r < M ? r : M - 1) * lda + 8 * lh;
    }
#pragma unroll
    for (int j = 0; j < kXNT; ++j) {
        const int r = n0 + 32 * j + li;
        pf[kXMT + j] = B + (size_t)(r < N ? r : N - 1) * ldb + 8 * lh;
    }
    if (COAL) {
        // TIMING EXPERIMENT ONLY (MI355PPO_FCX_COAL=1, results are garbage): the same number of 16-byte loads and the same
        // bytes per k-step, but four consecutive lanes read one row's 64 contiguous bytes -- what a coalesced load + LDS
        // transposition would ask of the memory pipeline.  Tests whether the per-lane row gather (every lane its own cache
        // line) is what bounds the kernel.
#pragma unroll
        for (int i = 0; i < kXMT; ++i) {
            const int r = m0 + 32 * i + (lane >> 2);
            pf[i] = A + (size_t)(r < M ? r : M - 1) * lda + 4 * (lane & 3);
        }
#pragma unroll
        for (int j = 0; j < kXNT; ++j) {
            const int r = n0 + 32 * j + (lane >> 2);
            pf[kXMT + j] = B + (size_t)(r < N ? r : N - 1) * ldb + 4 * (lane & 3);
        }
    }
    x_f32x16 acc[kXMT][kXNT];
#pragma unroll
    for (int i = 0; i < kXMT; ++i)
#pragma unroll
        for (int j = 0; j < kXNT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    x_u32x4 raw[2][kXF][2];                               // [stage parity][fragment][16-byte half]
    XTerms pc[2][kXF];                                    // split fragments of stage s live in pc[s & 1]
    auto fetch = [&](int st, int k0) {
#pragma unroll
        for (int q = 0; q < kXF; ++q) {
            raw[st][q][0] = *reinterpret_cast<const x_u32x4*>(pf[q] + k0);
            raw[st][q][1] = *reinterpret_cast<const x_u32x4*>(pf[q] + k0 + (COAL ? 16 * (size_t)(q < kXMT ? lda : ldb) : (size_t)4));
        }
    };
    const int nsteps = K >> 4;
    auto kof = [&](int s) { return (s < nsteps ? s : nsteps - 1) << 4; };       // past the end: re-read, never multiplied
    // one pipeline step for a stage of parity q: MFMAs of pc[q] (stage s), split raw[q ^ 1] (stage s+1) into pc[q ^ 1], fetch
    // stage s+2 into raw[q] (free: its stage was split one step ago).  Fixed parity = compile-time register indices.
    auto step = [&](auto qc, int s) {
        constexpr int q = decltype(qc)::value;
        fetch(q, kof(s + 2));
#pragma unroll
        for (int f = 0; f < kXF; ++f) pc[q ^ 1][f] = split8<SPLIT>(raw[q ^ 1][f][0], raw[q ^ 1][f][1], m8, m16);
        // term pairs outermost, the eight independent tiles innermost: no MFMA waits for the one before it
#pragma unroll
        for (int ta = 0; ta < 3; ++ta)
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) {
                if (NP == 6 && ta + tb > 2) continue;
#pragma unroll
                for (int i = 0; i < kXMT; ++i)
#pragma unroll
                    for (int j = 0; j < kXNT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pc[q][i].t[ta], pc[q][kXMT + j].t[tb], acc[i][j], 0, 0, 0);
            }
        // issue order: the 12 loads first, then NP x 8 x (1 MFMA, 4 or 6 VALU): the 264 VALU of the splits spread over the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * kXF, 0);
#pragma unroll
        for (int g = 0; g < NP * kXMT * kXNT; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NP == 9 ? 4 : 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: stage 0 split into pc[0], stage 1 raw in flight in raw[1]
    fetch(0, 0);
#pragma unroll
    for (int f = 0; f < kXF; ++f) pc[0][f] = split8<SPLIT>(raw[0][f][0], raw[0][f][1], m8, m16);
    fetch(1, kof(1));
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    for (; s + 2 <= nsteps; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s + 1);
    }
    if (s < nsteps) step(std::integral_constant<int, 0>{}, s);

    // ---- epilogue: accumulator element e of tile (i, j) is C[m0 + 32 i + (e & 3) + 8 (e >> 2) + 4 lh][n0 + 32 j + li].
    // Offsets are row * ldc + n with the row part shared by the four column tiles.  X_MASK: ALL 128 mask values of the
    // wave's block are requested before the first one is used (one wave per SIMD: nothing else hides a load's latency;
    // fetched tile by tile the epilogue cost 8 dependent round trips per 35 us of MFMAs).
    const bool wave_rows_ok = m0 + 64 <= M;
    if (EPI == X_MASK) {
        float mk[kXMT][kXNT][16];
#pragma unroll
        for (int i = 0; i < kXMT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float* row = cmask + (size_t)(m < M ? m : M - 1) * ldc;
#pragma unroll
                for (int j = 0; j < kXNT; ++j) {
                    const int n = n0 + 32 * j + li;
                    mk[i][j][e] = row[n < N ? n : N - 1];
                }
            }
#pragma unroll
        for (int i = 0; i < kXMT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float* row = C + (size_t)m * ldc;
#pragma unroll
                for (int j = 0; j < kXNT; ++j) {
                    const int n = n0 + 32 * j + li;
                    if ((wave_rows_ok || m < M) && n < N) row[n] = mk[i][j][e] > 0.0f ? acc[i][j][e] : 0.0f;
                }
            }
    } else {
        float bj[kXNT];
#pragma unroll
        for (int j = 0; j < kXNT; ++j) {
            const int n = n0 + 32 * j + li;
            bj[j] = bias[n < N ? n : N - 1];
        }
#pragma unroll
        for (int i = 0; i < kXMT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float* row = C + (size_t)m * ldc;
#pragma unroll
                for (int j = 0; j < kXNT; ++j) {
                    const int n = n0 + 32 * j + li;
                    float v = acc[i][j][e] + bj[j];
                    v = v > 0.0f ? v : 0.0f;
                    if ((wave_rows_ok || m < M) && n < N) row[n] = v;
                }
            }
    }
}

}  // namespace mi355ppo

using namespace mi355ppo;

static int fcx_check(const char* fn, const float* A, const float* B, const float* C, int M, int N, int K, int lda, int ldb, int ldc) {
    MI355_REQUIRE(A && B && C, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0, MI355PPO_EINVAL, "%s: M=%d N=%d K=%d (K must be a positive multiple of 16)", fn, M, N, K);
    MI355_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 4 == 0 && ldb % 4 == 0, MI355PPO_EINVAL,
                  "%s: leading dimensions lda=%d ldb=%d ldc=%d (lda, ldb: multiples of 4, >= K; ldc >= N)", fn, lda, ldb, ldc);
    MI355_REQUIRE(aligned(A, 16) && aligned(B, 16) && aligned(C, 4), MI355PPO_EALIGN, "%s: A and B must be 16-byte aligned", fn);
    MI355_REQUIRE((M + 63) / 64 <= 65535 * 64, MI355PPO_EINVAL, "%s: M=%d exceeds one launch", fn, M);
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_fc_fwd_relu_f32(const float* a, const float* W, const float* bias, float* h, int M, int N, int K,
                                                     void* stream) {
    const char* fn = "mi355ppo_fc_fwd_relu_f32";
    int rc = fcx_check(fn, a, W, h, M, N, K, K, K, N);
    if (rc) return rc;
    MI355_REQUIRE(bias && aligned(bias, 4), MI355PPO_EINVAL, "%s: bias missing or misaligned", fn);
    MI355_REQUIRE((M + 63) / 64 <= 65535, MI355PPO_EINVAL, "%s: M=%d exceeds one launch", fn, M);
#define FCX_FWD(NP, SP)                                                                                                       \
    hipLaunchKernelGGL((fcx_gemm_nt_kernel<X_BIAS_RELU, true, NP, SP>), dim3((N + 511) / 512, (M + 63) / 64), dim3(256), 0,    \
                       as_stream(stream), a, K, W, K, bias, (const float*)nullptr, h, N, M, N, K, 0xffff0000u, 0xffffff00u)
    const int np = bf16_term_pairs(), sp = bf16_split_mode();
    static const bool coal = getenv("MI355PPO_FCX_COAL") != nullptr;      // timing experiment, garbage results
    if (coal)
        hipLaunchKernelGGL((fcx_gemm_nt_kernel<X_BIAS_RELU, true, 6, 1, true>), dim3((N + 511) / 512, (M + 63) / 64), dim3(256), 0,
                           as_stream(stream), a, K, W, K, bias, (const float*)nullptr, h, N, M, N, K, 0xffff0000u, 0xffffff00u);
    else if (np == 9 && sp == 0) FCX_FWD(9, 0);
    else if (np == 9) FCX_FWD(9, 1);
    else if (sp == 0) FCX_FWD(6, 0);
    else FCX_FWD(6, 1);
#undef FCX_FWD
    return check_launch(fn);
}

extern "C" MI355PPO_API int mi355ppo_fc_dgrad_mask_f32(const float* dz, int lddz, const float* Wt, int ldwt, const float* act_in,
                                                       float* da, int M, int N, int K, void* stream) {
    const char* fn = "mi355ppo_fc_dgrad_mask_f32";
    int rc = fcx_check(fn, dz, Wt, da, M, N, K, lddz, ldwt, N);
    if (rc) return rc;
    MI355_REQUIRE(act_in && aligned(act_in, 4) && act_in != da, MI355PPO_EINVAL, "%s: act_in missing, misaligned or aliased with da", fn);
#define FCX_DGRAD(NP, SP)                                                                                                     \
    hipLaunchKernelGGL((fcx_gemm_nt_kernel<X_MASK, false, NP, SP>), dim3((N + 127) / 128, (M + 255) / 256), dim3(256), 0,      \
                       as_stream(stream), dz, lddz, Wt, ldwt, (const float*)nullptr, act_in, da, N, M, N, K, 0xffff0000u, 0xffffff00u)
    const int np = bf16_term_pairs(), sp = bf16_split_mode();
    if (np == 9 && sp == 0) FCX_DGRAD(9, 0);
    else if (np == 9) FCX_DGRAD(9, 1);
    else if (sp == 0) FCX_DGRAD(6, 0);
    else FCX_DGRAD(6, 1);
#undef FCX_DGRAD
    return check_launch(fn);
}
