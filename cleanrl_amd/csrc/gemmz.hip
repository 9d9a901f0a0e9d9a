// Kernel Z -- the exact-product bf16 GEMM of round 3 (replaces kernel X for the FC layer of the NatureCNN,
// cleanrl/ppo_atari_multigpu.py:144-145):  C[m][n] = sum_k A[m][k] * B[n][k],  A (M, K) f32 row-major (the activations or the
// incoming gradient), B (N, K) the weight matrix.
//
// What bounded kernel X (profiles/r03_pmc_*.csv, profiles/r03_conv_traffic_pairs_ab.jsonl): with "lane = row" fragment loads
// every lane of a 16-byte load touches its own cache line, and the vector-memory front end (TA) processes about one line per
// clock -- 48 such loads per k-step and CU kept it 83 % busy and the matrix pipe 37 % busy; the 264 split instructions per
// k-step did the rest.  Hence:
//   * B is split AHEAD, once per optimizer step, into MFMA fragment order (`zpack_kernel`: [k-step][32-column tile][term]
//     [lane][8 bf16] = one contiguous KiB per fragment load): no VALU for B in the GEMM, fully coalesced loads;
//   * A is loaded COALESCED -- four consecutive lanes read the 64 contiguous bytes (16 k) of one row, a wave instruction
//     covers 16 rows -- written to a wave-private LDS tile (row pitch 80 bytes: conflict-free for the 16-byte writes and for
//     the 16-byte "lane = row" reads), read back as fragments and split in registers (44 VALU per fragment).  Wave-private:
//     no workgroup barrier anywhere; LDS operations of one wave execute in order.
// Arithmetic: every f32 is the exact sum of three bf16 terms (bf16split.h); the six term pairs (i, j), i + j <= 2, are
// multiplied on `v_mfma_f32_32x32x16_bf16` with f32 accumulation; the three dropped pairs are below the rounding of one f32
// multiply.  MI355PPO_BF16_PAIRS=9 multiplies all nine (exact products).
//
// One wave per SIMD owns 64 x 128 of C (2 x 4 tiles, 128 accumulator registers).  Per k-step (16 k): 4 + 12 global loads, 4
// LDS writes, 4 LDS reads, 88 VALU, 48 MFMAs.
#include "common.h"
#include "bf16split.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float z_f32x16 __attribute__((ext_vector_type(16)));

enum { Z_BIAS_RELU = 0, Z_MASK = 1 };
constexpr int kZMT = 2, kZNT = 4;                     // 32-row / 32-column tiles per wave
constexpr int kZRows = 32 * kZMT;                     // A rows per wave
constexpr int kZLoads = kZRows / 16;                  // coalesced 16-byte loads per lane and k-step (16 rows per wave instruction)
constexpr int kZPitch = 20;                           // floats per LDS row: 16 k + 4 pad (80 bytes)
constexpr int kZTileBytes = 3 * 64 * 16;              // one 32-column tile of one k-step in the pack: 3 terms x 1 KiB

// pack[s][j][t][lane][e] (bf16) = term t of B[n = 32 j + (lane & 31)][k = 16 s + 8 (lane >> 5) + e]; rows n >= N are zero.
__global__ __launch_bounds__(256) void zpack_kernel(const float* __restrict__ B, int ldb, int N, int K, unsigned short* __restrict__ pack) {
    const int ntiles = (N + 31) / 32;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;                 // (s, j, lane, e)
    if (idx >= (long long)(K / 16) * ntiles * 512) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long long sj = idx >> 9;
    const int j = (int)(sj % ntiles), s = (int)(sj / ntiles);
    const int n = 32 * j + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
    const float x = n < N ? B[(size_t)n * ldb + k] : 0.0f;
    const unsigned xb = __float_as_uint(x);
    const unsigned t8 = xb & 0xffff0000u, t16 = xb & 0xffffff00u;
    const float mid = __uint_as_float(t16) - __uint_as_float(t8), lo = x - __uint_as_float(t16);
    const size_t o = ((size_t)sj * 3 * 64 + lane) * 8 + e;                           // term 0; terms 1, 2 follow at + 512, + 1024
    pack[o] = (unsigned short)(xb >> 16);
    pack[o + 512] = (unsigned short)(__float_as_uint(mid) >> 16);
    pack[o + 1024] = (unsigned short)(__float_as_uint(lo) >> 16);
}

// Item t of `items` sits behind MFMA number ((t + 1) * span) / items - 1 of a k-step (distinct slots for span >= items).
constexpr int z_item_at(int g, int items, int span) {
    for (int t = 0; t < items; ++t)
        if (((t + 1) * span) / items - 1 == g) return t;
    return -1;
}

// WAVES_N: the four waves of a workgroup sit side by side (64 x 512 of C: they read the same A rows -- the forward, where A is
// the 411 MB activation) or on top of each other (256 x 128: they stream the same B fragments -- the data gradient).
template <int EPI, bool WAVES_N, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void zgemm_kernel(
    const float* __restrict__ A, int lda, const unsigned char* __restrict__ pack, const float* __restrict__ bias,
    const float* __restrict__ cmask, float* __restrict__ C, int ldc, int M, int N, int K, unsigned m8, unsigned m16) {
    __shared__ __attribute__((aligned(16))) float lds[4 * kZRows * kZPitch];      // 4 waves x 64 rows x 80 bytes = 20 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = WAVES_N ? blockIdx.y * kZRows : (blockIdx.y * 4 + wave) * kZRows;
    const int n0 = WAVES_N ? (blockIdx.x * 4 + wave) * (32 * kZNT) : blockIdx.x * (32 * kZNT);
    if (m0 >= M || n0 >= N) return;                       // (whole wave; no barriers in this kernel)
    const int ntiles = (N + 31) / 32, j0 = n0 / 32;
    float* const wl = lds + wave * (kZRows * kZPitch);
    // coalesced A loads: load u of a k-step reads row m0 + 16 u + (lane >> 2), floats 4 (lane & 3) .. + 3 of the step's 16
    // (rows past M re-read the last row: their results are dropped at the store)
    unsigned voff[kZLoads];
#pragma unroll
    for (int u = 0; u < kZLoads; ++u) {
        const int r = m0 + 16 * u + (lane >> 2);
        voff[u] = (unsigned)(r < M ? r : M - 1) * (unsigned)lda * 4u + 16u * (unsigned)(lane & 3);      // bytes; A < 4 GiB (host-checked)
    }
    const unsigned char* const Ab = reinterpret_cast<const unsigned char*>(A);
    float* const wr_ptr = wl + (lane >> 2) * kZPitch + 4 * (lane & 3);               // + 16 u rows
    const float* const rd_ptr = wl + li * kZPitch + 8 * lh;                          // + 32 i rows
    const unsigned char* const pb = pack + (size_t)j0 * kZTileBytes + 16 * lane;
    const size_t step_bytes = (size_t)ntiles * kZTileBytes;

    z_f32x16 acc[kZMT][kZNT];
#pragma unroll
    for (int i = 0; i < kZMT; ++i)
#pragma unroll
        for (int j = 0; j < kZNT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    s_u32x4 stage[kZLoads];                               // A of a later k-step, as loaded (coalesced layout), on its way to LDS
    s_u32x4 raw[kZMT][2];                                 // fragments of the NEXT k-step as read back from LDS (f32, lane = row)
    unsigned ta[2][kZMT][3][4];                           // split A fragments: [k-step parity][fragment][term][4 x 2 bf16]
    s_u32x4 tb[2][kZNT][3];                               // B fragments straight from the pack
    const int nsteps = K >> 4;
    auto kclamp = [&](int s) { return s < nsteps ? s : nsteps - 1; };                // past the end: re-read, never multiplied
    auto load_a = [&](int s) {
        const unsigned char* base = Ab + (size_t)kclamp(s) * 64;
#pragma unroll
        for (int u = 0; u < kZLoads; ++u) stage[u] = *reinterpret_cast<const s_u32x4*>(base + voff[u]);
    };
    auto load_b = [&](int par, int s) {
        const unsigned char* p = pb + (size_t)kclamp(s) * step_bytes;
#pragma unroll
        for (int j = 0; j < kZNT; ++j) {
            const bool ok = j0 + j < ntiles;                                        // (wave-uniform) tiles past N: re-read tile j0
#pragma unroll
            for (int t = 0; t < 3; ++t) tb[par][j][t] = *reinterpret_cast<const s_u32x4*>(p + (ok ? j : 0) * kZTileBytes + t * 1024);
        }
    };
    auto to_lds = [&]() {
#pragma unroll
        for (int u = 0; u < kZLoads; ++u) *reinterpret_cast<s_u32x4*>(wr_ptr + 16 * u * kZPitch) = stage[u];
    };
    auto read_frags = [&]() {
#pragma unroll
        for (int i = 0; i < kZMT; ++i) {
            raw[i][0] = *reinterpret_cast<const s_u32x4*>(rd_ptr + 32 * i * kZPitch);
            raw[i][1] = *reinterpret_cast<const s_u32x4*>(rd_ptr + 32 * i * kZPitch + 4);
        }
    };
    // The split of one half fragment (4 elements = raw[i][hf]) in three pieces of 8 + 8 + 6 VALU instructions (bf16split.h,
    // fixed-position split): piece 0 the masks, piece 1 the two subtractions, piece 2 the packs into ta[par][i].
    unsigned t8[4], t16[4];
    float smid[4], slo[4];
    auto split_piece = [&](int par, auto ic, auto hc, auto pc) {
        constexpr int i = decltype(ic)::value, hf = decltype(hc)::value, piece = decltype(pc)::value;
        const s_u32x4 x = raw[i][hf];
        if constexpr (piece == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t8[j] = x[j] & m8;
                t16[j] = x[j] & m16;
            }
        } else if constexpr (piece == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                smid[j] = __uint_as_float(t16[j]) - __uint_as_float(t8[j]);
                slo[j] = __uint_as_float(x[j]) - __uint_as_float(t16[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                ta[par][i][0][2 * hf + (j >> 1)] = __builtin_amdgcn_perm(x[j + 1], x[j], 0x07060302u);
                ta[par][i][1][2 * hf + (j >> 1)] = split_pack(smid[j + 1], smid[j]);
                ta[par][i][2][2 * hf + (j >> 1)] = split_pack(slo[j + 1], slo[j]);
            }
        }
    };
    // One pipeline step of parity q.  Nothing a step computes depends on a load or LDS access of the SAME step:
    //   B of step s + 1 (global)                      -> tb[q ^ 1]
    //   `raw` = fragments of step s + 1 (read from LDS at the end of the previous step) -> split -> ta[q ^ 1]
    //   the MFMAs of step s on (ta[q], tb[q])
    //   `stage` = A of step s + 2 (loaded during the previous step) -> LDS;  A of step s + 3 (global) -> `stage`;
    //   fragments of step s + 2: LDS -> `raw`
    // The instruction order is pinned by hand (hipcc's sched_group_barrier solver did not reproduce this pipeline: it left the
    // split in front of the MFMAs and chained MFMAs on one accumulator): after MFMA g of the step's NM comes item
    // `item_after(g)` -- one of the 6 * kZMT split pieces, then the LDS writes, the A loads, the LDS reads -- behind a
    // sched_barrier; the last kTail MFMAs run bare and cover the LDS round trip.  Term pairs outermost, the eight independent
    // tiles innermost: no MFMA waits for the one before it.
    constexpr int NM = NP * kZMT * kZNT, kPieces = 6 * kZMT, kItems = kPieces + 3, kTail = 12;
    constexpr int PX[9] = {0, 0, 1, 0, 2, 1, 1, 2, 2}, PY[9] = {0, 1, 0, 2, 0, 1, 2, 1, 2};     // pairs by weight: the first six have x + y <= 2
    auto step = [&](auto qc, int s) {
        constexpr int q = decltype(qc)::value;
        load_b(q ^ 1, s + 1);
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int g = G, pi = g / (kZMT * kZNT), i = (g / kZNT) % kZMT, j = g % kZNT;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(s_bf16x8, (s_u32x4){ta[q][i][PX[pi]][0], ta[q][i][PX[pi]][1], ta[q][i][PX[pi]][2], ta[q][i][PX[pi]][3]}),
                    __builtin_bit_cast(s_bf16x8, tb[q][j][PY[pi]]), acc[i][j], 0, 0, 0);
                constexpr int t = z_item_at(g, kItems, NM - kTail);                 // the item behind MFMA g, or -1
                if constexpr (t >= 0) {
                    if constexpr (t < kPieces)
                        split_piece(q ^ 1, std::integral_constant<int, t / 6>{}, std::integral_constant<int, (t / 3) % 2>{}, std::integral_constant<int, t % 3>{});
                    else if constexpr (t == kPieces) to_lds();
                    else if constexpr (t == kPieces + 1) load_a(s + 3);
                    else read_frags();
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, NM>{});
    };
    auto split_all = [&](int par) {
        [&]<int... T>(std::integer_sequence<int, T...>) {
            (split_piece(par, std::integral_constant<int, T / 6>{}, std::integral_constant<int, (T / 3) % 2>{}, std::integral_constant<int, T % 3>{}), ...);
        }(std::make_integer_sequence<int, kPieces>{});
    };
    // prologue: step 0 split into ta[0] with its B terms in tb[0]; fragments of step 1 in `raw`; A of step 2 in `stage`
    load_a(0);
    load_b(0, 0);
    to_lds();
    load_a(1);
    read_frags();
    split_all(0);
    __builtin_amdgcn_sched_barrier(0);
    to_lds();
    load_a(2);
    read_frags();
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    for (; s + 2 <= nsteps; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s + 1);
    }
    if (s < nsteps) step(std::integral_constant<int, 0>{}, s);

    // ---- epilogue: accumulator element e of tile (i, j) is C[m0 + 32 i + (e & 3) + 8 (e >> 2) + 4 lh][n0 + 32 j + li]
    const bool wave_rows_ok = m0 + kZRows <= M;
    if (EPI == Z_MASK) {
        // all mask values of the wave's block are requested before the first one is used (one wave per SIMD: nothing else hides
        // a load's latency)
        float mk[kZMT][kZNT][16];
#pragma unroll
        for (int i = 0; i < kZMT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float* row = cmask + (size_t)(m < M ? m : M - 1) * ldc;
#pragma unroll
                for (int j = 0; j < kZNT; ++j) {
                    const int n = n0 + 32 * j + li;
                    mk[i][j][e] = row[n < N ? n : N - 1];
                }
            }
#pragma unroll
        for (int i = 0; i < kZMT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float* row = C + (size_t)m * ldc;
#pragma unroll
                for (int j = 0; j < kZNT; ++j) {
                    const int n = n0 + 32 * j + li;
                    if ((wave_rows_ok || m < M) && n < N) row[n] = mk[i][j][e] > 0.0f ? acc[i][j][e] : 0.0f;
                }
            }
    } else {
        float bj[kZNT];
#pragma unroll
        for (int j = 0; j < kZNT; ++j) {
            const int n = n0 + 32 * j + li;
            bj[j] = bias[n < N ? n : N - 1];
        }
#pragma unroll
        for (int i = 0; i < kZMT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float* row = C + (size_t)m * ldc;
#pragma unroll
                for (int j = 0; j < kZNT; ++j) {
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

static size_t zpack_bytes(int N, int K) { return (size_t)(K / 16) * (size_t)((N + 31) / 32) * kZTileBytes; }

extern "C" MI355PPO_API size_t mi355ppo_fc_pack_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || K % 16) return 0;
    return zpack_bytes(N, K);
}

extern "C" MI355PPO_API int mi355ppo_fc_pack_f32(const float* B, int ldb, int N, int K, void* pack, void* stream) {
    const char* fn = "mi355ppo_fc_pack_f32";
    MI355_REQUIRE(B && pack, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(N > 0 && K > 0 && K % 16 == 0 && ldb >= K, MI355PPO_EINVAL, "%s: N=%d K=%d ldb=%d (K a positive multiple of 16, ldb >= K)", fn, N, K, ldb);
    MI355_REQUIRE(aligned(B, 4) && aligned(pack, 16), MI355PPO_EALIGN, "%s: misaligned pointer (pack: 16 bytes)", fn);
    const long long total = (long long)(K / 16) * ((N + 31) / 32) * 512;
    hipLaunchKernelGGL(zpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), B, ldb, N, K,
                       static_cast<unsigned short*>(pack));
    return check_launch(fn);
}

static int zgemm_check(const char* fn, const float* A, const void* pack, const float* C, int M, int N, int K, int lda, int ldc) {
    MI355_REQUIRE(A && pack && C, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0, MI355PPO_EINVAL, "%s: M=%d N=%d K=%d (K must be a positive multiple of 16)", fn, M, N, K);
    MI355_REQUIRE(lda >= K && ldc >= N && lda % 4 == 0, MI355PPO_EINVAL, "%s: leading dimensions lda=%d ldc=%d (lda: a multiple of 4, >= K; ldc >= N)", fn, lda, ldc);
    MI355_REQUIRE(aligned(A, 16) && aligned(pack, 16) && aligned(C, 4), MI355PPO_EALIGN, "%s: A and pack must be 16-byte aligned", fn);
    MI355_REQUIRE((long long)M * lda * 4 < (1LL << 32), MI355PPO_EINVAL, "%s: A (%d x %d floats) must stay below 4 GiB (32-bit row offsets)", fn, M, lda);
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_fc_fwd_relu_packed_f32(const float* a, int lda, const void* pack, const float* bias, float* h,
                                                            int M, int N, int K, void* stream) {
    const char* fn = "mi355ppo_fc_fwd_relu_packed_f32";
    int rc = zgemm_check(fn, a, pack, h, M, N, K, lda, N);
    if (rc) return rc;
    MI355_REQUIRE(bias && aligned(bias, 4), MI355PPO_EINVAL, "%s: bias missing or misaligned", fn);
    MI355_REQUIRE((M + kZRows - 1) / kZRows <= 65535, MI355PPO_EINVAL, "%s: M=%d exceeds one launch", fn, M);
    const dim3 grid((N + 128 * 4 - 1) / (128 * 4), (M + kZRows - 1) / kZRows);
    if (bf16_term_pairs() == 9)
        hipLaunchKernelGGL((zgemm_kernel<Z_BIAS_RELU, true, 9>), grid, dim3(256), 0, as_stream(stream), a, lda,
                           static_cast<const unsigned char*>(pack), bias, (const float*)nullptr, h, N, M, N, K, 0xffff0000u, 0xffffff00u);
    else
        hipLaunchKernelGGL((zgemm_kernel<Z_BIAS_RELU, true, 6>), grid, dim3(256), 0, as_stream(stream), a, lda,
                           static_cast<const unsigned char*>(pack), bias, (const float*)nullptr, h, N, M, N, K, 0xffff0000u, 0xffffff00u);
    return check_launch(fn);
}

extern "C" MI355PPO_API int mi355ppo_fc_dgrad_mask_packed_f32(const float* dz, int lddz, const void* pack, const float* act_in, float* da,
                                                              int M, int N, int K, void* stream) {
    const char* fn = "mi355ppo_fc_dgrad_mask_packed_f32";
    int rc = zgemm_check(fn, dz, pack, da, M, N, K, lddz, N);
    if (rc) return rc;
    MI355_REQUIRE(act_in && aligned(act_in, 4) && act_in != da, MI355PPO_EINVAL, "%s: act_in missing, misaligned or aliased with da", fn);
    MI355_REQUIRE((M + 4 * kZRows - 1) / (4 * kZRows) <= 65535, MI355PPO_EINVAL, "%s: M=%d exceeds one launch", fn, M);
    const dim3 grid((N + 127) / 128, (M + 4 * kZRows - 1) / (4 * kZRows));
    if (bf16_term_pairs() == 9)
        hipLaunchKernelGGL((zgemm_kernel<Z_MASK, false, 9>), grid, dim3(256), 0, as_stream(stream), dz, lddz,
                           static_cast<const unsigned char*>(pack), (const float*)nullptr, act_in, da, N, M, N, K, 0xffff0000u, 0xffffff00u);
    else
        hipLaunchKernelGGL((zgemm_kernel<Z_MASK, false, 6>), grid, dim3(256), 0, as_stream(stream), dz, lddz,
                           static_cast<const unsigned char*>(pack), (const float*)nullptr, act_in, da, N, M, N, K, 0xffff0000u, 0xffffff00u);
    return check_launch(fn);
}
