// Kernel Y -- weight gradient of Linear(3136, 512) of the NatureCNN (cleanrl/ppo_atari_multigpu.py:144-145; autograd's
// dW = dz^T @ a for `hidden = relu(flatten @ W^T + b)`), gfx950, f32 matrix pipe:
//     dWp[n][k] = sum_m dz[m][n] * a[m][k]          n < 512, k < 3136, m < M (32,768 per minibatch)
//
// The reduction runs over the batch, so both operands are ALREADY in `v_mfma_f32_32x32x2_f32` operand layout as they lie in
// memory: lane (li, lh) of the A operand holds A[row li][kk lh] = dz[m + lh][n0 + li], of the B operand a[m + lh][k0 + li] --
// the two halves of a wave read 128 contiguous bytes of two consecutive batch rows.  No LDS, no transposition: a wave owns
// 64 x 224 of dWp (2 x 7 tiles = 224 accumulator registers, one wave per SIMD), streams its 64 + 224 columns of dz and a
// top to bottom with 4 loads per 14 MFMAs (896 pipe cycles; see `fetch` for the column assignment that makes them 16, 8, 8
// and 4 bytes wide), prefetched six row pairs ahead, and never synchronises with anybody.  A workgroup = 2 x 2 waves = 128 x 448; the batch is cut into S slabs (S = 9 at M = 32,768:
// 4 x 7 x 9 = 252 workgroups for 256 CUs, one round), each slab's workgroups write a partial, and a second kernel adds the
// partials in slab order (deterministic) and writes dW in the reference's (c, h, w) feature order -- the permutation torch
// ran as a separate copy.  Separate launch, not a ticketed last block: a kernel boundary costs ~3 us here, an agent-scope
// release/acquire pair per workgroup 1.4-2 us EACH (measured for K3).
//
// dz arrives already masked by the FC layer's own ReLU and the bias gradient already summed: both ride in the heads'
// backward pass (heads.hip), which reads h anyway.
//
// Workgroup -> XCD: the four n-tiles that read the same (k-tile, slab) block of `a` (the 411 MB operand) get the same XCD
// (workgroups are dealt to the 8 XCDs round robin), so that block leaves HBM once and is served three times by that L2.
#include "common.h"

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float y_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kYNT = 2, kYKT = 7;                         // tiles of 32 per wave: 64 (n) x 224 (k)
constexpr int kYWgN = 64 * 2, kYWgK = 224 * 2;            // 2 x 2 waves
constexpr int kYDepth = 9;                                // row pairs in flight per wave (36 loads)
constexpr int kYMaxSlabs = 9;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fcw_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ a, int lda, float* __restrict__ part, int M, int N, int K,
    int ntn, int ntk, int nslabs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // blockIdx -> (n-tile, k-tile, slab): the ntn workgroups of one (k-tile, slab) pair are ids that differ by multiples of 8
    int nt, combo;
    {
        const int b = blockIdx.x, x = b & 7, slot = b >> 3;
        nt = slot % ntn;
        combo = x + 8 * (slot / ntn);
    }
    if (combo >= ntk * nslabs) return;
    const int kt = combo % ntk, slab = combo / ntk;
    const int n0 = nt * kYWgN + (wave & 1) * 64, k0 = kt * kYWgK + (wave >> 1) * 224;
    if (n0 >= N || k0 >= K) return;                       // (whole wave; no barriers in this kernel)

    y_f32x16 acc[kYNT][kYKT];
#pragma unroll
    for (int i = 0; i < kYNT; ++i)
#pragma unroll
        for (int j = 0; j < kYKT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // The batch is dealt to the slabs in BLOCKS of 64 rows, round robin (slab s: blocks s, s + S, ...), not as S contiguous
    // ranges: at any moment all workgroups then work inside one window of S x 64 rows (7 MB of `a`), which HBM delivers once,
    // front to back, and the caches hand on to the other workgroups that need it.  With contiguous ranges every workgroup
    // walks its own 1.8 KB-wide column through its own 45 MB range -- 252 scattered streams.  (Measured: 0.95 ms either way,
    // against 0.82 ms for the same instruction stream on L2-resident rows and for the MFMAs alone; prefetch depth 6 .. 15 and
    // two half-size waves per SIMD change nothing -- see DESIGN 3.3.)  Rows past the last whole block: one masked pass.
    const int nfull = M >> 6, tail_rows = M & 63;
    const int nblocks = nfull > slab ? (nfull - slab + nslabs - 1) / nslabs : 0, npairs = nblocks * 32;
    // Which n (k) a tile row (column) stands for is free as long as the store agrees, so lane li takes CONSECUTIVE columns and
    // hands one to each tile: one 8-byte load of dz[m][n0 + 2 li ..] feeds the two n-tiles (tile i row li = n0 + 2 li + i), one
    // 16-, one 8- and one 4-byte load of a[m][..] the seven k-tiles (tiles 0-3 column li = k0 + 4 li + j, tiles 4-5 =
    // k0 + 128 + 2 li + j - 4, tile 6 = k0 + 192 + li): 4 load instructions per row pair instead of 9 four-byte ones.
    const float* pz = dz + n0 + 2 * li;
    const float* pa = a + k0;
    float fa[kYDepth][kYNT], fb[kYDepth][kYKT];
    auto fetch = [&](int slot, int p) {
        const int pp = p < npairs ? p : npairs - 1;        // past the end: re-read, never multiplied
        const size_t m = (size_t)(((slab + nslabs * (pp >> 5)) << 6) + 2 * (pp & 31) + lh);
        const float* ra = pa + m * lda;
        const float2 z = *reinterpret_cast<const float2*>(pz + m * lddz);
        const float4 b4 = *reinterpret_cast<const float4*>(ra + 4 * li);
        const float2 b2 = *reinterpret_cast<const float2*>(ra + 128 + 2 * li);
        fa[slot][0] = z.x; fa[slot][1] = z.y;
        fb[slot][0] = b4.x; fb[slot][1] = b4.y; fb[slot][2] = b4.z; fb[slot][3] = b4.w;
        fb[slot][4] = b2.x; fb[slot][5] = b2.y;
        fb[slot][6] = ra[192 + li];
    };
    auto multiply = [&](const float (&za)[kYNT], const float (&zb)[kYKT]) {
#pragma unroll
        for (int i = 0; i < kYNT; ++i)
#pragma unroll
            for (int j = 0; j < kYKT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(za[i], zb[j], acc[i][j], 0, 0, 0);
    };
    if (npairs > 0) {
#pragma unroll
        for (int u = 0; u < kYDepth; ++u) fetch(u, u);
        int p0 = 0;
        for (; p0 + kYDepth <= npairs; p0 += kYDepth) {      // whole groups: no conditions in the loop
#pragma unroll
            for (int u = 0; u < kYDepth; ++u) {
                multiply(fa[u], fb[u]);
                fetch(u, p0 + u + kYDepth);
                // keep each refill behind ITS multiplies: left alone the scheduler gathers the six refills at the top of the
                // rotated loop and the first multiplies wait for loads issued an instant earlier (measured: 0.99 ms)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < kYDepth; ++u)
            if (p0 + u < npairs) multiply(fa[u], fb[u]);       // (uniform) the last, partial group: already fetched
    }
    if (tail_rows != 0 && slab == nfull % nslabs) {           // the rows after the last whole block (< 64), rows >= M as zeros
        for (int m0 = nfull << 6; m0 < M; m0 += 2) {
            const int m = m0 + lh;
            const size_t mc = (size_t)(m < M ? m : M - 1);
            float za[kYNT], zb[kYKT];
            const float* ra = pa + mc * lda;
#pragma unroll
            for (int i = 0; i < kYNT; ++i) za[i] = m < M ? pz[mc * lddz + i] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) zb[j] = ra[4 * li + j];
            zb[4] = ra[128 + 2 * li]; zb[5] = ra[128 + 2 * li + 1]; zb[6] = ra[192 + li];
            multiply(za, zb);
        }
    }
    // partial [slab][n][k]: accumulator element e of tile (i, j) belongs to tile row r = (e & 3) + 8 (e >> 2) + 4 lh, i.e.
    // n = n0 + 2 r + i, and to tile column li, i.e. the k of the fetch above: 16-, 8- and 4-byte stores
    float* out = part + (size_t)slab * N * K;
#pragma unroll
    for (int i = 0; i < kYNT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = n0 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * lh) + i;
            float* row = out + (size_t)n * K + k0;
            *reinterpret_cast<float4*>(row + 4 * li) = make_float4(acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]);
            *reinterpret_cast<float2*>(row + 128 + 2 * li) = make_float2(acc[i][4][e], acc[i][5][e]);
            row[192 + li] = acc[i][6][e];
        }
}

// dW[n][f(k)] = sum over slabs (in slab order) of part[slab][n][k]; with C > 0 the columns are re-ordered from the trunk's
// (h, w, c) feature order to the reference's (c, h, w): k = hw * C + c  ->  c * (K / C) + hw.
__global__ __launch_bounds__(256) void fcw_reduce_kernel(const float* __restrict__ part, int nslabs, float* __restrict__ dW, int N, int K, int C) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)N * K;
    if (e >= total) return;
    float s = part[e];
    for (int p = 1; p < nslabs; ++p) s += part[(size_t)p * total + e];
    size_t o = e;
    if (C > 0) {
        const int n = (int)(e / K), k = (int)(e - (size_t)n * K);
        const int hw = k / C, c = k - hw * C;
        o = (size_t)n * K + (size_t)c * (K / C) + hw;
    }
    dW[o] = s;
}

static int fcw_slabs(int M) {
    int s = M / 512;                                      // at least 256 row pairs per slab
    return s < 1 ? 1 : (s > kYMaxSlabs ? kYMaxSlabs : s);
}

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API size_t mi355ppo_fc_wgrad_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)fcw_slabs(M) * N * K * sizeof(float);
}

extern "C" MI355PPO_API int mi355ppo_fc_wgrad_f32(const float* dz, int lddz, const float* a, float* dW, int M, int N, int K, int hwc_channels,
                                                  void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "mi355ppo_fc_wgrad_f32";
    MI355_REQUIRE(dz && a && dW, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(M > 0 && N > 0 && K > 0 && N % 64 == 0 && K % 224 == 0, MI355PPO_EINVAL,
                  "%s: M=%d N=%d K=%d (N must be a positive multiple of 64, K of 224: whole 64 x 224 wave tiles)", fn, M, N, K);
    MI355_REQUIRE(lddz >= N && lddz % 2 == 0, MI355PPO_EINVAL, "%s: lddz=%d (even, >= N=%d)", fn, lddz, N);
    MI355_REQUIRE(hwc_channels >= 0 && (hwc_channels == 0 || K % hwc_channels == 0), MI355PPO_EINVAL, "%s: hwc_channels=%d does not divide K=%d", fn,
                  hwc_channels, K);
    const size_t need = mi355ppo_fc_wgrad_workspace_bytes(M, N, K);
    MI355_REQUIRE(workspace && workspace_bytes >= need, MI355PPO_EWORKSPACE, "%s: workspace %zu bytes < required %zu", fn,
                  workspace ? workspace_bytes : (size_t)0, need);
    MI355_REQUIRE(aligned(dz, 8) && aligned(a, 16) && aligned(dW, 4) && aligned(workspace, 16), MI355PPO_EALIGN,
                  "%s: dz must be 8-byte, a and the workspace 16-byte aligned", fn);
    const int nslabs = fcw_slabs(M);
    const int ntn = (N + kYWgN - 1) / kYWgN, ntk = (K + kYWgK - 1) / kYWgK;
    const int combos = ntk * nslabs, groups = (combos + 7) / 8;                            // 8 (k-tile, slab) pairs per group, one per XCD
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(fcw_kernel, dim3((unsigned)(groups * ntn * 8)), dim3(256), 0, s, dz, lddz, a, K, part, M, N, K, ntn, ntk, nslabs);
    int rc = check_launch("fcw_kernel");
    if (rc) return rc;
    const size_t total = (size_t)N * K;
    hipLaunchKernelGGL(fcw_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, nslabs, dW, N, K, hwc_channels);
    return check_launch("fcw_reduce_kernel");
}
