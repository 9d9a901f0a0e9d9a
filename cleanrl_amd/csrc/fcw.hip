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
#include "bf16split.h"
#include "f16split.h"
#include <type_traits>

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
constexpr int kWSlabs = 5;
__global__ __launch_bounds__(256) void fcw_reduce_kernel(const float* __restrict__ part, int nslabs, float* __restrict__ dW, int N, int K, int C) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)N * K;
    if (e >= total) return;
    float s;
    if (nslabs == kWSlabs) {                          // kernel W's slab count: all loads in flight, the same order of additions
        float v[kWSlabs];
#pragma unroll
        for (int p = 0; p < kWSlabs; ++p) v[p] = part[(size_t)p * total + e];
        s = v[0];
#pragma unroll
        for (int p = 1; p < kWSlabs; ++p) s += v[p];
    } else if (nslabs == 8) {                         // kernel H's slab count (gemmh.hip)
        float v[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) v[p] = part[(size_t)p * total + e];
        s = v[0];
#pragma unroll
        for (int p = 1; p < 8; ++p) s += v[p];
    } else {
        s = part[e];
        for (int p = 1; p < nslabs; ++p) s += part[(size_t)p * total + e];
    }
    size_t o = e;
    if (C > 0) {
        const int n = (int)(e / K), k = (int)(e - (size_t)n * K);
        const int hw = k / C, c = k - hw * C;
        o = (size_t)n * K + (size_t)c * (K / C) + hw;
    }
    dW[o] = s;
}

// ------------------------------------------------------------------------------------------------------------ kernel W
// The same weight gradient on the bf16 matrix pipe (round 3): dWp[n][k] = sum_m dz[m][n] * a[m][k] with the six largest term
// pairs of the three-term bf16 split (bf16split.h; MI355PPO_BF16_PAIRS=9: all nine), f32 accumulation.
//
// The reduction index m is the SLOW index of both operands in memory, and the bf16 MFMA wants 8 consecutive m per lane: both
// operands are transposed on the way in.  Per k-step (16 batch rows) a wave loads its [16 m][128 n] block of dz and its
// [16 m][64 k] block of a COALESCED (a row of the block = 512 / 256 contiguous bytes), writes them to a wave-private LDS tile
// as they are, and reads fragments back TRANSPOSED: lane (li, lh) of fragment i takes column 32 i + li, rows 8 lh .. 8 lh + 7
// with eight 4-byte LDS reads (consecutive lanes = consecutive banks: conflict-free), then splits them in registers.  Wave-
// private LDS, double-buffered: no workgroup barrier.  A wave owns 128 (n) x 64 (k) of dWp (4 x 2 tiles, 128 accumulator
// registers, one wave per SIMD); the four waves of a workgroup cover n = 0 .. 511 of one 64-column block of k and read the same
// rows of `a` (the 411 MB operand leaves HBM once); the batch is dealt to S = 5 slabs in 16-row blocks (49 x 5 = 245
// workgroups at K = 3136: one round), partials summed in slab order by fcw_reduce_kernel.
// Per k-step: 12 global loads, 12 LDS writes, 48 LDS reads, 264 VALU, 48 MFMAs -- about seven issued instructions per MFMA:
// the wave is issue-bound at ~58 cycles per MFMA (tools/mfma_floor: 8 fillers -> 62), still 1.9 x the f32 pipe's 8 x 65 / 0.77
// cycles for the same 16 x 32 x 32 block.
typedef float w_f32x16 __attribute__((ext_vector_type(16)));
constexpr int kWMT = 4, kWNT = 2;                         // 32-row tiles of n (dz) x 32-column tiles of k (a) per wave
constexpr int kWn = 32 * kWMT, kWk = 32 * kWNT;           // 128 x 64
constexpr int kWLdsFloats = 16 * (kWn + kWk);             // one k-step of a wave: [16][128] + [16][64] floats = 12 KiB
constexpr unsigned kWOob = 0xFFFFF000u;
constexpr int kWRsrcWord3 = 0x00020000;

// schedule of a k-step: item t sits behind MFMA t.  kinds: 0 = read fragment `arg` from LDS (8 ds_read_b32), 1 = split piece
// `arg` = 6 * fragment + piece, 2 = LDS writes (third `arg` of the 12), 3 = global loads (third `arg`), -1 = nothing.
struct WItem { int kind, arg; };
// ppf = split pieces per fragment: 6 (bf16: two halves x {masks, subtractions, packs}) or 2 (f16: one piece per half, f16split.h)
constexpr WItem w_item(int t, int ppf) {
    // bf16: R0 R1 | S0.0 S0.1 S0.2 W0 S0.3 S0.4 S0.5 R2 | S1.* (W1) R3 | S2.* (W2) R4 | S3.* (L0) R5 | S4.* (L1) | S5.* (L2)
    // f16:  R0 R1 | S0.0 W0 S0.1 R2 | S1.0 W1 S1.1 R3 | S2.0 W2 S2.1 R4 | S3.0 L0 S3.1 R5 | S4.0 L1 S4.1 | S5.0 L2 S5.1
    if (t == 0) return {0, 0};
    if (t == 1) return {0, 1};
    int u = t - 2;
    for (int f = 0; f < 6; ++f) {
        const int len = ppf + 1 + (f < 4 ? 1 : 0);        // the split pieces, one W / L item, (one fragment read)
        if (u < len) {
            if (u < ppf / 2) return {1, ppf * f + u};
            if (u == ppf / 2) return f < 3 ? WItem{2, f} : WItem{3, f - 3};
            if (u < ppf + 1) return {1, ppf * f + u - 1};
            return {0, f + 2};
        }
        u -= len;
    }
    return {-1, 0};
}

// SPLIT: 0 = three bf16 terms, NP = 6 / 9 pairs; 1 = two f16 terms under the operands' power-of-two scales (amax records of dz and a),
// NP = 3 pairs, the partial un-scaled on its way out (f16split.h)
template <int NP, int SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fcw_bf16_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ a, float* __restrict__ part, int M, int N, int K, int nkb,
    unsigned m8, unsigned m16, const unsigned* __restrict__ dz_amax, const unsigned* __restrict__ a_amax) {
    constexpr int TERMS = SPLIT ? 2 : 3, PPF = SPLIT ? 2 : 6;
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * kWLdsFloats];        // 4 waves x 2 buffers x 12 KiB = 96 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    float sz = 1.0f, sa = 1.0f, un = 1.0f;                // SPLIT: the operands' scales and the factor that removes both from the partial
    if constexpr (SPLIT) {
        const int ez = f16_scale_exp(amax_load(dz_amax, lane)), ea = f16_scale_exp(amax_load(a_amax, lane));
        sz = f16_pow2(ez);
        sa = f16_pow2(ea);
        un = f16_unscale(ez, ea);
    }
    // Every workgroup reads ALL of its slab's dz (512 columns): in launch order the 49 workgroups of a slab land on all eight XCDs
    // and each L2 fetches the whole dz (8 x 67 MB of the launch's 0.95 GB of L2 misses, profiles/traffic.json).  XCD x takes a
    // contiguous range of the (slab, k block) order instead (kernel Z's scheme): an L2 then sees at most two slabs.
    unsigned wg = blockIdx.x;
    {
        const unsigned total = gridDim.x, x = wg & 7u, q = total >> 3, rem = total & 7u;
        wg = x * q + (x < rem ? x : rem) + (wg >> 3);
    }
    const int kb = (int)wg % nkb, slab = (int)wg / nkb;
    const int n0 = wave * kWn, k0 = kb * kWk;
    const int nblocks = M / 16;                           // 16-row blocks of the batch; slab s takes blocks s, s + S, s + 2 S, ...
    const int nsteps = (nblocks - slab + kWSlabs - 1) / kWSlabs;
    float* const wl = lds + wave * (2 * kWLdsFloats);
    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, (int)((unsigned)M * (unsigned)lddz * 4u), kWRsrcWord3);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, (int)((unsigned)M * (unsigned)K * 4u), kWRsrcWord3);
    // coalesced loads of a 16-row block: dz rows are 128 floats of this wave's n range (32 lanes x 16 bytes: 2 rows per
    // instruction, 8 instructions), a rows 64 floats (16 lanes: 4 rows per instruction, 4 instructions)
    const unsigned vo_dz = (unsigned)(lane >> 5) * (unsigned)lddz * 4u + (unsigned)n0 * 4u + 16u * (unsigned)(lane & 31);
    const unsigned vo_a = (unsigned)(lane >> 4) * (unsigned)K * 4u + (unsigned)k0 * 4u + 16u * (unsigned)(lane & 15);
    float* const wr_dz = wl + (lane >> 5) * kWn + 4 * (lane & 31);                  // + 2 u rows (u < 8)
    float* const wr_a = wl + 16 * kWn + (lane >> 4) * kWk + 4 * (lane & 15);        // + 4 u rows (u < 4)
    const float* const rd_dz = wl + (8 * lh) * kWn + li;                            // + e rows, + 32 i columns
    const float* const rd_a = wl + 16 * kWn + (8 * lh) * kWk + li;

    w_f32x16 acc[kWMT][kWNT];
#pragma unroll
    for (int i = 0; i < kWMT; ++i)
#pragma unroll
        for (int j = 0; j < kWNT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    s_u32x4 stage[12];                                    // one block as loaded: 8 x dz, 4 x a
    unsigned raw[2][8];                                   // two fragments as read back from LDS (f32, lane = column, 8 rows)
    unsigned tt[2][6][TERMS][4];                          // split fragments: [k-step parity][fragment: 4 x dz, 2 x a][term][4 x 2 bf16 / f16]
    auto sclamp = [&](int s) { return s < nsteps ? s : nsteps - 1; };               // past the end: re-read, never multiplied
    auto load_third = [&](int s, auto tc) {               // third tc of the 12 global loads of step s's block
        constexpr int t3 = decltype(tc)::value;
        const int blk = slab + sclamp(s) * kWSlabs;
        const unsigned so_dz = (unsigned)blk * 16u * (unsigned)lddz * 4u, so_a = (unsigned)blk * 16u * (unsigned)K * 4u;
#pragma unroll
        for (int u = 4 * t3; u < 4 * t3 + 4; ++u) {
            if (u < 8) stage[u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dz, vo_dz + (unsigned)(2 * u) * (unsigned)lddz * 4u, so_dz, 0));
            else stage[u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo_a + (unsigned)(4 * (u - 8)) * (unsigned)K * 4u, so_a, 0));
        }
    };
    auto write_third = [&](int buf, auto tc) {            // third tc of the 12 LDS writes of `stage` into buffer `buf`
        constexpr int t3 = decltype(tc)::value;
#pragma unroll
        for (int u = 4 * t3; u < 4 * t3 + 4; ++u) {
            if (u < 8) *reinterpret_cast<s_u32x4*>(wr_dz + buf * kWLdsFloats + 2 * u * kWn) = stage[u];
            else *reinterpret_cast<s_u32x4*>(wr_a + buf * kWLdsFloats + 4 * (u - 8) * kWk) = stage[u];
        }
    };
    auto read_frag = [&](int buf, auto fc) {              // fragment f (0..3: dz tile, 4..5: a tile) of buffer `buf` -> raw[f & 1]
        constexpr int f = decltype(fc)::value;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            raw[f & 1][e] = f < 4 ? __float_as_uint(rd_dz[buf * kWLdsFloats + e * kWn + 32 * f]) : __float_as_uint(rd_a[buf * kWLdsFloats + e * kWk + 32 * (f - 4)]);
    };
    unsigned t8[4], t16[4];
    float smid[4], slo[4];
    auto split_piece = [&](int par, auto fc, auto pc) {   // piece pc (0..5) of fragment f: halves of 4 elements x {masks, subtractions, packs}
        constexpr int f = decltype(fc)::value, hf = decltype(pc)::value / 3, piece = decltype(pc)::value % 3;
        if constexpr (piece == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t8[j] = raw[f & 1][4 * hf + j] & m8;
                t16[j] = raw[f & 1][4 * hf + j] & m16;
            }
        } else if constexpr (piece == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                smid[j] = __uint_as_float(t16[j]) - __uint_as_float(t8[j]);
                slo[j] = __uint_as_float(raw[f & 1][4 * hf + j]) - __uint_as_float(t16[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                tt[par][f][0][2 * hf + (j >> 1)] = __builtin_amdgcn_perm(raw[f & 1][4 * hf + j + 1], raw[f & 1][4 * hf + j], 0x07060302u);
                tt[par][f][1][2 * hf + (j >> 1)] = split_pack(smid[j + 1], smid[j]);
                tt[par][f][2][2 * hf + (j >> 1)] = split_pack(slo[j + 1], slo[j]);
            }
        }
    };
    auto split_piece_h = [&](int par, auto fc, auto hc) { // SPLIT = 1: half hf (4 elements) of fragment f, 10 VALU
        constexpr int f = decltype(fc)::value, hf = decltype(hc)::value;
        unsigned hi[2], lo[2];
        f16_split4((s_u32x4){raw[f & 1][4 * hf], raw[f & 1][4 * hf + 1], raw[f & 1][4 * hf + 2], raw[f & 1][4 * hf + 3]}, f < 4 ? sz : sa, hi, lo);
        tt[par][f][0][2 * hf] = hi[0]; tt[par][f][0][2 * hf + 1] = hi[1];
        tt[par][f][TERMS - 1][2 * hf] = lo[0]; tt[par][f][TERMS - 1][2 * hf + 1] = lo[1];
    };
    auto frag_bits = [&](int par, int f, int term) { return (s_u32x4){tt[par][f][term][0], tt[par][f][term][1], tt[par][f][term][2], tt[par][f][term][3]}; };
    constexpr int NM = NP * kWMT * kWNT;
    static_assert(SPLIT ? (NP == 3 || NP == 4) : (NP == 6 || NP == 9), "term pairs of the split");
    static_assert(NM >= 12 + 6 * PPF, "one MFMA per scheduled item");
    constexpr int PX[9] = {0, 0, 1, SPLIT ? 1 : 0, 2, 1, 1, 2, 2}, PY[9] = {0, 1, 0, SPLIT ? 1 : 2, 0, 1, 2, 1, 2};
    // One pipeline step of parity q (the MFMAs of step s on tt[q]); meanwhile: the fragments of step s + 1 are read from LDS
    // buffer q ^ 1 and split into tt[q ^ 1]; `stage` (the block of step s + 2, loaded during the previous step) is written to
    // LDS buffer q (its fragments were all read during the previous step); the block of step s + 3 is loaded into `stage`.
    // Order pinned by hand: item w_item(g) behind MFMA g, each followed by a sched_barrier.
    auto step = [&](auto qc, int s) {
        constexpr int q = decltype(qc)::value;
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int g = G, pi = g / (kWMT * kWNT), i = (g / kWNT) % kWMT, j = g % kWNT;
                if constexpr (SPLIT)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, frag_bits(q, i, PX[pi])),
                                                                       __builtin_bit_cast(s_f16x8, frag_bits(q, 4 + j, PY[pi])), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s_bf16x8, frag_bits(q, i, PX[pi])),
                                                                        __builtin_bit_cast(s_bf16x8, frag_bits(q, 4 + j, PY[pi])), acc[i][j], 0, 0, 0);
                constexpr WItem it = w_item(g, PPF);
                if constexpr (it.kind == 0) read_frag(q ^ 1, std::integral_constant<int, it.arg>{});
                else if constexpr (it.kind == 1 && SPLIT) split_piece_h(q ^ 1, std::integral_constant<int, it.arg / 2>{}, std::integral_constant<int, it.arg % 2>{});
                else if constexpr (it.kind == 1) split_piece(q ^ 1, std::integral_constant<int, it.arg / 6>{}, std::integral_constant<int, it.arg % 6>{});
                else if constexpr (it.kind == 2) write_third(q, std::integral_constant<int, it.arg>{});
                else if constexpr (it.kind == 3) load_third(s + 3, std::integral_constant<int, it.arg>{});
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, NM>{});
    };
    auto load_all = [&](int s) { load_third(s, std::integral_constant<int, 0>{}); load_third(s, std::integral_constant<int, 1>{}); load_third(s, std::integral_constant<int, 2>{}); };
    auto write_all = [&](int buf) { write_third(buf, std::integral_constant<int, 0>{}); write_third(buf, std::integral_constant<int, 1>{}); write_third(buf, std::integral_constant<int, 2>{}); };
    auto split_all = [&](int par, int buf) {              // all six fragments of LDS buffer `buf` -> tt[par] (prologue)
        // (ONE pack expansion: hipcc / clang mis-substitutes the outer pack element inside a nested pack-expanding lambda --
        // every inner call saw fragment 0, piece 0; found with a host replica of this bookkeeping)
        [&]<int... T>(std::integer_sequence<int, T...>) {
            ([&] {
                if constexpr (T % PPF == 0) read_frag(buf, std::integral_constant<int, T / PPF>{});
                if constexpr (SPLIT) split_piece_h(par, std::integral_constant<int, T / 2>{}, std::integral_constant<int, T % 2>{});
                else split_piece(par, std::integral_constant<int, T / 6>{}, std::integral_constant<int, T % 6>{});
            }(), ...);
        }(std::make_integer_sequence<int, 6 * PPF>{});
    };
    if (nsteps > 0) {
        // prologue: step 0 -> LDS buffer 0 -> tt[0]; step 1 -> LDS buffer 1; step 2 in `stage`
        load_all(0);
        write_all(0);
        load_all(1);
        split_all(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        write_all(1);
        load_all(2);
        __builtin_amdgcn_sched_barrier(0);
        int s = 0;
        for (; s + 2 <= nsteps; s += 2) {
            step(std::integral_constant<int, 0>{}, s);
            step(std::integral_constant<int, 1>{}, s + 1);
        }
        if (s < nsteps) step(std::integral_constant<int, 0>{}, s);
    }
    // partial [slab][n][k]: accumulator element e of tile (i, j) is row n0 + 32 i + (e & 3) + 8 (e >> 2) + 4 lh, column k0 + 32 j + li
    float* out = part + (size_t)slab * N * K;
#pragma unroll
    for (int i = 0; i < kWMT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = n0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
            float* row = out + (size_t)n * K + k0 + li;
#pragma unroll
            for (int j = 0; j < kWNT; ++j) row[32 * j] = SPLIT ? acc[i][j][e] * un : acc[i][j][e];
        }
}

static int fcw_slabs(int M) {
    int s = M / 512;                                      // at least 256 row pairs per slab
    return s < 1 ? 1 : (s > kYMaxSlabs ? kYMaxSlabs : s);
}

}  // namespace mi355ppo

using namespace mi355ppo;

// kernel W takes this layer's shape at minibatch sizes; the operands' alignment is checked at the launch
static bool fcw_w_shape(int M, int N, int K) {
    return N == 4 * kWn && K % kWk == 0 && M % 16 == 0 && M >= 1024 && (long long)M * K * 4 < (1LL << 32) - 8192;
}

extern "C" MI355PPO_API size_t mi355ppo_fc_wgrad_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int slabs = (fcw_w_shape(M, N, K) && kWSlabs > fcw_slabs(M)) ? kWSlabs : fcw_slabs(M);   // kernel Y's or kernel W's slabs, whichever is more
    if (gemmh_takes(M, N, K, N) && gemmh_slabs() > slabs) slabs = gemmh_slabs();                   // (kernel H, f16 split: 8)
    return (size_t)slabs * N * K * sizeof(float);
}

extern "C" MI355PPO_API int mi355ppo_fc_wgrad_kernel(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return fcw_w_shape(M, N, K) ? 'W' : 'Y';
}

// the kernel mi355ppo_fc_wgrad_f16x2_f32 runs for this shape (dense dz, lddz = N): 'H' (gemmh.hip, round 6), 'W' or 'Y'
extern "C" MI355PPO_API int mi355ppo_fc_wgrad_kernel_f16x2(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return gemmh_takes(M, N, K, N) ? 'H' : mi355ppo_fc_wgrad_kernel(M, N, K);
}

static int fc_wgrad_impl(const char* fn, const float* dz, int lddz, const float* a, float* dW, int M, int N, int K, int hwc_channels,
                         void* workspace, size_t workspace_bytes, const unsigned* dz_amax, const unsigned* a_amax, void* stream) {
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
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    // round 6, f16 split: kernel H (gemmh.hip) from 4,096 rows on -- both operands through a workgroup-wide LDS ring, split once
    if (dz_amax && a_amax && aligned(dz, 16) && gemmh_takes(M, N, K, lddz)) {
        int rch = gemmh_launch(dz, lddz, a, part, M, N, K, dz_amax, a_amax, s);
        if (rch) return rch;
        const size_t totalh = (size_t)N * K;
        hipLaunchKernelGGL(fcw_reduce_kernel, dim3((unsigned)((totalh + 255) / 256)), dim3(256), 0, s, part, gemmh_slabs(), dW, N, K, hwc_channels);
        return check_launch("fcw_reduce_kernel");
    }
    // kernel W (bf16 pipe) for this layer's shape at minibatch sizes; kernel Y (f32 pipe) otherwise
    if (fcw_w_shape(M, N, K) && aligned(dz, 16) && lddz % 4 == 0) {
        const int nkb = K / kWk;
        if (dz_amax)
            hipLaunchKernelGGL((fcw_bf16_kernel<3, 1>), dim3((unsigned)(nkb * kWSlabs)), dim3(256), 0, s, dz, lddz, a, part, M, N, K, nkb, 0xffff0000u, 0xffffff00u, dz_amax, a_amax);
        else if (bf16_term_pairs() == 9)
            hipLaunchKernelGGL((fcw_bf16_kernel<9, 0>), dim3((unsigned)(nkb * kWSlabs)), dim3(256), 0, s, dz, lddz, a, part, M, N, K, nkb, 0xffff0000u, 0xffffff00u, dz_amax, a_amax);
        else
            hipLaunchKernelGGL((fcw_bf16_kernel<6, 0>), dim3((unsigned)(nkb * kWSlabs)), dim3(256), 0, s, dz, lddz, a, part, M, N, K, nkb, 0xffff0000u, 0xffffff00u, dz_amax, a_amax);
        int rcw = check_launch("fcw_bf16_kernel");
        if (rcw) return rcw;
        const size_t totalw = (size_t)N * K;
        hipLaunchKernelGGL(fcw_reduce_kernel, dim3((unsigned)((totalw + 255) / 256)), dim3(256), 0, s, part, kWSlabs, dW, N, K, hwc_channels);
        return check_launch("fcw_reduce_kernel");
    }
    const int nslabs = fcw_slabs(M);
    const int ntn = (N + kYWgN - 1) / kYWgN, ntk = (K + kYWgK - 1) / kYWgK;
    const int combos = ntk * nslabs, groups = (combos + 7) / 8;                            // 8 (k-tile, slab) pairs per group, one per XCD
    hipLaunchKernelGGL(fcw_kernel, dim3((unsigned)(groups * ntn * 8)), dim3(256), 0, s, dz, lddz, a, K, part, M, N, K, ntn, ntk, nslabs);
    int rc = check_launch("fcw_kernel");
    if (rc) return rc;
    const size_t total = (size_t)N * K;
    hipLaunchKernelGGL(fcw_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, nslabs, dW, N, K, hwc_channels);
    return check_launch("fcw_reduce_kernel");
}

extern "C" MI355PPO_API int mi355ppo_fc_wgrad_f32(const float* dz, int lddz, const float* a, float* dW, int M, int N, int K, int hwc_channels,
                                                  void* workspace, size_t workspace_bytes, void* stream) {
    return fc_wgrad_impl("mi355ppo_fc_wgrad_f32", dz, lddz, a, dW, M, N, K, hwc_channels, workspace, workspace_bytes, nullptr, nullptr, stream);
}

// The same with kernel W on the two-term f16 split (f16split.h): dz_amax / a_amax = the operands' amax records.  Shapes kernel W does
// not take (mi355ppo_fc_wgrad_kernel(M, N, K) == 'Y') run the f32-pipe kernel Y as before; the records are then not read.
extern "C" MI355PPO_API int mi355ppo_fc_wgrad_f16x2_f32(const float* dz, int lddz, const float* a, float* dW, int M, int N, int K, int hwc_channels,
                                                        void* workspace, size_t workspace_bytes, const uint32_t* dz_amax, const uint32_t* a_amax,
                                                        void* stream) {
    const char* fn = "mi355ppo_fc_wgrad_f16x2_f32";
    MI355_REQUIRE(dz_amax && a_amax && aligned(dz_amax, 64) && aligned(a_amax, 64), MI355PPO_EINVAL, "%s: amax records missing or not 64-byte aligned", fn);
    return fc_wgrad_impl(fn, dz, lddz, a, dW, M, N, K, hwc_channels, workspace, workspace_bytes, dz_amax, a_amax, stream);
}
