// Kernel Z -- the exact-product bf16 GEMM / implicit-GEMM family of round 3:  C[m][n] = sum_k A[m][k] * B[n][k].
//   * GEMM rows (the FC layer of the NatureCNN, cleanrl/ppo_atari_multigpu.py:144-145): A (M, K) f32 row-major -- the
//     activations (forward) or the incoming gradient (data gradient), B (N, K) the weight matrix / its transpose;
//   * convolution rows (:139-142): row m = an output pixel (forward of layers 2 / 3) or a pixel of the data gradient's grid, k =
//     (tap row, tap column, channel) -- with channels-last storage the KW * C values of one tap row of the pixel's window are
//     contiguous, so a k-step is 64 contiguous bytes at a fixed offset from the window origin: no im2col buffer.  Zero padding
//     of the data gradients is the buffer instructions' range check (an invalid tap loads from an out-of-range offset: 0).
//
// What bounded kernels X / C (profiles/r03_pmc_busy_kernels_x_c.csv, profiles/r03_conv_traffic_pairs_ab.jsonl): with "lane =
// row" fragment loads every lane of a 16-byte load touches its own cache line, and the vector-memory front end (TA) processes
// about one line per clock -- it was 80 % busy with the matrix pipe 37 % busy (the same loads issued four lanes per row: FC
// forward 691 -> 531 us, results aside); the 264 split instructions per k-step of kernel X did the rest.  Hence:
//   * B is split AHEAD, once per optimizer step, into MFMA fragment order (`zpack_kernel`: [k-step][32-column tile][term]
//     [lane][8 bf16] = one contiguous KiB per fragment load): no VALU for B in the GEMM, fully coalesced loads;
//   * A is loaded COALESCED -- four consecutive lanes read the 64 contiguous bytes (16 k) of one row, a wave instruction
//     covers 16 rows -- written to a wave-private LDS tile (row pitch 80 bytes: conflict-free for the 16-byte writes and for
//     the 16-byte "lane = row" reads; tests/test_kernel_maps.py), read back as fragments and split in registers (44 VALU per
//     fragment).  Wave-private: no workgroup barrier anywhere; the LDS operations of one wave execute in order.
// Arithmetic: every f32 is the exact sum of three bf16 terms (bf16split.h); the six term pairs (i, j), i + j <= 2, are
// multiplied on `v_mfma_f32_32x32x16_bf16` with f32 accumulation; the three dropped pairs are below the rounding of one f32
// multiply.  MI355PPO_BF16_PAIRS=9 multiplies all nine (exact products).
//
// A wave owns (32 MT) rows x (32 NT) columns of C.  FC at minibatch size: MT 2 x NT 4 = 128 accumulator registers, one wave per
// SIMD; the convolutions (K = 256 .. 576 only) and the FC forward of smaller batches: MT 2 x NT 2 = 64 accumulator registers and
// about 200 VGPRs, two waves per SIMD from two DIFFERENT 4-wave workgroups per CU -- started at different times, so one wave's
// prologue / epilogue falls under the other's MFMAs (8-wave workgroups ran their two waves per SIMD in phase: 2 - 5 % slower,
// profiles/r03_zcfg_ab.jsonl).
#include "common.h"
#include "conv1q_pack.h"
#include "bf16split.h"
#include "f16split.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float z_f32x16 __attribute__((ext_vector_type(16)));

// Epilogues.  Z_BIAS_RELU: C = relu(acc + bias); Z_MASK / Z_MASK_CLS4: C = acc where mask > 0 (an f32 tensor of C's shape: the forward
// activation), else 0.  The *_BITS / Z_MASKB* variants carry the ReLU mask as ONE BIT per element instead (word w, bit b <-> element
// 32 w + b of the flat tensor; set where the activation is > 0): the forward writes the words beside its f32 output (a ballot per
// accumulator row, one 32-lane store per 32 x 32 tile), the data gradient fetches one word per row and column tile (a wave
// instruction per 64 x 32 tile instead of thirty-two) and selects by lane mask -- 1/32 of the mask bytes (the three data gradients
// read 2.8 GB of masks per 32,768-image minibatch otherwise).
// lane `l` of w := the wave-uniform value x (v_writelane_b32; hipcc 7.2 has no builtin for it).  s_nop 1: on gfx940 / gfx950 a VALU
// instruction that reads an SGPR needs two wait states behind the VALU instruction that wrote it (here: the v_cmp of the ballot), and
// the compiler's hazard recognizer does not look inside inline asm -- round 5's f16 instantiations scheduled the compare directly in
// front of the writelane and the mask words came out with bits of the PREVIOUS compare (tests/test_gpu_f16x2.py found it).
__device__ __forceinline__ int z_writelane(int w, unsigned x, int l) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(w) : "s"(x), "i"(l));
    return w;
}

// Z_RAW (GEMM rows, split K): blockIdx.z owns `steps_per` k-steps and stores its raw f32 partial of C into slab z of a workspace;
// zsplit_reduce_kernel adds the slabs in order, then bias + ReLU (rollout-sized batches: M / 64 x N / 64 wave tiles alone cannot fill
// the chip and each would walk all K / 16 = 196 k-steps in a row).
enum { Z_BIAS_RELU = 0, Z_MASK = 1, Z_MASK_CLS4 = 2, Z_MASKB = 3, Z_MASKB_CLS4 = 4, Z_BIAS_RELU_BITS = 5, Z_RAW = 6 };

// x where bit (lane) of the 64-bit lane mask {hi, lo} is set, else 0: one v_cndmask with the mask in an SGPR pair.
__device__ __forceinline__ float z_keep_where(float x, unsigned lo, unsigned hi) {
    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
    float r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));      // (s_nop: the mask comes from v_readlane -- see z_writelane)
    return r;
}
constexpr int kZPitch = 20;                           // floats per LDS row: 16 k + 4 pad (80 bytes)
constexpr int kZTileBytes = 3 * 64 * 16;              // one 32-column tile of one k-step in the pack: 3 terms x 1 KiB
constexpr unsigned kZOob = 0xFFFFF000u;               // buffer offset out of range for every tensor < 4 GiB - 4 KiB
constexpr int kZRsrcWord3 = 0x00020000;               // raw buffer, 32-bit elements (gfx9 / CDNA resource format)

// pack[s][j][t][lane][e] (bf16) = term t of B[n = 32 j + (lane & 31)][k = 16 s + 8 (lane >> 5) + e]; rows n >= N are zero.
__device__ __forceinline__ void zpack_store(unsigned short* __restrict__ pack, long long sj, int lane, int e, float x) {
    const unsigned xb = __float_as_uint(x);
    const unsigned t8 = xb & 0xffff0000u, t16 = xb & 0xffffff00u;
    const float mid = __uint_as_float(t16) - __uint_as_float(t8), lo = x - __uint_as_float(t16);
    const size_t o = ((size_t)sj * 3 * 64 + lane) * 8 + e;                           // term 0; terms 1, 2 follow at + 512, + 1024
    pack[o] = (unsigned short)(xb >> 16);
    pack[o + 512] = (unsigned short)(__float_as_uint(mid) >> 16);
    pack[o + 1024] = (unsigned short)(__float_as_uint(lo) >> 16);
}

// the same for two consecutive elements e0, e0 + 1 (e0 even) of one lane: three 4-byte stores
__device__ __forceinline__ void zpack_store2(unsigned short* __restrict__ pack, long long sj, int lane, int e0, float x0, float x1) {
    unsigned w[3];
    const float xs[2] = {x0, x1};
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned xb = __float_as_uint(xs[i]);
        const unsigned t8 = xb & 0xffff0000u, t16 = xb & 0xffffff00u;
        const float mid = __uint_as_float(t16) - __uint_as_float(t8), lo = xs[i] - __uint_as_float(t16);
        h[i] = xb >> 16; m[i] = __float_as_uint(mid) >> 16; l[i] = __float_as_uint(lo) >> 16;
    }
    w[0] = h[0] | (h[1] << 16); w[1] = m[0] | (m[1] << 16); w[2] = l[0] | (l[1] << 16);
    const size_t o = ((size_t)sj * 3 * 64 + lane) * 8 + e0;
    *reinterpret_cast<unsigned*>(pack + o) = w[0];
    *reinterpret_cast<unsigned*>(pack + o + 512) = w[1];
    *reinterpret_cast<unsigned*>(pack + o + 1024) = w[2];
}

__global__ __launch_bounds__(256) void zpack_kernel(const float* __restrict__ B, int ldb, int N, int K, unsigned short* __restrict__ pack) {
    const int ntiles = (N + 31) / 32;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;                 // (s, j, lane, e)
    if (idx >= (long long)(K / 16) * ntiles * 512) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long long sj = idx >> 9;
    const int j = (int)(sj % ntiles), s = (int)(sj / ntiles);
    const int n = 32 * j + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
    zpack_store(pack, sj, lane, e, n < N ? B[(size_t)n * ldb + k] : 0.0f);
}

// ---- the f16x2 pack (SPLIT = 1 of kernel Z, f16split.h): [64-byte header: word 0 = bits of max |B|, the rest 0][k-step][32-column tile]
// [hi, lo][64 lanes][8 f16], the terms of s B with s = 2^f16_scale_exp(max |B|) -- the kernel derives the same s from the header.
// `pack` below = the first byte behind the header.
__device__ __forceinline__ void zpack_store_h(unsigned short* __restrict__ pack, long long sj, int lane, int e, float x, float s) {
    unsigned short h, l;
    f16_split1(x, s, h, l);
    const size_t o = ((size_t)sj * 2 * 64 + lane) * 8 + e;                           // term 0; term 1 follows at + 512
    pack[o] = h;
    pack[o + 512] = l;
}
__device__ __forceinline__ void zpack_store2_h(unsigned short* __restrict__ pack, long long sj, int lane, int e0, float x0, float x1, float s) {
    unsigned short h0, l0, h1, l1;
    f16_split1(x0, s, h0, l0);
    f16_split1(x1, s, h1, l1);
    const size_t o = ((size_t)sj * 2 * 64 + lane) * 8 + e0;
    *reinterpret_cast<unsigned*>(pack + o) = (unsigned)h0 | ((unsigned)h1 << 16);
    *reinterpret_cast<unsigned*>(pack + o + 512) = (unsigned)l0 | ((unsigned)l1 << 16);
}
__device__ __forceinline__ void zpack_header(unsigned char* __restrict__ pack, unsigned amax_bits, int t) {      // threads t = 0 .. 15 of one block
    if (t < kF16PackHeader / 4) reinterpret_cast<unsigned*>(pack)[t] = t == 0 ? amax_bits : 0u;
}

__global__ __launch_bounds__(256) void zpack_h_kernel(const float* __restrict__ B, int ldb, int N, int K, const unsigned* __restrict__ b_amax,
                                                      unsigned char* __restrict__ pack) {
    const unsigned am = amax_load(b_amax, threadIdx.x & 63);                        // (before any exit: the wave reduction wants all lanes)
    const float sc = f16_pow2(f16_scale_exp(am));
    if (blockIdx.x == 0) zpack_header(pack, am, threadIdx.x);
    const int ntiles = (N + 31) / 32;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;                 // (s, j, lane, e)
    if (idx >= (long long)(K / 16) * ntiles * 512) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long long sj = idx >> 9;
    const int j = (int)(sj % ntiles), st = (int)(sj / ntiles);
    const int n = 32 * j + (lane & 31), k = 16 * st + 8 * (lane >> 5) + e;
    zpack_store_h(reinterpret_cast<unsigned short*>(pack + kF16PackHeader), sj, lane, e, n < N ? B[(size_t)n * ldb + k] : 0.0f, sc);
}

// max |x| of up to three tensors into their amax records (rec + MI355PPO_AMAX_WORDS * t), `per` blocks per tensor; the records were zeroed
struct ZAbsmax3 {
    const float* x[3];
    long long n[3];
};
__global__ __launch_bounds__(256) void zabsmax_kernel(ZAbsmax3 a, unsigned* __restrict__ rec, int per) {
    const int t = blockIdx.x / per, b = blockIdx.x - t * per;
    const float* __restrict__ x = a.x[t];
    const long long n = a.n[t];
    unsigned m = 0u;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long long n4 = n >> 2;
        for (long long i = (long long)b * 256 + threadIdx.x; i < n4; i += (long long)per * 256) {
            const uint4 v = reinterpret_cast<const uint4*>(x)[i];
            const unsigned p = max(max(v.x & 0x7fffffffu, v.y & 0x7fffffffu), max(v.z & 0x7fffffffu, v.w & 0x7fffffffu));
            m = max(m, p);
        }
        for (long long i = (n4 << 2) + (long long)b * 256 + threadIdx.x; i < n; i += (long long)per * 256) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    } else {
        for (long long i = (long long)b * 256 + threadIdx.x; i < n; i += (long long)per * 256) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    }
    amax_commit(rec + (size_t)MI355PPO_AMAX_WORDS * t, m, (unsigned)blockIdx.x * 4u + (threadIdx.x >> 6), threadIdx.x & 63);
}

// The same for the weight tensors of the NatureCNN packs WITHOUT atomics or a zeroed record: kAmaxSlots blocks per tensor, block b
// STORES the maximum of its share into slot b (every slot is written: the record needs no memset in front -- a memset node inside a
// captured update graph was the one launch of the pack sequence that a replay did not keep in order, tools/gpu/r5_graph_debug.py).
__global__ __launch_bounds__(1024) void zabsmax_store_kernel(ZAbsmax3 a, unsigned* __restrict__ rec) {
    __shared__ unsigned wmax[16];
    const int t = blockIdx.x / kAmaxSlots, b = blockIdx.x - t * kAmaxSlots;
    const float* __restrict__ x = a.x[t];
    const long long n = a.n[t];
    unsigned m = 0u;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long long n4 = n >> 2;
        for (long long i = (long long)b * 1024 + threadIdx.x; i < n4; i += (long long)kAmaxSlots * 1024) {
            const uint4 v = reinterpret_cast<const uint4*>(x)[i];
            m = max(m, max(max(v.x & 0x7fffffffu, v.y & 0x7fffffffu), max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
        }
        for (long long i = (n4 << 2) + (long long)b * 1024 + threadIdx.x; i < n; i += (long long)kAmaxSlots * 1024) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    } else {
        for (long long i = (long long)b * 1024 + threadIdx.x; i < n; i += (long long)kAmaxSlots * 1024) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    }
    const unsigned w = wave_umax(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x < 64) {
        const unsigned v = wave_umax(threadIdx.x < 16 ? wmax[threadIdx.x] : 0u);
        if (threadIdx.x == 0) rec[(size_t)MI355PPO_AMAX_WORDS * t + b * kAmaxStride] = v;
    }
}

// Every weight pack of the NatureCNN agent in ONE launch, straight from the parameters as torch stores them (round 4).  After an
// optimizer step the learner used to issue, per minibatch: conv_repack_kernel x 4 and two torch copies (the (h, w, c) reorder of
// Linear(3136,512).weight and its transpose) to build the f32 matrices, zpack_kernel x 6 over them, and kernel Q's digit pack -- 13
// launches of 3 - 12 us each, 1 % of a 32,768-image minibatch and 5 % of a 4,096-image one.  Piece p of the grid = the blocks
// [first[p], first[p] + count[p]): 0 layer-2 forward (64 x 512), 1 layer-3 forward (64 x 576), 2 layer-3 data gradient (64 x 576, taps
// flipped), 3 layer-2 data gradient (4 parity classes x 32 channels, 256), 4 FC forward (512 x 3,136, (h, w, c) order), 5 FC data
// gradient (3,136 x 512), 6 kernel Q's pack (one block).  The element maps are conv.hip's conv_repack_kernel modes 0 / 1 / 2 and
// cnn.py's fc_weight_hwc; the packs are bit-identical to the 13-launch route (tests/test_gpu_cnn.py).
struct ZNaturePacks {
    const float *W1, *W2, *W3, *Wfc;
    unsigned short* out[6];
    unsigned char* qpack;
    unsigned first[7], count[7];      // block range of piece p (6 = kernel Q's pack); the long blocks come first in the grid: Q's pack, the FC slices
    const unsigned* wrec;             // SPLIT = 1: the amax records of W2, W3, Wfc (MI355PPO_AMAX_WORDS apart), filled by zabsmax_kernel
};

// SPLIT = 1: the six kernel-Z packs in the f16x2 format (header + two f16 planes under the weight tensor's scale); kernel Q's pack as before
template <int SPLIT>
__global__ __launch_bounds__(256) void znature_pack_kernel(ZNaturePacks a) {
    int piece = 0;
#pragma unroll
    for (int p = 1; p < 7; ++p)
        if (blockIdx.x >= a.first[p] && blockIdx.x < a.first[p] + a.count[p]) piece = p;
    if (piece == 6) {
        conv1q_pack_body(a.W1, a.qpack);
        return;
    }
    float sc = 1.0f;
    unsigned short* const outp = a.out[piece] + (SPLIT ? kF16PackHeader / 2 : 0);       // the planes (behind the header)
    if constexpr (SPLIT) {                 // (block-uniform code so far: all lanes are active for the wave reduction)
        const int wt = piece >= 4 ? 2 : (piece == 1 || piece == 2) ? 1 : 0;             // pieces 0, 3: W2; 1, 2: W3; 4, 5: Wfc
        const unsigned am = amax_load(a.wrec + (size_t)MI355PPO_AMAX_WORDS * wt, threadIdx.x & 63);
        sc = f16_pow2(f16_scale_exp(am));
        if (blockIdx.x == a.first[piece]) zpack_header(reinterpret_cast<unsigned char*>(a.out[piece]), am, threadIdx.x);
    }
    if (piece >= 4) {
        // The FC weight (512 x 3,136 in (c, h, w) feature order) enters both packs in (h, w, c) order: element by element that is a 4-byte
        // read at stride 49 (or 3,136) floats per pack element -- 26 of the launch's 28 us.  Instead a block stages a slice whose rows are
        // CONTIGUOUS in the parameter through LDS (coalesced 16-byte loads) and walks the 49 (h, w) positions over it:
        //   piece 4, block (j, cb):  rows n = 32 j .. + 31, channels 16 cb .. + 15 = 784 contiguous floats per row -> units (s = 4 hw + cb, j)
        //   piece 5, block (s, ch):  rows n = 16 s .. + 15, channels 32 ch .. + 31 = 1,568 contiguous floats per row -> units (s, j' = 2 hw + ch)
        // (row pitch + 1 float: the walk's reads -- lanes along the rows / the channels, 49 floats apart -- touch 32 distinct banks)
        extern __shared__ __align__(16) float zn_lds[];
        const int b = (int)(blockIdx.x - a.first[piece]);
        const int rows = piece == 4 ? 32 : 16, run = piece == 4 ? 784 : 1568, pitch = run + 1;
        const int r0 = piece == 4 ? 32 * (b >> 2) : 16 * (b >> 1), c0 = piece == 4 ? 784 * (b & 3) : 1568 * (b & 1);
        const bool al16 = (reinterpret_cast<uintptr_t>(a.Wfc) & 15) == 0;      // (every slice row then starts on a 16-byte boundary: 3,136 and 784 are multiples of 4)
        const int total4 = rows * (run / 4);                                  // 6,272 16-byte chunks: 24.5 per thread, eight in flight at a time
        for (int base = 0; base < total4; base += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256 + threadIdx.x;
                if (i < total4) {
                    const int r = i / (run / 4), q = i - r * (run / 4);
                    const float* g = a.Wfc + (size_t)(r0 + r) * 3136 + c0 + 4 * q;
                    if (al16) v[u] = *reinterpret_cast<const float4*>(g);
                    else v[u] = make_float4(g[0], g[1], g[2], g[3]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256 + threadIdx.x;
                if (i < total4) {
                    const int r = i / (run / 4), q = i - r * (run / 4);
                    float* d = zn_lds + r * pitch + 4 * q;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
        __syncthreads();
        const int lane = threadIdx.x >> 2, e0 = (threadIdx.x & 3) * 2;          // the thread's lane of the unit and its element pair
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll 7
        for (int hw = 0; hw < 49; ++hw) {
            float x0, x1;
            long long sj;
            if (piece == 4) {              // B[n][(hw, c)]: row li of the slice, channels 8 lh + e of the block's 16
                const float* t = zn_lds + li * pitch + (8 * lh + e0) * 49 + hw;
                x0 = t[0]; x1 = t[49];
                sj = (long long)(4 * hw + (b & 3)) * 16 + (b >> 2);
            } else {                       // B[(hw, c)][n]: channel li of the block's 32, rows 8 lh + e of the slice
                const float* t = zn_lds + (8 * lh + e0) * pitch + li * 49 + hw;
                x0 = t[0]; x1 = t[pitch];
                sj = (long long)(b >> 1) * 98 + 2 * hw + (b & 1);
            }
            if constexpr (SPLIT) zpack_store2_h(outp, sj, lane, e0, x0, x1, sc);
            else zpack_store2(outp, sj, lane, e0, x0, x1);
        }
        return;
    }
    constexpr int kN[4] = {64, 64, 64, 128}, kK[4] = {512, 576, 576, 256};
    const int N = kN[piece], K = kK[piece], ntiles = N / 32;
    const long long idx = (long long)(blockIdx.x - a.first[piece]) * 256 + threadIdx.x;      // (s, j, lane, e)
    if (idx >= (long long)(K / 16) * ntiles * 512) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long long sj = idx >> 9;
    const int j = (int)(sj % ntiles), s = (int)(sj / ntiles);
    const int n = 32 * j + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
    float x;
    if (piece == 0) {                      // Bt[cout][(r, c, cin)] = W2[cout][cin][r][c]
        const int r = k >> 7, c = (k >> 5) & 3, ch = k & 31;
        x = a.W2[((n * 32 + ch) * 4 + r) * 4 + c];
    } else if (piece == 1) {               // ... = W3[cout][cin][r][c]
        const int r = k / 192, rem = k - r * 192, c = rem >> 6, ch = rem & 63;
        x = a.W3[((n * 64 + ch) * 3 + r) * 3 + c];
    } else if (piece == 2) {               // Bt[cin][(r, c, cout)] = W3[cout][cin][2 - r][2 - c]
        const int r = k / 192, rem = k - r * 192, c = rem >> 6, co = rem & 63;
        x = a.W3[((co * 64 + n) * 3 + (2 - r)) * 3 + (2 - c)];
    } else {                               // Bt[cls = 2 ph + pw][cin][(r, c, cout)] = W2[cout][cin][ph + 2 - 2 r][pw + 2 - 2 c]
        const int cls = n >> 5, ci = n & 31, r = k >> 7, c = (k >> 6) & 1, co = k & 63;
        x = a.W2[((co * 32 + ci) * 4 + ((cls >> 1) + 2 - 2 * r)) * 4 + ((cls & 1) + 2 - 2 * c)];
    }
    if constexpr (SPLIT) zpack_store_h(outp, sj, lane, e, x, sc);
    else zpack_store(outp, sj, lane, e, x);
}

// Item t of `items` sits behind MFMA number ((t + 1) * span) / items - 1 of a k-step (distinct slots for span >= items).
constexpr int z_item_at(int g, int items, int span) {
    for (int t = 0; t < items; ++t)
        if (((t + 1) * span) / items - 1 == g) return t;
    return -1;
}

// ---- row geometries: where row m of A starts, and where k-step s sits relative to that ----------------------------------
// GEMM rows: row m at m * lda floats, k-step s at 16 s floats.
struct ZRowsLinear {
    static constexpr bool CONV = false, PAD = false, CLS = false;
    static constexpr int K = 0, SPR = 1, PITCHB = 0, PER_IMG = 1, GX = 1, S = 1, OFF = 0, H = 1, W = 1, C = 16, KH = 1, KW = 1, DH = 1, DW = 1,
                         DC = 1, DM = 1;
};
// Convolution rows (FixedGeom's parameters, conv.hip): source (images, H, W, C), window KH x KW, grid GY x GX per image with
// stride S and origin offset OFF (negative: zero padding); destination (images, DH, DW, DC), grid pixel -> destination pixel
// (gy * DM, gx * DM) (+ the class offset of Z_MASK_CLS4).
//
// Border classes (AX, the data gradients): a grid coordinate g reaches taps t with 0 <= g * S + OFF + t < extent only.  The
// coordinates with the same valid tap range [T0, T1) form a class (G0 .. G0 + NG - 1); a (row class, column class) pair is a
// border class of the grid.  With AX the rows of the GEMM are ordered class by class -- row r of class (cy, cx) = image
// r / (NG[cy] NG[cx]), pixel r % (NG[cy] NG[cx]) of the class's rectangle -- so that a wave's 64 rows share ONE valid tap
// window and the k-loop runs over the valid taps only: no MFMA multiplies a padding zero (the layer-3 data gradient's padded
// windows hold 1.65 x its valid taps, the layer-2 one's 1.23 x).
struct ZAxisDgrad3 {        // 9 positions, 3 taps, origin offset -2, source extent 7
    static constexpr int NC = 5, G0[5] = {0, 1, 2, 7, 8}, NG[5] = {1, 1, 5, 1, 1}, T0[5] = {2, 1, 0, 0, 0}, T1[5] = {3, 3, 3, 2, 1};
};
struct ZAxisDgrad2 {        // 10 positions, 2 taps, origin offset -1, source extent 9
    static constexpr int NC = 3, G0[3] = {0, 1, 9}, NG[3] = {1, 8, 1}, T0[3] = {1, 0, 0}, T1[3] = {2, 2, 1};
};
template <class AX>
constexpr bool z_axis_ok(int G, int KT, int S, int OFF, int EXT) {      // the tables above against the definition
    int covered = 0;
    for (int c = 0; c < AX::NC; ++c)
        for (int g = AX::G0[c]; g < AX::G0[c] + AX::NG[c]; ++g, ++covered)
            for (int t = 0; t < KT; ++t)
                if ((g * S + OFF + t >= 0 && g * S + OFF + t < EXT) != (t >= AX::T0[c] && t < AX::T1[c])) return false;
    return covered == G;
}
struct ZNoAxis {};

template <int H_, int W_, int C_, int KH_, int KW_, int GY_, int GX_, int S_, int OFF_, int DH_, int DW_, int DC_, int DM_, class AX_ = ZNoAxis>
struct ZRowsConv {
    using AX = AX_;
    static constexpr bool CLS = !std::is_same_v<AX_, ZNoAxis>;
    static constexpr bool CONV = true, PAD = OFF_ < 0 && !CLS;
    static constexpr int H = H_, W = W_, C = C_, KH = KH_, KW = KW_, GY = GY_, GX = GX_, S = S_, OFF = OFF_, DH = DH_, DW = DW_,
                         DC = DC_, DM = DM_;
    static constexpr int RUN = KW * C, SPR = RUN / 16, K = KH * RUN, PITCHB = W * C * 4, PER_IMG = GY * GX;
    static_assert(C % 16 == 0 && KH * KW <= 32, "a k-step lies inside one tap; the validity mask is 32 bits");
};
using ZConv2 = ZRowsConv<20, 20, 32, 4, 4, 9, 9, 2, 0, 9, 9, 64, 1>;
using ZConv3 = ZRowsConv<9, 9, 64, 3, 3, 7, 7, 1, 0, 7, 7, 64, 1>;
using ZDgrad3 = ZRowsConv<7, 7, 64, 3, 3, 9, 9, 1, -2, 9, 9, 64, 1, ZAxisDgrad3>;        // source = dz3, destination = da2
using ZDgrad2 = ZRowsConv<9, 9, 64, 2, 2, 10, 10, 1, -1, 20, 20, 32, 2, ZAxisDgrad2>;    // source = dz2, destination = da1 (4 parity classes = 4 column tiles)
static_assert(z_axis_ok<ZAxisDgrad3>(9, 3, 1, -2, 7) && z_axis_ok<ZAxisDgrad2>(10, 2, 1, -1, 9), "border-class tables");

// Order in which the forward convolutions walk their k-steps: visited index v -> k-step (tap row * SPR + 16-float chunk of the tap row;
// also the step's position in the pack).  A source line is wanted by every (output pixel, tap) pair that lands on it -- 4 pairs in
// layer 2 (4 x 4 taps, stride 2: the taps of equal row and column parity), 9 in layer 3 -- and in plain (tap row, tap column) order
// those requests lie 4 to 24 k-steps apart while the 256 waves of an XCD stream 1 MB per k-step through its 4 MB L2: the layer-2 /
// layer-3 forwards fetched their input 2.0 / 2.2 times, the layer-2 one at the fabric's copy rate (profiles/traffic.json).  Hence
// PHASES: all taps that share source lines are walked back to back -- layer 2: (tap-row parity, tap-column parity) = 4 phases of
// 2 x 2 taps x 2 chunks; layer 3: the two 128-byte lines of a source pixel = 2 phases of 3 x 3 taps x 2 chunks -- so that a phase
// re-requests ONE set of lines (10 KB per wave, 2.6 MB per XCD) every other step.  The two chunks of a line stay adjacent: a k-step
// pair (the B ring's unit) is two consecutive steps of the pack.
template <class RG>
__device__ __forceinline__ int z_kstep(int v) {
    if constexpr (RG::CONV && !RG::CLS && RG::KH == 4 && RG::KW == 4 && RG::S == 2 && RG::C == 32) {
        const int g = v >> 3, w = v & 7;                                        // phase (row parity, column parity); step in the phase
        const int ty = (g >> 1) + 2 * (w >> 2), tx = (g & 1) + 2 * ((w >> 1) & 1);
        return ty * RG::SPR + 2 * tx + (w & 1);
    } else if constexpr (RG::CONV && !RG::CLS && RG::KH == 3 && RG::KW == 3 && RG::S == 1 && RG::C == 64) {
        const int lp = v >= 18 ? 1 : 0, w = v - 18 * lp, combo = w >> 1;        // which line of the pixel; (tap row, tap column)
        const int ty = combo / 3, tx = combo - 3 * ty;
        return ty * RG::SPR + 4 * tx + 2 * lp + (w & 1);
    } else {
        return v;
    }
}

// Border classes of a grid in the order the kernel walks them: heaviest first (pixels x valid taps), so that the short tiles
// of the corner classes fill the tail of the launch.
template <class AX>
struct ZClassOrder {
    int c[AX::NC * AX::NC];
    constexpr ZClassOrder() : c{} {
        constexpr int n = AX::NC * AX::NC;
        auto weight = [](int k) { return AX::NG[k / AX::NC] * AX::NG[k % AX::NC] * (AX::T1[k / AX::NC] - AX::T0[k / AX::NC]) * (AX::T1[k % AX::NC] - AX::T0[k % AX::NC]); };
        for (int i = 0; i < n; ++i) c[i] = i;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j)
                if (weight(c[j]) > weight(c[i])) { const int t = c[i]; c[i] = c[j]; c[j] = t; }
    }
};
// wave tiles of a class-ordered launch: PER_IMG per group of `rows` images
template <class RG>
static long long z_class_tiles(long long images, int rows) { return (images + rows - 1) / rows * RG::PER_IMG; }

struct ZArgs {
    const void* A;              // source tensor
    unsigned a_bytes;           // its size (buffer range check)
    int lda;                    // GEMM rows: leading dimension of A in floats
    const unsigned char* pack;  // B, pre-split (zpack_kernel)
    const float* bias;          // Z_BIAS_RELU
    const float* mask;          // Z_MASK*: C is zeroed where mask <= 0 (same indexing as C)
    const unsigned* bits_in;    // Z_MASKB*: the same mask as bits (word w, bit b = element 32 w + b of C's flat tensor)
    unsigned* bits_out;         // Z_BIAS_RELU_BITS: (C > 0) as bits, written beside C
    float* C;
    unsigned c_bytes;           // size of C (and of the mask, which has C's shape): buffer range check
    int ldc;                    // GEMM rows: leading dimension of C; convolution rows: unused (DC)
    long long M;                // rows: matrix rows, or images * GY * GX
    long long images;           // convolution rows: images
    int N, K;
    unsigned m8, m16;           // 0xffff0000, 0xffffff00: in SGPRs (as literals every v_and would be an 8-byte instruction)
    int steps_per;              // Z_RAW: k-steps per K split (blockIdx.z); 0 otherwise
    int super_rows;             // GEMM rows, stacked waves: row blocks per supertile of the workgroup order (0: launch order)
    const unsigned* a_amax;     // SPLIT = 1 (two-term f16 split, f16split.h): amax record of A; B's maximum sits in the pack's header
    unsigned* c_amax;           // SPLIT = 1: amax record of C to fold the epilogue's values into, or null
};

// WAVES_N: the waves of a workgroup sit side by side (they read the same A rows) instead of on top of each other (they stream
// the same B fragments).
// OCC: waves per SIMD the kernel is compiled for (of one workgroup or of several per CU).
// BLDS: the stacked waves of a workgroup (WAVES_N = false) multiply the SAME B fragments.  Every wave streaming them for itself
//   kept the vector-memory front end busier than the matrix pipe (6 of the 10 KiB a 64 x 64 wave tile pulls per k-step are B:
//   TA 0.74 - 0.83 busy, matrix pipe 0.34 - 0.55, profiles/r03_pmc_*.csv; with the B loads compiled out the layer-2 forward ran
//   1,133 -> 845 us).  With BLDS the workgroup fetches the B of a k-step PAIR once -- every wave 1 / NWAVES of its 2 NT x 3 KiB
//   pieces, two pairs ahead -- into a two-slot LDS ring, and all waves read their fragments from there (ds_read_b128, lane-
//   linear: conflict-free): one s_barrier per k-step pair, B's share of the L1 traffic / NWAVES.  The waves of a workgroup must
//   walk the same k-steps: no wave leaves early (rows past the batch are clamped and their stores dropped), and with border
//   classes the NWAVES waves take the SAME class tile of NWAVES consecutive image groups.  K / 16 must be even.
// SPLIT: 0 = three bf16 terms per operand, NP = 6 (or 9) term pairs on v_mfma_f32_32x32x16_bf16 (bf16split.h);
//        1 = two f16 terms per operand under per-tensor power-of-two scales, NP = 3 (or 4) pairs on v_mfma_f32_32x32x16_f16
//            (f16split.h): half the matrix instructions, 10 instead of 22 split instructions per four elements.  The pack is then
//            [64-byte header][k-step][tile][hi, lo][64 lanes][8 f16], the accumulators are un-scaled in the epilogue.
// (Measured and rejected for SPLIT = 1, each bit-identical: A's global loads issued TWO k-steps ahead of their LDS write -- layer-2 forward
//  -25 us, the data gradients +15 us, bench +-0: profiles/r05_kernel_z_prefetch_two_steps_ab.jsonl; three waves per SIMD for the ring
//  kernels (168 VGPRs, 17 - 26 spilled): 1.7 x slower, profiles/r05_tile_shape_experiments.txt.)
template <class RG, int MT, int NT, int NWAVES, int EPI, bool WAVES_N, int NP, int OCC, bool BLDS, int SPLIT>
__global__ __launch_bounds__(64 * NWAVES) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void z_kernel(ZArgs a) {
    constexpr int ROWS = 32 * MT, LOADS = ROWS / 16;
    constexpr int TERMS = SPLIT ? 2 : 3, kTile = TERMS * 1024;                        // one 32-column tile of one k-step in the pack
    constexpr int kPairPieces = 2 * NT * TERMS, kPairBytes = kPairPieces * 1024, kShare = kPairPieces / NWAVES;      // B of a k-step pair
    static_assert(!BLDS || (!WAVES_N && kPairPieces % NWAVES == 0), "B ring: stacked waves, whole KiB pieces per wave");
    __shared__ __attribute__((aligned(16))) float lds[NWAVES * ROWS * kZPitch + (BLDS ? 2 * kPairBytes / 4 : 0)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // (the wave index in an SGPR)
    const int li = lane & 31, lh = lane >> 5;
    const int N = a.N;
    // SPLIT: the operands' scales -- A's from its amax record, B's from the pack header (the load's latency hides under the prologue's)
    float sa = 1.0f, un = 1.0f;
    if constexpr (SPLIT) {
        const int ea = f16_scale_exp(amax_load(a.a_amax, lane));
        const int eb = f16_scale_exp(*reinterpret_cast<const unsigned*>(a.pack));
        sa = f16_pow2(ea);
        un = f16_unscale(ea, eb);
    }
    // Workgroup -> (column block, row block).  The convolutions re-read their source through the L2 (windows overlap; the
    // border classes of an image group share its taps), and every XCD has its own L2: workgroups are dealt to the XCDs round
    // robin in launch order, so XCD x runs the CONTIGUOUS range [start_x, start_x + count_x) of the logical order -- neighbours in
    // that order (the column blocks of one row block, then the next row block) meet in one L2 at about the same time.
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if constexpr (RG::CONV) {
        const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y;
        const unsigned x = L & 7u, q = total >> 3, rem = total & 7u;
        const unsigned logical = x * q + (x < rem ? x : rem) + (L >> 3);
        bx = logical % gridDim.x;
        by = logical / gridDim.x;
    } else if constexpr (!WAVES_N && EPI != Z_RAW) {
        // GEMM rows with stacked waves (the FC data gradient: 25 column blocks x M / 256 row blocks, K = 512): in launch order the
        // column blocks of a row block land on all eight XCDs, so every L2 streams the whole 9.6 MB of B (and all the A rows in
        // flight) once per ~10 row blocks -- 1.8 GB of L2 misses per launch for 0.49 GB of algorithmic bytes
        // (profiles/traffic.json).  Instead: XCD x takes a CONTIGUOUS range of the logical order, and the logical order walks
        // supertiles of `super_rows` row blocks x all column blocks, row block fastest: the supertile's A rows (super_rows x
        // 512 KB) stay in the L2 while B streams through once per supertile.
        if (a.super_rows > 0) {
            const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y;
            const unsigned x = L & 7u, q = total >> 3, rem = total & 7u;
            const unsigned logical = x * q + (x < rem ? x : rem) + (L >> 3);
            const unsigned per = (unsigned)a.super_rows * gridDim.x;
            const unsigned sup = logical / per, r = logical - sup * per;
            const unsigned left = gridDim.y - sup * (unsigned)a.super_rows;
            const unsigned rows = left < (unsigned)a.super_rows ? left : (unsigned)a.super_rows;      // (the last supertile may be short)
            bx = r / rows;
            by = sup * (unsigned)a.super_rows + (r - bx * rows);
        }
    }
    const int n0 = WAVES_N ? (bx * NWAVES + wave) * (32 * NT) : bx * (32 * NT);
    long long wtile = WAVES_N ? (long long)by : (long long)by * NWAVES + wave;        // this wave's row tile
    // Border classes: rows run group by group -- a group = ROWS images = PER_IMG tiles: class (cy, cx) owns NG[cy] NG[cx] of them,
    // heaviest class first -- so that all classes of an image are multiplied while its taps sit in the L2.  Row r of a class's
    // tiles in a group = image r / npix of the group, pixel r % npix of the class rectangle.
    int c_gy0 = 0, c_gx0 = 0, c_nx = 1, c_npix = 1, c_ty0 = 0, c_nty = 1, c_us0 = 0, c_spr = 1;
    unsigned c_mg_npix = 0, c_mg_nx = 0;                  // reciprocals for r / npix, p / nx (exact for r * d < 2^32)
    long long c_img0 = 0;                                 // first image of the group
    if constexpr (RG::CLS) {
        using AX = typename RG::AX;
        constexpr ZClassOrder<AX> ord{};
        long long grp = wtile / RG::PER_IMG;
        int t = (int)(wtile - grp * RG::PER_IMG), cls = ord.c[0];
        if constexpr (BLDS) {                             // the waves of a workgroup: tile t of NWAVES consecutive groups
            const unsigned quad = by / (unsigned)RG::PER_IMG;
            t = (int)(by - quad * (unsigned)RG::PER_IMG);
            grp = (long long)quad * NWAVES + wave;
        }
        c_img0 = grp * ROWS;
        if constexpr (!BLDS)
            if (c_img0 >= a.images) return;               // (BLDS: a group past the batch multiplies the last image, stores dropped)
        for (int i = 0; i < AX::NC * AX::NC; ++i) {
            cls = ord.c[i];
            const int np = AX::NG[cls / AX::NC] * AX::NG[cls % AX::NC];
            if (t < np) break;
            t -= np;
        }
        wtile = t;                                        // the tile's index among the class's tiles of this group
        const int cy = cls / AX::NC, cx = cls % AX::NC;
        c_gy0 = AX::G0[cy]; c_gx0 = AX::G0[cx]; c_nx = AX::NG[cx]; c_npix = AX::NG[cy] * c_nx;
        c_ty0 = AX::T0[cy]; c_nty = AX::T1[cy] - c_ty0;
        c_us0 = AX::T0[cx] * (RG::C / 16); c_spr = (AX::T1[cx] - AX::T0[cx]) * (RG::C / 16);
        c_mg_npix = c_npix > 1 ? (unsigned)((1ull << 32) / (unsigned)c_npix + 1) : 0u;
        c_mg_nx = c_nx > 1 ? (unsigned)((1ull << 32) / (unsigned)c_nx + 1) : 0u;
    }
    const long long M = RG::CLS ? (long long)ROWS * c_npix : a.M;       // (border classes: the class's rows in this group, whole tiles)
    const long long m0 = wtile * ROWS;
    if constexpr (!BLDS)
        if (m0 >= M || n0 >= N) return;                   // (whole wave; no barriers without BLDS)
    // class row r -> (image, grid y, grid x); images past the batch (last group) report ok = false and are clamped
    auto cls_pixel = [&](unsigned r, unsigned& img, int& gy, int& gx) -> bool {  // (selects, no branches: d == 1 has no 32-bit reciprocal)
        const unsigned q = __umulhi(r, c_mg_npix);
        const unsigned loc = c_npix > 1 ? q : r;
        const unsigned p = r - loc * (unsigned)c_npix;
        const unsigned q2 = __umulhi(p, c_mg_nx);
        const unsigned py = c_nx > 1 ? q2 : p;
        gy = c_gy0 + (int)py;
        gx = c_gx0 + (int)(p - py * (unsigned)c_nx);
        const long long im = c_img0 + loc;
        const bool ok = im < a.images;
        img = (unsigned)(ok ? im : a.images - 1);
        return ok;
    };
    const int ntiles = (N + 31) / 32, j0 = n0 / 32;
    float* const wl = lds + wave * (ROWS * kZPitch);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)a.a_bytes, kZRsrcWord3);
    // coalesced A loads: load u of a k-step reads row m0 + 16 u + (lane >> 2), floats 4 (lane & 3) .. + 3 of the step's 16.
    // voff = byte offset of that row's k = 0 (+ the lane's 16-byte chunk); vm = validity bits (tap row * KW + tap column).
    unsigned voff[LOADS], vm[LOADS];
#pragma unroll
    for (int u = 0; u < LOADS; ++u) {
        const long long r = m0 + 16 * u + (lane >> 2);
        const long long rc = r < M ? r : M - 1;           // rows past M re-read the last row: their results are dropped at the store
        if constexpr (RG::CONV) {
            unsigned img;
            int gy, gx;
            if constexpr (RG::CLS) {
                cls_pixel((unsigned)rc, img, gy, gx);
            } else {
                img = (unsigned)(rc / RG::PER_IMG);
                const unsigned rem = (unsigned)(rc - (long long)img * RG::PER_IMG);
                gy = (int)(rem / RG::GX);
                gx = (int)(rem - (rem / RG::GX) * RG::GX);
            }
            const int sy0 = gy * RG::S + RG::OFF, sx0 = gx * RG::S + RG::OFF;
            voff[u] = (unsigned)((((int)img * RG::H + sy0) * RG::W + sx0) * RG::C * 4) + 16u * (unsigned)(lane & 3);   // may wrap for PAD taps
            unsigned m = 0u;
            if constexpr (RG::PAD) {
#pragma unroll
                for (int ty = 0; ty < RG::KH; ++ty)
#pragma unroll
                    for (int tx = 0; tx < RG::KW; ++tx)
                        if (sy0 + ty >= 0 && sy0 + ty < RG::H && sx0 + tx >= 0 && sx0 + tx < RG::W) m |= 1u << (ty * RG::KW + tx);
            }
            vm[u] = m;
        } else {
            voff[u] = (unsigned)rc * (unsigned)a.lda * 4u + 16u * (unsigned)(lane & 3);          // A < 4 GiB (host-checked)
            vm[u] = 0u;
        }
    }
    float* const wr_ptr = wl + (lane >> 2) * kZPitch + 4 * (lane & 3);               // + 16 u rows
    const float* const rd_ptr = wl + li * kZPitch + 8 * lh;                          // + 32 i rows
    const unsigned char* const pb = a.pack + (SPLIT ? kF16PackHeader : 0) + (size_t)j0 * kTile + 16 * lane;
    const size_t step_bytes = (size_t)ntiles * kTile;
    const unsigned m8 = a.m8, m16 = a.m16;

    z_f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    s_u32x4 stage[LOADS];                                 // A of a later k-step, as loaded (coalesced layout), on its way to LDS
    s_u32x4 raw[MT][2];                                   // fragments of the NEXT k-step as read back from LDS (f32, lane = row)
    unsigned ta[2][MT][TERMS][4];                         // split A fragments: [k-step parity][fragment][term][4 x 2 bf16 / f16]
    s_u32x4 tb[2][NT][TERMS];                             // B fragments straight from the pack
    const int k0 = EPI == Z_RAW ? (int)blockIdx.z * a.steps_per : 0;              // first k-step of this K split
    const int nsteps = RG::CLS ? c_nty * c_spr : RG::CONV ? RG::K / 16 : EPI == Z_RAW ? ((a.K >> 4) - k0 < a.steps_per ? (a.K >> 4) - k0 : a.steps_per) : a.K >> 4;
    auto kclamp = [&](int s) { return s < nsteps ? s : nsteps - 1; };                // past the end: re-read, never multiplied
    // border classes: the k-steps of a tile are the chunks us0 .. us0 + spr - 1 of tap rows ty0 .. ty0 + nty - 1; the loads of A
    // and of B each walk them with their own cursor (the loads are issued in step order; past the end a cursor stays put)
    struct Cursor { int n, ty, us; };
    Cursor ca{0, c_ty0, c_us0}, cb{0, c_ty0, c_us0};
    auto advance = [&](Cursor& c) {
        if (c.n + 1 < nsteps) {
            ++c.n;
            if (++c.us == c_us0 + c_spr) { c.us = c_us0; ++c.ty; }
        }
    };
    auto load_a_into = [&](s_u32x4 (&stage)[LOADS], int s) {
        const int sc = kclamp(s);
        if constexpr (RG::CLS) {
            const unsigned off = (unsigned)(ca.ty * RG::PITCHB + ca.us * 64);       // every tap of the class window is valid for every row
#pragma unroll
            for (int u = 0; u < LOADS; ++u) stage[u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[u] + off, 0, 0));
            advance(ca);
        } else if constexpr (RG::CONV) {
            const int sm = z_kstep<RG>(sc);
            const int ty = sm / RG::SPR, us = sm - ty * RG::SPR;                     // (uniform: scalar unit)
            const int off = ty * RG::PITCHB + us * 64;
            if constexpr (RG::PAD) {
                const int tap = ty * RG::KW + (us * 16) / RG::C;
#pragma unroll
                for (int u = 0; u < LOADS; ++u) {
                    const unsigned vo = ((vm[u] >> tap) & 1u) ? voff[u] + (unsigned)off : kZOob;
                    stage[u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0));
                }
            } else {
#pragma unroll
                for (int u = 0; u < LOADS; ++u) stage[u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[u], off, 0));
            }
        } else {
#pragma unroll
            for (int u = 0; u < LOADS; ++u) stage[u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[u], (k0 + sc) * 64, 0));
        }
    };
    auto load_a = [&](int s) { load_a_into(stage, s); };
    auto load_b = [&](int par, int s) {
        const unsigned char* p = pb + (size_t)(RG::CLS ? cb.ty * RG::SPR + cb.us : k0 + z_kstep<RG>(kclamp(s))) * step_bytes;
        if constexpr (RG::CLS) advance(cb);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bool ok = j0 + j < ntiles;                                        // (wave-uniform) tiles past N: re-read tile j0
#pragma unroll
            for (int t = 0; t < TERMS; ++t) tb[par][j][t] = *reinterpret_cast<const s_u32x4*>(p + (ok ? j : 0) * kTile + t * 1024);
        }
    };
    // ---- BLDS: the workgroup's B ring.  Piece x of a pair = (k-step half h, column tile j, term t), x = (h NT + j) TERMS + t, one KiB
    // each; wave w fetches pieces w kShare .. + kShare - 1.  `bcur` / `bnxt` = byte offsets of the two slots.
    unsigned char* const ring = reinterpret_cast<unsigned char*>(lds + NWAVES * ROWS * kZPitch);
    s_u32x4 bst[BLDS ? kShare : 1];
    size_t bsrc[BLDS ? kShare : 1];                       // where this wave's pieces sit relative to the pair's first k-step
    unsigned bcur = 0, bnxt = kPairBytes;
    Cursor cp{0, c_ty0, c_us0};                           // first k-step of the next pair to fetch (stays on the last pair)
    if constexpr (BLDS) {
#pragma unroll
        for (int u = 0; u < kShare; ++u) {
            const int x = wave * kShare + u, h = x / (NT * TERMS), jt = x - h * (NT * TERMS), j = jt / TERMS;
            const int jt_ok = j0 + j < ntiles ? jt : jt - TERMS * j;                 // column tiles past N: re-read tile j0 (never stored)
            bsrc[u] = (size_t)h * step_bytes + (size_t)jt_ok * 1024;
        }
    }
    auto load_bpair = [&]() {
        const unsigned char* p = pb + (size_t)(RG::CLS ? cp.ty * RG::SPR + cp.us : z_kstep<RG>(cp.n)) * step_bytes;      // (a pair = two consecutive steps of the pack)
#pragma unroll
        for (int u = 0; u < kShare; ++u) bst[u] = *reinterpret_cast<const s_u32x4*>(p + bsrc[u]);
        if (cp.n + 2 < nsteps) {
            cp.n += 2;
            cp.us += 2;
            if (cp.us == c_us0 + c_spr) { cp.us = c_us0; ++cp.ty; }
        }
    };
    auto write_bpair = [&](unsigned slot) {
#pragma unroll
        for (int u = 0; u < kShare; ++u) *reinterpret_cast<s_u32x4*>(ring + slot + (wave * kShare + u) * 1024 + 16 * lane) = bst[u];
    };
    auto read_b = [&](int par, unsigned slot, int h) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < TERMS; ++t) tb[par][j][t] = *reinterpret_cast<const s_u32x4*>(ring + slot + ((h * NT + j) * TERMS + t) * 1024 + 16 * lane);
    };
    auto ring_barrier = [&]() {                           // this wave's ring writes have landed; then every wave's
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto to_lds_from = [&](const s_u32x4 (&stage)[LOADS]) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) *reinterpret_cast<s_u32x4*>(wr_ptr + 16 * u * kZPitch) = stage[u];
    };
    auto to_lds = [&]() { to_lds_from(stage); };
    auto read_frags = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
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
    // SPLIT = 1: the f16 split of one half fragment is ONE piece of 10 VALU instructions (f16split.h)
    auto split_piece_h = [&](int par, auto ic, auto hc) {
        constexpr int i = decltype(ic)::value, hf = decltype(hc)::value;
        unsigned hi[2], lo[2];
        f16_split4(raw[i][hf], sa, hi, lo);
        ta[par][i][0][2 * hf] = hi[0]; ta[par][i][0][2 * hf + 1] = hi[1];
        ta[par][i][TERMS - 1][2 * hf] = lo[0]; ta[par][i][TERMS - 1][2 * hf + 1] = lo[1];
    };
    // One pipeline step of parity q.  Nothing a step computes depends on a load or LDS access of the SAME step:
    //   B of step s + 1 (global)                      -> tb[q ^ 1]
    //   `raw` = fragments of step s + 1 (read from LDS at the end of the previous step) -> split -> ta[q ^ 1]
    //   the MFMAs of step s on (ta[q], tb[q])
    //   `stage` = A of step s + 2 (loaded during the previous step) -> LDS;  A of step s + 3 (global) -> `stage`;
    //   fragments of step s + 2: LDS -> `raw`
    // The instruction order is pinned by hand (hipcc's sched_group_barrier solver did not reproduce this pipeline: it left the
    // split in front of the MFMAs and chained MFMAs on one accumulator): behind MFMA g of the step's NM comes the item
    // z_item_at(g) -- one of the 6 MT split pieces, then the LDS writes, the A loads, the LDS reads -- and a sched_barrier; the
    // last kTail MFMAs run bare and cover the LDS round trip.  Term pairs outermost, the MT x NT independent tiles innermost: no
    // MFMA waits for the one before it.
    // BLDS: two more items in the even steps -- this wave's pieces of pair p + 1 to the ring (loaded two steps earlier), its pieces
    // of pair p + 2 from global memory -- and the odd steps open with the ring barrier.
    constexpr int NM = NP * MT * NT, kPieces = (SPLIT ? 2 : 6) * MT, kItems = kPieces + 3 + (BLDS ? 2 : 0), kTail = NM > 2 * kItems ? NM / 4 : NM - kItems;
    static_assert(NM - kTail >= kItems, "one MFMA per scheduled item");
    static_assert(SPLIT ? (NP == 3 || NP == 4) : (NP == 6 || NP == 9), "term pairs of the split");
    constexpr int PX[9] = {0, 0, 1, SPLIT ? 1 : 0, 2, 1, 1, 2, 2}, PY[9] = {0, 1, 0, SPLIT ? 1 : 2, 0, 1, 2, 1, 2};     // pairs by weight: bf16 -- the first six have x + y <= 2; f16 -- hi hi, hi lo, lo hi (, lo lo)
    auto step = [&](auto qc, int s) {
        constexpr int q = decltype(qc)::value;
        if constexpr (BLDS) {
            if constexpr (q == 1) ring_barrier();
            read_b(q ^ 1, q == 0 ? bcur : bnxt, q ^ 1);                            // B of step s + 1: second half of this pair / first of the next
        } else {
            load_b(q ^ 1, s + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int g = G, pi = g / (MT * NT), i = (g / NT) % MT, j = g % NT;
                if constexpr (SPLIT)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(s_f16x8, (s_u32x4){ta[q][i][PX[pi]][0], ta[q][i][PX[pi]][1], ta[q][i][PX[pi]][2], ta[q][i][PX[pi]][3]}),
                        __builtin_bit_cast(s_f16x8, tb[q][j][PY[pi]]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(s_bf16x8, (s_u32x4){ta[q][i][PX[pi]][0], ta[q][i][PX[pi]][1], ta[q][i][PX[pi]][2], ta[q][i][PX[pi]][3]}),
                        __builtin_bit_cast(s_bf16x8, tb[q][j][PY[pi]]), acc[i][j], 0, 0, 0);
                constexpr int t = z_item_at(g, kItems, NM - kTail);                 // the item behind MFMA g, or -1
                if constexpr (t >= 0) {
                    if constexpr (t < kPieces && SPLIT)
                        split_piece_h(q ^ 1, std::integral_constant<int, t / 2>{}, std::integral_constant<int, t % 2>{});
                    else if constexpr (t < kPieces)
                        split_piece(q ^ 1, std::integral_constant<int, t / 6>{}, std::integral_constant<int, (t / 3) % 2>{}, std::integral_constant<int, t % 3>{});
                    else if constexpr (t == kPieces) to_lds();
                    else if constexpr (t == kPieces + 1) load_a(s + 3);
                    else if constexpr (t == kPieces + 2) read_frags();
                    else if constexpr (t == kPieces + 3) { if constexpr (q == 0) write_bpair(bnxt); }
                    else { if constexpr (q == 0) load_bpair(); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, NM>{});
    };
    auto split_all = [&](int par) {
        [&]<int... T>(std::integer_sequence<int, T...>) {
            if constexpr (SPLIT) (split_piece_h(par, std::integral_constant<int, T / 2>{}, std::integral_constant<int, T % 2>{}), ...);
            else (split_piece(par, std::integral_constant<int, T / 6>{}, std::integral_constant<int, (T / 3) % 2>{}, std::integral_constant<int, T % 3>{}), ...);
        }(std::make_integer_sequence<int, kPieces>{});
    };
    // prologue: step 0 split into ta[0] with its B terms in tb[0]; fragments of step 1 in `raw`; A of step 2 in `stage`.  The
    // loads of all three steps go out together (one memory latency per tile instead of three in a row: the accumulators are
    // not live yet, the two extra staging sets cost nothing)
    {
        s_u32x4 st0[LOADS], st1[LOADS];
        load_a_into(st0, 0);
        if constexpr (BLDS) load_bpair();                 // pair 0 -> ring slot 0
        else load_b(0, 0);
        load_a_into(st1, 1);
        load_a(2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BLDS) {
            write_bpair(bcur);
            load_bpair();                                 // pair 1 stays in registers until step 0
            ring_barrier();
            read_b(0, bcur, 0);
        }
        to_lds_from(st0);
        read_frags();
        split_all(0);
        __builtin_amdgcn_sched_barrier(0);
        to_lds_from(st1);
        read_frags();
        __builtin_amdgcn_sched_barrier(0);
    }
    int s = 0;
    for (; s + 2 <= nsteps; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s + 1);
        if constexpr (BLDS) { const unsigned t = bcur; bcur = bnxt; bnxt = t; }
    }
    if constexpr (!BLDS)
        if (s < nsteps) step(std::integral_constant<int, 0>{}, s);

    // ---- epilogue: accumulator element e of tile (i, j) is C[row m0 + 32 i + (e & 3) + 8 (e >> 2) + 4 lh][column n0 + 32 j + li].
    // Byte offset of (row m, column tile j): GEMM rows and the convolutions whose destination pixel IS the row (channels last,
    // DC = N): (m * ldc + n) * 4; Z_MASK_CLS4 (layer-2 data gradient): column tile j = stride-parity class (j >> 1, j & 1) of
    // grid pixel m -> destination pixel (2 gy + (j >> 1), 2 gx + (j & 1)), channel li.  All accesses go through buffer resources
    // (C and the mask are below 4 GiB, host-checked): rows past M and columns past N get an out-of-range offset -- their loads
    // return 0, their stores are dropped -- so the epilogue has no branches and no per-access waits.
    const unsigned c_bytes = a.c_bytes;
    float* const c_base = EPI == Z_RAW ? a.C + (size_t)blockIdx.z * (size_t)(c_bytes >> 2) : a.C;     // Z_RAW: slab z of the workspace
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(c_base, 0, (int)c_bytes, kZRsrcWord3);
    constexpr bool kCls4 = EPI == Z_MASK_CLS4 || EPI == Z_MASKB_CLS4, kMaskF32 = EPI == Z_MASK || EPI == Z_MASK_CLS4,
                   kMaskBits = EPI == Z_MASKB || EPI == Z_MASKB_CLS4, kBitsOut = EPI == Z_BIAS_RELU_BITS;
    static_assert(!(kMaskBits || kBitsOut) || ROWS == 64, "bit masks: one word per lane = one row of the wave's 64");
    const __amdgpu_buffer_rsrc_t rsrc_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask), 0, kMaskF32 ? (int)c_bytes : 0, kZRsrcWord3);
    const void* const bits_base = kMaskBits ? (const void*)a.bits_in : (const void*)a.bits_out;
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(bits_base), 0, (kMaskBits || kBitsOut) ? (int)(c_bytes >> 5) : 0, kZRsrcWord3);
    const int ldc = RG::CONV ? RG::DC : a.ldc;
    auto row_off = [&](long long m) -> unsigned {
        if constexpr (RG::CLS) {                          // computed for every row, selected at the end: an early exit would put a
            unsigned img;                                 // branch (and its own wait) around every load of the epilogue
            int gy, gx;
            const bool ok = cls_pixel((unsigned)m, img, gy, gx);
            const unsigned off = (unsigned)((((int)img * RG::DH + gy * RG::DM) * RG::DW + gx * RG::DM) * RG::DC) * 4u;
            return ok ? off : kZOob;
        }
        if (m >= M) return kZOob;
        if constexpr (kCls4) {
            const long long img = m / RG::PER_IMG;
            const int rem = (int)(m - img * RG::PER_IMG), gy = rem / RG::GX, gx = rem - gy * RG::GX;
            return (unsigned)((((int)img * RG::DH + gy * RG::DM) * RG::DW + gx * RG::DM) * RG::DC) * 4u;
        } else {
            return (unsigned)m * (unsigned)ldc * 4u;
        }
    };
    // Row offsets of the epilogue.  Border classes: a row's (image, pixel) costs two multiply-high divisions and a dozen selects --
    // computed ONCE per lane for row m0 + lane and fetched per accumulator row with a lane permute (the 32 evaluations per tile were
    // ~700 VALU instructions beside a 312-MFMA k-loop in the layer-2 data gradient).
    unsigned ro_lane = 0u;
    if constexpr (RG::CLS && ROWS == 64) ro_lane = row_off(m0 + lane);
    auto ro_of = [&](int r) -> unsigned {                 // r = row of the tile (its lh-dependent part included)
        if constexpr (RG::CLS && ROWS == 64) return (unsigned)__shfl((int)ro_lane, r, 64);
        else return row_off(m0 + r);
    };
    unsigned coff[NT];                                    // byte offset of this lane's column in tile j (out of range past N)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if constexpr (kCls4) coff[j] = (unsigned)((((j0 + j) >> 1) * RG::DW + ((j0 + j) & 1)) * RG::DC + li) * 4u;
        else coff[j] = n0 + 32 * j + li < N ? (unsigned)(n0 + 32 * j + li) * 4u : kZOob;
    }
    auto at = [&](unsigned ro, int j) -> unsigned { return (ro == kZOob || coff[j] == kZOob) ? kZOob : ro + coff[j]; };
    // SPLIT: the accumulators carry both operands' scales; `fin` removes them (a power of two: exact).  `cmax` collects the bit
    // patterns of |C| for C's amax record -- of every value the wave computed: rows / columns past the edge are re-reads of real rows
    // and columns (their stores are dropped), so they cannot exceed the tensor's maximum.
    auto fin = [&](float x) -> float { if constexpr (SPLIT) return x * un; else return x; };
    float cmax = 0.0f;                                    // (one v_max_f32 with the |x| source modifier per value; a NaN is not recorded)
    auto seen = [&](float v) { if constexpr (SPLIT && EPI != Z_RAW) cmax = __builtin_fmaxf(cmax, __builtin_fabsf(v)); };
    if constexpr (kMaskBits) {
        // lane L holds the mask word of row m0 + L for each of the wave's column tiles: C's element (row, column n) is bit n % 32 of
        // word (byte offset of the element) / 128 -- every 32-column tile of a row starts on a 128-byte boundary of C
        unsigned wm[NT];
        {
            const unsigned ro = (RG::CLS && ROWS == 64) ? ro_lane : row_off(m0 + lane);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                unsigned o = kZOob;
                if constexpr (kCls4) o = ro == kZOob ? kZOob : (ro + (unsigned)((((j0 + j) >> 1) * RG::DW + ((j0 + j) & 1)) * RG::DC) * 4u) >> 5;
                else o = (ro == kZOob || n0 + 32 * j >= N) ? kZOob : (ro + (unsigned)(n0 + 32 * j) * 4u) >> 5;
                wm[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, o, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned ro = ro_of(32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)wm[j], 32 * i + (e & 3) + 8 * (e >> 2));
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)wm[j], 32 * i + (e & 3) + 8 * (e >> 2) + 4);
                    const float v = z_keep_where(fin(acc[i][j][e]), lo, hi);  // lanes 0..31 (lh = 0): bit li of `lo`, lanes 32..63: of `hi`
                    seen(v);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_c, at(ro, j), 0, 0);
                }
            }
    } else if constexpr (kMaskF32) {
        // The mask values are requested before the first one is used -- of the wave's whole block with one wave per SIMD (nothing
        // else hides a load's latency there), of one 32-row tile at a time with two waves per SIMD (registers).
        constexpr int IB = OCC == 1 ? MT : 1;                                       // 32-row tiles per batch
#pragma unroll
        for (int i0 = 0; i0 < MT; i0 += IB) {
            unsigned mk[IB][NT][16];
#pragma unroll
            for (int ib = 0; ib < IB; ++ib)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned ro = ro_of(32 * (i0 + ib) + (e & 3) + 8 * (e >> 2) + 4 * lh);
#pragma unroll
                    for (int j = 0; j < NT; ++j) mk[ib][j][e] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_m, at(ro, j), 0, 0);
                }
#pragma unroll
            for (int ib = 0; ib < IB; ++ib)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned ro = ro_of(32 * (i0 + ib) + (e & 3) + 8 * (e >> 2) + 4 * lh);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float v = __uint_as_float(mk[ib][j][e]) > 0.0f ? fin(acc[i0 + ib][j][e]) : 0.0f;
                        seen(v);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_c, at(ro, j), 0, 0);
                    }
                }
        }
    } else if constexpr (EPI == Z_RAW) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned ro = ro_of(32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh);
#pragma unroll
                for (int j = 0; j < NT; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fin(acc[i][j][e])), rsrc_c, at(ro, j), 0, 0);
            }
    } else {
        float bj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + 32 * j + li;
            bj[j] = a.bias[n < N ? n : N - 1];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int wv[NT];                                   // kBitsOut: lane L (< 32) collects the mask word of row m0 + 32 i + L
#pragma unroll
            for (int j = 0; j < NT; ++j) wv[j] = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned ro = ro_of(32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float v = fin(acc[i][j][e]) + bj[j];
                    if constexpr (SPLIT) v = v < 0.0f ? 0.0f : v;      // (a NaN from an overflowed f16 operand -- an understated amax record -- stays a NaN)
                    else v = v > 0.0f ? v : 0.0f;
                    seen(v);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_c, at(ro, j), 0, 0);
                    if constexpr (kBitsOut) {             // lanes 0..31 of the ballot: the 32 columns of row (e & 3) + 8 (e >> 2); 32..63: of that row + 4
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(v > 0.0f);
                        wv[j] = z_writelane(wv[j], (unsigned)bal, (e & 3) + 8 * (e >> 2));
                        wv[j] = z_writelane(wv[j], (unsigned)(bal >> 32), (e & 3) + 8 * (e >> 2) + 4);
                    }
                }
            }
            if constexpr (kBitsOut) {
                const unsigned ro = lh == 0 ? row_off(m0 + 32 * i + li) : kZOob;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const unsigned o = (ro == kZOob || n0 + 32 * j >= N) ? kZOob : (ro + (unsigned)(n0 + 32 * j) * 4u) >> 5;
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)wv[j], rsrc_b, o, 0, 0);
                }
            }
        }
    }
    if constexpr (SPLIT && EPI != Z_RAW)
        if (a.c_amax) amax_commit(a.c_amax, __float_as_uint(cmax), blockIdx.x + 3u * blockIdx.y + (unsigned)wave, lane);      // (wave-uniform condition)
}

template <class RG, int MT, int NT, int NWAVES, int EPI, bool WAVES_N, int OCC = NWAVES / 4, bool BLDS = false>
static int z_launch(const ZArgs& a, hipStream_t s, const char* what) {
    long long row_blocks = (a.M + 32 * MT - 1) / (32 * MT);
    const long long col_blocks = (a.N + 32 * NT - 1) / (32 * NT);
    if constexpr (RG::CLS) row_blocks = z_class_tiles<RG>(a.M / RG::PER_IMG, 32 * MT);
    dim3 grid = WAVES_N ? dim3((unsigned)((col_blocks + NWAVES - 1) / NWAVES), (unsigned)row_blocks)
                        : dim3((unsigned)col_blocks, (unsigned)((row_blocks + NWAVES - 1) / NWAVES));
    if constexpr (BLDS && RG::CLS) {      // a workgroup = one class tile of NWAVES consecutive image groups
        const long long groups = (a.M / RG::PER_IMG + 32 * MT - 1) / (32 * MT);
        grid.y = (unsigned)((groups + NWAVES - 1) / NWAVES * RG::PER_IMG);
    }
    if constexpr (EPI == Z_RAW) grid.z = (unsigned)(((a.K >> 4) + a.steps_per - 1) / a.steps_per);
    if (grid.y > 65535u) {            // (hardware grid limit) 4.1 M rows at the smallest row block: beyond every caller's sizes
        set_error("%s: %lld rows exceed one launch", what, a.M);
        return MI355PPO_EINVAL;
    }
    if (a.a_amax)         // the two-term f16 split (the *_f16x2 entry points): `pack` is an f16x2 pack
        hipLaunchKernelGGL((z_kernel<RG, MT, NT, NWAVES, EPI, WAVES_N, 3, OCC, BLDS, 1>), grid, dim3(64 * NWAVES), 0, s, a);
    else if (bf16_term_pairs() == 9)
        hipLaunchKernelGGL((z_kernel<RG, MT, NT, NWAVES, EPI, WAVES_N, 9, OCC, BLDS, 0>), grid, dim3(64 * NWAVES), 0, s, a);
    else
        hipLaunchKernelGGL((z_kernel<RG, MT, NT, NWAVES, EPI, WAVES_N, 6, OCC, BLDS, 0>), grid, dim3(64 * NWAVES), 0, s, a);
    return check_launch(what);
}

}  // namespace mi355ppo

using namespace mi355ppo;

// The convolutions share B through the workgroup's LDS ring (BLDS) from 2,048 images on; below, a launch is a single round of
// workgroups and the ring's barriers cost what its saved L1 traffic buys (1,024 images: layer 3 forward 31.5 -> 33.8 us).
// (Round 6: the same-box A/B switch MI355PPO_Z_BLDS is gone; profiles/r03_blds_ab.jsonl holds its runs.)
static bool z_blds(long long images) { return images >= 2048; }

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
    MI355_REQUIRE((long long)M * lda * 4 < (1LL << 32) - 8192 && (long long)M * ldc * 4 < (1LL << 32) - 8192, MI355PPO_EINVAL,
                  "%s: A (%d x %d floats) and C (%d x %d) must stay below 4 GiB (32-bit buffer offsets)", fn, M, lda, M, ldc);
    return MI355PPO_OK;
}

static ZArgs zargs(const void* A, long long a_bytes, int lda, const void* pack, const float* bias, const float* mask, float* C,
                   long long c_bytes, int ldc, long long M, int N, int K, long long images = 0) {
    ZArgs a;
    a.A = A; a.a_bytes = (unsigned)a_bytes; a.lda = lda; a.pack = static_cast<const unsigned char*>(pack); a.bias = bias; a.mask = mask;
    a.bits_in = nullptr; a.bits_out = nullptr;
    a.a_amax = nullptr; a.c_amax = nullptr;
    a.steps_per = 0;
    a.super_rows = 0;
    a.C = C; a.c_bytes = (unsigned)c_bytes; a.ldc = ldc; a.M = M; a.images = images; a.N = N; a.K = K; a.m8 = 0xffff0000u; a.m16 = 0xffffff00u;
    return a;
}

// All six kernel-Z packs and kernel Q's pack of the NatureCNN agent from its parameters in torch's layouts, one launch
// (znature_pack_kernel).  Buffers: mi355ppo_fc_pack_bytes(64, 512), (64, 576), (64, 576), (128, 256), (512, 3136), (3136, 512) and
// mi355ppo_cnn_conv1q_pack_bytes() bytes; a null output skips its piece (W1 / qpack, W2 / its two packs, ... may be absent together).
static int nature_packs_impl(const char* fn, const float* W1, const float* W2, const float* W3, const float* Wfc, void* qpack, void* conv2_fwd,
                             void* conv3_fwd, void* conv3_dgrad, void* conv2_dgrad, void* fc_fwd, void* fc_dgrad, unsigned* w_amax, void* stream) {
    void* outs[6] = {conv2_fwd, conv3_fwd, conv3_dgrad, conv2_dgrad, fc_fwd, fc_dgrad};
    const float* srcs[6] = {W2, W3, W3, W2, Wfc, Wfc};
    static const int kN[6] = {64, 64, 64, 128, 512, 3136}, kK[6] = {512, 576, 576, 256, 3136, 512};
    ZNaturePacks a;
    a.W1 = W1; a.W2 = W2; a.W3 = W3; a.Wfc = Wfc;
    a.qpack = static_cast<unsigned char*>(qpack);
    a.wrec = w_amax;
    for (int p = 0; p < 6; ++p) {
        MI355_REQUIRE(!outs[p] || srcs[p], MI355PPO_EINVAL, "%s: pack %d requested without its weight", fn, p);
        MI355_REQUIRE(aligned(outs[p], 16) && aligned(srcs[p], 4), MI355PPO_EALIGN, "%s: packs must be 16-byte aligned", fn);
        a.out[p] = static_cast<unsigned short*>(outs[p]);
        a.count[p] = !outs[p] ? 0u : p >= 4 ? 64u : (unsigned)((long long)kK[p] * kN[p] / 256);   // the FC packs: 64 LDS-staged slices each; else K x N elements, 256 per block
    }
    MI355_REQUIRE(!qpack || W1, MI355PPO_EINVAL, "%s: kernel Q's pack requested without the layer-1 weight", fn);
    MI355_REQUIRE(aligned(qpack, 16) && aligned(W1, 4), MI355PPO_EALIGN, "%s: packs must be 16-byte aligned", fn);
    a.count[6] = qpack ? 1u : 0u;
    unsigned at = 0;
    static const int order[7] = {6, 4, 5, 0, 1, 2, 3};      // grid order: the one-block digit pack (a 12-us latency chain) and the FC slices start first
    for (int i = 0; i < 7; ++i) {
        a.first[order[i]] = at;
        at += a.count[order[i]];
    }
    MI355_REQUIRE(at > 0, MI355PPO_EINVAL, "%s: nothing to pack", fn);
    constexpr size_t kLds = (size_t)32 * 785 * sizeof(float);      // the larger of the two FC slices (16 x 1,569 floats is 64 bytes less)
    static thread_local int lds_set_dev = -1;                       // hipFuncSetAttribute once per (thread, device): legal inside a capture afterwards
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev != lds_set_dev) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(znature_pack_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(znature_pack_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("%s: hipFuncSetAttribute(%zu bytes of LDS) failed", fn, kLds);
            return MI355PPO_EHIP;
        }
        lds_set_dev = dev;
    }
    if (w_amax) {
        // the weights' maxima first -- one pass over W2 / W3 / Wfc (6.7 MB), every slot of the three records stored --, then the packs
        MI355_REQUIRE(W2 && W3 && Wfc, MI355PPO_EINVAL, "%s: the f16x2 packs need W2, W3 and Wfc", fn);
        MI355_REQUIRE(aligned(w_amax, 64), MI355PPO_EALIGN, "%s: the amax records must be 64-byte aligned", fn);
        ZAbsmax3 m;
        m.x[0] = W2; m.n[0] = 64 * 32 * 4 * 4; m.x[1] = W3; m.n[1] = 64 * 64 * 3 * 3; m.x[2] = Wfc; m.n[2] = 512 * 3136;
        hipLaunchKernelGGL(zabsmax_store_kernel, dim3(3 * kAmaxSlots), dim3(1024), 0, as_stream(stream), m, w_amax);
        int rc = check_launch("zabsmax_store_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(znature_pack_kernel<1>, dim3(at), dim3(256), (fc_fwd || fc_dgrad) ? kLds : 0, as_stream(stream), a);
    } else {
        hipLaunchKernelGGL(znature_pack_kernel<0>, dim3(at), dim3(256), (fc_fwd || fc_dgrad) ? kLds : 0, as_stream(stream), a);
    }
    return check_launch(fn);
}

extern "C" MI355PPO_API int mi355ppo_nature_packs_f32(const float* W1, const float* W2, const float* W3, const float* Wfc, void* qpack,
                                                      void* conv2_fwd, void* conv3_fwd, void* conv3_dgrad, void* conv2_dgrad, void* fc_fwd,
                                                      void* fc_dgrad, void* stream) {
    return nature_packs_impl("mi355ppo_nature_packs_f32", W1, W2, W3, Wfc, qpack, conv2_fwd, conv3_fwd, conv3_dgrad, conv2_dgrad, fc_fwd, fc_dgrad,
                             nullptr, stream);
}

// The same with the six kernel-Z packs in the f16x2 format (mi355ppo_fc_pack_f16x2_bytes of the same shapes) -- what the *_f16x2 entry
// points take.  `w_amax`: 3 amax records (3 * MI355PPO_AMAX_WORDS uint32, 64-byte aligned) that receive max |W2|, |W3|, |Wfc| (every slot
// stored here: one pass over the three tensors, then the pack launch).
extern "C" MI355PPO_API int mi355ppo_nature_packs_f16x2_f32(const float* W1, const float* W2, const float* W3, const float* Wfc, void* qpack,
                                                            void* conv2_fwd, void* conv3_fwd, void* conv3_dgrad, void* conv2_dgrad, void* fc_fwd,
                                                            void* fc_dgrad, uint32_t* w_amax, void* stream) {
    const char* fn = "mi355ppo_nature_packs_f16x2_f32";
    MI355_REQUIRE(w_amax, MI355PPO_EINVAL, "%s: null pointer", fn);
    return nature_packs_impl(fn, W1, W2, W3, Wfc, qpack, conv2_fwd, conv3_fwd, conv3_dgrad, conv2_dgrad, fc_fwd, fc_dgrad, w_amax, stream);
}

// h[m][n] = relu(bias[n] + part[0][m][n] + part[1][m][n] + ...): the K splits of Z_RAW added in order (deterministic)
__global__ __launch_bounds__(256) void zsplit_reduce_kernel(const float4* __restrict__ part, int splits, size_t slab4, const float* __restrict__ bias,
                                                            int n4, float4* __restrict__ h, size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 acc = part[i];
    for (int z = 1; z < splits; ++z) {
        const float4 p = part[(size_t)z * slab4 + i];
        acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    const float* const b = bias + 4 * (i % (size_t)n4);
    acc.x += b[0]; acc.y += b[1]; acc.z += b[2]; acc.w += b[3];
    h[i] = make_float4(acc.x > 0.0f ? acc.x : 0.0f, acc.y > 0.0f ? acc.y : 0.0f, acc.z > 0.0f ? acc.z : 0.0f, acc.w > 0.0f ? acc.w : 0.0f);
}

// K splits of the small-batch forward: (row tile, column tile, split) wave tiles for ONE wave per SIMD (1,024: measured 30.8 us at
// 1,024 rows, 22.6 at 512 -- the library GEMM's 31.3 / 22.6; two waves per SIMD = twice the partials: 38.7 / 29.3 us,
// tools/gpu/fcsplit2.sh), at least 4 k-steps per split.  Below 512 rows fewer, longer splits win (a split of 4 k-steps is all
// prologue and epilogue, and every split is one more partial to fold): 2 M wave tiles, at least 256 -- 128 rows 24.1 -> 18.3 us,
// 256 rows 22.5 -> 20.0 us (profiles/r04_fcsplit_sweep.txt).  The result depends on the split (f32 summation order), so the choice is a
// function of the shape alone: the same M always splits the same way.
static int zsplit_steps_per(int M, int N, int K) {
    const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
    const int total = K / 16;
    int target = 2 * M;                                   // (tuning runs: profiles/r04_fcsplit_sweep.txt)
    if (target < 256) target = 256;
    if (target > 1024) target = 1024;
    long long want = (target + tiles - 1) / tiles;
    if (want < 1) want = 1;
    int per = (int)((total + want - 1) / want);
    if (per < 4) per = 4;
    if (per > total) per = total;
    return per;
}

extern "C" MI355PPO_API size_t mi355ppo_fc_fwd_workspace_bytes(int M, int N, int K) {
    // Below 8,192 rows (round 3: 4,096) -- config B's minibatch of 4,096 rows is 512 whole-K wave tiles on 2,048 wave slots: two K splits
    // 118 -> 79 us; at 8,192 rows a split buys nothing (134 vs 135 us; profiles/r04_fc_split_minibatch_ab.txt).
    constexpr int below = 8192;
    if (M <= 0 || N <= 0 || K <= 0 || K % 16 || M >= below) return 0;          // (from there on whole-K wave tiles: no workspace)
    const int per = zsplit_steps_per(M, N, K), splits = (K / 16 + per - 1) / per;
    return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

// which f16x2 FC launches kernel G (gemmg.hip) takes -- the one place that decides: the forward from 16,384 rows on (below, its 128-row x
// 256-column blocks do not fill the chip), the bit-masked data gradient from 1,024 (profiles/r06_kernel_g_sizes.txt)
static bool fc_g_takes(int M, int N, int K, bool dgrad) {
    return M > 0 && N > 0 && N % 32 == 0 && K >= 128 && K % 64 == 0 && gemmg_on(M, dgrad ? 1024 : 16384);
}
extern "C" MI355PPO_API int mi355ppo_fc_packed_kernel_f16x2(int M, int N, int K, int dgrad) { return fc_g_takes(M, N, K, dgrad != 0) ? 'G' : 'Z'; }

static int fc_fwd_impl(const char* fn, const float* a, int lda, const void* pack, const float* bias, float* h, int M, int N, int K,
                       const unsigned* a_amax, unsigned* h_amax, void* stream) {
    int rc = zgemm_check(fn, a, pack, h, M, N, K, lda, N);
    if (rc) return rc;
    MI355_REQUIRE(bias && aligned(bias, 4), MI355PPO_EINVAL, "%s: bias missing or misaligned", fn);
    ZArgs za = zargs(a, (long long)M * lda * 4, lda, pack, bias, nullptr, h, (long long)M * N * 4, N, M, N, K);
    za.a_amax = a_amax; za.c_amax = h_amax;
    // round 6, f16 split: kernel G (gemmg.hip) from 16,384 rows on -- both operands through workgroup-wide LDS rings, bit-identical results
    if (a_amax && fc_g_takes(M, N, K, false)) {
        rc = gemmg_launch(fn, 0, a, lda, pack, bias, nullptr, h, M, N, K, a_amax, h_amax, as_stream(stream));
        if (rc <= 0) return rc;
    }
    // 64 x 128 wave tiles, one wave per SIMD, for the large batches; below 16,384 rows those do not fill the chip (M / 64 workgroups):
    // 64 x 64 wave tiles, two 4-wave workgroups per CU (measured on one box, profiles/r03_zcfg_ab.jsonl: 32,768 rows 471 vs 531 us,
    // 8,192 rows 205 vs 145 us, 4,096 rows 182 vs 128 us)
    // (round 5, f16 split: the two-wave shape at 32,768 rows again 428 vs 317 us, the data gradient 583 vs 477 us -- 535 with the B ring --,
    //  bit-identical: profiles/r05_tile_shape_experiments.txt)
    if (M < 16384) return z_launch<ZRowsLinear, 2, 2, 4, Z_BIAS_RELU, true, 2>(za, as_stream(stream), fn);
    return z_launch<ZRowsLinear, 2, 4, 4, Z_BIAS_RELU, true>(za, as_stream(stream), fn);
}

extern "C" MI355PPO_API int mi355ppo_fc_fwd_relu_packed_f32(const float* a, int lda, const void* pack, const float* bias, float* h,
                                                            int M, int N, int K, void* stream) {
    return fc_fwd_impl("mi355ppo_fc_fwd_relu_packed_f32", a, lda, pack, bias, h, M, N, K, nullptr, nullptr, stream);
}

// The same with a workspace of mi355ppo_fc_fwd_workspace_bytes(M, N, K) bytes: batches below 8,192 rows (a rollout step's 1,024
// envs) split K over blockIdx.z -- raw partials into the workspace, then one pass that adds them in order, bias, ReLU.  Without a
// workspace (or when none is needed) this IS mi355ppo_fc_fwd_relu_packed_f32.
static int fc_fwd_ws_impl(const char* fn, const float* a, int lda, const void* pack, const float* bias, float* h, int M, int N, int K, void* ws,
                          size_t ws_bytes, const unsigned* a_amax, unsigned* h_amax, void* stream) {
    const size_t need = mi355ppo_fc_fwd_workspace_bytes(M, N, K);
    if (need == 0 || !ws || N % 4) return fc_fwd_impl(fn, a, lda, pack, bias, h, M, N, K, a_amax, h_amax, stream);
    MI355_REQUIRE(!h_amax, MI355PPO_EINVAL, "%s: the K-split forward (M=%d) does not record h's maximum (no consumer splits h: pass null)", fn, M);
    int rc = zgemm_check(fn, a, pack, h, M, N, K, lda, N);
    if (rc) return rc;
    MI355_REQUIRE(bias && aligned(bias, 4) && aligned(h, 16) && aligned(ws, 16), MI355PPO_EINVAL, "%s: bias missing, or h / workspace not 16-byte aligned", fn);
    MI355_REQUIRE(ws_bytes >= need, MI355PPO_EINVAL, "%s: workspace of %zu bytes, %zu needed (mi355ppo_fc_fwd_workspace_bytes)", fn, ws_bytes, need);
    ZArgs za = zargs(a, (long long)M * lda * 4, lda, pack, bias, nullptr, static_cast<float*>(ws), (long long)M * N * 4, N, M, N, K);
    za.a_amax = a_amax;
    za.steps_per = zsplit_steps_per(M, N, K);
    const int splits = (K / 16 + za.steps_per - 1) / za.steps_per;
    rc = z_launch<ZRowsLinear, 2, 2, 4, Z_RAW, true, 2>(za, as_stream(stream), fn);
    if (rc) return rc;
    const size_t total4 = (size_t)M * N / 4;
    hipLaunchKernelGGL(zsplit_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       static_cast<const float4*>(ws), splits, total4, bias, N / 4, reinterpret_cast<float4*>(h), total4);
    return check_launch(fn);
}

extern "C" MI355PPO_API int mi355ppo_fc_fwd_relu_packed_ws_f32(const float* a, int lda, const void* pack, const float* bias, float* h,
                                                               int M, int N, int K, void* ws, size_t ws_bytes, void* stream) {
    return fc_fwd_ws_impl("mi355ppo_fc_fwd_relu_packed_ws_f32", a, lda, pack, bias, h, M, N, K, ws, ws_bytes, nullptr, nullptr, stream);
}

// The K-split half of mi355ppo_fc_fwd_relu_packed_ws_f32 alone: raw partials of a (M, K) x pack -> (splits, M, N) in `ws`; the caller
// folds them (heads.hip's fused FC-fold + heads + sampling kernel of the rollout).  Returns the number of splits in *splits (>= 2).
namespace mi355ppo {
int z_fc_raw_launch(const char* fn, const float* a, int lda, const void* pack, int M, int N, int K, void* ws, size_t ws_bytes, int* splits,
                    hipStream_t stream, const unsigned* a_amax) {
    const size_t need = mi355ppo_fc_fwd_workspace_bytes(M, N, K);
    MI355_REQUIRE(need > 0 && N % 4 == 0, MI355PPO_EINVAL, "%s: M=%d rows need no K split (use the unfused entry points from 8,192 rows on)", fn, M);
    int rc = zgemm_check(fn, a, pack, static_cast<const float*>(ws), M, N, K, lda, N);
    if (rc) return rc;
    MI355_REQUIRE(ws && aligned(ws, 16) && ws_bytes >= need, MI355PPO_EWORKSPACE, "%s: workspace of %zu bytes, %zu needed (mi355ppo_fc_fwd_workspace_bytes)",
                  fn, ws ? ws_bytes : (size_t)0, need);
    ZArgs za = zargs(a, (long long)M * lda * 4, lda, pack, nullptr, nullptr, static_cast<float*>(ws), (long long)M * N * 4, N, M, N, K);
    za.a_amax = a_amax;                                                   // non-null: `pack` is an f16x2 pack (the *_f16x2 entry points)
    za.steps_per = zsplit_steps_per(M, N, K);
    *splits = (K / 16 + za.steps_per - 1) / za.steps_per;
    return z_launch<ZRowsLinear, 2, 2, 4, Z_RAW, true, 2>(za, stream, fn);
}
}  // namespace mi355ppo

static int fc_dgrad_impl(const char* fn, const float* dz, int lddz, const void* pack, const float* act_in, const unsigned* bits, float* da,
                         int M, int N, int K, void* stream, const unsigned* dz_amax = nullptr, unsigned* da_amax = nullptr) {
    int rc = zgemm_check(fn, dz, pack, da, M, N, K, lddz, N);
    if (rc) return rc;
    ZArgs za = zargs(dz, (long long)M * lddz * 4, lddz, pack, nullptr, act_in, da, (long long)M * N * 4, N, M, N, K);
    za.a_amax = dz_amax; za.c_amax = da_amax;
    // (the B ring -- z_launch<..., 1, true> for even K / 16 -- measured 630 -> 720 us here: one wave per SIMD has nothing to run while
    // it waits at the ring barrier; profiles/r03_blds_ab.jsonl)
    // supertiles of 4 row blocks: L2-miss reads 1.75 -> 0.62 GB per launch at 32,768 rows, 612 -> 605 us (2 / 8 row blocks: 598 / 602 us;
    // same-box A/B with a run-time switch, profiles/r03_raster_ab.jsonl, r03_pmc_fetch_raster_{before,after}.csv)
    za.super_rows = 4;
    if (bits) {
        MI355_REQUIRE(N % 32 == 0 && aligned(bits, 4) && aligned(da, 128), MI355PPO_EINVAL, "%s: bit masks need N %% 32 == 0 (N=%d) and da on a 128-byte boundary", fn, N);
        za.bits_in = bits;
        if (dz_amax && fc_g_takes(M, N, K, true)) {         // round 6, f16 split: kernel G (gemmg.hip), bit-identical results
            rc = gemmg_launch(fn, 1, dz, lddz, pack, nullptr, bits, da, M, N, K, dz_amax, da_amax, as_stream(stream));
            if (rc <= 0) return rc;
        }
        return z_launch<ZRowsLinear, 2, 4, 4, Z_MASKB, false>(za, as_stream(stream), fn);
    }
    MI355_REQUIRE(act_in && aligned(act_in, 4) && act_in != da, MI355PPO_EINVAL, "%s: act_in missing, misaligned or aliased with da", fn);
    return z_launch<ZRowsLinear, 2, 4, 4, Z_MASK, false>(za, as_stream(stream), fn);
}

extern "C" MI355PPO_API int mi355ppo_fc_dgrad_mask_packed_f32(const float* dz, int lddz, const void* pack, const float* act_in, float* da,
                                                              int M, int N, int K, void* stream) {
    return fc_dgrad_impl("mi355ppo_fc_dgrad_mask_packed_f32", dz, lddz, pack, act_in, nullptr, da, M, N, K, stream);
}

// The same with the ReLU mask as bits (word w, bit b <-> element 32 w + b of the flat (M, N) activation; N % 32 == 0), as written by
// mi355ppo_cnn_conv_fwd_packed_bits_f32 for the layer below.
extern "C" MI355PPO_API int mi355ppo_fc_dgrad_maskbits_packed_f32(const float* dz, int lddz, const void* pack, const uint32_t* mask_bits,
                                                                  float* da, int M, int N, int K, void* stream) {
    const char* fn = "mi355ppo_fc_dgrad_maskbits_packed_f32";
    MI355_REQUIRE(mask_bits, MI355PPO_EINVAL, "%s: null pointer", fn);
    return fc_dgrad_impl(fn, dz, lddz, pack, nullptr, mask_bits, da, M, N, K, stream);
}

// ---- convolutions of layers 2 and 3 on kernel Z.  `pack` = mi355ppo_fc_pack_f32 of the layer's (N, K) f32 matrix from
// mi355ppo_cnn_repack_weights_f32: mode 0 for the forward (N = 64 output channels, K = (tap row, tap column, input channel)),
// mode 1 for the layer-3 data gradient (N = 64 input channels, K = (r, c, output channel), taps flipped), mode 2 for the
// layer-2 data gradient (N = 4 stride-parity classes x 32 input channels, K = (r, c, output channel)).
static bool conv_r_takes(long long images, int layer, bool dgrad) {      // which f16x2 launches kernel R (convr.hip) takes -- the one place that decides
    return convr_on(images, dgrad && layer == 2 ? 512 : 1);      // (round 6: the per-launch switches MI355PPO_CONV_R2F / R3F / R2 / R3 are gone)
}

static int conv_fwd_packed_impl(const char* fn, const float* src, const void* pack, const float* bias, float* dst, unsigned* bits,
                                int64_t images, int layer, void* stream, const unsigned* src_amax = nullptr, unsigned* dst_amax = nullptr) {
    MI355_REQUIRE(src && pack && bias && dst, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(layer == 2 || layer == 3, MI355PPO_EINVAL, "%s: layer=%d must be 2 or 3", fn, layer);
    MI355_REQUIRE(images > 0, MI355PPO_EINVAL, "%s: images=%lld must be positive", fn, (long long)images);
    MI355_REQUIRE(aligned(src, 16) && aligned(pack, 16) && aligned(dst, 128) && aligned(bias, 4) && aligned(bits, 4), MI355PPO_EALIGN,
                  "%s: src / pack must be 16-byte aligned, dst 128-byte aligned", fn);
    const long long srcb = (long long)images * (layer == 2 ? 20 * 20 * 32 : 9 * 9 * 64) * 4;
    MI355_REQUIRE(srcb < (1LL << 32) - 8192, MI355PPO_EINVAL, "%s: the source (%lld bytes) must stay below 4 GiB (32-bit buffer offsets)", fn, srcb);   // (the destination is smaller)
    hipStream_t st = as_stream(stream);
    // Rollout-sized launches are a single, partly filled round of wave tiles, each walking all K / 16 k-steps: below 768 images 32-row
    // tiles (twice the waves, half the MFMAs per k-step) -- layer 2 / layer 3 at 256 images 23.4 -> 17.2 / 26.0 -> 18.5 us, at 128
    // images 22.3 -> 17.6 / 25.2 -> 17.8 us, bit-identical; at 1,024 images the 64-row tiles win (44.7 vs 50.9 us;
    // profiles/r04_small_tiles_ab.txt).
    const bool small = !bits && images < 768;
    if (layer == 2) {
        if (src_amax && conv_r_takes(images, 2, false))
            return convr_fwd2(fn, src, (unsigned)srcb, pack, bias, dst, (unsigned)((long long)images * 81 * 64 * 4), bits, images, src_amax, dst_amax, st);
        ZArgs za = zargs(src, srcb, 0, pack, bias, nullptr, dst, (long long)images * 81 * 64 * 4, 64, (long long)images * 81, 64, ZConv2::K);
        za.bits_out = bits;
        za.a_amax = src_amax; za.c_amax = dst_amax;
        if (small) return z_launch<ZConv2, 1, 2, 4, Z_BIAS_RELU, false, 2>(za, st, fn);
        if (bits) return z_blds(images) ? z_launch<ZConv2, 2, 2, 4, Z_BIAS_RELU_BITS, false, 2, true>(za, st, fn) : z_launch<ZConv2, 2, 2, 4, Z_BIAS_RELU_BITS, false, 2>(za, st, fn);
        return z_blds(images) ? z_launch<ZConv2, 2, 2, 4, Z_BIAS_RELU, false, 2, true>(za, st, fn) : z_launch<ZConv2, 2, 2, 4, Z_BIAS_RELU, false, 2>(za, st, fn);
    }
    if (src_amax && conv_r_takes(images, 3, false))      // f16 split: kernel R (convr.hip), the source of an image group resident in LDS
        return convr_fwd3(fn, src, (unsigned)srcb, pack, bias, dst, (unsigned)((long long)images * 49 * 64 * 4), bits, images, src_amax, dst_amax, st);
    ZArgs za = zargs(src, srcb, 0, pack, bias, nullptr, dst, (long long)images * 49 * 64 * 4, 64, (long long)images * 49, 64, ZConv3::K);
    za.bits_out = bits;
    za.a_amax = src_amax; za.c_amax = dst_amax;
    if (small) return z_launch<ZConv3, 1, 2, 4, Z_BIAS_RELU, false, 2>(za, st, fn);
    if (bits) return z_blds(images) ? z_launch<ZConv3, 2, 2, 4, Z_BIAS_RELU_BITS, false, 2, true>(za, st, fn) : z_launch<ZConv3, 2, 2, 4, Z_BIAS_RELU_BITS, false, 2>(za, st, fn);
    return z_blds(images) ? z_launch<ZConv3, 2, 2, 4, Z_BIAS_RELU, false, 2, true>(za, st, fn) : z_launch<ZConv3, 2, 2, 4, Z_BIAS_RELU, false, 2>(za, st, fn);
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv_fwd_packed_f32(const float* src, const void* pack, const float* bias, float* dst,
                                                             int64_t images, int layer, void* stream) {
    return conv_fwd_packed_impl("mi355ppo_cnn_conv_fwd_packed_f32", src, pack, bias, dst, nullptr, images, layer, stream);
}

// The same, also writing (dst > 0) as bits (images * 81 * 2 words for layer 2, images * 49 * 2 for layer 3): the mask the data
// gradient of the layer ABOVE needs (mi355ppo_cnn_conv_dgrad_packed_bits_f32 / mi355ppo_fc_dgrad_maskbits_packed_f32).
extern "C" MI355PPO_API int mi355ppo_cnn_conv_fwd_packed_bits_f32(const float* src, const void* pack, const float* bias, float* dst,
                                                                  uint32_t* mask_bits, int64_t images, int layer, void* stream) {
    const char* fn = "mi355ppo_cnn_conv_fwd_packed_bits_f32";
    MI355_REQUIRE(mask_bits, MI355PPO_EINVAL, "%s: null pointer", fn);
    return conv_fwd_packed_impl(fn, src, pack, bias, dst, mask_bits, images, layer, stream);
}

static int conv_dgrad_packed_impl(const char* fn, const float* dz, const void* pack, const float* act_in, const unsigned* bits, float* dsrc,
                                  int64_t images, int layer, void* stream, const unsigned* dz_amax = nullptr, unsigned* dsrc_amax = nullptr) {
    MI355_REQUIRE(dz && pack && (act_in || bits) && dsrc, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(layer == 2 || layer == 3, MI355PPO_EINVAL, "%s: layer=%d must be 2 or 3", fn, layer);
    MI355_REQUIRE(images > 0, MI355PPO_EINVAL, "%s: images=%lld must be positive", fn, (long long)images);
    MI355_REQUIRE(aligned(dz, 16) && aligned(pack, 16) && aligned(dsrc, 128) && aligned(act_in, 16) && aligned(bits, 4), MI355PPO_EALIGN,
                  "%s: dz / pack / act_in must be 16-byte aligned, dsrc 128-byte aligned", fn);
    MI355_REQUIRE(act_in != dsrc, MI355PPO_EINVAL, "%s: act_in must not alias dsrc", fn);
    const long long srcb = (long long)images * (layer == 2 ? 9 * 9 * 64 : 7 * 7 * 64) * 4;
    const long long dstb = (long long)images * (layer == 2 ? 20 * 20 * 32 : 9 * 9 * 64) * 4;
    MI355_REQUIRE(srcb < (1LL << 32) - 8192 && dstb < (1LL << 32) - 8192, MI355PPO_EINVAL,
                  "%s: dz (%lld bytes) and dsrc (%lld bytes) must stay below 4 GiB (32-bit buffer offsets)", fn, srcb, dstb);
    hipStream_t st = as_stream(stream);
    if (layer == 3) {      // da2 (images, 9, 9, 64) = full correlation of dz3 with the flipped taps, masked by a2 > 0
        if (dz_amax && bits && conv_r_takes(images, 3, true)) return convr_dgrad3(fn, dz, (unsigned)srcb, pack, bits, dsrc, (unsigned)dstb, images, dz_amax, dsrc_amax, st);
        ZArgs za = zargs(dz, srcb, 0, pack, nullptr, act_in, dsrc, (long long)images * 81 * 64 * 4, 64, (long long)images * 81, 64, ZDgrad3::K, images);
        za.bits_in = bits;
        za.a_amax = dz_amax; za.c_amax = dsrc_amax;
        if (bits) return z_blds(images) ? z_launch<ZDgrad3, 2, 2, 4, Z_MASKB, false, 2, true>(za, st, fn) : z_launch<ZDgrad3, 2, 2, 4, Z_MASKB, false, 2>(za, st, fn);
        return z_blds(images) ? z_launch<ZDgrad3, 2, 2, 4, Z_MASK, false, 2, true>(za, st, fn) : z_launch<ZDgrad3, 2, 2, 4, Z_MASK, false, 2>(za, st, fn);
    }
    // da1 (images, 20, 20, 32): the four stride-parity classes are the four column tiles of one 128-column GEMM over the 10 x 10 grid
    if (dz_amax && bits && conv_r_takes(images, 2, true)) return convr_dgrad2(fn, dz, (unsigned)srcb, pack, bits, dsrc, (unsigned)dstb, images, dz_amax, dsrc_amax, st);
    ZArgs za = zargs(dz, srcb, 0, pack, nullptr, act_in, dsrc, (long long)images * 400 * 32 * 4, 0, (long long)images * 100, 128, ZDgrad2::K, images);
    za.bits_in = bits;
    za.a_amax = dz_amax; za.c_amax = dsrc_amax;
    if (bits) return z_blds(images) ? z_launch<ZDgrad2, 2, 2, 4, Z_MASKB_CLS4, false, 2, true>(za, st, fn) : z_launch<ZDgrad2, 2, 2, 4, Z_MASKB_CLS4, false, 2>(za, st, fn);
    return z_blds(images) ? z_launch<ZDgrad2, 2, 2, 4, Z_MASK_CLS4, false, 2, true>(za, st, fn) : z_launch<ZDgrad2, 2, 2, 4, Z_MASK_CLS4, false, 2>(za, st, fn);
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv_dgrad_packed_f32(const float* dz, const void* pack, const float* act_in, float* dsrc,
                                                               int64_t images, int layer, void* stream) {
    const char* fn = "mi355ppo_cnn_conv_dgrad_packed_f32";
    MI355_REQUIRE(act_in, MI355PPO_EINVAL, "%s: null pointer", fn);
    return conv_dgrad_packed_impl(fn, dz, pack, act_in, nullptr, dsrc, images, layer, stream);
}

// The same with the ReLU mask of the layer's INPUT activation as bits (layer 2: images * 400 words from mi355ppo_cnn_conv1q_fwd_bits;
// layer 3: images * 81 * 2 words from mi355ppo_cnn_conv_fwd_packed_bits_f32 of layer 2).
extern "C" MI355PPO_API int mi355ppo_cnn_conv_dgrad_packed_bits_f32(const float* dz, const void* pack, const uint32_t* mask_bits, float* dsrc,
                                                                    int64_t images, int layer, void* stream) {
    const char* fn = "mi355ppo_cnn_conv_dgrad_packed_bits_f32";
    MI355_REQUIRE(mask_bits, MI355PPO_EINVAL, "%s: null pointer", fn);
    return conv_dgrad_packed_impl(fn, dz, pack, nullptr, mask_bits, dsrc, images, layer, stream);
}

// ---------------------------------------------------------------------------------------------------------------- f16x2 entry points
// The same kernels on the two-term f16 split (f16split.h; SPLIT = 1 of z_kernel): `pack` is an f16x2 pack (its header carries the
// weight tensor's maximum), every split operand comes with its amax record, every result a consumer will split again gets its record
// filled.  Same shapes, alignments and epilogues as the entry points above.
extern "C" MI355PPO_API int mi355ppo_cnn_conv_packed_kernel_f16x2(int64_t images, int layer, int dgrad) {
    if (!(images > 0 && (layer == 2 || layer == 3) && conv_r_takes(images, layer, dgrad != 0))) return 'Z';
    return (layer == 2 && dgrad != 0 && convrb_takes(images)) ? 'B' : 'R';      // 'B': kernel RB (convrb.hip), kernel R's layer-2 data gradient by border class
}

extern "C" MI355PPO_API size_t mi355ppo_fc_pack_f16x2_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || K % 16) return 0;
    return (size_t)kF16PackHeader + (size_t)(K / 16) * (size_t)((N + 31) / 32) * 2048;
}

// amax |= max |x| over n floats (the record must have been zeroed, or hold the maximum of another part of the same tensor)
extern "C" MI355PPO_API int mi355ppo_absmax_f32(const float* x, int64_t n, uint32_t* amax, void* stream) {
    const char* fn = "mi355ppo_absmax_f32";
    MI355_REQUIRE(x && amax, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(n > 0, MI355PPO_EINVAL, "%s: n=%lld must be positive", fn, (long long)n);
    MI355_REQUIRE(aligned(x, 4) && aligned(amax, 64), MI355PPO_EALIGN, "%s: the amax record must be 64-byte aligned", fn);
    ZAbsmax3 m;
    m.x[0] = x; m.n[0] = n; m.x[1] = m.x[2] = x; m.n[1] = m.n[2] = 0;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zabsmax_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), m, amax, (int)blocks);
    return check_launch(fn);
}

// B (N, K) f32 -> its f16x2 pack; `b_amax` = B's amax record (mi355ppo_absmax_f32 of B, or anything >= max |B|)
extern "C" MI355PPO_API int mi355ppo_fc_pack_f16x2_f32(const float* B, int ldb, int N, int K, const uint32_t* b_amax, void* pack, void* stream) {
    const char* fn = "mi355ppo_fc_pack_f16x2_f32";
    MI355_REQUIRE(B && pack && b_amax, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(N > 0 && K > 0 && K % 16 == 0 && ldb >= K, MI355PPO_EINVAL, "%s: N=%d K=%d ldb=%d (K a positive multiple of 16, ldb >= K)", fn, N, K, ldb);
    MI355_REQUIRE(aligned(B, 4) && aligned(pack, 16) && aligned(b_amax, 64), MI355PPO_EALIGN, "%s: misaligned pointer (pack: 16 bytes, record: 64)", fn);
    const long long total = (long long)(K / 16) * ((N + 31) / 32) * 512;
    hipLaunchKernelGGL(zpack_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), B, ldb, N, K, b_amax,
                       static_cast<unsigned char*>(pack));
    return check_launch(fn);
}

// relu(a @ B^T + bias); rows below 8,192 with a workspace: K split over the grid (then h_amax must be null).  h_amax: optional.
extern "C" MI355PPO_API int mi355ppo_fc_fwd_relu_packed_f16x2_f32(const float* a, int lda, const void* pack, const float* bias, float* h, int M,
                                                                  int N, int K, void* ws, size_t ws_bytes, const uint32_t* a_amax,
                                                                  uint32_t* h_amax, void* stream) {
    const char* fn = "mi355ppo_fc_fwd_relu_packed_f16x2_f32";
    MI355_REQUIRE(a_amax && aligned(a_amax, 64) && aligned(h_amax, 64), MI355PPO_EINVAL, "%s: a's amax record missing, or a record not 64-byte aligned", fn);
    return fc_fwd_ws_impl(fn, a, lda, pack, bias, h, M, N, K, ws, ws_bytes, a_amax, h_amax, stream);
}

// (dz @ B^T) masked by the ReLU of the layer below: `mask_bits` if given, else `act_in` > 0
extern "C" MI355PPO_API int mi355ppo_fc_dgrad_packed_f16x2_f32(const float* dz, int lddz, const void* pack, const float* act_in,
                                                               const uint32_t* mask_bits, float* da, int M, int N, int K, const uint32_t* dz_amax,
                                                               uint32_t* da_amax, void* stream) {
    const char* fn = "mi355ppo_fc_dgrad_packed_f16x2_f32";
    MI355_REQUIRE(dz_amax && aligned(dz_amax, 64) && aligned(da_amax, 64), MI355PPO_EINVAL, "%s: dz's amax record missing, or a record not 64-byte aligned", fn);
    MI355_REQUIRE(act_in || mask_bits, MI355PPO_EINVAL, "%s: neither act_in nor mask_bits", fn);
    return fc_dgrad_impl(fn, dz, lddz, pack, mask_bits ? nullptr : act_in, mask_bits, da, M, N, K, stream, dz_amax, da_amax);
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv_fwd_packed_f16x2_f32(const float* src, const void* pack, const float* bias, float* dst,
                                                                   uint32_t* mask_bits, int64_t images, int layer, const uint32_t* src_amax,
                                                                   uint32_t* dst_amax, void* stream) {
    const char* fn = "mi355ppo_cnn_conv_fwd_packed_f16x2_f32";
    MI355_REQUIRE(src_amax && aligned(src_amax, 64) && aligned(dst_amax, 64), MI355PPO_EINVAL, "%s: src's amax record missing, or a record not 64-byte aligned", fn);
    return conv_fwd_packed_impl(fn, src, pack, bias, dst, mask_bits, images, layer, stream, src_amax, dst_amax);
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv_dgrad_packed_f16x2_f32(const float* dz, const void* pack, const float* act_in,
                                                                     const uint32_t* mask_bits, float* dsrc, int64_t images, int layer,
                                                                     const uint32_t* dz_amax, uint32_t* dsrc_amax, void* stream) {
    const char* fn = "mi355ppo_cnn_conv_dgrad_packed_f16x2_f32";
    MI355_REQUIRE(dz_amax && aligned(dz_amax, 64) && aligned(dsrc_amax, 64), MI355PPO_EINVAL, "%s: dz's amax record missing, or a record not 64-byte aligned", fn);
    return conv_dgrad_packed_impl(fn, dz, pack, mask_bits ? nullptr : act_in, mask_bits, dsrc, images, layer, stream, dz_amax, dsrc_amax);
}
