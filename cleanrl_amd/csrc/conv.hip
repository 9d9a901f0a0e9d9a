// NatureCNN convolutions as f32-MFMA implicit GEMMs on channels-last tensors (gfx950).
//
// Replaces the conv stack of Agent.network (cleanrl/ppo_atari_multigpu.py:136-147: Conv2d(4,32,8,s4) ReLU
// Conv2d(32,64,4,s2) ReLU Conv2d(64,64,3,s1) ReLU) -- forward, data gradient and weight gradient -- and fuses
// into it what the reference runs as separate passes over the largest tensors of the update:
//   * `b_obs[mb_inds]` + `x / 255.0` (:320,154): conv1 reads the uint8 rollout rows through mb_inds and
//     converts in registers (the 3.7 GB f32 staging tensor of the K5 path is never written);
//   * bias add + ReLU (epilogue of every forward kernel);
//   * ReLU backward (epilogue mask of the data-gradient kernel that PRODUCES the gradient) and the bias
//     gradient (side sum of the weight-gradient kernel).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, exact f32 products with one rounding per
// fma, 157 TFLOP/s chip peak (the same rate as the f32 vector unit).  No reduced precision anywhere.
//
// The GEMM view: C[m][n] = sum_k A[m][k] * Bt[n][k]
//   m  = a destination pixel (img, gy, gx);  n = output channel;  k = (r, c, ch) tap-major, channel-minor.
//   A[m][k] = src[img, gy*SS + OFF + r, gx*SS + OFF + c, ch] (0 outside the tensor): with channels-last
//   storage a (pixel, r) pair is ONE contiguous run of KW*C elements, so no im2col buffer exists anywhere.
//   The same GEMM is the forward conv (SS = stride, OFF = 0), the stride-1 data gradient (OFF = -(K-1), flipped
//   weights) and the stride-2 data gradient (four parity classes of destination pixels, each a 2x2-tap stride-1
//   problem with its own weight matrix).  The weight gradient is dWt[n][k] = sum_m dz[m][n] * A[m][k], one partial per
//   workgroup / wave, summed in a fixed order by conv_wgrad_reduce1/2 (deterministic) and scattered into torch's
//   (Cout, Cin, KH, KW) layout.
//
// Kernels in this file (the LDS-tiled first generation G and the staged weight-gradient kernels W / D were removed in
// round 2; DESIGN.md section 3.2 keeps their measurements as the tuning log):
// Kernel S (conv_stream_kernel): forward / data gradient with run-time geometry -- the fallback for tensors beyond the
//   4 GiB that kernel F's 32-bit buffer offsets address;
// Kernel F (conv_fixed_kernel): forward / data gradient with compile-time geometry, the weight matrix resident in LDS,
//   A fragments streamed global -> register ring, buffer loads / stores for padding and tails;
// Kernel R (conv_wgrad_rows_kernel): layer-1 weight gradient on the f32 pipe, wave-autonomous (private double-buffered LDS
//   slab, no barrier) -- the A/B counterpart of kernel P (conv1p.hip, bf16 pipe), which is the default;
// Kernel T (conv_wgrad_taps_kernel): layer-2/3 weight gradient, operands straight from global memory in MFMA operand
//   layout, a sliding register window of source columns, no LDS at all.
// Each is documented where it is defined; DESIGN.md section 3.2 has the measurements and the tuning log.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace mi355ppo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvGeom {
    int H, W, C;          // source tensor per image: (H, W, C) channels-last
    int KH, KW;           // taps
    int GY, GX;           // destination grid enumerated by the GEMM rows (per image GY*GX rows)
    int SS, OFF;          // source y = gy*SS + OFF + r, x = gx*SS + OFF + c
    int DH, DW, DC;       // destination tensor per image (DH, DW, DC); DC = channels per pixel (row stride)
    int DM, DAY, DAX;     // destination pixel = (gy*DM + DAY, gx*DM + DAX)
    int K;                // KH*KW*C
    int N;                // output channels computed (== BN)
    int classes;          // 1, or 4 = stride-2 data-gradient parity classes (blockIdx.y)
    int logC;             // log2(C) (f32 sources)
    long long P;          // total GEMM rows = images * GY * GX
};

// uint8 taps enter the MFMAs as the exact integers 0..255 (one v_cvt_f32_ubyteN each) and the 1/255 of
// `x / 255.0` (:154) is folded into the layer-1 weight matrix by the repack kernel (forward) or applied once to the
// reduced sum (weight gradient): sum_k (x_k/255) w_k == sum_k x_k (w_k/255) up to one f32 rounding per term -- the
// same error class as the summation-order differences between any two f32 convolutions -- and the per-tap VALU work
// next to each MFMA drops from 5 instructions to 1 (VALU issue on a SIMD comes out of the matrix pipe's time).
__device__ __forceinline__ float u8_tap(uint32_t w, int b) { return (float)((w >> (8 * b)) & 0xffu); }
constexpr float kInv255 = 1.0f / 255.0f;

enum { EPI_BIAS_RELU = 0, EPI_MASK = 1, EPI_RAW = 2 };

// ------------------------------------------------------------------------------------------ kernel S
// conv_stream_kernel: the GEMM of the file header with the weight matrix, not the pixels, resident on chip.  The whole
// weight matrix Bt (N x K, <= 148 KB) is loaded ONCE per workgroup into LDS and stays there; every wave then
// streams its own 32-pixel tiles: a lane fetches its A fragments (16 bytes = 4 consecutive k of ITS pixel, or 16
// uint8 taps) straight from global memory into a register ring that runs D chunks ahead -- across tile
// boundaries -- so HBM latency is covered by D x (4..16) MFMAs.  No LDS traffic for A, no workgroup barriers in
// the main loop, waves drift freely; the matrix pipe sees an MFMA stream interrupted only by the per-tile
// epilogue.  Tiles are handed out round-robin over all waves of the grid (persistent workgroups).
constexpr int kRing = 8;

// MT = pixel tiles (of 32) per wave iteration: with MT = 2 every B fragment read from LDS feeds two MFMAs and the
// per-chunk bookkeeping (addresses, waits, fences) is shared by twice as many MFMAs.
template <int NJT, bool U8IN, int EPI, bool PAD, bool CLS4, int MT, int NW>
__global__ __launch_bounds__(64 * NW) void conv_stream_kernel(const void* __restrict__ src_v, const int64_t* __restrict__ inds,
                                                          const float* __restrict__ Bt_all, const float* __restrict__ bias,
                                                          const float* __restrict__ mask_src, float* __restrict__ dst,
                                                          ConvGeom g, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];       // [32*NJT][K + 4] (+ 8 floats of slack) [+ bias]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int cls = blockIdx.y;
    const int ldb = g.K + 4;
    const int day = g.classes == 4 ? (cls >> 1) : g.DAY;
    const int dax = g.classes == 4 ? (cls & 1) : g.DAX;
    {
        const float4* __restrict__ Bt4 = reinterpret_cast<const float4*>(Bt_all + (size_t)cls * (32 * NJT) * g.K);
        const unsigned k4 = (unsigned)g.K >> 2, total = (unsigned)(32 * NJT) * k4;
        for (unsigned e = tid; e < total; e += 64 * NW) {
            const unsigned row = e / k4, c = e - row * k4;
            *reinterpret_cast<float4*>(&Bs[row * ldb + c * 4]) = Bt4[e];
        }
        // bias goes through LDS too: a value that arrives by a global load and is first used inside the tile loop
        // makes hipcc guard every epilogue store with s_waitcnt vmcnt(1), which drains the prefetch ring
        if (EPI == EPI_BIAS_RELU && tid < 32 * NJT) Bs[(32 * NJT) * ldb + 8 + tid] = bias[tid];
    }
    __syncthreads();

    const unsigned per_img = (unsigned)(g.GY * g.GX), GX = (unsigned)g.GX;
    const unsigned P = (unsigned)g.P;
    const int runlen = g.KW * g.C, rowpitch = g.W * g.C;
    const int nwv = gridDim.x * NW;
    const int nblocks = U8IN ? (g.KH / kRing) : (g.K / 8 / kRing);       // ring rounds per tile
    constexpr int TP = 32 * MT;                                          // pixels per wave iteration

    // ---- load cursor (runs kRing chunks ahead of the MFMAs); (r, rem, koff) are wave-uniform
    int l_tile = blockIdx.x * NW + wave;
    int l_r = 0, l_rem = 0, l_koff = 0;
    long long l_base[MT];
    int l_sy0[MT], l_sx0[MT];
    bool l_ok[MT];
    auto setup = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned p = (unsigned)l_tile * (unsigned)TP + (unsigned)(32 * m + li);
            l_ok[m] = (l_tile < ntiles) && (p < P);
            const unsigned pp = l_ok[m] ? p : 0u;
            const unsigned img = pp / per_img, rem = pp - img * per_img;
            const unsigned gy = rem / GX, gx = rem - gy * GX;
            l_sy0[m] = (int)gy * g.SS + g.OFF;
            l_sx0[m] = (int)gx * g.SS + g.OFF;
            const long long simg = (U8IN && inds) ? inds[img] : (long long)img;
            l_base[m] = ((simg * g.H + l_sy0[m]) * g.W + l_sx0[m]) * (long long)g.C + (U8IN ? 16 : 4) * lh;
        }
    };
    auto advance = [&](bool may_wrap) {
        if (U8IN) {
            ++l_r;
            l_koff += rowpitch;
        } else {
            l_rem += 8;
            l_koff += 8;
            if (l_rem == runlen) { l_rem = 0; ++l_r; l_koff = l_r * rowpitch; }
        }
        if (may_wrap && l_r == g.KH) {
            l_r = 0; l_rem = 0; l_koff = 0;
            l_tile += nwv;
            setup();
        }
    };
    // The ring is loaded UNCONDITIONALLY (invalid taps read a clamped, always-mapped address and are zeroed when
    // consumed, bit d of `vmask`), and slot d is refilled only after its MFMAs have been issued: the slots then
    // keep fixed registers and the compiler can count the loads (s_waitcnt vmcnt(N)) instead of draining the queue
    // at every loop back-edge.
    u32x4 ring[MT][kRing];
    unsigned vmask[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) vmask[m] = 0u;
    auto fetch = [&](int d) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            bool ok = l_ok[m];
            if (!U8IN && PAD) {
                const int sy = l_sy0[m] + l_r, sx = l_sx0[m] + (l_rem >> g.logC);
                ok = ok && sy >= 0 && sy < g.H && sx >= 0 && sx < g.W;
            }
            const long long off = ok ? (l_base[m] + l_koff) : (long long)((U8IN ? 16 : 4) * lh);
            if (U8IN) ring[m][d] = *reinterpret_cast<const u32x4*>(static_cast<const uint8_t*>(src_v) + off);
            else ring[m][d] = *reinterpret_cast<const u32x4*>(static_cast<const float*>(src_v) + off);
            if (PAD) vmask[m] = (vmask[m] & ~(1u << d)) | ((ok ? 1u : 0u) << d);
        }
    };
    setup();
#pragma unroll
    for (int d = 0; d < kRing; ++d) {
        fetch(d);
        advance(d == kRing - 1);
    }

    float bias_r[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) bias_r[jt] = (EPI == EPI_BIAS_RELU) ? Bs[(32 * NJT) * ldb + 8 + jt * 32 + li] : 0.0f;
    constexpr int NB = U8IN ? 4 : 1;                  // float4 B fragments per chunk and channel tile
    for (int tile = blockIdx.x * NW + wave; tile < ntiles; tile += nwv) {
        f32x16 acc[MT][NJT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][jt][e] = 0.0f;
        const float* __restrict__ Bp = Bs + li * ldb + (U8IN ? 16 : 4) * lh;
        float4 bcur[NJT][NB], bnxt[NJT][NB];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int q = 0; q < NB; ++q) bcur[jt][q] = *reinterpret_cast<const float4*>(Bp + jt * 32 * ldb + 4 * q);
        for (int blk = 0; blk < nblocks; ++blk) {
#pragma unroll
            for (int d = 0; d < kRing; ++d) {
                Bp += U8IN ? 32 : 8;                  // B fragments of the NEXT chunk are read under this chunk's MFMAs
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int q = 0; q < NB; ++q) bnxt[jt][q] = *reinterpret_cast<const float4*>(Bp + jt * 32 * ldb + 4 * q);
                u32x4 A[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    A[m] = ring[m][d];
                    if (PAD && !((vmask[m] >> d) & 1u)) A[m] = (u32x4){0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int q = 0; q < NB; ++q) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {       // k-pair c of this fragment
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const float a = U8IN ? u8_tap(A[m][q], c) : __uint_as_float(A[m][c]);
#pragma unroll
                            for (int jt = 0; jt < NJT; ++jt) {
                                const float b = c == 0 ? bcur[jt][q].x : c == 1 ? bcur[jt][q].y : c == 2 ? bcur[jt][q].z : bcur[jt][q].w;
                                acc[m][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m][jt], 0, 0, 0);
                            }
                        }
                    }
                }
                fetch(d);                            // refill the slot just consumed (kRing chunks ahead)
                advance(d == kRing - 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the refill HERE: hipcc otherwise sinks all kRing loads to the
                                                     // loop tail and the first one is awaited one instruction later
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int q = 0; q < NB; ++q) bcur[jt][q] = bnxt[jt][q];
            }
        }
        // ---- epilogue: destination offset of THIS lane's pixel, fetched per accumulator row by a wave shuffle
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int myoff = -1;
            {
                const unsigned p = (unsigned)tile * (unsigned)TP + (unsigned)(32 * m + li);
                if (p < P) {
                    const unsigned img = p / per_img, rem = p - img * per_img;
                    const unsigned gy = rem / GX, gx = rem - gy * GX;
                    myoff = (int)(((img * (unsigned)g.DH + (gy * (unsigned)g.DM + (unsigned)day)) * (unsigned)g.DW +
                                   (gx * (unsigned)g.DM + (unsigned)dax)) * (unsigned)g.DC);
                }
            }
            int offs[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) offs[e] = __shfl(myoff, (e & 3) + 8 * (e >> 2) + 4 * lh, 64);
            if (EPI == EPI_MASK) {
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    // CLS4: channel tile jt is parity class (jt>>1, jt&1) of the stride-2 data gradient -> its own pixel
                    const int noff = CLS4 ? ((jt >> 1) * g.DW + (jt & 1)) * g.DC + li : jt * 32 + li;
                    float mk[16];                        // the 16 mask loads of a tile are in flight together
#pragma unroll
                    for (int e = 0; e < 16; ++e) mk[e] = mask_src[(size_t)(offs[e] >= 0 ? offs[e] : 0) + noff];
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (offs[e] >= 0) dst[(size_t)offs[e] + noff] = mk[e] > 0.0f ? acc[m][jt][e] : 0.0f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int jt = 0; jt < NJT; ++jt) {
                        float v = acc[m][jt][e];
                        if (EPI == EPI_BIAS_RELU) {
                            v = v + bias_r[jt];
                            v = v > 0.0f ? v : 0.0f;
                        }
                        if (offs[e] >= 0) dst[(size_t)offs[e] + jt * 32 + li] = v;
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ kernel F
// conv_fixed_kernel: kernel S with the layer geometry as template constants and buffer (bounds-checked, 32-bit
// offset) memory instructions -- what remains between two MFMAs is one buffer_load per 4*NJT*MT MFMAs and the LDS
// fragment reads; all tap addressing is immediates / SGPR offsets:
//   * the chunk loop of a tile is fully unrolled (8 .. 72 chunks), so tap row r, offset within the run and ring slot
//     are compile-time; the refill of chunk c+8 comes from this tile or, past its end, from the next tile whose
//     per-lane base offsets were computed one tile ahead;
//   * zero padding of the data-gradient problems is the hardware's range check: an invalid tap loads from offset
//     0xFFFFF000 (> num_records) and returns 0 -- no masks, no selects when the fragment is consumed;
//   * the epilogue stores through a buffer resource as well: rows beyond P get the same out-of-range offset and are
//     dropped by the range check, so there is no per-row predication.
// f32 sources and every destination must be < 4 GiB (checked by the launcher).  The uint8 source of layer 1 (the
// whole rollout buffer, which may exceed 4 GiB) keeps 64-bit global loads; it needs no padding.
template <int H_, int W_, int C_, int KH_, int KW_, int GY_, int GX_, int SS_, int OFF_, int DH_, int DW_, int DC_, int DM_>
struct FixedGeom {
    static constexpr int H = H_, W = W_, C = C_, KH = KH_, KW = KW_, GY = GY_, GX = GX_, SS = SS_, OFF = OFF_, DH = DH_,
                         DW = DW_, DC = DC_, DM = DM_;
    static constexpr int K = KH_ * KW_ * C_, RUN = KW_ * C_, PITCH = W_ * C_, PER_IMG = GY_ * GX_;
};
using GeomConv1 = FixedGeom<84, 84, 4, 8, 8, 20, 20, 4, 0, 20, 20, 32, 1>;
using GeomConv2 = FixedGeom<20, 20, 32, 4, 4, 9, 9, 2, 0, 9, 9, 64, 1>;
using GeomConv3 = FixedGeom<9, 9, 64, 3, 3, 7, 7, 1, 0, 7, 7, 64, 1>;
using GeomDgrad3 = FixedGeom<7, 7, 64, 3, 3, 9, 9, 1, -2, 9, 9, 64, 1>;       // source = dz3, destination = da2
using GeomDgrad2 = FixedGeom<9, 9, 64, 2, 2, 10, 10, 1, -1, 20, 20, 32, 2>;   // source = dz2, destination = da1 (4 classes)

constexpr int kC3_total = 81 * 4096;         // floats of the per-class layer-3 data-gradient matrices: (1+2+3+2+1)^2 * 64 * 64
constexpr int kC2_total = 16 * 128 * 64;     // floats of the per-class layer-2 data-gradient matrices: (1+2+1)^2 taps * (4*32) rows * 64
constexpr unsigned kOob = 0xFFFFF000u;      // buffer offset that is out of range for every tensor < 4 GiB - 4 KiB
constexpr int kRsrcWord3 = 0x00020000;      // raw buffer, 32-bit elements (gfx9 / CDNA resource format)

// Up to four problem classes per launch (blockIdx.y): same tap shape and grid, different source window origin,
// destination origin and weight matrix -- the border classes of the layer-3 data gradient (see launch_dgrad3_classes).
struct ClsParams {
    int offy[4], offx[4];      // source y = gy*SS + offy + r, x = gx*SS + offx + c
    int day[4], dax[4];        // destination pixel = (gy*DM + day, gx*DM + dax)
    int bt_off[4];             // element offset of the class's weight matrix in Bt_all
};

template <class G, int NJT, bool U8IN, int EPI, bool PAD, bool CLS4, int MT, int NW = 8>
__global__ __launch_bounds__(64 * NW) void conv_fixed_kernel(const void* __restrict__ src_v, const int64_t* __restrict__ inds,
                                                         const float* __restrict__ Bt_all, const float* __restrict__ bias,
                                                         const float* __restrict__ mask_src, float* __restrict__ dst,
                                                         unsigned P, int ntiles, unsigned src_bytes, unsigned dst_bytes,
                                                         ClsParams cp) {
    const int cls = blockIdx.y;
    const int offy = cp.offy[cls], offx = cp.offx[cls], day = cp.day[cls], dax = cp.dax[cls];
    constexpr int K = G::K, LDB = K + 4;
    constexpr int CH = U8IN ? G::KH : K / 8;          // chunks per tile (uint8: one 32-byte tap row per chunk)
    constexpr int RPC = U8IN ? 1 : G::RUN / 8;        // chunks per contiguous run
    constexpr int TP = 32 * MT;
    constexpr int NB = U8IN ? 4 : 1;
    static_assert(CH % kRing == 0, "chunk count must be a multiple of the ring depth");
    extern __shared__ __attribute__((aligned(16))) float Bs[];       // [32*NJT][K + 4] (+ 8 slack) [+ bias]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    {
        const float4* __restrict__ Bt4 = reinterpret_cast<const float4*>(Bt_all + cp.bt_off[cls]);
        constexpr unsigned k4 = K / 4, total = (unsigned)(32 * NJT) * k4;
        for (unsigned e = tid; e < total; e += 64 * NW) {
            const unsigned row = e / k4, c = e - row * k4;
            *reinterpret_cast<float4*>(&Bs[row * LDB + c * 4]) = Bt4[e];
        }
        if (EPI == EPI_BIAS_RELU && tid < 32 * NJT) Bs[(32 * NJT) * LDB + 8 + tid] = bias[tid];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src_v), 0, (int)src_bytes, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsrc_dst = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsrc_msk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mask_src), 0,
                                                                              EPI == EPI_MASK ? (int)dst_bytes : 0, kRsrcWord3);
    const int nwv = gridDim.x * NW;

    // per lane and pixel tile: byte offset of tap (0,0) of the lane's pixel (+16 bytes for the upper half-wave), or the
    // 64-bit pointer for uint8 sources; PAD: validity bit r*KW + c
    struct Pix {
        unsigned voff[MT];
        const uint8_t* ptr[MT];
        unsigned vm[MT];
    };
    auto setup = [&](int tile, Pix& px) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned p = (unsigned)tile * (unsigned)TP + (unsigned)(32 * m + li);
            const bool ok = (tile < ntiles) && (p < P);
            const unsigned pp = ok ? p : 0u;
            const unsigned img = pp / (unsigned)G::PER_IMG, rem = pp - img * (unsigned)G::PER_IMG;
            const unsigned gy = rem / (unsigned)G::GX, gx = rem - gy * (unsigned)G::GX;
            const int sy0 = (int)gy * G::SS + offy, sx0 = (int)gx * G::SS + offx;
            if (U8IN) {
                const long long simg = inds ? inds[img] : (long long)img;
                px.ptr[m] = static_cast<const uint8_t*>(src_v) + ((simg * G::H + sy0) * G::W + sx0) * (long long)G::C + 16 * lh;
                px.voff[m] = 0u;
            } else {
                px.ptr[m] = nullptr;
                px.voff[m] = (unsigned)((((int)img * G::H + sy0) * G::W + sx0) * G::C * 4 + 16 * lh);   // may wrap for PAD taps
            }
            unsigned vm = 0u;
            if constexpr (PAD) {
                static_assert(G::KH * G::KW <= 32, "validity mask is 32 bits");
#pragma unroll
                for (int r = 0; r < G::KH; ++r)
#pragma unroll
                    for (int c = 0; c < G::KW; ++c) {
                        const int sy = sy0 + r, sx = sx0 + c;
                        if (ok && sy >= 0 && sy < G::H && sx >= 0 && sx < G::W) vm |= 1u << (r * G::KW + c);
                    }
            }
            px.vm[m] = vm;
        }
    };
    u32x4 ring[MT][kRing];
    // chunk c (compile-time) of the tile described by px -> ring slot c % kRing
    auto fetch = [&](const Pix& px, auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int r = c / RPC, rem = (c % RPC) * 8;                 // tap row, element offset within the run
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (U8IN) {
                ring[m][c % kRing] = *reinterpret_cast<const u32x4*>(px.ptr[m] + r * G::PITCH);
            } else if constexpr (PAD) {
                constexpr int tc = rem / G::C;                            // tap column of this chunk
                const bool ok = (px.vm[m] >> (r * G::KW + tc)) & 1u;
                const unsigned vo = ok ? px.voff[m] + (unsigned)((r * G::PITCH + rem) * 4) : kOob;
                ring[m][c % kRing] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_src, vo, 0, 0);
            } else {
                ring[m][c % kRing] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_src, px.voff[m] + (unsigned)(rem * 4), r * G::PITCH * 4, 0);
            }
        }
    };

    Pix cur, nxt;
    int tile = blockIdx.x * NW + wave;
    setup(tile, cur);
    setup(tile + nwv, nxt);
    // prologue: chunks 0 .. kRing-1 of the first tile
    [&]<int... I>(std::integer_sequence<int, I...>) { (fetch(cur, std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, kRing>{});

    float bias_r[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) bias_r[jt] = (EPI == EPI_BIAS_RELU) ? Bs[(32 * NJT) * LDB + 8 + jt * 32 + li] : 0.0f;
    const float* __restrict__ Bp0 = Bs + li * LDB + (U8IN ? 16 : 4) * lh;

    for (; tile < ntiles; tile += nwv) {
        f32x16 acc[MT][NJT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][jt][e] = 0.0f;
        float4 bcur[NJT][NB], bnxt[NJT][NB];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int q = 0; q < NB; ++q) bcur[jt][q] = *reinterpret_cast<const float4*>(Bp0 + jt * 32 * LDB + 4 * q);

        auto chunk = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int cn = c + 1 < CH ? c + 1 : c;                   // B fragments of the next chunk (clamped at the end)
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    bnxt[jt][q] = *reinterpret_cast<const float4*>(Bp0 + jt * 32 * LDB + cn * (U8IN ? 32 : 8) + 4 * q);
#pragma unroll
            for (int q = 0; q < NB; ++q) {
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float a = U8IN ? u8_tap(ring[m][c % kRing][q], kp) : __uint_as_float(ring[m][c % kRing][kp]);
#pragma unroll
                        for (int jt = 0; jt < NJT; ++jt) {
                            const float b = kp == 0 ? bcur[jt][q].x : kp == 1 ? bcur[jt][q].y : kp == 2 ? bcur[jt][q].z : bcur[jt][q].w;
                            acc[m][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m][jt], 0, 0, 0);
                        }
                    }
                }
            }
            if constexpr (U8IN) {
                if constexpr (c + kRing < CH) fetch(cur, std::integral_constant<int, c + kRing>{});
                else fetch(nxt, std::integral_constant<int, c + kRing - CH>{});
            } else if constexpr (c % 4 == 3) {
                // f32: four consecutive chunks of a lane pair are ONE 128-byte line.  Refill them back to back (slots
                // c-3 .. c, just consumed) so that the line is requested from L2 once and the other three loads hit it
                // while it is still pending / resident; spread over four chunk times the L1 (32 KiB, 8 waves x 64 lines
                // in flight) evicts it in between and every 16-byte read becomes its own L2 request.
                [&]<int... J>(std::integer_sequence<int, J...>) {
                    ((c - 3 + J + kRing < CH ? fetch(cur, std::integral_constant<int, (c - 3 + J + kRing < CH ? c - 3 + J + kRing : 0)>{})
                                             : fetch(nxt, std::integral_constant<int, (c - 3 + J + kRing < CH ? 0 : c - 3 + J + kRing - CH)>{})), ...);
                }(std::make_integer_sequence<int, 4>{});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int q = 0; q < NB; ++q) bcur[jt][q] = bnxt[jt][q];
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { (chunk(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, CH>{});

        // ---- epilogue
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            unsigned myoff = kOob;                   // BYTE offset of channel 0 of this lane's destination pixel
            {
                const unsigned p = (unsigned)tile * (unsigned)TP + (unsigned)(32 * m + li);
                if (p < P) {
                    const unsigned img = p / (unsigned)G::PER_IMG, rem = p - img * (unsigned)G::PER_IMG;
                    const unsigned gy = rem / (unsigned)G::GX, gx = rem - gy * (unsigned)G::GX;
                    myoff = ((img * (unsigned)G::DH + gy * (unsigned)G::DM + (unsigned)day) * (unsigned)G::DW + gx * (unsigned)G::DM + (unsigned)dax) * (unsigned)(G::DC * 4);
                }
            }
            // CLS4: channel tile jt is parity class (jt>>1, jt&1) of the stride-2 data gradient -> its own pixel
            auto noff_of = [](int jt) { return CLS4 ? ((jt >> 1) * G::DW + (jt & 1)) * G::DC * 4 : jt * 128; };
            if constexpr (EPI == EPI_MASK) {
                // The ReLU-mask values are loaded in batches of 8 accumulator rows BEFORE the stores of the batch: written
                // as load -> select -> store per element, every load sits behind a store that may alias it as far as
                // hipcc can tell, so it emits 16 * NJT dependent load / s_waitcnt vmcnt(0) / store round trips per tile --
                // a third of the time of the short-K border classes of the layer-3 data gradient.
#pragma unroll
                for (int e0 = 0; e0 < 16; e0 += 8) {
                    unsigned off[8], mk[8][NJT];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u;
                        off[u] = (unsigned)__shfl((int)myoff, (e & 3) + 8 * (e >> 2) + 4 * lh, 64) + (unsigned)(li * 4);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int jt = 0; jt < NJT; ++jt)
                            mk[u][jt] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_msk, off[u], noff_of(jt), 0);   // 0 when out of range
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int jt = 0; jt < NJT; ++jt) {
                            const float v = __uint_as_float(mk[u][jt]) > 0.0f ? acc[m][jt][e0 + u] : 0.0f;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_dst, off[u], noff_of(jt), 0);   // dropped when out of range
                        }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned off = (unsigned)__shfl((int)myoff, (e & 3) + 8 * (e >> 2) + 4 * lh, 64) + (unsigned)(li * 4);
#pragma unroll
                    for (int jt = 0; jt < NJT; ++jt) {
                        float v = acc[m][jt][e];
                        if (EPI == EPI_BIAS_RELU) {
                            v = v + bias_r[jt];
                            v = v > 0.0f ? v : 0.0f;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_dst, off, noff_of(jt), 0);   // dropped when out of range
                    }
                }
            }
        }
        cur = nxt;
        setup(tile + 2 * nwv, nxt);
    }
}

// ------------------------------------------------------------------------------------------ weight gradient
// ---- weight gradient of layers 2 and 3, third generation (kernel T, "taps"): no LDS, no barrier, no address table.
// With 32 or 64 input channels an MFMA tile (32 columns of dW) is ONE tap (kh, kw) x 32 input channels, so the B operand
// of tile (kh, kw) for an output pixel is the dword `src[y*SS + kh][x*SS + kw][c0 + li]` -- 32 consecutive channels of one
// source pixel, a whole 128-byte line per half-wave, already in operand layout as it lies in memory.  Consecutive output
// pixels of a row share KW - SS of their KW source columns, so a wave keeps a sliding window of source columns in registers
// and loads only the SS new columns (x its tap rows) per step, straight from global memory / L1, D steps ahead of use.
//   * the two pixels of an MFMA k-pair are the SAME output pixel of TWO images (lh = image parity): no odd-grid padding
//     pair, and both half-waves use identical compile-time offsets from their own image pointers;
//   * wave w owns channel half ci = w & 1 of dW's rows and a quarter of the taps x input channels: layer 3 (3x3, 64 ch)
//     all 9 taps of input-channel half w >> 1; layer 2 (4x4, 32 ch) tap rows 2(w >> 1), 2(w >> 1) + 1, all 4 columns;
//   * the whole image (49 / 81 steps) is unrolled: every load is (per-image pointer + immediate); the image-pair loop is
//     unrolled by two so that the prefetch of the next pair lands in the other half of the (logical) register arrays.
// One partial dW per workgroup (same reduce kernels as every weight-gradient kernel).
//   PAIR (layer 2, default; measured 1.52 -> 1.41 ms per entry point, bit-identical dW): with 32 input channels
//   two neighbouring source columns are 256 contiguous bytes, so ONE 8-byte load per lane fetches both -- lanes 0-15 hold
//   channels (2l, 2l+1) of column x, lanes 16-31 of column x+1 -- and the two tiles it feeds are {tap kw, tap kw+1} x
//   {even, odd channels}: a permutation of dW's columns (both taps multiply the same dz fragment), undone when the partial
//   is written.  Two loads per step instead of four beside the same 8 MFMAs.
template <class G, int KHW, int CSPLIT, int D, bool PAIR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_taps_kernel(
    const float* __restrict__ src, const float* __restrict__ dz,
    float* __restrict__ part_w,      // [grid][Cout][K]
    float* __restrict__ part_b,      // [grid][Cout]
    int images) {
    constexpr int S = G::PER_IMG, GX = G::GX, KW = G::KW, SS = G::SS, W = G::W, Cin = G::C, Cout = G::DC, K = G::K;
    constexpr int TPW = KHW * KW, kSrcImg = G::H * G::W * G::C, kDzImg = S * Cout;
    static_assert(Cout == 64 && Cin == 32 * CSPLIT && (CSPLIT == 2 ? KHW == G::KH : 2 * KHW == G::KH) && D < GX, "wave -> tile map");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int ci = wave & 1, sub = wave >> 1;
    const int kh0 = CSPLIT == 1 ? sub * KHW : 0, c0 = CSPLIT == 2 ? sub * 32 : 0;
    const int npairs = (images + 1) >> 1, gstep = gridDim.x;

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    float bsum = 0.0f;

    auto img_of = [&](int pair) { const int i = 2 * pair + lh; return i < images ? i : images - 1; };
    static_assert(!PAIR || (CSPLIT == 1 && KW == 4 && SS == 2 && (W & 1) == 0), "PAIR is the layer-2 configuration");
    auto psrc = [&](int pair) { return src + (long long)img_of(pair) * kSrcImg + (kh0 * W * Cin + c0 + (PAIR ? 2 * li : li)); };
    auto pdz = [&](int pair) { return dz + (long long)img_of(pair) * kDzImg + (ci * 32 + li); };

    // Register "arrays" (every index below is a compile-time constant once the loops are unrolled; kept small, or hipcc
    // leaves them in scratch): the dz ring continues across pairs, which needs 2 S % (D + 1) == 0 with the pair loop
    // unrolled by two; the window rows alternate by output-row parity, which continues across pairs because GY is odd.
    constexpr int DR = D + 1;
    static_assert((2 * S) % DR == 0 && (G::GY & 1) == 1, "ring / row-parity continuity across image pairs");
    float a[DR];                       // dz fragments: step s of pair parity P lives in a[(s + P * S) % DR]
    float bw[2][KHW][W];               // source-column window [(output row + P) & 1][tap row][source column]
    float2 bw2[2][KHW][W / 2];         // PAIR: [..][tap row][source column pair]
    auto issue = [&](int P, int t, const float* ps, const float* pd) {      // the loads step t of a pair needs and step t-1 did not
        const int gy = t / GX, gx = t % GX;
        a[(t + P * S) % DR] = pd[t * Cout];
        if constexpr (PAIR) {
#pragma unroll
            for (int r = 0; r < KHW; ++r)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    if (gx == 0 || pp == 1)
                        bw2[(gy + P) & 1][r][gx + pp] = *reinterpret_cast<const float2*>(ps + ((gy * SS + r) * W + 2 * (gx + pp)) * Cin);
        } else {
#pragma unroll
            for (int r = 0; r < KHW; ++r)
#pragma unroll
                for (int c = 0; c < KW; ++c)
                    if (gx == 0 || c >= KW - SS) bw[(gy + P) & 1][r][gx * SS + c] = ps[((gy * SS + r) * W + gx * SS + c) * Cin];
        }
    };
    auto run = [&](int P, int pair_raw) {
        // a workgroup with an odd number of pairs runs one dead pair (all dz fragments forced to 0) rather than leaving the
        // loop between its two halves: a mid-loop exit makes hipcc shuffle the accumulators through scratch at the join
        const bool valid = pair_raw < npairs;
        const int pair = valid ? pair_raw : npairs - 1;
        const int nxt = pair + gstep < npairs ? pair + gstep : pair;         // clamped: the refills past the end are never consumed
        const float* const ps = psrc(pair);
        const float* const pd = pdz(pair);
        const float* const psn = psrc(nxt);
        const float* const pdn = pdz(nxt);
        const bool live = valid && 2 * pair + lh < images;                   // (also: the second image of the last pair of an odd batch)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int gy = s / GX, gx = s % GX;
            const float av = live ? a[(s + P * S) % DR] : 0.0f;
            bsum += av;
            if constexpr (PAIR) {
#pragma unroll
                for (int r = 0; r < KHW; ++r)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const float2 v = bw2[(gy + P) & 1][r][gx + pp];
                        acc[r * KW + 2 * pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, v.x, acc[r * KW + 2 * pp], 0, 0, 0);
                        acc[r * KW + 2 * pp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, v.y, acc[r * KW + 2 * pp + 1], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int r = 0; r < KHW; ++r)
#pragma unroll
                    for (int c = 0; c < KW; ++c)
                        acc[r * KW + c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[(gy + P) & 1][r][gx * SS + c], acc[r * KW + c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + D < S) issue(P, s + D, ps, pd);
            else issue(P ^ 1, s + D - S, psn, pdn);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int pair = blockIdx.x;
    if (pair >= npairs) return;                                              // (grid <= pairs: never taken)
    {
        const float* const ps = psrc(pair);
        const float* const pd = pdz(pair);
#pragma unroll
        for (int t = 0; t < D; ++t) issue(0, t, ps, pd);
    }
    for (; pair < npairs; pair += 2 * gstep) {
        run(0, pair);
        run(1, pair + gstep);
    }

    float* pw = part_w + (size_t)blockIdx.x * Cout * K;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        // PAIR: tile (r, 2pp + which) holds, in lane li, tap column 2pp + (li >> 4), channel 2 (li & 15) + which
        const int kcol = PAIR ? ((kh0 + t / KW) * KW + 2 * ((t % KW) >> 1) + (li >> 4)) * Cin + 2 * (li & 15) + (t & 1)
                              : ((kh0 + t / KW) * KW + t % KW) * Cin + c0 + li;
#pragma unroll
        for (int e = 0; e < 16; ++e) pw[(size_t)(ci * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * K + kcol] = acc[t][e];
    }
    const float both = bsum + __shfl_xor(bsum, 32, 64);
    if (lh == 0 && sub == 0) part_b[(size_t)blockIdx.x * Cout + ci * 32 + li] = both;
}

// dW (torch layout (N, C, KH, KW)) and db from the per-workgroup partials, fixed summation order, two stages so
// that the first (which reads all partials) has (N*K/256) x nchunks workgroups instead of N*K/256.
constexpr int kRedChunk = 32;     // partials folded per stage-1 workgroup row

__global__ __launch_bounds__(256) void conv_wgrad_reduce1(const float* __restrict__ part_w, int nparts, int NK,
                                                          float* __restrict__ mid,          // mid: [nchunks][NK]
                                                          const float* __restrict__ part_b, int nparts_b, int N,
                                                          float* __restrict__ mid_b) {      // mid_b: [nchunks][N]
    if (blockIdx.x == gridDim.x - 1) {                // the extra block column folds the bias partials of this chunk
        const int n = threadIdx.x;
        if (n >= N) return;
        const int p0 = blockIdx.y * kRedChunk;
        const int p1 = p0 + kRedChunk < nparts_b ? p0 + kRedChunk : nparts_b;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = p0; p < p1; ++p) s4[p & 3] += part_b[(size_t)p * N + n];
        mid_b[blockIdx.y * N + n] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        return;
    }
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= NK) return;
    const int p0 = blockIdx.y * kRedChunk;
    const int p1 = p0 + kRedChunk < nparts ? p0 + kRedChunk : nparts;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int p = p0;
    if (p1 - p0 == kRedChunk) {                       // a whole chunk: its 32 loads in flight together, added in the order of the loops below
        float v[kRedChunk];
#pragma unroll
        for (int u = 0; u < kRedChunk; ++u) v[u] = part_w[(size_t)(p0 + u) * NK + e];
#pragma unroll
        for (int u = 0; u < kRedChunk; ++u) s8[u & 7] += v[u];
        p = p1;
    }
    for (; p + 8 <= p1; p += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += part_w[(size_t)(p + u) * NK + e];
    }
    for (; p < p1; ++p) s8[p & 7] += part_w[(size_t)p * NK + e];
    mid[(size_t)blockIdx.y * NK + e] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce2(const float* __restrict__ mid, int nchunks,
                                                          const float* __restrict__ mid_b, int nchunks_b, int N, int C,
                                                          int KH, int KW, float scale, float* __restrict__ dW,
                                                          float* __restrict__ db) {
    const int K = KH * KW * C;
    const int e = blockIdx.x * 256 + threadIdx.x;     // index into [N][K] (tap-major, channel-minor)
    if (e < N * K) {
        float s = 0.0f;
        int q = 0;
        for (; q + 16 <= nchunks; q += 16) {          // 16 loads in flight, the same order of additions
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = mid[(size_t)(q + u) * N * K + e];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; q < nchunks; ++q) s += mid[(size_t)q * N * K + e];
        const int n = e / K, k = e - n * K;
        const int r = k / (KW * C), rem = k - r * (KW * C), c = rem / C, ch = rem - c * C;
        dW[((n * C + ch) * KH + r) * KW + c] = s * scale;
    }
    if (blockIdx.x == 0 && threadIdx.x < N && db) {
        float s = 0.0f;
        for (int c = 0; c < nchunks_b; ++c) s += mid_b[c * N + threadIdx.x];
        db[threadIdx.x] = s;
    }
}

// Both stages in ONE launch when the partials are at most 8 chunks (256 workgroups' partials) -- taken for layer 1 only, see the launcher: a workgroup = 32 consecutive
// elements x 8 chunks; thread (chunk c, element x) folds chunk c exactly as conv_wgrad_reduce1 does, the chunk sums meet in LDS, and the threads of chunk 0
// add them in conv_wgrad_reduce2's order and scatter into torch's layout -- the same additions in the same order (bit-identical), one dependent launch
// (5.9 us + the gap in front of it) less per layer and minibatch.  The last workgroup folds the bias partials the same way.
__global__ __launch_bounds__(256) void conv_wgrad_reduce12(const float* __restrict__ part_w, int nparts, int NK, const float* __restrict__ part_b, int N, int C,
                                                           int KH, int KW, float scale, float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float cs[8][64];
    const int c = threadIdx.x >> 5, x = threadIdx.x & 31, nchunks = (nparts + kRedChunk - 1) / kRedChunk;
    const int p0 = c * kRedChunk, p1 = p0 + kRedChunk < nparts ? p0 + kRedChunk : nparts;
    if (blockIdx.x == gridDim.x - 1) {                // bias: channels x and x + 32
        for (int n = x; n < N; n += 32) {
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int p = p0; p < p1; ++p) s4[p & 3] += part_b[(size_t)p * N + n];
            cs[c][n] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        __syncthreads();
        if (c == 0 && db)
            for (int n = x; n < N; n += 32) {
                float s = 0.0f;
                for (int q = 0; q < nchunks; ++q) s += cs[q][n];
                db[n] = s;
            }
        return;
    }
    const int e = blockIdx.x * 32 + x;                // NK % 32 == 0 (host-checked)
    if (c < nchunks) {
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int p = p0;
        if (p1 - p0 == kRedChunk) {                   // a whole chunk: its 32 loads in flight together, added in the order of the loops below
            float v[kRedChunk];
#pragma unroll
            for (int u = 0; u < kRedChunk; ++u) v[u] = part_w[(size_t)(p0 + u) * NK + e];
#pragma unroll
            for (int u = 0; u < kRedChunk; ++u) s8[u & 7] += v[u];
            p = p1;
        }
        for (; p + 8 <= p1; p += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] += part_w[(size_t)(p + u) * NK + e];
        }
        for (; p < p1; ++p) s8[p & 7] += part_w[(size_t)p * NK + e];
        cs[c][x] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
    __syncthreads();
    if (c == 0) {
        float s = 0.0f;
        for (int q = 0; q < nchunks; ++q) s += cs[q][x];
        const int K = KH * KW * C;
        const int n = e / K, k = e - n * K;
        const int r = k / (KW * C), rem = k - r * (KW * C), cc = rem / C, ch = rem - cc * C;
        dW[((n * C + ch) * KH + r) * KW + cc] = s * scale;
    }
}

// ------------------------------------------------------------------------------------------ weight repack
// mode 0 (forward):           Bt[n=cout][(r,c,cin)]          = W[cout][cin][r][c]
// mode 1 (dgrad, stride 1):   Bt[n=cin][(r,c,cout)]          = W[cout][cin][KH-1-r][KW-1-c]
// mode 2 (dgrad, stride 2):   Bt[cls][n=cin][(r,c,cout)]     = W[cout][cin][ph+2-2r][pw+2-2c], cls = 2*ph+pw, r,c in {0,1}
// mode 5 (dgrad, stride 2, per border class): the same, one [4*cin][(r',c',cout)] matrix per class of the class grid
__global__ __launch_bounds__(256) void conv_repack_kernel(const float* __restrict__ W, float* __restrict__ Bt, int Cout,
                                                          int Cin, int KH, int KW, int mode, float scale) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (mode == 0) {
        const int K = KH * KW * Cin;
        if (e >= Cout * K) return;
        const int n = e / K, k = e - n * K;
        const int r = k / (KW * Cin), rem = k - r * (KW * Cin), c = rem / Cin, ch = rem - c * Cin;
        Bt[e] = W[((n * Cin + ch) * KH + r) * KW + c] * scale;
    } else if (mode == 1) {
        const int K = KH * KW * Cout;
        if (e >= Cin * K) return;
        const int n = e / K, k = e - n * K;
        const int r = k / (KW * Cout), rem = k - r * (KW * Cout), c = rem / Cout, co = rem - c * Cout;
        Bt[e] = W[((co * Cin + n) * KH + (KH - 1 - r)) * KW + (KW - 1 - c)];
    } else if (mode == 3) {
        // per-class matrices of the layer-3 data gradient: class (a,b), [n=cin][(r',c',cout)], taps flipped as in mode 1
        const int r0t[5] = {2, 1, 0, 0, 0}, nrt[5] = {1, 2, 3, 2, 1};
        int off = 0;
        for (int a = 0; a < 5; ++a)
            for (int b = 0; b < 5; ++b) {
                const int K = nrt[a] * nrt[b] * Cout, sz = Cin * K;
                if (e >= off && e < off + sz) {
                    const int e2 = e - off, n = e2 / K, k = e2 - n * K;
                    const int rr = k / (nrt[b] * Cout), rem = k - rr * (nrt[b] * Cout), cc = rem / Cout, co = rem - cc * Cout;
                    const int r = r0t[a] + rr, c = r0t[b] + cc;
                    Bt[e] = W[((co * Cin + n) * KH + (KH - 1 - r)) * KW + (KW - 1 - c)];
                    return;
                }
                off += sz;
            }
    } else if (mode == 5) {
        // per-class matrices of the layer-2 data gradient: border class (a,b) of the 10x10 class grid, [4 parity classes x
        // n=cin][(r',c',cout)], taps as in mode 2 restricted to the class's valid window
        const int r0t[3] = {1, 0, 0}, nrt[3] = {1, 2, 1};
        int off = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const int K = nrt[a] * nrt[b] * Cout, sz = 4 * Cin * K;
                if (e >= off && e < off + sz) {
                    const int e2 = e - off, row = e2 / K, k = e2 - row * K;
                    const int cls = row / Cin, n = row - cls * Cin;
                    const int rr = k / (nrt[b] * Cout), rem = k - rr * (nrt[b] * Cout), cc = rem / Cout, co = rem - cc * Cout;
                    const int r = r0t[a] + rr, c = r0t[b] + cc;
                    const int ph = cls >> 1, pw = cls & 1;
                    Bt[e] = W[((co * Cin + n) * KH + (ph + 2 - 2 * r)) * KW + (pw + 2 - 2 * c)];
                    return;
                }
                off += sz;
            }
    } else {
        const int K = 4 * Cout;                      // 2x2 taps
        if (e >= 4 * Cin * K) return;
        const int cls = e / (Cin * K), e2 = e - cls * (Cin * K);
        const int n = e2 / K, k = e2 - n * K;
        const int r = k / (2 * Cout), rem = k - r * (2 * Cout), c = rem / Cout, co = rem - c * Cout;
        const int ph = cls >> 1, pw = cls & 1;
        Bt[e] = W[((co * Cin + n) * KH + (ph + 2 - 2 * r)) * KW + (pw + 2 - 2 * c)];
    }
}

constexpr int kDefaultVariant = 2;
constexpr long long kBufLimit = (1LL << 32) - 8192;   // largest tensor kernel F's 32-bit buffer offsets can address
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace mi355ppo

using namespace mi355ppo;

// layer: 1 = Conv2d(4,32,8,s4) on 84x84, 2 = Conv2d(32,64,4,s2) on 20x20, 3 = Conv2d(64,64,3,s1) on 9x9
static bool layer_dims(int layer, int* Cin, int* Cout, int* KH, int* SS, int* Hin, int* Hout) {
    switch (layer) {
        case 1: *Cin = 4; *Cout = 32; *KH = 8; *SS = 4; *Hin = 84; *Hout = 20; return true;
        case 2: *Cin = 32; *Cout = 64; *KH = 4; *SS = 2; *Hin = 20; *Hout = 9; return true;
        case 3: *Cin = 64; *Cout = 64; *KH = 3; *SS = 1; *Hin = 9; *Hout = 7; return true;
    }
    return false;
}

extern "C" MI355PPO_API int mi355ppo_cnn_repack_weights_f32(const float* W, float* Bt, int layer, int mode, void* stream) {
    const char* fn = "mi355ppo_cnn_repack_weights_f32";
    int Cin, Cout, KH, SS, Hin, Hout;
    MI355_REQUIRE(W && Bt, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(layer_dims(layer, &Cin, &Cout, &KH, &SS, &Hin, &Hout), MI355PPO_EINVAL, "%s: layer=%d must be 1..3", fn, layer);
    if (mode == 4 && layer == 1) return mi355ppo_cnn_conv1q_pack(W, Bt, stream);   // integer-digit pack of kernel Q (conv1q.hip)
    MI355_REQUIRE(mode == 0 || ((mode == 1 || mode == 3) && layer == 3) || ((mode == 2 || mode == 5) && layer == 2), MI355PPO_EINVAL,
                  "%s: mode %d is not defined for layer %d", fn, mode, layer);
    const int total = mode == 3 ? kC3_total : mode == 5 ? kC2_total : Cout * Cin * KH * KH;
    hipLaunchKernelGGL(conv_repack_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), W, Bt, Cout, Cin, KH,
                       KH, mode, layer == 1 ? kInv255 : 1.0f);     // layer 1 consumes raw uint8 taps
    return check_launch("conv_repack_kernel");
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) and calling thread: the launch path itself
// then consists of nothing but the launch (legal inside a stream capture).
static hipError_t ensure_dynamic_lds(const void* fn, size_t bytes) {
    struct Seen { const void* fn; int dev; size_t bytes; };
    static thread_local Seen seen[64];
    static thread_local int nseen = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    for (int i = 0; i < nseen; ++i)
        if (seen[i].fn == fn && seen[i].dev == dev && seen[i].bytes >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && nseen < 64) seen[nseen++] = Seen{fn, dev, bytes};
    return e;
}

template <int NJT, bool U8IN, int EPI, bool PAD, bool CLS4, int MT, int NW>
static int launch_stream_cfg(const void* src, const int64_t* inds, const float* Bt, const float* bias, const float* mask_src,
                         float* dst, const ConvGeom& g, hipStream_t s) {
    const size_t smem = ((size_t)(32 * NJT) * (g.K + 4) + 8 + 32 * NJT) * sizeof(float);   // weights + slack + bias
    auto k = conv_stream_kernel<NJT, U8IN, EPI, PAD, CLS4, MT, NW>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), smem);
    if (e != hipSuccess) {
        set_error("conv_stream_kernel: hipFuncSetAttribute(%zu bytes of LDS): %s", smem, hipGetErrorString(e));
        return MI355PPO_EHIP;
    }
    const int ntiles = (int)((g.P + 32 * MT - 1) / (32 * MT));
    // persistent workgroups of 8 waves: one per CU when the weights fill the LDS, two when they are small
    // (one 8-wave workgroup per CU: the kernels use 160-240 VGPRs, so two would not be co-resident anyway)
    int wgs = 256 / g.classes;
    const int need = (ntiles + NW - 1) / NW;
    if (wgs > need) wgs = need;
    hipLaunchKernelGGL(k, dim3((unsigned)wgs, (unsigned)g.classes), dim3(64 * NW), smem, s, src, inds, Bt, bias, mask_src, dst, g,
                       ntiles);
    return check_launch("conv_stream_kernel");
}

template <class G, int NJT, bool U8IN, int EPI, bool PAD, bool CLS4, int MT, int NW = 8, int WGS_PER_CU = 1>
static int launch_fixed_cfg(const void* src, const int64_t* inds, const float* Bt, const float* bias, const float* mask_src,
                            float* dst, long long P, long long src_bytes, long long dst_bytes, hipStream_t s,
                            const ClsParams* cpp = nullptr, int ncls = 1) {
    ClsParams cp{};
    if (cpp) cp = *cpp;
    else { cp.offy[0] = cp.offx[0] = G::OFF; }
    if ((!U8IN && src_bytes > kBufLimit) || dst_bytes > kBufLimit) {
        set_error("conv_fixed_kernel: tensors must stay below 4 GiB (source %lld, destination %lld bytes): split the batch",
                  src_bytes, dst_bytes);
        return MI355PPO_EINVAL;
    }
    const size_t smem = ((size_t)(32 * NJT) * (G::K + 4) + 8 + 32 * NJT) * sizeof(float);
    auto k = conv_fixed_kernel<G, NJT, U8IN, EPI, PAD, CLS4, MT, NW>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), smem);
    if (e != hipSuccess) {
        set_error("conv_fixed_kernel: hipFuncSetAttribute(%zu bytes of LDS): %s", smem, hipGetErrorString(e));
        return MI355PPO_EHIP;
    }
    const int ntiles = (int)((P + 32 * MT - 1) / (32 * MT));
    int wgs = 256 * WGS_PER_CU / ncls;
    const int need = (ntiles + NW - 1) / NW;
    if (wgs > need) wgs = need;
    hipLaunchKernelGGL(k, dim3((unsigned)wgs, (unsigned)ncls), dim3(64 * NW), smem, s, src, inds, Bt, bias, mask_src, dst, (unsigned)P,
                       ntiles, (unsigned)(U8IN ? 0 : src_bytes), (unsigned)dst_bytes, cp);
    return check_launch("conv_fixed_kernel");
}
// 64-pixel wave tiles for minibatch-sized problems, 32-pixel tiles when that leaves waves of the chip idle (rollout)
template <class G, int NJT, bool U8IN, int EPI, bool PAD, bool CLS4 = false, int MTBIG = 2>
static int launch_fixed(const void* src, const int64_t* inds, const float* Bt, const float* bias, const float* mask_src,
                        float* dst, long long P, long long src_bytes, long long dst_bytes, hipStream_t s) {
    if (MTBIG == 1 || P < 64LL * 4 * 2048)
        return launch_fixed_cfg<G, NJT, U8IN, EPI, PAD, CLS4, 1>(src, inds, Bt, bias, mask_src, dst, P, src_bytes, dst_bytes, s);
    return launch_fixed_cfg<G, NJT, U8IN, EPI, PAD, CLS4, MTBIG>(src, inds, Bt, bias, mask_src, dst, P, src_bytes, dst_bytes, s);
}

// ---- layer-3 data gradient without its structural zeros.  da2 (9x9) = full correlation of dz3 (7x7) with the flipped
// 3x3 taps: a destination row iy only has the taps r with 0 <= iy-2+r < 7, so 288 of the 729 (pixel, tap) products of an
// image are zeros of the padding.  Destination rows / columns fall into five classes {0},{1},{2..6},{7},{8} with tap
// windows of 1,2,3,2,1 taps; a (row class, column class) pair is a dense, un-padded convolution with its own tap
// window.  Classes with the same window SHAPE share a kernel instantiation and go into one launch (blockIdx.y).
static const int kC3_r0[5] = {2, 1, 0, 0, 0}, kC3_nr[5] = {1, 2, 3, 2, 1}, kC3_p0[5] = {0, 1, 2, 7, 8};   // rows per class: 1,1,5,1,1
static int c3_bt_offset(int a, int b) {      // element offset of class (a,b)'s matrix [64][nr*nc*64] in the mode-3 repack
    int off = 0;
    for (int aa = 0; aa < 5; ++aa)
        for (int bb = 0; bb < 5; ++bb) {
            if (aa == a && bb == b) return off;
            off += 64 * kC3_nr[aa] * kC3_nr[bb] * 64;
        }
    return off;
}

template <int NR, int NC, int GY, int GX>
static int launch_dgrad3_group(const float* dz, const float* Bt, const float* act_in, float* dsrc, long long images,
                               long long srcb, long long dstb, hipStream_t s) {
    using G = FixedGeom<7, 7, 64, NR, NC, GY, GX, 1, 0, 9, 9, 64, 1>;
    ClsParams cp{};
    int n = 0;
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            if (kC3_nr[a] == NR && kC3_nr[b] == NC) {
                cp.offy[n] = kC3_p0[a] - 2 + kC3_r0[a];
                cp.offx[n] = kC3_p0[b] - 2 + kC3_r0[b];
                cp.day[n] = kC3_p0[a];
                cp.dax[n] = kC3_p0[b];
                cp.bt_off[n] = c3_bt_offset(a, b);
                ++n;
            }
    const long long P = images * GY * GX;
    if (GY * GX >= 5 && P >= 64LL * 4 * 2048 / n)
        return launch_fixed_cfg<G, 2, false, EPI_MASK, false, false, 2>((const void*)dz, nullptr, Bt, nullptr, act_in, dsrc, P, srcb, dstb, s, &cp, n);
    return launch_fixed_cfg<G, 2, false, EPI_MASK, false, false, 1>((const void*)dz, nullptr, Bt, nullptr, act_in, dsrc, P, srcb, dstb, s, &cp, n);
}
static int launch_dgrad3_classes(const float* dz, const float* Bt, const float* act_in, float* dsrc, long long images,
                                 long long srcb, long long dstb, hipStream_t s) {
    int rc;
    if ((rc = launch_dgrad3_group<3, 3, 5, 5>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad3_group<2, 3, 1, 5>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad3_group<3, 2, 5, 1>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad3_group<1, 3, 1, 5>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad3_group<3, 1, 5, 1>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad3_group<2, 2, 1, 1>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad3_group<1, 2, 1, 1>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad3_group<2, 1, 1, 1>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    return launch_dgrad3_group<1, 1, 1, 1>(dz, Bt, act_in, dsrc, images, srcb, dstb, s);
}

// ---- layer-2 data gradient without its structural zeros.  da1 (20x20) = four stride-parity classes of 2x2-tap
// correlations of dz2 (9x9) on a 10x10 class grid; class-grid row gy only has the taps r with 0 <= gy-1+r < 9, so 19 % of
// the (pixel, tap) products of an image are zeros of the padding.  Grid rows / columns fall into three classes {0},
// {1..8},{9} with tap windows of 1,2,1 taps; a (row class, column class) pair is a dense, un-padded problem.  The four
// parity classes stay one 128-channel GEMM inside every border class (CLS4).
static const int kC2_r0[3] = {1, 0, 0}, kC2_nr[3] = {1, 2, 1}, kC2_p0[3] = {0, 1, 9};
static int c2_bt_offset(int a, int b) {      // element offset of class (a,b)'s matrix [128][nr*nc*64] in the mode-5 repack
    int off = 0;
    for (int aa = 0; aa < 3; ++aa)
        for (int bb = 0; bb < 3; ++bb) {
            if (aa == a && bb == b) return off;
            off += 128 * kC2_nr[aa] * kC2_nr[bb] * 64;
        }
    return off;
}
template <int NR, int NC, int GY, int GX>
static int launch_dgrad2_group(const float* dz, const float* Bt, const float* act_in, float* dsrc, long long images,
                               long long srcb, long long dstb, hipStream_t s) {
    using G = FixedGeom<9, 9, 64, NR, NC, GY, GX, 1, 0, 20, 20, 32, 2>;
    ClsParams cp{};
    int n = 0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            if (kC2_nr[a] == NR && kC2_nr[b] == NC && (kC2_nr[a] == 2 ? 8 : 1) == GY && (kC2_nr[b] == 2 ? 8 : 1) == GX) {
                cp.offy[n] = kC2_p0[a] - 1 + kC2_r0[a];
                cp.offx[n] = kC2_p0[b] - 1 + kC2_r0[b];
                cp.day[n] = 2 * kC2_p0[a];
                cp.dax[n] = 2 * kC2_p0[b];
                cp.bt_off[n] = c2_bt_offset(a, b);
                ++n;
            }
    const long long P = images * GY * GX;
    return launch_fixed_cfg<G, 4, false, EPI_MASK, false, true, 1>((const void*)dz, nullptr, Bt, nullptr, act_in, dsrc, P, srcb, dstb, s, &cp, n);
}
static int launch_dgrad2_classes(const float* dz, const float* Bt, const float* act_in, float* dsrc, long long images,
                                 long long srcb, long long dstb, hipStream_t s) {
    int rc;
    if ((rc = launch_dgrad2_group<2, 2, 8, 8>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad2_group<1, 2, 1, 8>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    if ((rc = launch_dgrad2_group<2, 1, 8, 1>(dz, Bt, act_in, dsrc, images, srcb, dstb, s))) return rc;
    return launch_dgrad2_group<1, 1, 1, 1>(dz, Bt, act_in, dsrc, images, srcb, dstb, s);
}

template <int NJT, bool U8IN, int EPI, bool PAD, bool CLS4 = false, int MT = 2>
static int launch_stream(const void* src, const int64_t* inds, const float* Bt, const float* bias, const float* mask_src,
                         float* dst, const ConvGeom& g, hipStream_t s) {
    if (NJT == 4 || MT == 1) return launch_stream_cfg<NJT, U8IN, EPI, PAD, CLS4, 1, 8>(src, inds, Bt, bias, mask_src, dst, g, s);
    // small problems (rollout batches): 32-pixel tiles give every wave of the chip something to do
    if (g.P < 64LL * 4 * 2048) return launch_stream_cfg<NJT, U8IN, EPI, PAD, CLS4, 1, 8>(src, inds, Bt, bias, mask_src, dst, g, s);
    return launch_stream_cfg<NJT, U8IN, EPI, PAD, CLS4, 2, 8>(src, inds, Bt, bias, mask_src, dst, g, s);
}

static int conv_fwd_impl(const void* src, const int64_t* inds, const float* Bt, const float* bias, float* dst,
                         int64_t images, int layer, int variant, void* stream) {
    const char* fn = "mi355ppo_cnn_conv_fwd_f32";
    int Cin, Cout, KH, SS, Hin, Hout;
    MI355_REQUIRE(src && Bt && bias && dst, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(layer_dims(layer, &Cin, &Cout, &KH, &SS, &Hin, &Hout), MI355PPO_EINVAL, "%s: layer=%d must be 1..3", fn, layer);
    MI355_REQUIRE(images > 0 && images <= (1 << 22), MI355PPO_EINVAL, "%s: images=%lld out of range (1..4194304)", fn,
                  (long long)images);
    MI355_REQUIRE(layer == 1 || inds == nullptr, MI355PPO_EINVAL, "%s: inds (row gather) is only defined for layer 1", fn);
    MI355_REQUIRE(variant == 0 || variant == 2 || variant == 4 || (variant == 6 && layer == 1), MI355PPO_EINVAL,
                  "%s: unknown variant %d", fn, variant);
    MI355_REQUIRE(aligned(src, 16) && aligned(Bt, 16) && aligned(dst, 16) && aligned(inds, 8), MI355PPO_EALIGN,
                  "%s: src/Bt/dst must be 16-byte aligned", fn);
    if (variant == 6) return mi355ppo_cnn_conv1q_fwd(src, inds, Bt, bias, dst, images, stream);   // Bt = the mode-4 pack
    ConvGeom g;
    g.H = g.W = Hin; g.C = Cin; g.KH = g.KW = KH; g.GY = g.GX = Hout; g.SS = SS; g.OFF = 0;
    g.DH = g.DW = Hout; g.DC = Cout; g.DM = 1; g.DAY = g.DAX = 0; g.K = KH * KH * Cin; g.N = Cout; g.classes = 1;
    g.logC = ilog2(Cin); g.P = (long long)images * Hout * Hout;
    MI355_REQUIRE(g.P * Cout < (1LL << 31), MI355PPO_EINVAL, "%s: destination exceeds 2^31 elements", fn);
    hipStream_t s = as_stream(stream);
    const long long srcb = (long long)images * Hin * Hin * Cin * 4, dstb = g.P * Cout * 4;
    if (variant == 0)   // kernel F addresses f32 tensors with 32-bit buffer offsets; beyond 4 GiB fall back to kernel S
        variant = ((layer > 1 && srcb > kBufLimit) || dstb > kBufLimit) ? 4 : kDefaultVariant;
    if (variant == 2) {
        if (layer == 1) {
            // layer 1 has the shortest tiles (8 chunks): on gfx9 loads and stores share one out-of-order vmcnt, so the tile
            // epilogue's stores force a full drain of the prefetch ring at every tile boundary; three 4-wave workgroups per
            // CU (3 waves per SIMD at 152 VGPRs) give the matrix pipe two other waves to run meanwhile.
            if (g.P >= 64LL * 4 * 2048)
                return launch_fixed_cfg<GeomConv1, 1, true, EPI_BIAS_RELU, false, false, 2, 4, 3>(src, inds, Bt, bias, nullptr, dst, g.P, 0, dstb, s);
            return launch_fixed<GeomConv1, 1, true, EPI_BIAS_RELU, false>(src, inds, Bt, bias, nullptr, dst, g.P, 0, dstb, s);
        }
        if (layer == 2) return launch_fixed<GeomConv2, 2, false, EPI_BIAS_RELU, false>(src, inds, Bt, bias, nullptr, dst, g.P, srcb, dstb, s);
        return launch_fixed<GeomConv3, 2, false, EPI_BIAS_RELU, false>(src, inds, Bt, bias, nullptr, dst, g.P, srcb, dstb, s);
    }
    if (variant == 4) {
        if (layer == 1) return launch_stream<1, true, EPI_BIAS_RELU, false>(src, inds, Bt, bias, nullptr, dst, g, s);
        return launch_stream<2, false, EPI_BIAS_RELU, false>(src, inds, Bt, bias, nullptr, dst, g, s);
    }
    set_error("%s: unknown variant %d", fn, variant);
    return MI355PPO_EINVAL;
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv_fwd_f32(const void* src, const int64_t* inds, const float* Bt, const float* bias,
                                                      float* dst, int64_t images, int layer, void* stream) {
    return conv_fwd_impl(src, inds, Bt, bias, dst, images, layer, 0, stream);
}
extern "C" MI355PPO_API int mi355ppo_cnn_conv_fwd_f32_variant(const void* src, const int64_t* inds, const float* Bt,
                                                              const float* bias, float* dst, int64_t images, int layer,
                                                              int variant, void* stream) {
    return conv_fwd_impl(src, inds, Bt, bias, dst, images, layer, variant, stream);
}

// The three forward layers in one call (rollout inference: the host-side cost of a launch matters there).
extern "C" MI355PPO_API int mi355ppo_cnn_trunk_fwd_f32(const void* obs_u8, const int64_t* inds, const float* bt1, const float* b1,
                                                       const float* bt2, const float* b2, const float* bt3, const float* b3,
                                                       float* a1, float* a2, float* a3, int64_t images, int conv1_variant,
                                                       void* stream) {
    int rc = conv_fwd_impl(obs_u8, inds, bt1, b1, a1, images, 1, conv1_variant, stream);
    if (rc) return rc;
    rc = conv_fwd_impl(a1, nullptr, bt2, b2, a2, images, 2, 0, stream);
    if (rc) return rc;
    return conv_fwd_impl(a2, nullptr, bt3, b3, a3, images, 3, 0, stream);
}

static int conv_dgrad_impl(const float* dz, const float* Bt, const float* act_in, float* dsrc, int64_t images, int layer,
                           int variant, void* stream) {
    const char* fn = "mi355ppo_cnn_conv_dgrad_f32";
    int Cin, Cout, KH, SS, Hin, Hout;
    MI355_REQUIRE(dz && Bt && act_in && dsrc, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE((layer == 2 || layer == 3) && layer_dims(layer, &Cin, &Cout, &KH, &SS, &Hin, &Hout), MI355PPO_EINVAL,
                  "%s: layer=%d must be 2 or 3 (conv1's input needs no gradient)", fn, layer);
    MI355_REQUIRE(images > 0 && images <= (1 << 22), MI355PPO_EINVAL, "%s: images=%lld out of range (1..4194304)", fn,
                  (long long)images);
    MI355_REQUIRE((variant == 0 || variant == 2 || variant == 4 || (variant == 3 && layer == 2) || (variant == 5 && layer == 3) ||
                   (variant == 6 && layer == 2)), MI355PPO_EINVAL, "%s: unknown variant %d", fn, variant);
    MI355_REQUIRE(aligned(dz, 16) && aligned(Bt, 16) && aligned(act_in, 16) && aligned(dsrc, 16), MI355PPO_EALIGN,
                  "%s: pointers must be 16-byte aligned", fn);
    MI355_REQUIRE((long long)images * Hin * Hin * Cin < (1LL << 31), MI355PPO_EINVAL, "%s: destination exceeds 2^31 elements", fn);
    ConvGeom g;
    g.H = g.W = Hout; g.C = Cout;                  // the "source" of this GEMM is dz: (Hout, Hout, Cout)
    g.DH = g.DW = Hin; g.DC = Cin; g.N = Cin; g.logC = ilog2(Cout);
    hipStream_t s = as_stream(stream);
    if (variant == 0)
        variant = ((long long)images * Hout * Hout * Cout * 4 > kBufLimit || (long long)images * Hin * Hin * Cin * 4 > kBufLimit) ? 4 : kDefaultVariant;
    if (layer == 3) {
        g.KH = g.KW = 3; g.GY = g.GX = Hin; g.SS = 1; g.OFF = -2; g.DM = 1; g.DAY = g.DAX = 0; g.classes = 1;
        g.K = 9 * Cout; g.P = (long long)images * Hin * Hin;
        const long long srcb3 = (long long)images * Hout * Hout * Cout * 4, dstb3 = (long long)images * Hin * Hin * Cin * 4;
        if (variant == 5)   // Bt must be the mode-3 (per-class) repack
            return launch_dgrad3_classes(dz, Bt, act_in, dsrc, images, srcb3, dstb3, s);
        if (variant == 2)
            return launch_fixed<GeomDgrad3, 2, false, EPI_MASK, true>((const void*)dz, nullptr, Bt, nullptr, act_in, dsrc, g.P, srcb3, dstb3, s);
        if (variant == 4)
            return launch_stream<2, false, EPI_MASK, true>((const void*)dz, nullptr, Bt, nullptr, act_in, dsrc, g, s);
        set_error("%s: unknown variant %d for layer 3", fn, variant);
        return MI355PPO_EINVAL;
    } else {
        g.KH = g.KW = 2; g.GY = g.GX = Hin / 2; g.SS = 1; g.OFF = -1; g.DM = 2; g.DAY = g.DAX = 0; g.classes = 4;
        g.K = 4 * Cout; g.P = (long long)images * (Hin / 2) * (Hin / 2);
        const long long srcb2 = (long long)images * Hout * Hout * Cout * 4, dstb2 = (long long)images * Hin * Hin * Cin * 4;
        if (variant == 6)       // Bt must be the mode-5 (per border class) repack
            return launch_dgrad2_classes(dz, Bt, act_in, dsrc, images, srcb2, dstb2, s);
        if (variant == 2)       // one pass for all four parity classes: they read the SAME dz taps (only weights and
                                // destination pixel differ), so the 4x32 class channels form one 128-wide GEMM
            return launch_fixed<GeomDgrad2, 4, false, EPI_MASK, true, true, 1>((const void*)dz, nullptr, Bt, nullptr, act_in, dsrc, g.P, srcb2, dstb2, s);
        if (variant == 4) {
            g.classes = 1;
            return launch_stream<4, false, EPI_MASK, true, true, 1>((const void*)dz, nullptr, Bt, nullptr, act_in, dsrc, g, s);
        }
        if (variant == 3)
            return launch_stream<1, false, EPI_MASK, true>((const void*)dz, nullptr, Bt, nullptr, act_in, dsrc, g, s);
        set_error("%s: unknown variant %d for layer 2", fn, variant);
        return MI355PPO_EINVAL;
    }
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv_dgrad_f32(const float* dz, const float* Bt, const float* act_in, float* dsrc,
                                                        int64_t images, int layer, void* stream) {
    return conv_dgrad_impl(dz, Bt, act_in, dsrc, images, layer, 0, stream);
}
extern "C" MI355PPO_API int mi355ppo_cnn_conv_dgrad_f32_variant(const float* dz, const float* Bt, const float* act_in,
                                                                float* dsrc, int64_t images, int layer, int variant,
                                                                void* stream) {
    return conv_dgrad_impl(dz, Bt, act_in, dsrc, images, layer, variant, stream);
}

static int wgrad_grid(int64_t images) { return images < 512 ? (int)images : 512; }
// partial sums the workspace holds: the workgroups of kernels P / T, or kernel V's slabs where those are more (batches of 208 .. 240 images
// at layer 2: 256 slabs -- round 5: sized for `images` partials before, kernel V's last slabs landed in the bias partials)
static int wgrad_parts(int64_t images, int layer) {
    const int g = wgrad_grid(images) * (layer == 1 ? 4 : 1), v = layer == 1 ? 0 : convw_parts(images, layer), u = convu_max_parts();
    return g > v ? (g > u ? g : u) : (v > u ? v : u);
}

extern "C" MI355PPO_API size_t mi355ppo_cnn_conv_wgrad_workspace_bytes(int64_t images, int layer) {
    int Cin, Cout, KH, SS, Hin, Hout;
    if (images <= 0 || !layer_dims(layer, &Cin, &Cout, &KH, &SS, &Hin, &Hout)) return 0;
    const size_t wparts = (size_t)wgrad_parts(images, layer);                       // layer 1: one partial per wave
    const size_t nchunks = (wparts + kRedChunk - 1) / kRedChunk;
    return ((wparts + nchunks) * (size_t)Cout * KH * KH * Cin + (wparts + nchunks) * Cout) * sizeof(float);
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv_wgrad_kernel(int64_t images, int layer) {
    if (images <= 0 || layer < 1 || layer > 3) return 0;
    return layer == 1 ? 'P' : convw_applies(images, layer) ? 'V' : 'T';
}

// 'U', 'V', 'P' or 'T': the kernel mi355ppo_cnn_conv_wgrad_f16x2_f32 (layers 2, 3) / mi355ppo_cnn_conv1_wgrad_f16x2 (layer 1) runs for this batch --
// the launchers' own decision (convu.hip::convu_takes, convw.hip::convw_applies), not a restatement of it
extern "C" MI355PPO_API int mi355ppo_cnn_conv_wgrad_kernel_f16x2(int64_t images, int layer) {
    if (images <= 0 || layer < 1 || layer > 3) return 0;
    if (convu_takes(images, layer)) return 'U';
    return layer == 1 ? 'P' : convw_applies(images, layer, true) ? 'V' : 'T';
}

static int conv_wgrad_impl(const char* fn, const void* src, const int64_t* inds, const float* dz, float* dW, float* db, int64_t images, int layer,
                           void* workspace, size_t workspace_bytes, const unsigned* src_amax, const unsigned* dz_amax, void* stream) {
    int Cin, Cout, KH, SS, Hin, Hout;
    MI355_REQUIRE(src && dz && dW && db, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(layer_dims(layer, &Cin, &Cout, &KH, &SS, &Hin, &Hout), MI355PPO_EINVAL, "%s: layer=%d must be 1..3", fn, layer);
    MI355_REQUIRE(images > 0 && images <= (1 << 24), MI355PPO_EINVAL, "%s: images=%lld out of range", fn, (long long)images);
    MI355_REQUIRE(layer == 1 || inds == nullptr, MI355PPO_EINVAL, "%s: inds (row gather) is only defined for layer 1", fn);
    const size_t need = mi355ppo_cnn_conv_wgrad_workspace_bytes(images, layer);
    MI355_REQUIRE(workspace && workspace_bytes >= need, MI355PPO_EWORKSPACE, "%s: workspace %zu bytes < required %zu", fn,
                  workspace ? workspace_bytes : (size_t)0, need);
    MI355_REQUIRE(aligned(src, 16) && aligned(dz, 16) && aligned(dW, 4) && aligned(db, 4) && aligned(workspace, 16) &&
                      aligned(inds, 8), MI355PPO_EALIGN, "%s: src/dz/workspace must be 16-byte aligned", fn);
    const int K = KH * KH * Cin;
    // Workspace layout (sized by mi355ppo_cnn_conv_wgrad_workspace_bytes for the kernel with the most partials):
    //   part_w [lparts][Cout*K] | part_b [lparts][Cout] | mid [ceil(lparts/32)][Cout*K] | mid_b [ceil(lparts/32)][Cout]
    const int lparts = wgrad_parts(images, layer);
    const int total_w = Cout * K;
    float* part_w = static_cast<float*>(workspace);
    float* part_b = part_w + (size_t)lparts * total_w;
    float* mid = part_b + (size_t)lparts * Cout;
    float* mid_b = mid + (size_t)((lparts + kRedChunk - 1) / kRedChunk) * total_w;
    // Layer 1: kernel P (bf16 pipe, exact products, conv1p.hip).  Layers 2, 3: kernel V (bf16 pipe, convw.hip) for batches it
    // can cut into slabs, else kernel T (taps, f32 pipe; paired 8-byte loads on layer 2).
    hipStream_t s = as_stream(stream);
    float pscale = 1.0f;                // kernel P: the factor its partial sums still carry
    int grid = wgrad_grid(images);      // workgroups launched
    int wparts = grid;                  // partials they write (weights and bias alike)
    if (int u1parts = 0; layer == 1 && convu1_launch(static_cast<const unsigned char*>(src), inds, dz, part_w, part_b, images, &u1parts, s, dz_amax) == 0) {
        wparts = u1parts;               // kernel U's layer-1 variant (f16 split, one image per pass resident in LDS): one partial per workgroup, 1 / 255 left
    } else if (layer == 1) {            // kernel P: one partial per wave
        wparts = grid * 4;
        const int rc = conv1p_launch(static_cast<const unsigned char*>(src), inds, dz, part_w, part_b, (int)images, grid, s, dz_amax, &pscale);
        if (rc) return rc;
    } else if (int uparts = 0; convu_launch(static_cast<const float*>(src), dz, part_w, part_b, images, layer, &uparts, s, dz_amax, src_amax) == 0) {
        wparts = uparts;                // kernel U (f16 split, layer 3: both operands resident in LDS, convu.hip): one partial per workgroup
    } else if (int vparts = 0; convw_launch(static_cast<const float*>(src), dz, part_w, part_b, images, layer, &vparts, s, dz_amax, src_amax) != 1) {
        wparts = vparts;                // kernel V (bf16 pipe, convw.hip) took it: one partial per slab (an error surfaces in check_launch below)
    } else if (layer == 2) {            // kernel T: a workgroup walks image PAIRS
        wparts = grid = wgrad_grid((images + 1) / 2);
        auto kp = conv_wgrad_taps_kernel<GeomConv2, 2, 1, 5, true>;
        hipLaunchKernelGGL(kp, dim3(grid), dim3(256), 0, s, static_cast<const float*>(src), dz, part_w, part_b, (int)images);
    } else {
        wparts = grid = wgrad_grid((images + 1) / 2);
        auto k = conv_wgrad_taps_kernel<GeomConv3, 3, 2, 6>;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, s, static_cast<const float*>(src), dz, part_w, part_b, (int)images);
    }
    int rc = check_launch("conv_wgrad_kernel");
    if (rc) return rc;
    const int nchunks = (wparts + kRedChunk - 1) / kRedChunk;
    // both stages in one launch (the same additions in the same order) where it is ahead: layer 1's 8,192 elements (weight gradient entry point 575 -> 564 us at
    // 32,768 images, 98 -> 92 at 4,096); for the 33 - 37 k elements of layers 2 / 3 its 128-byte rows read slower than stage 1's 1-KB rows (+4 us), so they keep two launches
    if (nchunks <= 8 && total_w % 32 == 0 && Cout <= 64 && total_w <= 8192) {
        hipLaunchKernelGGL(conv_wgrad_reduce12, dim3(total_w / 32 + 1), dim3(256), 0, s, part_w, wparts, total_w, part_b, Cout, Cin, KH, KH,
                           layer == 1 ? kInv255 * pscale : 1.0f, dW, db);
        return check_launch("conv_wgrad_reduce12");
    }
    hipLaunchKernelGGL(conv_wgrad_reduce1, dim3((total_w + 255) / 256 + 1, nchunks), dim3(256), 0, s, part_w, wparts, total_w, mid,
                       part_b, wparts, Cout, mid_b);
    rc = check_launch("conv_wgrad_reduce1");
    if (rc) return rc;
    hipLaunchKernelGGL(conv_wgrad_reduce2, dim3((total_w + 255) / 256), dim3(256), 0, s, mid, nchunks, mid_b, nchunks, Cout, Cin,
                       KH, KH, layer == 1 ? kInv255 * pscale : 1.0f, dW, db);
    return check_launch("conv_wgrad_reduce2");
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv_wgrad_f32(const void* src, const int64_t* inds, const float* dz, float* dW,
                                                        float* db, int64_t images, int layer, void* workspace,
                                                        size_t workspace_bytes, void* stream) {
    return conv_wgrad_impl("mi355ppo_cnn_conv_wgrad_f32", src, inds, dz, dW, db, images, layer, workspace, workspace_bytes, nullptr, nullptr, stream);
}

// Layers 2 / 3 with kernel V on the two-term f16 split (f16split.h): src_amax / dz_amax = the operands' amax records.  Batches kernel V
// does not take (mi355ppo_cnn_conv_wgrad_kernel(images, layer) != 'V') run the f32-pipe kernel T as before; the records are then not read.
extern "C" MI355PPO_API int mi355ppo_cnn_conv_wgrad_f16x2_f32(const float* src, const float* dz, float* dW, float* db, int64_t images, int layer,
                                                              void* workspace, size_t workspace_bytes, const uint32_t* src_amax,
                                                              const uint32_t* dz_amax, void* stream) {
    const char* fn = "mi355ppo_cnn_conv_wgrad_f16x2_f32";
    MI355_REQUIRE(layer == 2 || layer == 3, MI355PPO_EINVAL, "%s: layer=%d must be 2 or 3 (layer 1: mi355ppo_cnn_conv1_wgrad_f16x2)", fn, layer);
    MI355_REQUIRE(src_amax && dz_amax && aligned(src_amax, 64) && aligned(dz_amax, 64), MI355PPO_EINVAL, "%s: amax records missing or not 64-byte aligned", fn);
    return conv_wgrad_impl(fn, src, nullptr, dz, dW, db, images, layer, workspace, workspace_bytes, src_amax, dz_amax, stream);
}

// Layer 1 (kernel P) with dz in two f16 terms under its tensor's scale: the uint8 frames are exact f16 operands, only dz needs its amax record
// (filled by mi355ppo_cnn_conv_dgrad_packed_f16x2_f32 of layer 2).  Same arguments as mi355ppo_cnn_conv_wgrad_f32 with layer = 1.
extern "C" MI355PPO_API int mi355ppo_cnn_conv1_wgrad_f16x2(const void* src_u8, const int64_t* inds, const float* dz, float* dW, float* db,
                                                           int64_t images, void* workspace, size_t workspace_bytes, const uint32_t* dz_amax,
                                                           void* stream) {
    const char* fn = "mi355ppo_cnn_conv1_wgrad_f16x2";
    MI355_REQUIRE(dz_amax && aligned(dz_amax, 64), MI355PPO_EINVAL, "%s: dz's amax record missing or not 64-byte aligned", fn);
    return conv_wgrad_impl(fn, src_u8, inds, dz, dW, db, images, 1, workspace, workspace_bytes, nullptr, dz_amax, stream);
}
