// Kernel Q -- layer-1 forward convolution of the NatureCNN on the INTEGER matrix pipe (gfx950).
//
// relu(conv1(obs[inds] / 255) + bias), cleanrl/ppo_atari_multigpu.py:136-137,154 (Conv2d(4, 32, 8, stride=4) + ReLU on
// uint8 frames), the same contract as kernel F's layer-1 forward in conv.hip -- but the source operand of this layer is
// uint8, i.e. already an exact integer, and gfx950's `v_mfma_i32_32x32x32_i8` runs at 32x the rate of the f32 MFMA
// (2x bf16), accumulates EXACTLY in int32 and takes a whole 32-byte tap row (8 taps x 4 channels) per instruction:
//
//   * weights: each f32 weight is rounded once to a 31-bit signed fixed-point number on its output channel's scale
//     (q = rint(w * 2^(30-E_n)), 2^(E_n-1) <= max_k |w[k][n]| < 2^E_n: an error of <= 2^(E_n-31), i.e. <= 2^-30 of the
//     channel's largest weight -- 1/64 of that weight's own f32 ulp and far below the rounding noise of a 256-term f32
//     sum; the channel's large weights are represented exactly) and written as four signed
//     radix-256 digits  q = d0*2^24 + d1*2^16 + d2*2^8 + d3,  d in [-128, 127];
//   * inputs: v - 128 (one `v_xor 0x80808080` per four bytes) is int8; the constant 128 * sum_k d[k][n] is added to the
//     int32 accumulator, so D_j = sum_k v[k] * d_j[k][n] comes out exact, no cancellation anywhere;
//   * y = (((D3 * 2^-8 + D2) * 2^-8 + D1) * 2^-8 + D0) * (2^(E_n-6) / 255) + bias: three f32 roundings in the Horner
//     chain, one for the scale, one for the bias -- instead of 256 roundings in an f32 fma chain.  Measured against a
//     float64 convolution the result is CLOSER than the f32-MFMA kernel's (tests/test_gpu_cnn.py).
//
// Structure: persistent 4-wave workgroups (three per CU; four from 4,096 images on: the WPS instances below), a wave owns
// 32-pixel x 32-channel tiles.  The four digit matrices sit in LDS in operand layout (32 KB per workgroup).  Holding them in
// registers was measured first (profiles/r02_*): 128 VGPRs of digits leave two waves per SIMD (3.6 TB/s).
// Round 6: the compiler's wait for the next tile's rows sat behind the tile's 17 epilogue stores (vmcnt(0) at the loop
// top: a write's round trip per tile) -- the loop is now entered with nothing pending, which leaves vmcnt(24 - r) --, and the
// epilogue lost a third of its instructions (store offsets as immediates, exact fused multiply-adds).  No barrier after the prologue.  Per tile and tap row a lane issues one 16-byte
// global load (its half of the pixel's 32-byte tap row, straight from the uint8 rollout rows through mb_inds), four LDS
// reads, four xors and four MFMAs; the eight rows of the next tile are requested as the current ones are consumed.
// 1024 matrix-pipe cycles per tile against 8192 for the f32 MFMA: the kernel is bound by HBM (28,224 B read + 51,200 B
// written per image), not by the pipe.
#include "common.h"
#include "conv1q_pack.h"
#include "f16split.h"

#pragma clang fp contract(off)

namespace mi355ppo {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4q __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void conv1q_pack_kernel(const float* __restrict__ W, unsigned char* __restrict__ pack) {
    conv1q_pack_body(W, pack);
}

// lane `l` of w := the wave-uniform value x
__device__ __forceinline__ int q_writelane(int w, unsigned x, int l) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(w) : "s"(x), "i"(l));      // (s_nop: two wait states behind the v_cmp that wrote x -- see gemmz.hip's z_writelane)
    return w;
}

// BITS: also writes (dst > 0) as one bit per element -- word p = the 32 channels of pixel p -- the ReLU mask the layer-2 data
// gradient needs (mi355ppo_cnn_conv_dgrad_packed_bits_f32 reads 1,600 bytes per image instead of the 51,200-byte activation).
// WPS: waves per SIMD the instance is compiled for = 4-wave workgroups per CU.  3 (168 VGPRs) at rollout sizes; 4 (128 VGPRs: 122 used, the epilogue's
// per-channel constants read from LDS at the top of every epilogue instead of living in six registers across the matrix phase) from 4,096 images on:
// -2.6 % at 32,768 images, +6 % at 1,024, where a fourth workgroup per CU is a fourth 32-KB digit copy for a quarter fewer tiles (profiles/r06_nt_ab.txt).
template <int NW, bool BITS, int WPS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void conv1q_fwd_kernel(const unsigned char* __restrict__ src, const int64_t* __restrict__ inds,
                                                             const unsigned char* __restrict__ pack, const float* __restrict__ bias,
                                                             float* __restrict__ dst, unsigned* __restrict__ bits, unsigned P, int ntiles, unsigned dst_bytes,
                                                             unsigned* __restrict__ amax) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform: the tile index lives in an SGPR)
    const int li = lane & 31, lh = lane >> 5;
    // digit matrices, in operand layout, -> LDS [row][digit][lane]
    __shared__ i32x4 Bl[kQRows][kQDigits][64];
    {
        const i32x4* __restrict__ p4 = reinterpret_cast<const i32x4*>(pack);
        for (int e = tid; e < kQRows * kQDigits * 64; e += 64 * NW) (&Bl[0][0][0])[e] = p4[e];
    }
    __shared__ int Cl[kQDigits + 2][32];           // digit offsets, scale, bias per channel
    if (tid < 32) {
#pragma unroll
        for (int d = 0; d < kQDigits; ++d) Cl[d][tid] = reinterpret_cast<const int*>(pack + kQAccOff)[d * 32 + tid];
        Cl[kQDigits][tid] = reinterpret_cast<const int*>(pack + kQScaleOff)[tid];
        Cl[kQDigits + 1][tid] = __float_as_int(bias[tid]);
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc_dst = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, kQRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsrc_bits = __builtin_amdgcn_make_buffer_rsrc(bits, 0, BITS ? (int)(dst_bytes >> 5) : 0, kQRsrcWord3);
    const int nwv = gridDim.x * NW;

    // pointer to byte 16*lh of tap row 0 of the lane's pixel of `tile` (pixels past P read pixel 0: results dropped).
    // A 32-pixel tile touches at most two images: their rollout rows are looked up with SCALAR loads (lgkmcnt), so the
    // address of the next tile never waits behind this tile's vector stores (vmcnt).
    auto setup = [&](int tile) -> const unsigned char* {
        const int tu = __builtin_amdgcn_readfirstlane(tile);
        const bool live = tu < ntiles;
        const unsigned p0 = live ? (unsigned)tu * 32u : 0u;
        const unsigned img0 = p0 / (unsigned)kQPerImg;
        const unsigned last = (P - 1u) / (unsigned)kQPerImg;
        const unsigned img1 = img0 < last ? img0 + 1u : img0;
        const long long s0 = inds ? inds[img0] : (long long)img0;
        const long long s1 = inds ? inds[img1] : (long long)img1;
        const unsigned p = p0 + (unsigned)li;
        const unsigned pp = (live && p < P) ? p : p0;
        const unsigned img = pp / (unsigned)kQPerImg, rem = pp - img * (unsigned)kQPerImg;
        const unsigned gy = rem / (unsigned)kQG, gx = rem - gy * (unsigned)kQG;
        const long long simg = img == img0 ? s0 : s1;
        return src + ((simg * kQH + gy * 4) * kQW + gx * 4) * (long long)kQC + 16 * lh;
    };
    const unsigned lanebase = (unsigned)(512 * lh + 4 * li);
    unsigned vmax = 0u;                    // bits of the largest value this lane stored (>= 0 after the ReLU): dst's amax record (f16split.h)
    u32x4q ring[kQRows];
    // workgroups are dealt to the 8 XCDs round robin: give each XCD a contiguous range of tile groups, so that the waves
    // that share source rows (vertical window overlap, neighbouring tiles of one image) also share an L2
    int wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) wg = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    int tile = wg * NW + wave;
    const unsigned char* cur = setup(tile);
    const unsigned char* nxt = setup(tile + nwv);
#pragma unroll
    for (int r = 0; r < kQRows; ++r) ring[r] = *reinterpret_cast<const u32x4q*>(cur + r * kQPitch);
    // vmcnt counts loads and stores in ONE in-order queue.  A tile's ring loads are issued in FRONT of the previous tile's 17 epilogue stores, so the
    // wait for ring[r] at the top of a tile is vmcnt(24 - r): the loads, not the stores behind them.  The compiler merges the loop header's state
    // with the loop entry's, though (the first tile's loads with nothing behind them: vmcnt(7 - r)), and on the back edge that count means "and the
    // stores acknowledged": it emitted vmcnt(0) there, and every wave sat out a write's round trip per tile (round 2: ~2,000 cycles per 1,800-cycle
    // tile).  A use of the first tile's rows in front of the loop -- the compiler waits for them HERE -- leaves the back edge's counts at its top.
#pragma unroll
    for (int r = 0; r < kQRows; ++r) asm volatile("" : "+v"(ring[r]));

    for (; tile < ntiles; tile += nwv) {
        i32x16 acc[kQDigits];
#pragma unroll
        for (int d = 0; d < kQDigits; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[d][e] = 0;                  // (the 128 * sum_k digit offsets are added in the epilogue:
                                                                          //  a zero start is an inline operand, a splat costs 16 VGPRs)
#pragma unroll
        for (int r = 0; r < kQRows; ++r) {
            const u32x4q t = ring[r] ^ 0x80808080u;                       // v - 128 as int8
            const i32x4 a = {(int)t.x, (int)t.y, (int)t.z, (int)t.w};
#pragma unroll
            for (int d = 0; d < kQDigits; ++d) acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Bl[r][d][lane], acc[d], 0, 0, 0);
            ring[r] = *reinterpret_cast<const u32x4q*>(nxt + r * kQPitch);   // the same row of the wave's next tile
            __builtin_amdgcn_sched_barrier(0);                            // keep the row's four LDS reads in the row
        }
        // ---- epilogue: Horner over the digits, scale, bias, ReLU; accumulator row e of the lane is pixel
        // (e & 3) + 8 * (e >> 2) + 4 * lh of the tile, column = channel li
        unsigned myoff = kQOob;
        {
            const unsigned p = (unsigned)tile * 32u + (unsigned)li;
            if (p < P) myoff = p * 128u;                                  // pixel-major (N,20,20,32) f32: 128 bytes per pixel
        }
        int wv = 0;                                                       // BITS: lane L (< 32) collects the mask word of pixel L of the tile
        int acc0[kQDigits];
#pragma unroll
        for (int d = 0; d < kQDigits; ++d) acc0[d] = Cl[d][li];
        const float scale = __int_as_float(Cl[kQDigits][li]), bias_r = __int_as_float(Cl[kQDigits + 1][li]);
        // accumulator row e -> pixel 32 tile + (e & 3) + 8 (e >> 2) + 4 lh: byte offset = (4096 tile + 512 lh + 4 li: one add per tile) + 128 ((e & 3) +
        // 8 (e >> 2)) (the instruction's immediate).  Pixels past P lie past dst_bytes = 128 P: the buffer drops their stores -- which is why the tile's
        // base travels in the VECTOR offset: a buffer instruction's scalar offset is not part of its range check.
        const unsigned tl = (unsigned)__builtin_amdgcn_readfirstlane(tile) * 4096u + lanebase;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned off = tl + (unsigned)(128 * ((e & 3) + 8 * (e >> 2)));
            float t = (float)(acc[3][e] + acc0[3]);                       // exact integers D_j = sum_k v[k] * d_j[k][n]
            // t * 2^-8 is exact (a power of two, no underflow: |t| is 0 or >= 2^-24), so the fused multiply-add rounds exactly as the multiply followed by
            // the add did -- one rounding per step, the same bits, one instruction instead of two (the file compiles with contraction off: explicit)
            t = __builtin_fmaf(t, 0.00390625f, (float)(acc[2][e] + acc0[2]));
            t = __builtin_fmaf(t, 0.00390625f, (float)(acc[1][e] + acc0[1]));
            t = __builtin_fmaf(t, 0.00390625f, (float)(acc[0][e] + acc0[0]));
            float v = t * scale;
            v = v + bias_r;
            v = v > 0.0f ? v : 0.0f;
            vmax = __float_as_uint(v) > vmax ? __float_as_uint(v) : vmax;     // (pixels past P re-read pixel 0: real values)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_dst, off, 0, MI355_AUX_STREAM_ST);   // dropped when out of range
            if constexpr (BITS) {      // ballot lanes 0..31: the 32 channels of pixel (e & 3) + 8 (e >> 2); lanes 32..63: of that pixel + 4
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(v > 0.0f);
                wv = q_writelane(wv, (unsigned)bal, (e & 3) + 8 * (e >> 2));
                wv = q_writelane(wv, (unsigned)(bal >> 32), (e & 3) + 8 * (e >> 2) + 4);
            }
        }
        if constexpr (BITS)            // myoff = 128 p (or out of range): word p sits at byte 4 p
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)wv, rsrc_bits, (lh == 0 && myoff != kQOob) ? myoff >> 5 : kQOob, 0, 0);
        cur = nxt;
        nxt = setup(tile + 2 * nwv);
    }
    if (amax) amax_commit(amax, vmax, blockIdx.x * NW + wave, lane);      // (uniform; one atomic per wave of the persistent grid)
}

static int g_q_cus = 0;
#ifndef MI355_Q_FOUR_WAVES_FROM
#define MI355_Q_FOUR_WAVES_FROM 4096
#endif
constexpr long long kQFourWavesFrom = MI355_Q_FOUR_WAVES_FROM;

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API size_t mi355ppo_cnn_conv1q_pack_bytes(void) { return (size_t)kQPackBytes; }

extern "C" MI355PPO_API int mi355ppo_cnn_conv1q_pack(const float* W, void* pack, void* stream) {
    const char* fn = "mi355ppo_cnn_conv1q_pack";
    MI355_REQUIRE(W && pack, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(aligned(W, 4) && aligned(pack, 16), MI355PPO_EALIGN, "%s: pack must be 16-byte aligned", fn);
    hipLaunchKernelGGL(conv1q_pack_kernel, dim3(1), dim3(256), 0, as_stream(stream), W, static_cast<unsigned char*>(pack));
    return check_launch(fn);
}

static int conv1q_fwd_impl(const char* fn, const void* src_u8, const int64_t* inds, const void* pack, const float* bias, float* dst,
                           unsigned* bits, int64_t images, void* stream, unsigned* amax = nullptr) {
    MI355_REQUIRE(src_u8 && pack && bias && dst, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(images > 0 && images <= (1 << 22), MI355PPO_EINVAL, "%s: images=%lld out of range (1..4194304)", fn,
                  (long long)images);
    MI355_REQUIRE(aligned(src_u8, 16) && aligned(pack, 16) && aligned(dst, 128) && aligned(inds, 8) && aligned(bias, 4) && aligned(bits, 4),
                  MI355PPO_EALIGN, "%s: src/pack must be 16-byte aligned, dst 128-byte aligned", fn);
    const long long P = (long long)images * kQPerImg, dstb = P * 128;
    MI355_REQUIRE(dstb <= (1LL << 32) - 8192, MI355PPO_EINVAL, "%s: destination of %lld bytes exceeds the 32-bit buffer range", fn, dstb);
    if (g_q_cus == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        g_q_cus = cus;
    }
    const int ntiles = (int)((P + 31) / 32);
    constexpr int NW = 4;
    const int wps = images >= kQFourWavesFrom ? 4 : 3;        // 4-wave workgroups per CU = waves per SIMD
    long long wgs = (long long)g_q_cus * wps;
    if (wgs * NW > ntiles) wgs = (ntiles + NW - 1) / NW;
#define MI355_Q_LAUNCH(BITS_, WPS_)                                                                                                            \
    hipLaunchKernelGGL((conv1q_fwd_kernel<NW, BITS_, WPS_>), dim3((unsigned)wgs), dim3(64 * NW), 0, as_stream(stream),                         \
                       static_cast<const unsigned char*>(src_u8), inds, static_cast<const unsigned char*>(pack), bias, dst, bits, (unsigned)P, \
                       ntiles, (unsigned)dstb, amax)
    if (bits) { if (wps == 4) MI355_Q_LAUNCH(true, 4); else MI355_Q_LAUNCH(true, 3); }
    else { if (wps == 4) MI355_Q_LAUNCH(false, 4); else MI355_Q_LAUNCH(false, 3); }
#undef MI355_Q_LAUNCH
    return check_launch(fn);
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv1q_fwd(const void* src_u8, const int64_t* inds, const void* pack, const float* bias,
                                                    float* dst, int64_t images, void* stream) {
    return conv1q_fwd_impl("mi355ppo_cnn_conv1q_fwd", src_u8, inds, pack, bias, dst, nullptr, images, stream);
}

extern "C" MI355PPO_API int mi355ppo_cnn_conv1q_fwd_bits(const void* src_u8, const int64_t* inds, const void* pack, const float* bias,
                                                         float* dst, uint32_t* mask_bits, int64_t images, void* stream) {
    const char* fn = "mi355ppo_cnn_conv1q_fwd_bits";
    MI355_REQUIRE(mask_bits, MI355PPO_EINVAL, "%s: null pointer", fn);
    return conv1q_fwd_impl(fn, src_u8, inds, pack, bias, dst, mask_bits, images, stream);
}

// The same (mask_bits may be null) that also folds the stored activations into `dst_amax`, dst's amax record (MI355PPO_AMAX_WORDS uint32,
// zeroed by the caller): what the f16x2 forward of layer 2 scales its A operand by.
extern "C" MI355PPO_API int mi355ppo_cnn_conv1q_fwd_amax(const void* src_u8, const int64_t* inds, const void* pack, const float* bias,
                                                         float* dst, uint32_t* mask_bits, int64_t images, uint32_t* dst_amax, void* stream) {
    const char* fn = "mi355ppo_cnn_conv1q_fwd_amax";
    MI355_REQUIRE(dst_amax && aligned(dst_amax, 64), MI355PPO_EINVAL, "%s: amax record missing or not 64-byte aligned", fn);
    return conv1q_fwd_impl(fn, src_u8, inds, pack, bias, dst, mask_bits, images, stream, dst_amax);
}
