// Shared host/device helpers for libmi355ppo (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/mi355ppo.h"

#define MI355_WAVE 64
// Cache-policy bits (the `aux` operand of the buffer loads / stores: sc0 = 1, nt = 2, sc1 = 16) of the accesses that stream a tensor through a
// launch once.  Build-time knobs (tools/build_variant.py + tools/gpu/lib_ab.sh; same-box A/B at 32,768 images, profiles/r06_nt_ab.txt):
//   * the operands of the weight-gradient kernels U / H (read once, nothing written beside them): non-temporal loads, -2.5 % on the layer-1 /
//     layer-2 weight gradients, -1 % on the FC one (a read-only stream: 6.1 instead of 5.5 TB/s, profiles/r06_hbm_probe.jsonl);
//   * kernel R's source prefetch and kernel G's A loads: default policy (nt: +-0 on R, +5 % on the FC data gradient, whose A rows are re-read
//     from the L2 by the other column blocks of a supertile);
//   * the epilogue stores of kernels Q / R / G: default policy (nt: +-0).
#ifndef MI355_AUX_WGRAD_LD
#define MI355_AUX_WGRAD_LD 2
#endif
#ifndef MI355_AUX_STREAM_LD
#define MI355_AUX_STREAM_LD 0
#endif
#ifndef MI355_AUX_STREAM_ST
#define MI355_AUX_STREAM_ST 0
#endif
// Row / element math shared by the device kernels and their host-pointer twins (host_twins.hip): one definition, compiled for
// both sides, so "same math" is a property of the source and not of two restatements.
#define MI355_HD __host__ __device__ __forceinline__

namespace mi355ppo {

void set_error(const char* fmt, ...);

// kernel P (conv1p.hip): layer-1 weight gradient on the bf16 matrix pipe; one partial per wave (conv.hip reduces them)
int conv1p_launch(const unsigned char* src, const int64_t* inds, const float* dz, float* part_w, float* part_b, int images, int grid,
                  hipStream_t s, const unsigned* dz_amax, float* partial_scale);
// kernel V (convw.hip): layers 2 / 3 weight + bias gradient on the bf16 pipe; returns 1 when the batch does not qualify
bool convw_applies(int64_t images, int layer, bool f16 = false);      // (f16: the two-tile shape of the f16 split -- 227 instead of 170 slabs on layer 3)
int convw_parts(int64_t images, int layer);
int convw_launch(const float* src, const float* dz, float* part_w, float* part_b, int64_t images, int layer, int* nparts, hipStream_t s,
                 const unsigned* dz_amax = nullptr, const unsigned* src_amax = nullptr);      // amax records: the two-term f16 split (f16split.h)

// kernel Z (gemmz.hip): the K-split raw partials of the rollout-sized FC forward, (splits, M, N) f32 into `ws`
// convr.hip: kernel R, the input-resident layer-3 forward and layer-2 / layer-3 data gradients on the f16 split (bit-identical to kernel Z's)
bool convr_on(long long images, long long min_images);
int convr_fwd3(const char* fn, const float* src, unsigned src_bytes, const void* pack, const float* bias, float* dst, unsigned dst_bytes, unsigned* bits,
               long long images, const unsigned* src_amax, unsigned* dst_amax, hipStream_t st);
int convr_fwd2(const char* fn, const float* src, unsigned src_bytes, const void* pack, const float* bias, float* dst, unsigned dst_bytes, unsigned* bits,
               long long images, const unsigned* src_amax, unsigned* dst_amax, hipStream_t st);
int convr_dgrad3(const char* fn, const float* dz, unsigned dz_bytes, const void* pack, const unsigned* bits, float* dsrc, unsigned dsrc_bytes,
                 long long images, const unsigned* dz_amax, unsigned* dsrc_amax, hipStream_t st);
int convr_dgrad2(const char* fn, const float* dz, unsigned dz_bytes, const void* pack, const unsigned* bits, float* dsrc, unsigned dsrc_bytes,
                 long long images, const unsigned* dz_amax, unsigned* dsrc_amax, hipStream_t st);
// Kernels R / RB: the weights' LDS ring filled by LDS-DMA (buffer_load_dwordx4 ... lds, one slot ahead, hand-counted vmcnt waits) instead of global -> registers ->
// ds_write (two slots ahead): bit-identical, layer-3 forward / data gradient -5 %, layer-2 launches -1.5 % at 32,768 images (profiles/r06_ring_dma_ab.txt).
// -DMI355_RING_DMA=0 (tools/build_variant.py) builds the register ring for A/Bs.
#ifndef MI355_RING_DMA
#define MI355_RING_DMA 1
#endif
// convrb.hip: kernel RB, the layer-2 data gradient with a group's rows dealt to tiles by border class (three images per group, two thirds of
// kernel R's matrix instructions, bit-identical results); convr_dgrad2 hands over from MI355_RB_MIN_IMAGES images on
#ifndef MI355_RB_MIN_IMAGES
#define MI355_RB_MIN_IMAGES 3072
#endif
bool convrb_takes(long long images);
int convrb_dgrad2(const char* fn, const float* dz, unsigned dz_bytes, const void* pack, const unsigned* bits, float* dsrc, unsigned dsrc_bytes,
                  long long images, const unsigned* dz_amax, unsigned* dsrc_amax, hipStream_t st);
// convu.hip: kernel U, the layer-3 weight gradient on the f16 split with both operands of an image group resident in LDS (-> 0 launched, 1 not applicable)
int convu_max_parts();
bool convu_takes(int64_t images, int layer);       // the f16x2 weight gradient of this size and layer (1, 2, 3) runs on kernel U
int convu_launch(const float* src, const float* dz, float* part_w, float* part_b, int64_t images, int layer, int* nparts, hipStream_t s,
                 const unsigned* dz_amax, const unsigned* src_amax);
int convu1_launch(const unsigned char* frames, const int64_t* inds, const float* dz, float* part_w, float* part_b, int64_t images, int* nparts,
                  hipStream_t s, const unsigned* dz_amax);
// gemmg.hip: kernel G, the FC forward / data gradient on the f16 split with both operands through workgroup-wide LDS rings (bit-identical to
// kernel Z's; epi 0: relu(A B^T + bias), 1: (A B^T) under the ReLU mask bits; -> 0 launched, 1 not applicable, < 0 error)
bool gemmg_on(long long rows, long long min_rows);
int gemmg_launch(const char* fn, int epi, const float* A, int lda, const void* pack, const float* bias, const unsigned* bits, float* C, int M, int N,
                 int K, const unsigned* a_amax, unsigned* c_amax, hipStream_t s);
// gemmh.hip: kernel H, the FC weight gradient on the f16 split with both operands through a workgroup-wide LDS ring (gemmh_slabs() partials
// [slab][N][K] into `part`; fcw_reduce_kernel adds them)
bool gemmh_takes(int M, int N, int K, int lddz);
int gemmh_slabs();
int gemmh_launch(const float* dz, int lddz, const float* a, float* part, int M, int N, int K, const unsigned* dz_amax, const unsigned* a_amax, hipStream_t s);
int z_fc_raw_launch(const char* fn, const float* a, int lda, const void* pack, int M, int N, int K, void* ws, size_t ws_bytes, int* splits,
                    hipStream_t stream, const unsigned* a_amax = nullptr);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Term pairs of the three-term bf16 split multiplied by kernels Z (gemmz.hip), W (fcw.hip) and V (convw.hip): 9 = all 3 x 3 (every f32
// product exact), 6 = the pairs (a_i, b_j) with i + j <= 2 -- the three dropped pairs (mid x lo, lo x mid, lo x lo) are
// together below 2^-21 of the product (typically 2^-24: under the rounding of ONE f32 multiply), see DESIGN.md.  Read once.
inline int bf16_term_pairs() {
    static const int n = [] {
        const char* e = getenv("MI355PPO_BF16_PAIRS");
        return (e && e[0] == '9') ? 9 : 6;
    }();
    return n;
}

// The library's kernel switches, ONE variable per kernel family (A/B runs and tests; read at every call, never cached -- the library keeps no state):
//   unset / "1": the kernel takes the sizes it wins at (default_min);  "0": never;  "min:<n>": from n rows / images on.
inline bool kernel_switch(const char* name, long long size, long long default_min) {
    const char* e = getenv(name);
    if (!e || !e[0]) return size >= default_min;
    if (e[0] == '0' && !e[1]) return false;
    if (e[0] == 'm' && e[1] == 'i' && e[2] == 'n' && e[3] == ':') return size >= atoll(e + 4);
    return size >= default_min;
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Launch-site check.  hipGetLastError is cheap, does not synchronise, and is legal during capture.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MI355PPO_EHIP;
    }
    return MI355PPO_OK;
}

#define MI355_REQUIRE(cond, code, ...)          \
    do {                                        \
        if (!(cond)) {                          \
            ::mi355ppo::set_error(__VA_ARGS__); \
            return (code);                      \
        }                                       \
    } while (0)

// ------------------------------------------------------------------------------------ host + device
#ifdef __HIPCC__

MI355_HD uint32_t mulhi_u32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

// Philox4x32-10 (Salmon et al., SC'11).  Counter-based: the caller chooses the counter so that the
// stream does not depend on the launch geometry (nor on host vs device: the twins draw the same words).
struct Philox {
    uint32_t k0, k1;
    MI355_HD Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    MI355_HD uint4 operator()(uint64_t ctr_lo, uint64_t ctr_hi) const {
        uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = mulhi_u32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = mulhi_u32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            const uint32_t n0 = hi1 ^ c1 ^ a, n1 = lo1, n2 = hi0 ^ c3 ^ b, n3 = lo0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};

// uint32 -> uniform in (0,1): the top 23 bits k give (k + 0.5) * 2^-23.  k + 0.5 = (2k+1)/2 needs 24 significand
// bits, so every value is exact in f32 and lies in [2^-24, 1 - 2^-24]: neither 0 nor 1 can be produced (with 24 bits
// of x, (2^24-1) + 0.5 would round to 2^24 and return exactly 1).
MI355_HD float u32_to_unit_open(uint32_t x) {
    return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

// ------------------------------------------------------------------------------------ device

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, MI355_WAVE);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, MI355_WAVE);
    return v;
}

// Sum `v` over a block of NWAVES*64 threads in a fixed order; result valid in thread 0.
// `lds` must hold NWAVES doubles.  Ends with a barrier-safe state (callers may reuse `lds` after
// the next __syncthreads()).
template <int NWAVES>
__device__ __forceinline__ double block_sum(double v, double* lds) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) r += lds[w];
    }
    __syncthreads();
    return r;
}

#endif  // __HIPCC__
}  // namespace mi355ppo
