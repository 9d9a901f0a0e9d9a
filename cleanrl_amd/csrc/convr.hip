// Kernel R -- the INPUT-RESIDENT convolution of round 5 (two-term f16 split only): the layer-2 / layer-3 forwards and data gradients of the
// NatureCNN (cleanrl/ppo_atari_multigpu.py:141-142 and their backward, :358) with the source tensor of a GROUP of images
// held in LDS, already split into f16 hi / lo planes.
//
// What bounded kernel Z on these launches (profiles/r05_pmc_*.csv: matrix pipe 0.24 busy, ten VALU instructions per MFMA, TA 0.7 busy,
// waves waiting on memory 0.36): its rows are im2col rows, so every source element is fetched from the L1 / L2, written to LDS, read back
// and SPLIT once per tap that touches it -- 9 times in the 3 x 3 layers, 4 times in the layer-2 data gradient -- inside the k-loop, on the
// matrix instructions' critical path.  Here a workgroup
//   * loads the f32 source of G consecutive images ONCE (contiguous, 16 bytes per lane), splits every element ONCE (f16split.h, the
//     tensor's scale from its amax record) and stores the hi / lo halves into a padded pixel grid in LDS: record of a pixel = its C hi
//     halves (2 C bytes) | its C lo halves | 16 B pad.  The pad makes the record pitch an odd number of sixteen-byte slots (17 / 9): "lane = pixel" fragment
//     reads of 16 lanes with distinct pixel numbers mod 16 hit 16 distinct slots (ds_read_b128: conflict-free).  Zero padding of the data
//     gradients = a border of zero records written once per launch;
//   * walks the k-steps with NO address arithmetic and NO VALU in the loop: every fragment is one ds_read_b128 at (lane's window origin +
//     compile-time offset of the tap / channel chunk), the weights come from the f16x2 pack through the workgroup's two-buffer LDS ring
//     (kernel Z's BLDS with slots of 4 - 6 k-steps: one barrier per slot), the operands of step v + 1 are requested before the matrix
//     instructions of step v, and the loop is 2 MT + 2 NT reads per 3 MT NT matrix instructions;
//   * requests the NEXT group's source two 16-byte loads per k-step into registers (one workgroup per CU holds the LDS), splits and stores
//     it at the top of the next group.
// Eight or twelve waves per CU (the instances below RGeom); which row sits in which lane is a compile-time table that keeps the 16-lane groups of a
// fragment read on 16 different bank slots (RRowTable).  What was measured on the way (each step bit-identical): profiles/
// r05_tile_shape_experiments.txt (the s_memtime stamps of round 5's MI355PPO_R_TRACE builds: profiles/r05_kernel_r_trace.txt; the stamp code is gone since round 6).
// Accumulator layout and epilogue are kernel Z's (lane = channel, accumulator e = row (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the tile):
// a 4-byte store instruction writes two whole 128-byte lines.  (The transposed layout -- lane = pixel, sixteen channels per lane in four
// 16-byte runs -- needs a quarter of the store instructions, but each touches 32 lines a quarter at a time: measured 300 - 480 cycles per
// store instruction, the epilogue a third of the layer-2 data gradient's group time.)
// Arithmetic: the same products in the same order as kernel Z's SPLIT = 1 instances (k-steps in the order of z_kstep / the border classes'
// ascending taps; per step hi hi, hi lo, lo hi; padded taps add exact zeros): the results are bit-identical to kernel Z's
// (tools/conv_traffic hashes, tests/test_gpu_f16x2.py).
#include "common.h"
#include "f16split.h"
#include "convr_geom.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float r_f32x16 __attribute__((ext_vector_type(16)));
typedef float r_f32x4 __attribute__((ext_vector_type(4)));

enum { R_BIAS_RELU = 0, R_BIAS_RELU_BITS = 1, R_MASKB = 2, R_MASKB_CLS4 = 3 };

constexpr unsigned kROob = 0xFFFFF000u;               // buffer offset out of range for every tensor < 4 GiB - 4 KiB
constexpr int kRRsrcWord3 = 0x00020000;               // raw buffer, 32-bit elements
constexpr int kRWaitVm0 = 0x0F70;                     // s_waitcnt vmcnt(0) (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)

// lane `l` of w := the wave-uniform value x; x where bit (lane) of {hi, lo} is set, else 0 (gemmz.hip's z_writelane / z_keep_where: the
// s_nop covers the two wait states a VALU read of an SGPR needs behind the VALU write the compiler cannot see inside the asm)
__device__ __forceinline__ int r_writelane(int w, unsigned x, int l) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(w) : "s"(x), "i"(l));
    return w;
}
__device__ __forceinline__ float r_keep_where(float x, unsigned lo, unsigned hi) {
    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
    float r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
    return r;
}

// one LDS-DMA piece: 16 bytes per lane from the buffer (per-lane offset `voff`, wave-uniform `soff`) to LDS at `dst` + 16 lane (buffer_load_dwordx4 ... lds).
// (A plain function on purpose: with the builtin inside the kernel TEMPLATE the host pass drops the kernels' launch stubs without a diagnostic.)
__device__ __forceinline__ void r_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

struct RArgs {
    const float* A;             // source tensor (images, IH, IW, 64)
    unsigned a_bytes;
    const unsigned char* pack;  // f16x2 pack of the weights (header + [k-step][tile][hi, lo][lane][8 f16])
    const float* bias;          // R_BIAS_RELU*
    const unsigned* bits_in;    // R_MASKB*: the ReLU mask of the destination's activation, one bit per element
    unsigned* bits_out;         // R_BIAS_RELU_BITS
    float* C;
    unsigned c_bytes;
    long long images;
    int groups;
    const unsigned* a_amax;     // amax record of A
    unsigned* c_amax;           // amax record of C to fold into, or null
};

#if MI355_RING_DMA
// vector loads a wave issues behind the LDS-DMA pieces of ring slot `slot` + 1 (requested at the slot's first step, in front of that step's prefetch round) and in
// front of its wait (top of the slot's last step): the prefetch rounds (r_kernel's pre_lo) and the mask words of the steps in between
template <class RG, int EPI, bool SPREAD>
constexpr int r_dma_younger(int slot) {
    constexpr int NI = RG::NI, kPreSpan = RG::KSTEPS - 3;
    int n = 0;
    for (int v = RG::SS * slot; v < RG::SS * slot + RG::SS - 1; ++v) {
        int lo[2] = {0, 0};
        for (int t = 0; t < 2; ++t) {
            const int w = v + t, m = SPREAD ? ((w - 1) * NI + kPreSpan - 1) / kPreSpan : (w - 1) * 2;
            lo[t] = w < 1 ? 0 : (m > NI ? NI : m);
        }
        n += (lo[1] - lo[0]) + (((EPI == 2 || EPI == 3) && v == RG::KSTEPS - 4) ? RG::NTW : 0);      // (R_MASKB, R_MASKB_CLS4)
    }
    return n;
}
#endif

template <class RG, int EPI, bool SPREAD>
__global__ __launch_bounds__(64 * RG::NW) __attribute__((amdgpu_waves_per_eu((RG::NW + 3) / 4 * RG::WGS, (RG::NW + 3) / 4 * RG::WGS))) void r_kernel(RArgs a) {
    constexpr int NT = RG::NT, NTW = RG::NTW, MT = RG::MT, NW = RG::NW, NI = RG::NI, THREADS = RG::THREADS;
    constexpr int SS = RG::SS, NSLOT = RG::KSTEPS / SS;
    constexpr int kPieces = SS * NT * 2;                  // KiB pieces of a ring slot: (step h of the slot, tile j, term t), x = (h NT + j) 2 + t
    constexpr int kFetchers = kPieces % NW == 0 ? NW : (kPieces % (NW / 2) == 0 ? NW / 2 : (kPieces % 8 == 0 && NW >= 8 ? 8 : 4));      // waves that move ring pieces
    constexpr int kShare = kPieces / kFetchers;
    static_assert(kPieces % kFetchers == 0 && kFetchers <= NW, "whole pieces per fetching wave");
    __shared__ __attribute__((aligned(16))) unsigned char lds[RG::ABYTES + 2 * RG::SLOTB];
    unsigned char* const ring = lds + RG::ABYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int rw = RG::NS == 1 ? wave : wave % RG::RW, jg = RG::NS == 1 ? 0 : wave / RG::RW;      // row-wave; first column tile / NTW (wave-uniform)

    const int ea = f16_scale_exp(amax_load(a.a_amax, lane));
    const int eb = f16_scale_exp(*reinterpret_cast<const unsigned*>(a.pack));
    const float sa = f16_pow2(ea), un = f16_unscale(ea, eb);

    // ---- rows of a group.  Slot (wave MT + i) 32 + x holds row table.row[...] of the group (empty slots repeat a row of their sixteen-lane set: the same values
    // stored twice).  A-fragment reads: lane -> slot x = lane % 32 of tile i.  Epilogue: accumulator e of tile i -> slot x = (e & 3) +
    // 8 (e >> 2) + 4 lh; `roff[i][e]` = byte offset of that row's first channel in C for image 0 of the group (+ the lane's channel).
    constexpr RRowTable<RG> table{};
    auto row_pixel = [&](int slot, int& gi, int& pp, int& gy, int& gx) __attribute__((always_inline)) {
        const int rc = table.src[slot];
        gi = rc / RG::OP; pp = rc - gi * RG::OP; gy = pp / RG::OW; gx = pp - gy * RG::OW;
    };
    auto row_coff = [&](int gi, int pp, int gy, int gx) __attribute__((always_inline)) -> unsigned {      // C offset of the row (tile j = 0) within the group
        if constexpr (EPI == R_MASKB_CLS4) return (unsigned)(((gi * (2 * RG::OH) + 2 * gy) * (2 * RG::OW) + 2 * gx) * 32) * 4u;
        else return (unsigned)((gi * RG::OP + pp) * (32 * NT)) * 4u;
    };
    const unsigned char* win[MT];                         // window origin of the lane's fragment row (+ the lane half's 8 channels)
    // (two 16-bit offsets per register where a group's C stays below 64 KiB -- the single-class layers: 16 registers instead of 32)
    constexpr bool kPackRoff = (EPI == R_MASKB_CLS4 ? RG::G * 4 * RG::OP * 32 : RG::G * RG::OP * 32 * NT) * 4 < 65536;
    unsigned roff[MT][kPackRoff ? 8 : 16], rlane = 0u, rword[MT];         // rlane: C offset of slot row `lane` of the wave's 32 MT rows (mask words in); rword[i]: of slot row li of tile i (mask words out)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int gi, pp, gy, gx;
        row_pixel((rw * MT + i) * 32 + li, gi, pp, gy, gx);
        win[i] = lds + gi * RG::IMGB + RG::pidx(RG::S * gy, RG::S * gx) * RG::PIX + 16 * lh;
        rword[i] = row_coff(gi, pp, gy, gx);
        if (MT == 1 || lh == i) rlane = rword[i];                         // (MT = 2: lanes 0..31 tile 0, lanes 32..63 tile 1; MT = 1: both halves tile 0)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            row_pixel((rw * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, gi, pp, gy, gx);
            const unsigned o = row_coff(gi, pp, gy, gx) + 4u * (unsigned)li;
            if constexpr (kPackRoff) roff[i][e >> 1] = (e & 1) ? (roff[i][e >> 1] | (o << 16)) : o;
            else roff[i][e] = o;
        }
    }
    static_assert(MT <= 2, "mask words: one lane per row of the wave");
    unsigned char* const ring_l = ring + 16 * lane;
    const unsigned char* const ring_r = ring_l + jg * (NTW * 2048);        // this wave's column tiles of a k-step

    // ---- the group's source: unit u = 16 bytes = 4 channels of a pixel; thread tid takes units it * THREADS + tid
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)a.a_bytes, kRRsrcWord3);
    unsigned udst[NI];                                    // LDS byte address of the unit's hi half (lo: + RG::LO); ~0u: no such unit
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int u = it * THREADS + tid;
        const int pix = u / RG::UPP, c4 = u - pix * RG::UPP, img = pix / (RG::IH * RG::IW), q = pix - img * (RG::IH * RG::IW), qy = q / RG::IW, qx = q - qy * RG::IW;
        udst[it] = u < RG::UNITS ? (unsigned)(img * RG::IMGB + RG::pidx(qy + RG::HL, qx + RG::HL) * RG::PIX + c4 * 8) : ~0u;
    }
    // Loads of the next group's source issued before k-step v: rounds [pre_lo(v), pre_lo(v + 1)) go out in step v.  Two per step from the second
    // step on (the small sources of the data gradients: 38 - 42 KB are in flight at once without holding anybody), or -- SPREAD, the forwards'
    // 104 KB -- evenly over the k-loop: a CU takes in ~11 bytes per cycle from HBM when every CU asks, so 24 KB per step (12 waves x 2 loads) is
    // five times what arrives, the queue fills within two steps and every wave then waits AT ITS NEXT LOAD until the 104 KB have drained --
    // s_memtime: k-steps 2 .. 7 of the layer-2 forward took 11,200 ticks instead of 2,000 (profiles/r05_kernel_r_trace.txt).  One round
    // every ~3 steps asks for what arrives.
    constexpr int kPreSpan = RG::KSTEPS - 3;              // SPREAD: steps 1 .. kPreSpan carry the NI rounds
    auto pre_lo = [](int v) constexpr -> int {
        const int n = SPREAD ? ((v - 1) * NI + kPreSpan - 1) / kPreSpan : (v - 1) * 2;
        return v < 1 ? 0 : (n > NI ? NI : n);
    };
    static_assert(pre_lo(RG::KSTEPS) == NI, "the next group's source is requested inside one k-loop");
    s_u32x4 pre[NI];
    // (issued unconditionally -- past the last group with out-of-range offsets, which load zeros without touching memory: loads behind a
    //  branch leave the compiler without a count of the outstanding ones, and every later wait for a weight piece became "all of them")
    auto prefetch = [&](int grp, int it0, int n) __attribute__((always_inline)) {        // units past the tensor (last group) load zeros: their rows are never stored
        const bool any = grp < a.groups;
        const unsigned base = (unsigned)grp * (unsigned)(RG::UNITS * 16);
#pragma unroll
        for (int it = it0; it < it0 + n && it < NI; ++it) {
            const int u = it * THREADS + tid;
            pre[it] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (any && u < RG::UNITS) ? base + (unsigned)u * 16u : kROob, 0, MI355_AUX_STREAM_LD));
        }
    };
    auto fill = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            unsigned hi[2], lo[2];
            f16_split4(pre[it], sa, hi, lo);
            if (udst[it] != ~0u) {
                *reinterpret_cast<uint2*>(lds + udst[it]) = make_uint2(hi[0], hi[1]);
                *reinterpret_cast<uint2*>(lds + udst[it] + RG::LO) = make_uint2(lo[0], lo[1]);
            }
        }
    };
    if constexpr (RG::HL > 0) {                           // the zero border (and everything else, once)
        for (int o = tid * 16; o < RG::ABYTES; o += THREADS * 16) *reinterpret_cast<s_u32x4*>(lds + o) = (s_u32x4){0u, 0u, 0u, 0u};
    }

    // ---- epilogue constants
    const void* const bits_base = EPI == R_BIAS_RELU_BITS ? (const void*)a.bits_out : (const void*)a.bits_in;
    // C's descriptor of a group starts AT the group (+ the wave's column tiles): the per-value offsets are then lane constants + an immediate, no address
    // arithmetic per store; the range shrinks with the base, so rows of images past the batch still fall out of it (a scalar offset would not be checked)
    auto rsrc_c_of = [&](int g, unsigned gbase) __attribute__((always_inline)) {
        const bool ok = g >= 0 && g < a.groups;
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.C) + (ok ? gbase : 0u), 0, ok ? (int)(a.c_bytes - gbase) : 0, kRRsrcWord3);
    };
    auto rsrc_b_of = [&](int g) __attribute__((always_inline)) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(bits_base), 0, (EPI != R_BIAS_RELU && g >= 0 && g < a.groups) ? (int)(a.c_bytes >> 5) : 0, kRRsrcWord3); };
    float bj[NTW];                                        // R_BIAS_RELU*: the lane's bias element per column tile
#pragma unroll
    for (int j = 0; j < NTW; ++j) bj[j] = (EPI == R_BIAS_RELU || EPI == R_BIAS_RELU_BITS) ? a.bias[32 * (jg * NTW + j) + li] : 0.0f;
    constexpr unsigned kGroupC = (EPI == R_MASKB_CLS4 ? RG::G * 4 * RG::OP * 32 : RG::G * RG::OP * 32 * NT) * 4;      // bytes of C per group
    auto joff = [](int j) __attribute__((always_inline)) -> int {                        // byte offset of column tile j from the row's first channel
        return EPI == R_MASKB_CLS4 ? ((j >> 1) * (2 * RG::OW) + (j & 1)) * 32 * 4 : 32 * j * 4;
    };
    float cmax = 0.0f;

    // the pack through a buffer resource: per-lane offset 16 lane in ONE register, the piece's offset in an SGPR (64-bit per-piece addresses
    // of the unrolled loop cost 70 registers)
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.pack), 0, kF16PackHeader + RG::KSTEPS * RG::STEPB, kRRsrcWord3);
    const unsigned lane16f = (kFetchers == NW || wave < kFetchers) ? 16u * (unsigned)lane : kROob;      // (waves that move no ring pieces load zeros from nowhere: no branch around a load)
    // The weights' ring: slot s = the visited k-steps SS s .. SS s + SS - 1, 2 buffers.  Wave w moves pieces w kShare .. + kShare - 1 of a slot: global ->
    // registers TWO slots ahead of the write (vector loads return in order and the next group's source -- a trip to HBM -- travels in the
    // same queue: one slot of distance left the ring waiting), registers -> buffer (s + 1) & 1 at the first step of slot s -- every wave
    // read that buffer (slot s - 1) for the last time before the barrier of slot s - 1 --, and the barrier of slot s (at its LAST step)
    // stands between these writes and the first reads of slot s + 1 (issued during that last step, for the step after it).  The weights
    // are the same for every group: with an even number of slots (the sets' parity carries over) the first two slots of the NEXT group are fetched during the last two
    // slots of this one (the sets are free by then), so a group starts with its weights in registers.
    s_u32x4 bst[2][kShare];                               // slot s travels in set s & 1
    auto load_slot = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kShare; ++u) {
            const int x = wave * kShare + u, h = x / (NT * 2), jt = x - h * (NT * 2);       // (wave-uniform: scalar arithmetic)
            bst[slot & 1][u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_p, lane16f, kF16PackHeader + r_kstep<RG>(SS * slot + h) * RG::STEPB + jt * 1024, 0));
        }
    };
    auto write_slot = [&](int slot) __attribute__((always_inline)) {                     // -> buffer slot & 1
        if (kFetchers == NW || wave < kFetchers)
#pragma unroll
            for (int u = 0; u < kShare; ++u) *reinterpret_cast<s_u32x4*>(ring_l + (slot & 1) * RG::SLOTB + (wave * kShare + u) * 1024) = bst[slot & 1][u];
    };
    auto ring_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
#if MI355_RING_DMA
    // The ring by LDS-DMA (buffer_load_dwordx4 ... lds): a piece = one wave instruction, global -> LDS without registers and without the ds_write pass.  Slot s + 1
    // is requested at the first step of slot s into the buffer every wave read for the last time before the barrier of slot s - 1 (one slot of distance; slot NSLOT =
    // the next group's slot 0, waited for in front of the epilogue); the issuing wave waits for ITS pieces in front of the barrier at the last step of slot s: vmcnt
    // counts in order, so "all but the r_dma_younger(s) vector loads issued behind the pieces" -- the prefetch rounds and mask words of the slot's steps but the last.
    static_assert(NSLOT % 2 == 0, "the next group's slot 0 goes to buffer 0 while the last slot is read from buffer 1");
    auto dma_slot = [&](int slot) __attribute__((always_inline)) {                       // -> buffer slot & 1
        if (kFetchers == NW || wave < kFetchers) {
#pragma unroll
            for (int u = 0; u < kShare; ++u) {
                const int x = wave * kShare + u, h = x / (NT * 2), jt = x - h * (NT * 2);
                r_dma16(rsrc_p, ring + (slot & 1) * RG::SLOTB + x * 1024, 16u * (unsigned)lane, kF16PackHeader + r_kstep<RG>(SS * (slot % NSLOT) + h) * RG::STEPB + jt * 1024);
            }
        }
    };
    auto dma_wait = [&](int slot) __attribute__((always_inline)) {       // s_waitcnt vmcnt(dma_younger(slot)): an immediate -- the chain folds once the k-loop is unrolled
        const int n = r_dma_younger<RG, EPI, SPREAD>(slot);
#define R_VMCNT_CASE(k) if (n == k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory");
        R_VMCNT_CASE(0) R_VMCNT_CASE(1) R_VMCNT_CASE(2) R_VMCNT_CASE(3) R_VMCNT_CASE(4) R_VMCNT_CASE(5) R_VMCNT_CASE(6) R_VMCNT_CASE(7)
        R_VMCNT_CASE(8) R_VMCNT_CASE(9) R_VMCNT_CASE(10) R_VMCNT_CASE(11) R_VMCNT_CASE(12)
#undef R_VMCNT_CASE
        if (n > 12) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (never more than a few rounds per slot: a full wait is always correct)
    };
#endif

    constexpr bool kCarry = NSLOT % 2 == 0 && NSLOT >= 4;  // slots 0 and 1 of the next group travel in the sets across the group boundary
    r_f32x16 acc[MT][NTW];
    unsigned wm[NTW];                                  // R_MASKB*: lane L holds the mask word of slot row L of the wave's rows, per column tile
    int wv = 0;                                           // R_BIAS_RELU_BITS: lane L (< 32) collects the mask word of slot row L of the tile in flight
    s_u32x4 pa[2][MT][2], wb[2][NTW][2];                   // [k-step parity]: pixel fragments [tile][hi, lo]; weight fragments [tile][hi, lo]
    auto read_a = [&](int par, int v) __attribute__((always_inline)) {
        const int off = r_tapoff<RG>(r_kstep<RG>(v));
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            pa[par][i][0] = *reinterpret_cast<const s_u32x4*>(win[i] + off);
            pa[par][i][1] = *reinterpret_cast<const s_u32x4*>(win[i] + off + RG::LO);
        }
    };
    auto read_b = [&](int par, int v) __attribute__((always_inline)) {                   // step v = step v % SS of slot v / SS
        const int buf = (v / SS) & 1, h = v % SS;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) wb[par][j][t] = *reinterpret_cast<const s_u32x4*>(ring_r + buf * RG::SLOTB + ((h * NT + j) * 2 + t) * 1024);
    };
    // ---- epilogue (kernel Z's), one value: accumulator e of tile (i, j) = row slot (e & 3) + 8 (e >> 2) + 4 lh of tile i, channel 32 j + li.
    // Every offset carries the group's base, so rows of images past the batch fall out of the buffer's range: stores dropped, mask words
    // read as zero.  Values of one (i, j) come in the order e = 0 .. 15 (the mask word out is assembled across them).
    auto load_masks = [&](unsigned gbase, const __amdgpu_buffer_rsrc_t rb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) wm[j] = __builtin_amdgcn_raw_buffer_load_b32(rb, (gbase + rlane + (unsigned)joff(j)) >> 5, 0, 0);
    };
    auto epi_elem = [&](int i, int j, int e, unsigned gbase, const __amdgpu_buffer_rsrc_t rc, const __amdgpu_buffer_rsrc_t rb) __attribute__((always_inline)) {
        // (the table entry is made opaque IN PLACE: row offset + column-tile offset is loop-invariant, and hoisted out of the group loop it costs a register
        //  per (value, column tile) -- 64 in the layer-2 data gradient, which then spills; left here, joff(j) becomes the store's immediate)
        asm volatile("" : "+v"(roff[i][kPackRoff ? e >> 1 : e]));
        const unsigned rpk = roff[i][kPackRoff ? e >> 1 : e];
        const unsigned ro = (kPackRoff ? ((e & 1) ? rpk >> 16 : rpk & 0xffffu) : rpk) + (unsigned)joff(j);      // (from the group's base: rsrc_c_of)
        float v;
        if constexpr (EPI == R_MASKB || EPI == R_MASKB_CLS4) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)wm[j], (MT == 2 ? 32 * i : 0) + (e & 3) + 8 * (e >> 2));
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)wm[j], (MT == 2 ? 32 * i : 0) + (e & 3) + 8 * (e >> 2) + 4);
            v = r_keep_where(acc[i][j][e] * un, lo, hi);                // lanes 0..31 (lh = 0): bit li of `lo`, lanes 32..63: of `hi`
        } else {
            v = __builtin_fmaf(acc[i][j][e], un, bj[j]);                      // (un is a power of two: the product is exact, the fused form rounds as mul + add did)
            v = v < 0.0f ? 0.0f : v;                                          // (a NaN stays a NaN, as kernel Z's SPLIT epilogue)
        }
        cmax = __builtin_fmaxf(cmax, __builtin_fabsf(v));                     // (rows past the batch: zeros, or relu(bias) of a real channel -- see below)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, ro, 0, MI355_AUX_STREAM_ST);
        if constexpr (EPI == R_BIAS_RELU_BITS) {          // lanes 0..31 of the ballot: the 32 channels of slot row (e & 3) + 8 (e >> 2); 32..63: of that row + 4
            if (e == 0) wv = 0;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(v > 0.0f);
            wv = r_writelane(wv, (unsigned)bal, (e & 3) + 8 * (e >> 2));
            wv = r_writelane(wv, (unsigned)(bal >> 32), (e & 3) + 8 * (e >> 2) + 4);
            if (e == 15) __builtin_amdgcn_raw_buffer_store_b32((unsigned)wv, rb, lh == 0 ? (gbase + rword[i] + (unsigned)joff(j)) >> 5 : kROob, 0, 0);
        }
    };
    static_assert(RG::NS == 1 || EPI != R_MASKB_CLS4 || NTW == 2, "stride-parity classes: a wave's column tiles are one class row (joff additive)");
    auto group = [&](int grp) __attribute__((always_inline)) {
        const unsigned gbase = (unsigned)grp * kGroupC + (unsigned)joff(jg * NTW);      // (joff is additive over the waves' column-tile groups: static_assert below)
        const __amdgpu_buffer_rsrc_t rb_cur = rsrc_b_of(grp);
        // every wave is done with the previous group's records and ring slots
#if MI355_RING_DMA
        ring_barrier();                                   // (a bare s_barrier: __syncthreads' fence would drain vmcnt -- the previous group's stores -- while an LDS-DMA may be pending)
        fill();
        __builtin_amdgcn_sched_barrier(0);
        ring_barrier();                                   // the records are written (slot 0 landed before the previous epilogue / the prologue's wait)
#else
        __syncthreads();
        fill();
        if constexpr (!kCarry) { load_slot(0); load_slot(1); }
        __builtin_amdgcn_sched_barrier(0);
        write_slot(0);
        load_slot(2);
        ring_barrier();
#endif
        read_a(0, 0);
        read_b(0, 0);
#pragma unroll
        for (int v = 0; v < RG::KSTEPS; ++v) {
            const int q = v & 1, slot = v / SS, h = v % SS;
            // the operands of step v + 1 are requested before the matrix instructions of step v go out (a wave cannot run ahead of the matrix
            // pipe: whatever is issued behind a step's MFMAs starts when they end)
#if MI355_RING_DMA
            if (v + 1 < RG::KSTEPS) {
                if (h == SS - 1) {
                    dma_wait(slot);                       // this wave's pieces of slot + 1 have landed
                    ring_barrier();
                }
                read_b(q ^ 1, v + 1);
                read_a(q ^ 1, v + 1);
            }
            if (h == 0) {
                dma_slot(slot + 1);
                __builtin_amdgcn_sched_barrier(0);        // (r_dma_younger assumes the pieces are issued in front of this step's prefetch round)
            }
#else
            if (v + 1 < RG::KSTEPS) {
                if (h == SS - 1) ring_barrier();          // slot + 1 has landed in its buffer (written at the first step of this slot)
                read_b(q ^ 1, v + 1);
                read_a(q ^ 1, v + 1);
            }
            if (h == 0) {
                if (slot + 1 < NSLOT) write_slot(slot + 1);
                if (slot + 3 < NSLOT) load_slot(slot + 3);
                else if (kCarry && slot + 3 - NSLOT < 2) load_slot(slot + 3 - NSLOT);     // (the last group fetches them for nobody)
            }
#endif
            // the next group's source, a few loads at a time (pre_lo above): the whole group at once (104 KB per CU in the layer-3
            // forward) exceeds what a CU keeps in flight and held the issuing waves -- and the matrix pipe behind them -- for 7,000 cycles
            if (pre_lo(v + 1) > pre_lo(v)) prefetch(grp + gridDim.x, pre_lo(v), pre_lo(v + 1) - pre_lo(v));
            if constexpr (EPI == R_MASKB || EPI == R_MASKB_CLS4)
                if (v == RG::KSTEPS - 4) load_masks(gbase, rb_cur);      // this group's mask words, for its epilogue
            __builtin_amdgcn_sched_barrier(0);
            // hi hi, hi lo (weights), lo hi (pixels): kernel Z's order of the three term pairs, tiles innermost
#pragma unroll
            for (int pi = 0; pi < 3; ++pi)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        if (v == 0 && pi == 0) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;      // (the MFMA's inline zero)
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, pa[q][i][pi == 2 ? 1 : 0]),
                                                                           __builtin_bit_cast(s_f16x8, wb[q][j][pi == 1 ? 1 : 0]), acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        // Every load this wave has in flight -- the next group's source, its first two ring slots, this group's mask words -- is waited for HERE, in
        // front of the epilogue's stores: vmcnt counts loads and stores in one in-order queue, so a wait for a load placed BEHIND the 32 - 64
        // stores (where the compiler puts it: at the fill's first use) holds the wave until the stores are acknowledged as well -- a bubble of
        // a write's round trip per group with nothing to overlap it (one workgroup per CU).  With the loads known complete, fill and k-loop
        // start while the stores drain.
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(kRWaitVm0);
        __builtin_amdgcn_sched_barrier(0);
        // the group's values (kernel Z's epilogue orders: rows outermost for the masked gradients, tiles outermost for the forward's mask words)
        const __amdgpu_buffer_rsrc_t rc = rsrc_c_of(grp, gbase);
        if constexpr (EPI == R_MASKB || EPI == R_MASKB_CLS4) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) epi_elem(i, j, e, gbase, rc, rb_cur);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) epi_elem(i, j, e, gbase, rc, rb_cur);
        }
    };

    int grp = blockIdx.x;
    if (grp < a.groups) {
        prefetch(grp, 0, NI);
#if MI355_RING_DMA
        dma_slot(0);
#else
        if constexpr (kCarry) { load_slot(0); load_slot(1); }
#endif
        __builtin_amdgcn_s_waitcnt(kRWaitVm0);              // (the loop is entered as the back edge enters it: nothing pending, so its top needs no wait -- see group())
    }
    for (; grp < a.groups; grp += gridDim.x) group(grp);
    if (a.c_amax) amax_commit(a.c_amax, __float_as_uint(cmax), blockIdx.x * NW + (unsigned)wave, lane);
}

template <class RG, int EPI, int SPREAD_MODE = 0>      // 1: the prefetch spread over the k-loop (the data gradients' small sources: measured +-0, not instantiated)
static int r_launch(const RArgs& a0, hipStream_t s, const char* what) {
    RArgs a = a0;
    a.groups = (int)((a.images + RG::G - 1) / RG::G);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        cus = n;
    }
    const int grid = a.groups < cus * RG::WGS ? a.groups : cus * RG::WGS;
    // (the forwards' prefetch is spread over the k-loop -- SPREAD_MODE 1; same-box A/B against two loads per step: profiles/r05_kernel_r_spread_ab.jsonl)
    hipLaunchKernelGGL((r_kernel<RG, EPI, SPREAD_MODE != 0>), dim3((unsigned)grid), dim3(64 * RG::NW), 0, s, a);
    return check_launch(what);
}

// Which launches kernel R takes (profiles/r05_kernel_r_sizes.txt: same-box A/B at 256 .. 32,768 images): both forwards and the layer-3 data
// gradient at every size (at or below kernel Z's time everywhere; at 32,768 images 0.72 x for the layer-3 forward, 0.90 x for the layer-2
// forward -- its 51-KB source allows two images = 162 of 192 row slots per workgroup), the layer-2 data gradient from `min_images` = 512 on
// (0.76 x at 32,768; 1.13 x at 256: one persistent workgroup per CU with two images each leaves half the chip idle there).
// MI355PPO_CONV_R=0: never (A/B runs; the results are bit-identical either way); =min:<n>: every layer from n images on (common.h::kernel_switch).
bool convr_on(long long images, long long min_images) { return kernel_switch("MI355PPO_CONV_R", images, min_images); }

int convr_fwd3(const char* fn, const float* src, unsigned src_bytes, const void* pack, const float* bias, float* dst, unsigned dst_bytes, unsigned* bits,
               long long images, const unsigned* src_amax, unsigned* dst_amax, hipStream_t st) {
    RArgs a{};
    a.A = src; a.a_bytes = src_bytes; a.pack = static_cast<const unsigned char*>(pack); a.bias = bias; a.bits_out = bits; a.C = dst; a.c_bytes = dst_bytes;
    a.images = images; a.a_amax = src_amax; a.c_amax = dst_amax;
    return bits ? r_launch<RConv3, R_BIAS_RELU_BITS, 1>(a, st, fn) : r_launch<RConv3, R_BIAS_RELU, 1>(a, st, fn);
}

int convr_fwd2(const char* fn, const float* src, unsigned src_bytes, const void* pack, const float* bias, float* dst, unsigned dst_bytes, unsigned* bits,
               long long images, const unsigned* src_amax, unsigned* dst_amax, hipStream_t st) {
    RArgs a{};
    a.A = src; a.a_bytes = src_bytes; a.pack = static_cast<const unsigned char*>(pack); a.bias = bias; a.bits_out = bits; a.C = dst; a.c_bytes = dst_bytes;
    a.images = images; a.a_amax = src_amax; a.c_amax = dst_amax;
    return bits ? r_launch<RConv2, R_BIAS_RELU_BITS, 1>(a, st, fn) : r_launch<RConv2, R_BIAS_RELU, 1>(a, st, fn);
}

int convr_dgrad3(const char* fn, const float* dz, unsigned dz_bytes, const void* pack, const unsigned* bits, float* dsrc, unsigned dsrc_bytes,
                 long long images, const unsigned* dz_amax, unsigned* dsrc_amax, hipStream_t st) {
    RArgs a{};
    a.A = dz; a.a_bytes = dz_bytes; a.pack = static_cast<const unsigned char*>(pack); a.bits_in = bits; a.C = dsrc; a.c_bytes = dsrc_bytes;
    a.images = images; a.a_amax = dz_amax; a.c_amax = dsrc_amax;
    return r_launch<RDgrad3, R_MASKB>(a, st, fn);
}

int convr_dgrad2(const char* fn, const float* dz, unsigned dz_bytes, const void* pack, const unsigned* bits, float* dsrc, unsigned dsrc_bytes,
                 long long images, const unsigned* dz_amax, unsigned* dsrc_amax, hipStream_t st) {
    if (convrb_takes(images)) return convrb_dgrad2(fn, dz, dz_bytes, pack, bits, dsrc, dsrc_bytes, images, dz_amax, dsrc_amax, st);      // kernel RB (convrb.hip)
    RArgs a{};
    a.A = dz; a.a_bytes = dz_bytes; a.pack = static_cast<const unsigned char*>(pack); a.bits_in = bits; a.C = dsrc; a.c_bytes = dsrc_bytes;
    a.images = images; a.a_amax = dz_amax; a.c_amax = dsrc_amax;
    return r_launch<RDgrad2, R_MASKB_CLS4>(a, st, fn);
}

}  // namespace mi355ppo
