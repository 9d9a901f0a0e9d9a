// Kernel RB -- the layer-2 data gradient of the NatureCNN (the backward of cleanrl/ppo_atari_multigpu.py:141, `loss.backward()` :358) with the
// rows of an image group dealt to tiles BY BORDER CLASS (round 6).  Kernel R's structure (convr.hip: the group's dz2 resident in LDS as f16
// hi / lo records, split once; weights from the f16x2 pack through the workgroup's two-buffer LDS ring; kernel Z's epilogue) with the
// geometry of convrb_geom.h:
//   * THREE images per group (padded lines share their border record: 117 records per image), 300 rows in ten 32-row tiles;
//   * six interior tiles walk all 16 k-steps, the four rim tiles (bottom / top line, right / left column of the 10 x 10 grid of window
//     origins) only the 8 k-steps of the two taps that can lie inside the image -- the skipped products are exact zeros in kernel R / Z;
//   * 8 waves = 4 stride-parity classes (one column tile each) x 2 row-waves; a wave holds 3 interior + 2 rim accumulator tiles (80
//     registers) and issues, in every k-step, 4 tiles x 3 term pairs = 12 matrix instructions (kernel R: 2 x 2 x 3): 192 per group of
//     THREE images where kernel R issues 192 per group of two;
//   * pixel fragments single-buffered: a pair of tiles' fragments for step v + 1 is requested right behind that pair's matrix instructions
//     of step v (six matrix instructions of distance), weight fragments double-buffered as in kernel R;
//   * the epilogue's row offsets (5 tiles x 16 values per lane half) come from a 1.25-KB table in LDS instead of 80 registers.
// Arithmetic: per accumulator the products of kernel Z's SPLIT = 1 instance in its order (k-steps ascending, per step hi hi, hi lo, lo hi)
// minus products with a zero-border operand: bit-identical results (tools/conv_traffic hashes, tests/test_gpu_f16x2.py).
#include "common.h"
#include "f16split.h"
#include "convrb_geom.h"

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float rb_f32x16 __attribute__((ext_vector_type(16)));

constexpr unsigned kRBOob = 0xFFFFF000u;              // buffer offset out of range for every tensor < 4 GiB - 4 KiB
constexpr int kRBRsrcWord3 = 0x00020000;              // raw buffer, 32-bit elements
constexpr int kRBWaitVm0 = 0x0F70;                    // s_waitcnt vmcnt(0) (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)

// x where bit (lane) of {hi, lo} is set, else 0 (convr.hip's r_keep_where: the s_nop covers the wait states a VALU read of an SGPR needs
// behind a VALU write the compiler cannot see inside the asm)
__device__ __forceinline__ float rb_keep_where(float x, unsigned lo, unsigned hi) {
    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
    float r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
    return r;
}

struct RBArgs {
    const float* A;             // dz2 (images, 9, 9, 64)
    unsigned a_bytes;
    const unsigned char* pack;  // f16x2 pack of the class matrices (header + [k-step][class][hi, lo][lane][8 f16])
    const unsigned* bits_in;    // ReLU mask of a1, one bit per element of (images, 20, 20, 32)
    float* C;                   // da1 (images, 20, 20, 32)
    unsigned c_bytes;
    long long images;
    int groups;
    int grid;                   // workgroups of the launch (the group stride)
    const unsigned* a_amax;     // amax record of dz2
    unsigned* c_amax;           // amax record of da1 to fold into, or null
};

__global__ __launch_bounds__(RBGeom::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void rb_kernel(RBArgs a) {
    using RG = RBGeom;
    constexpr int NT = RG::NT, MT = RG::MT, NW = RG::NW, NI = RG::NI, THREADS = RG::THREADS, SS = RG::SS, NSLOT = RG::NSLOT, NI3 = RG::NI3;
    constexpr int kPieces = SS * NT * 2;                  // KiB pieces of a ring slot: (step h of the slot, class j, term t), x = (h NT + j) 2 + t
    constexpr int kShare = kPieces / NW;                  // every wave moves kShare pieces of a slot
    static_assert(kPieces % NW == 0 && NSLOT % 2 == 0 && NSLOT >= 4, "whole pieces per wave; the sets' parity carries over the group boundary");
    __shared__ __attribute__((aligned(16))) unsigned char lds[RG::LDSB];
    unsigned char* const ring = lds + RG::ABYTES;
    unsigned* const rofft = reinterpret_cast<unsigned*>(lds + RG::ABYTES + 2 * RG::SLOTB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int rw = wave & 1, jg = wave >> 1;               // row-wave; class (wave-uniform)

    const int ea = f16_scale_exp(amax_load(a.a_amax, lane));
    const int eb = f16_scale_exp(*reinterpret_cast<const unsigned*>(a.pack));
    const float sa = f16_pow2(ea), un = f16_unscale(ea, eb);

    // ---- rows.  C offset of row (image gi of the group, window origin (gy, gx)) for class 0: pixel (2 gy, 2 gx) of the (20, 20, 32) image
    constexpr RBRowTable table{};
    auto row_coff = [](int id) __attribute__((always_inline)) -> unsigned {
        const int gi = id / 100, gy = (id % 100) / 10, gx = id % 10;
        return (unsigned)(((gi * 20 + 2 * gy) * 20 + 2 * gx) * 32) * 4u;
    };
    unsigned win[NI3], winr[NSLOT];                       // LDS byte offset of the lane's fragment row's window origin (+ the lane half's 8 channels): interior tiles; the rim tile of ring slot s
    auto win_of = [&](int tile) __attribute__((always_inline)) -> unsigned {
        const int id = table.src[(rw * MT + tile) * 32 + li];
        return (unsigned)((id / 100) * RG::IMGB + RG::pidx((id % 100) / 10, id % 10) * RG::PIX + 16 * lh);
    };
#pragma unroll
    for (int i = 0; i < NI3; ++i) win[i] = win_of(i);
    {
        const unsigned w3 = win_of(NI3), w4 = win_of(NI3 + 1);
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) winr[sl] = (rw == 0 ? (sl >> 1) : (sl & 1)) ? w4 : w3;      // RG::rim_of
    }
    // the epilogue's table: word ((rw MT + i) 2 + lh) 16 + e = C offset of the row in slot (e & 3) + 8 (e >> 2) + 4 lh of tile i
    if (tid < RG::RW * MT * 32) {
        const int t = tid >> 5, h = (tid >> 4) & 1, e = tid & 15;          // t = rw MT + i
        rofft[tid] = row_coff(table.src[t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h]);
    }
    unsigned char* const ring_l = ring + 16 * lane;
    const unsigned char* const ring_r = ring_l + jg * 2048;                // this wave's class of a k-step

    // ---- the group's source: unit u = 16 bytes = 4 channels of a pixel; thread tid takes units it * THREADS + tid
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)a.a_bytes, kRBRsrcWord3);
    unsigned udst[NI / 2];                                // LDS byte address / 8 of two units' hi halves (lo: + LO), 16 bits each; 0xffff: no such unit
    static_assert(NI % 2 == 0 && RG::ABYTES / 8 < 0xffff, "two 16-bit record offsets per register");
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int u = it * THREADS + tid;
        const int pix = u / RG::UPP, c4 = u - pix * RG::UPP, img = pix / (RG::IH * RG::IW), q = pix - img * (RG::IH * RG::IW), qy = q / RG::IW, qx = q - qy * RG::IW;
        const unsigned d = u < RG::UNITS ? (unsigned)(img * RG::IMGB + RG::pidx(qy + 1, qx + 1) * RG::PIX + c4 * 8) >> 3 : 0xffffu;
        udst[it >> 1] = (it & 1) ? (udst[it >> 1] | (d << 16)) : d;
    }
    // the next group's source: two loads per k-step from the second step on (unconditional: past the last group with out-of-range offsets, which
    // load zeros without touching memory -- see convr.hip)
    auto pre_lo = [](int v) constexpr -> int {
        const int n = (v - 1) * 2;
        return v < 1 ? 0 : (n > NI ? NI : n);
    };
    static_assert(pre_lo(RG::KSTEPS) == NI, "the next group's source is requested inside one k-loop");
    s_u32x4 pre[NI];
    auto prefetch = [&](int grp, int it0, int n) __attribute__((always_inline)) {
        const bool any = grp < a.groups;
        const unsigned base = (unsigned)grp * (unsigned)(RG::UNITS * 16);
#pragma unroll
        for (int it = it0; it < it0 + n && it < NI; ++it) {
            const int u = it * THREADS + tid;
            pre[it] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (any && u < RG::UNITS) ? base + (unsigned)u * 16u : kRBOob, 0, MI355_AUX_STREAM_LD));
        }
    };
    auto fill = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            unsigned hi[2], lo[2];
            f16_split4(pre[it], sa, hi, lo);
            // (the packed word is made opaque in place: unpacked outside the group loop the eight addresses cost registers the k-loop does not have,
            //  and a scratch reload here waits -- vmcnt is one in-order queue -- for the previous group's stores)
            if ((it & 1) == 0) asm volatile("" : "+v"(udst[it >> 1]));
            const unsigned d = (it & 1) ? udst[it >> 1] >> 16 : udst[it >> 1] & 0xffffu;
            if ((it + 1) * THREADS <= RG::UNITS || d != 0xffffu) {
                *reinterpret_cast<uint2*>(lds + 8u * d) = make_uint2(hi[0], hi[1]);
                *reinterpret_cast<uint2*>(lds + 8u * d + RG::LO) = make_uint2(lo[0], lo[1]);
            }
        }
    };
    // the zero border (and everything else, once)
    for (int o = tid * 16; o < RG::ABYTES; o += THREADS * 16) *reinterpret_cast<s_u32x4*>(lds + o) = (s_u32x4){0u, 0u, 0u, 0u};

    // ---- epilogue constants: C's descriptor of a group starts AT the group (+ the wave's class): offsets are table words + the lane's channel
    constexpr unsigned kGroupC = RG::G * 4 * RG::OP * 32 * 4;             // bytes of da1 per group
    const unsigned cls_off = (unsigned)(((jg >> 1) * 20 + (jg & 1)) * 32 * 4);
    auto rsrc_c_of = [&](int g, unsigned gbase) __attribute__((always_inline)) {
        const bool ok = g >= 0 && g < a.groups;
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.C) + (ok ? gbase : 0u), 0, ok ? (int)(a.c_bytes - gbase) : 0, kRBRsrcWord3);
    };
    auto rsrc_b_of = [&](int g) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.bits_in), 0, (g >= 0 && g < a.groups) ? (int)(a.c_bytes >> 5) : 0, kRBRsrcWord3);
    };
    float cmax = 0.0f;

    // ---- the weights' ring (kernel R's: slot s = k-steps SS s .. SS s + SS - 1 = one tap, two buffers, registers two slots ahead of the write)
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.pack), 0, kF16PackHeader + RG::KSTEPS * RG::STEPB, kRBRsrcWord3);
    const unsigned lane16 = 16u * (unsigned)lane;
    s_u32x4 bst[2][kShare];                               // slot s travels in set s & 1
    auto load_slot = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kShare; ++u) {
            const int x = wave * kShare + u, h = x / (NT * 2), jt = x - h * (NT * 2);       // (wave-uniform: scalar arithmetic)
            bst[slot & 1][u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_p, lane16, kF16PackHeader + (SS * slot + h) * RG::STEPB + jt * 1024, 0));
        }
    };
    auto write_slot = [&](int slot) __attribute__((always_inline)) {                     // -> buffer slot & 1
#pragma unroll
        for (int u = 0; u < kShare; ++u) *reinterpret_cast<s_u32x4*>(ring_l + (slot & 1) * RG::SLOTB + (wave * kShare + u) * 1024) = bst[slot & 1][u];
    };
    auto ring_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
#if MI355_RING_DMA
    // The ring by LDS-DMA (buffer_load_dwordx4 ... lds): a piece = one wave instruction, global -> LDS without registers and without the ds_write pass.  Slot s + 1
    // is requested at the first step of slot s into the buffer every wave read for the last time before the barrier of slot s - 1; the issuing wave waits for ITS
    // pieces (vmcnt counts in order: `younger` = the vector loads it issued behind them) in front of the barrier at the last step of slot s.
    auto dma_slot = [&](int slot) __attribute__((always_inline)) {                       // -> buffer slot & 1
#pragma unroll
        for (int u = 0; u < kShare; ++u) {
            const int x = wave * kShare + u, h = x / (NT * 2), jt = x - h * (NT * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_p, (__attribute__((address_space(3))) void*)(ring + (slot & 1) * RG::SLOTB + x * 1024), 16, lane16,
                                                     kF16PackHeader + (SS * (slot % NSLOT) + h) * RG::STEPB + jt * 1024, 0, 0);
        }
    };
#endif

    rb_f32x16 acc[MT];
    unsigned wm[3];                                       // lane L of wm[k]: the mask word of wave row 64 k + L (this wave's class)
    s_u32x4 pa[4][2], wb[2][2];                           // pixel fragments of the step's four tiles [pair position][hi, lo]; weight fragments [k-step parity][hi, lo]
    auto read_a = [&](int pos, int tile, int v) __attribute__((always_inline)) {         // tile's fragment of k-step v -> position pos (tile NI3: the step's rim tile)
        const int off = rb_tapoff(v);
        const unsigned w = tile < NI3 ? win[tile] : winr[v / SS];
        pa[pos][0] = *reinterpret_cast<const s_u32x4*>(lds + w + off);
        pa[pos][1] = *reinterpret_cast<const s_u32x4*>(lds + w + off + RG::LO);
    };
    auto read_b = [&](int par, int v) __attribute__((always_inline)) {
        const int buf = (v / SS) & 1, h = v % SS;
#pragma unroll
        for (int t = 0; t < 2; ++t) wb[par][t] = *reinterpret_cast<const s_u32x4*>(ring_r + buf * RG::SLOTB + (h * NT * 2 + t) * 1024);
    };
    // mask words in: lane L of wm[k] takes wave row 64 k + L = slot x = L % 32 of tile 2 k + L / 32, whose C offset is table word (lane half (x >> 2) & 1,
    // value (x & 3) + 4 (x >> 3)) of that tile; wave rows 160 .. 191 do not exist
    const unsigned mword = (unsigned)((rw * MT * 2 + ((li >> 2) & 1)) * 16 + (li & 3) + 4 * (li >> 3) + 32 * lh);
    auto load_masks = [&](unsigned gbase, const __amdgpu_buffer_rsrc_t rb) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned ro = rofft[mword + (k < 2 || lh == 0 ? 64 * k : 0)];
            wm[k] = __builtin_amdgcn_raw_buffer_load_b32(rb, (k < 2 || lh == 0) ? (gbase + ro) >> 5 : kRBOob, 0, 0);
        }
    };
    // six matrix instructions of a pair of tiles: (hi hi, hi lo, lo hi) interleaved over the two tiles (a tile's three are dependent)
    auto pair = [&](int q, int t0, int p0, bool z0, int t1, int p1, bool z1) __attribute__((always_inline)) {
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) {
            if (pi == 0 && z0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t0][e] = 0.0f;                  // (the MFMA's inline zero)
            }
            acc[t0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, pa[p0][pi == 2 ? 1 : 0]), __builtin_bit_cast(s_f16x8, wb[q][pi == 1 ? 1 : 0]), acc[t0], 0, 0, 0);
            if (pi == 0 && z1) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t1][e] = 0.0f;
            }
            acc[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, pa[p1][pi == 2 ? 1 : 0]), __builtin_bit_cast(s_f16x8, wb[q][pi == 1 ? 1 : 0]), acc[t1], 0, 0, 0);
        }
    };

    // The rim accumulator a ring slot feeds is the same in both row-waves' code -- tile 3 in slots 0, 1, tile 4 in slots 2, 3: row-wave 0's bottom, bottom,
    // top, top.  Row-wave 1 (right, left, right, left) SWAPS the two accumulators in front of slots 1 and 3 (tile 4 zeroed at the group's start): slot 0
    // right -> tile 3; swap; slot 1 left -> tile 3, slot 2 right -> tile 4; swap; slot 3 left -> tile 4.  32 register moves per group in place of two
    // copies of the k-loop (which left the register allocator spilling the prefetched source inside the second copy) or of branches around the matrix
    // instructions (whose accumulators the compiler then merged with copies behind every branch).
    auto group = [&](int grp) __attribute__((always_inline)) {
        const unsigned gbase = (unsigned)grp * kGroupC + cls_off;
        const __amdgpu_buffer_rsrc_t rb_cur = rsrc_b_of(grp);
        // every wave is done with the previous group's records and ring slots
#if MI355_RING_DMA
        ring_barrier();                                   // (a bare s_barrier: __syncthreads' fence drains vmcnt -- the previous group's stores -- while an LDS-DMA may be pending)
#else
        __syncthreads();
#endif
        fill();
        __builtin_amdgcn_sched_barrier(0);
#if MI355_RING_DMA
        ring_barrier();                                   // the records are written (slot 0 landed before the previous epilogue / the prologue's wait)
#else
        write_slot(0);
        load_slot(2);
        ring_barrier();
#endif
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[NI3 + 1][e] = 0.0f;
        read_b(0, 0);
        read_a(0, 0, 0); read_a(1, 1, 0); read_a(2, 2, 0); read_a(3, NI3, 0);
#pragma unroll
        for (int v = 0; v < RG::KSTEPS; ++v) {
            const int q = v & 1, slot = v / SS, h = v % SS;
#if MI355_RING_DMA
            if (v + 1 < RG::KSTEPS) {
                if (h == SS - 1) {
                    // this wave's pieces of slot + 1: everything but the vector loads issued behind them (the prefetch rounds of steps SS slot + 1 .. v - 1)
                    constexpr int younger[NSLOT] = {4, 2, 0, 0};
                    static_assert(pre_lo(SS - 1) - pre_lo(1) == 4 && pre_lo(2 * SS - 1) - pre_lo(SS) == 2 && pre_lo(2 * SS) == NI, "the hand-counted waits");
                    if (younger[slot] == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (younger[slot] == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    ring_barrier();
                }
                read_b(q ^ 1, v + 1);
            }
            if (h == 0) {
                dma_slot(slot + 1);                       // (slot NSLOT = the next group's slot 0; the last group fetches it for nobody)
                __builtin_amdgcn_sched_barrier(0);        // (the counts above assume the pieces are issued in front of this step's prefetch round)
            }
#else
            if (v + 1 < RG::KSTEPS) {
                if (h == SS - 1) ring_barrier();          // slot + 1 has landed in its buffer (written at the first step of this slot)
                read_b(q ^ 1, v + 1);
            }
            if (h == 0) {
                if (slot + 1 < NSLOT) write_slot(slot + 1);
                if (slot + 3 < NSLOT) load_slot(slot + 3);
                else if (slot + 3 - NSLOT < 2) load_slot(slot + 3 - NSLOT);      // the next group's first two slots (the last group fetches them for nobody)
            }
#endif
            if (pre_lo(v + 1) > pre_lo(v)) prefetch(grp + a.grid, pre_lo(v), pre_lo(v + 1) - pre_lo(v));
            if (v == RG::KSTEPS - 4) load_masks(gbase, rb_cur);
            if (h == 0 && (slot & 1) && rw != 0) {          // row-wave 1: the rim accumulators change places
                const rb_f32x16 t = acc[NI3];
                acc[NI3] = acc[NI3 + 1];
                acc[NI3 + 1] = t;
            }
            __builtin_amdgcn_sched_barrier(0);
            pair(q, 0, 0, v == 0, 1, 1, v == 0);
            __builtin_amdgcn_sched_barrier(0);
            if (v + 1 < RG::KSTEPS) { read_a(0, 0, v + 1); read_a(1, 1, v + 1); }
            __builtin_amdgcn_sched_barrier(0);
            pair(q, 2, 2, v == 0, NI3 + (slot >> 1), 3, v == 0);
            __builtin_amdgcn_sched_barrier(0);
            if (v + 1 < RG::KSTEPS) { read_a(2, 2, v + 1); read_a(3, NI3, v + 1); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // every load in flight -- the next group's source, its first two ring slots, this group's mask words -- is waited for in front of the
        // epilogue's stores (vmcnt counts loads and stores in one in-order queue: convr.hip)
        __builtin_amdgcn_s_waitcnt(kRBWaitVm0);
        __builtin_amdgcn_sched_barrier(0);
        const __amdgpu_buffer_rsrc_t rc = rsrc_c_of(grp, gbase);
        const unsigned* const rt = rofft + ((rw * MT) * 2 + lh) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            unsigned ro[16];
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const s_u32x4 w = *reinterpret_cast<const s_u32x4*>(rt + i * 32 + 4 * e4);
                ro[4 * e4] = w[0]; ro[4 * e4 + 1] = w[1]; ro[4 * e4 + 2] = w[2]; ro[4 * e4 + 3] = w[3];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int wr = 32 * i + (e & 3) + 8 * (e >> 2);                 // wave row of lanes 0 .. 31's value; lanes 32 .. 63: + 4
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)wm[wr >> 6], wr & 63);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)wm[(wr + 4) >> 6], (wr + 4) & 63);
                const float v = rb_keep_where(acc[i][e] * un, lo, hi);
                cmax = __builtin_fmaxf(cmax, __builtin_fabsf(v));
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, ro[e] + 4u * (unsigned)li, 0, MI355_AUX_STREAM_ST);
            }
            __builtin_amdgcn_sched_barrier(0);            // (one tile's offsets in registers at a time: hoisted together they are 80)
        }
    };

    int grp = blockIdx.x;
    if (grp < a.groups) {
        prefetch(grp, 0, NI);
#if MI355_RING_DMA
        dma_slot(0);
#else
        load_slot(0);
        load_slot(1);
#endif
        __builtin_amdgcn_s_waitcnt(kRBWaitVm0);
    }
    for (; grp < a.groups; grp += a.grid) group(grp);
    if (a.c_amax) amax_commit(a.c_amax, __float_as_uint(cmax), blockIdx.x * NW + (unsigned)wave, lane);
}

// Which sizes kernel RB takes from kernel R's RDgrad2: one persistent workgroup per CU with three images each -- from 3,072 images on every CU
// has at least four groups.  MI355_RB_OFF (build-time, tools/build_variant.py): never -- the A/B library of tools/gpu/lib_ab.sh.
bool convrb_takes(long long images) {
#ifdef MI355_RB_OFF
    (void)images;
    return false;
#else
    return images >= MI355_RB_MIN_IMAGES;
#endif
}

int convrb_dgrad2(const char* fn, const float* dz, unsigned dz_bytes, const void* pack, const unsigned* bits, float* dsrc, unsigned dsrc_bytes,
                  long long images, const unsigned* dz_amax, unsigned* dsrc_amax, hipStream_t st) {
    RBArgs a{};
    a.A = dz; a.a_bytes = dz_bytes; a.pack = static_cast<const unsigned char*>(pack); a.bits_in = bits; a.C = dsrc; a.c_bytes = dsrc_bytes;
    a.images = images; a.a_amax = dz_amax; a.c_amax = dsrc_amax;
    a.groups = (int)((images + RBGeom::G - 1) / RBGeom::G);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        cus = n;
    }
    a.grid = a.groups < cus ? a.groups : cus;
    hipLaunchKernelGGL(rb_kernel, dim3((unsigned)a.grid), dim3(RBGeom::THREADS), 0, st, a);
    return check_launch(fn);
}

}  // namespace mi355ppo
