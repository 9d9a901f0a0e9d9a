// Kernel H -- the weight gradient of Linear(3136, 512) (cleanrl/ppo_atari_multigpu.py:144 and its backward, :358) on the two-term f16 split with
// BOTH operands streamed through a workgroup-wide LDS ring, split once, fragments by LDS transpose reads; round 6 (kernel U's idea applied to
// kernel W's problem):
//     dWp[n][k] = sum over the batch rows m of dz[m][n] * a[m][k]          n < 512, k < 3136, m < M (32,768 per minibatch)
// The reduction index m is the SLOW index of both operands.  Kernel W (fcw.hip) transposes both through wave-private LDS with 4-byte reads and
// splits every fragment in registers: 264 VALU instructions and 48 LDS reads per 48 matrix instructions, one wave per SIMD, matrix pipe 0.38
// busy (profiles/r06_g1_pmc_busy.csv).  Here a workgroup of eight waves (two per SIMD) owns a 256 (n) x 224 (k) block of dWp for one SLAB of
// the batch and
//   * streams the slab in slots of 32 rows: 512 threads load the slot's [32][256] block of dz and [32][224] block of a coalesced (16 bytes per
//     lane, 8 loads), split every element ONCE (each tensor's scale from its amax record) and store hi / lo halves as row records -- dz: 512 B hi
//     | 512 B lo | 64 B pad = 1,088 B, a: 448 B hi | 448 B lo | 64 B pad = 960 B: pitches of 64 / 192 (mod 256), so that the four consecutive rows
//     of a transpose-read block sit on four different quarters of the 64 banks;
//   * reads MFMA fragments with `ds_read_b64_tr_b16` (a 16-lane group reads a [4 rows][16 columns] block and every lane receives one column's
//     four rows: "8 consecutive m of one column" is two such reads) -- no VALU and no address arithmetic in the loop: 4 + 28 reads per 21 matrix
//     instructions per wave (wave w: n tile w of the block x its seven k tiles, 112 accumulator registers);
//   * two-buffer ring, one barrier per slot, global loads two slots ahead of their LDS write (kernel G's protocol).
// The batch is cut into 8 slabs = the 8 XCDs (workgroup L -> XCD L % 8 -> slab L % 8): the 28 workgroups of a slab run on one XCD and walk the
// slab's rows in step, so both operands leave HBM once and are served to the other workgroups by that L2.  One partial per slab; fcw_reduce_kernel
// adds them in slab order (deterministic) and writes dW in the reference's (c, h, w) feature order.
// Arithmetic: exact products of the f16 terms (hi hi, hi lo, lo hi), f32 accumulation in another order than kernel W's (8 slabs of contiguous rows
// instead of 5 interleaved ones): held to float64 with kernel W's bars (tests/test_gpu_f16x2.py), not bit-compared.
#include "common.h"
#include "f16split.h"

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float h_f32x16 __attribute__((ext_vector_type(16)));
typedef short h_s16x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kHOob = 0xFFFFF000u;
constexpr int kHRsrcWord3 = 0x00020000;

struct HGeom {
    static constexpr int NW = 8, THREADS = 64 * NW, CO = 32 * NW, TPW = 7, CI = 32 * TPW, SS = 2, ROWS = 16 * SS;
    static constexpr int PD = 4 * CO + 64, LOD = 2 * CO, PA = 4 * CI + 64, LOA = 2 * CI;       // row records: hi | lo | 64 B pad
    static constexpr int DBUF = ROWS * PD, ABUF = ROWS * PA, SLOT = DBUF + ABUF;
    static constexpr int UD = ROWS * CO / 4, UA = ROWS * CI / 4, NID = (UD + THREADS - 1) / THREADS, NIA = (UA + THREADS - 1) / THREADS, NI = NID + NIA;
    static constexpr int SLABS = 8;
    static_assert(PD % 256 == 64 && PA % 256 == 192 && 2 * SLOT <= 160 * 1024 && UD % THREADS == 0, "shape");
};

__global__ __launch_bounds__(HGeom::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void h_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ a, float* __restrict__ part, int M, int N, int K, int rows_per_slab, int co_blocks,
    unsigned dz_bytes, unsigned a_bytes, const unsigned* __restrict__ dz_amax, const unsigned* __restrict__ a_amax) {
    using HG = HGeom;
    constexpr int TPW = HG::TPW, SS = HG::SS, NID = HG::NID, NI = HG::NI, PD = HG::PD, PA = HG::PA;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HG::SLOT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15, r = i >> 2, c4 = i & 3, half = g >> 1;          // 16-lane group; its lane's block row / column chunk

    const int ed = f16_scale_exp(amax_load(dz_amax, lane)), es = f16_scale_exp(amax_load(a_amax, lane));
    const float sd = f16_pow2(ed), ss = f16_pow2(es), un = f16_unscale(ed, es);

    const int slab = blockIdx.x % HG::SLABS, blk = blockIdx.x / HG::SLABS;
    const int cob = blk % co_blocks, cib = blk / co_blocks;
    const int n0 = cob * HG::CO, k0 = cib * HG::CI;
    const int m_lo = slab * rows_per_slab, nslot = rows_per_slab / HG::ROWS;               // (even: host-checked)

    // ---- a slot's units: 16 bytes = 4 columns of a row; thread tid takes dz units it * THREADS + tid (it < NID), then a units
    const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, (int)dz_bytes, kHRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, (int)a_bytes, kHRsrcWord3);
    unsigned goff[NI], loff[NI];                          // byte offset in the tensor at the slab's first row (kHOob: no such unit); in a buffer
    const unsigned rowb_d = (unsigned)lddz * 4u, rowb_a = (unsigned)K * 4u;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        if (it < NID) {
            const int u = it * HG::THREADS + tid, row = u / (HG::CO / 4), q = u - row * (HG::CO / 4);
            goff[it] = (unsigned)(m_lo + row) * rowb_d + (unsigned)(n0 + 4 * q) * 4u;
            loff[it] = (unsigned)(row * PD + q * 8);
        } else {
            const int u = (it - NID) * HG::THREADS + tid, row = u / (HG::CI / 4), q = u - row * (HG::CI / 4);
            goff[it] = u < HG::UA ? (unsigned)(m_lo + row) * rowb_a + (unsigned)(k0 + 4 * q) * 4u : kHOob;
            loff[it] = u < HG::UA ? (unsigned)(HG::DBUF + row * PA + q * 8) : ~0u;
        }
    }
    s_u32x4 pre[2][NI];                                   // slot s travels in set s & 1
    // (rows past M -- the last slab, slots past the batch -- fall out of the tensors' ranges and load zeros; slots past the slab are never multiplied)
    auto load_slot = [&](int set, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const unsigned so = (unsigned)(s * HG::ROWS) * (it < NID ? rowb_d : rowb_a);
            const unsigned o = goff[it] == kHOob ? kHOob : goff[it] + so;
            pre[set][it] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(it < NID ? rsrc_d : rsrc_a, o, 0, MI355_AUX_WGRAD_LD));
        }
    };
    auto write_slot = [&](int set, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            unsigned hi[2], lo[2];
            f16_split4(pre[set][it], it < NID ? sd : ss, hi, lo);
            if (loff[it] != ~0u) {
                unsigned char* const d = lds + buf * HG::SLOT + loff[it];
                *reinterpret_cast<uint2*>(d) = make_uint2(hi[0], hi[1]);
                *reinterpret_cast<uint2*>(d + (it < NID ? HG::LOD : HG::LOA)) = make_uint2(lo[0], lo[1]);
            }
        }
    };
    auto ring_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ---- fragments.  Lane (g, r, c4) passes the address of 4 contiguous halves: row r of its group's block, columns 16 (g & 1) + 4 c4 .. + 3 of the
    // tile; block b = 2 half + t (t: first / second read of a fragment) = rows 4 b .. 4 b + 3 of the k-step: consecutive records, 64 / 192 bytes
    // apart mod 256 -- the 32 lanes of a pass (4 rows x 2 groups x 4 chunks) cover all 64 banks once.
    const unsigned char* const fd = lds + (8 * half + r) * PD + (32 * wave + 16 * (g & 1) + 4 * c4) * 2;
    const unsigned char* const fa = lds + HG::DBUF + (8 * half + r) * PA + (16 * (g & 1) + 4 * c4) * 2;
    typedef h_s16x4 __attribute__((address_space(3))) * lds_v4;
    auto tr2 = [&](const unsigned char* p0, const unsigned char* p1) __attribute__((always_inline)) -> s_u32x4 {      // 8 m of this lane's column
        const h_s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0));
        const h_s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p1));
        const uint2 ux = __builtin_bit_cast(uint2, x), uy = __builtin_bit_cast(uint2, y);
        return (s_u32x4){ux.x, ux.y, uy.x, uy.y};
    };
    s_u32x4 afr[2][2], bfr[2][2];                         // [parity][hi, lo]: dz fragment of a step; a fragment of a (step, tile) pair
    auto load_a = [&](int par, int buf, int h) __attribute__((always_inline)) {
        const unsigned char* const p = fd + buf * HG::SLOT + 16 * h * PD;
        afr[par][0] = tr2(p, p + 4 * PD);
        afr[par][1] = tr2(p + HG::LOD, p + 4 * PD + HG::LOD);
    };
    auto load_b = [&](int par, int buf, int h, int j) __attribute__((always_inline)) {
        const unsigned char* const p = fa + buf * HG::SLOT + 16 * h * PA + 64 * j;
        bfr[par][0] = tr2(p, p + 4 * PA);
        bfr[par][1] = tr2(p + HG::LOA, p + 4 * PA + HG::LOA);
    };

    h_f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

    // one slot: its (step, tile) pairs in one software pipeline -- the reads of pair q + 1 go out before the matrix instructions of pair q; at the
    // first pair the other buffer gets the next slot (from `wset`), whose set then takes the loads of the slot after next
    auto slot_body = [&](int buf, int wset, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < SS * TPW; ++q) {
            const int h = q / TPW, j = q - h * TPW;
            if (q == 0) {
                write_slot(wset, buf ^ 1);
                load_slot(wset, s + 3);
            }
            if (q + 1 < SS * TPW) {
                const int h1 = (q + 1) / TPW, j1 = (q + 1) - h1 * TPW;
                if (j1 == 0) load_a(h1 & 1, buf, h1);
                load_b((q + 1) & 1, buf, h1, j1);
            } else {
                ring_barrier();                           // the next slot has landed in the other buffer
                load_a(0, buf ^ 1, 0);
                load_b(0, buf ^ 1, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const s_u32x4 a_hi = afr[h & 1][0], a_lo = afr[h & 1][1], b_hi = bfr[q & 1][0], b_lo = bfr[q & 1][1];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, a_hi), __builtin_bit_cast(s_f16x8, b_hi), acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, a_hi), __builtin_bit_cast(s_f16x8, b_lo), acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, a_lo), __builtin_bit_cast(s_f16x8, b_hi), acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert((SS * TPW) % 2 == 0, "the pair parity carries over from slot to slot");

    // ---- prologue: slots 0 and 1 in the sets, slot 0 in buffer 0, slot 2 on its way, the fragments of pair 0 requested
    load_slot(0, 0);
    load_slot(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    write_slot(0, 0);
    load_slot(0, 2);
    ring_barrier();
    load_a(0, 0, 0);
    load_b(0, 0, 0, 0);
#pragma clang loop unroll(disable)
    for (int s = 0; s < nslot; s += 2) {
        slot_body(0, 1, s);                               // slot s in buffer 0: set 1 (slot s + 1) -> buffer 1, then loads slot s + 3
        slot_body(1, 0, s + 1);                           // slot s + 1 in buffer 1: set 0 (slot s + 2) -> buffer 0, then loads slot s + 4
    }

    // ---- this slab's partial: accumulator e of tile j = row n0 + 32 wave + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column k0 + 32 j + lane % 32
    float* const pw = part + (size_t)slab * N * K;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) pw[(size_t)(n0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * lh) * K + k0 + 32 * j + li] = acc[j][e] * un;
}

// MI355PPO_FC_H=0: the FC weight gradient stays on kernel W (A/B runs); =min:<n>: from n rows on (common.h::kernel_switch).  Kernel H takes Linear(3136, 512)'s shape
// from 4,096 rows on (config B's minibatch: 82 -> 75 us; 2,048 rows: 53 = 53 us; profiles/r06_kernel_h_ab.txt).
bool gemmh_takes(int M, int N, int K, int lddz) {
    using HG = HGeom;
    return kernel_switch("MI355PPO_FC_H", M, 4096) && N % HG::CO == 0 && K % HG::CI == 0 && lddz % 4 == 0 &&
           ((long long)M + 1024) * K * 4 < (1LL << 32) - 8192 && ((long long)M + 1024) * lddz * 4 < (1LL << 32) - 8192;      // (slots past a slab's end are requested, never multiplied)
}
int gemmh_slabs() { return HGeom::SLABS; }

// partials [slab][N][K] into `part` (gemmh_slabs() of them); the caller reduces
int gemmh_launch(const float* dz, int lddz, const float* a, float* part, int M, int N, int K, const unsigned* dz_amax, const unsigned* a_amax, hipStream_t s) {
    using HG = HGeom;
    const int per = (M + HG::SLABS - 1) / HG::SLABS;
    const int rows_per_slab = (per + 2 * HG::ROWS - 1) / (2 * HG::ROWS) * (2 * HG::ROWS);          // whole slot pairs
    const int co_blocks = N / HG::CO, ci_blocks = K / HG::CI;
    hipLaunchKernelGGL(h_kernel, dim3((unsigned)(HG::SLABS * co_blocks * ci_blocks)), dim3(HG::THREADS), 0, s, dz, lddz, a, part, M, N, K, rows_per_slab, co_blocks,
                       (unsigned)((long long)M * lddz * 4), (unsigned)((long long)M * K * 4), dz_amax, a_amax);
    return check_launch("h_kernel");
}

}  // namespace mi355ppo
