// Kernel G -- the FC layer's forward and data gradient of the NatureCNN (Linear(3136, 512) and its backward,
// cleanrl/ppo_atari_multigpu.py:144,358) on the two-term f16 split with BOTH operands streamed through workgroup-wide LDS rings, round 6.
//
// What bounded kernel Z on these launches (profiles/r05_pmc_*.csv: matrix pipe 0.36 - 0.54 busy, TA 0.72 busy, one wave per SIMD): its four
// waves sit side by side on 64 rows x 512 columns, every wave loads, transposes and SPLITS the same A rows for itself inside the k-loop (101
// VALU instructions per 24 MFMAs, four times over) and streams its own B fragments from the L2 -- the whole 6.4-MB pack once per 64 rows,
// 3.3 GB of L2 -> CU traffic per launch.  Here a workgroup of eight waves (two per SIMD) owns a 128-row x 256-column block and
//   * streams A in SLOTS of two k-steps (32 k): 512 threads load the slot's 128 rows x 128 bytes once (16 bytes per lane, two loads), split
//     every element ONCE (f16split.h, A's scale from its amax record) and store hi / lo halves as row records of 144 bytes (64 B hi | 64 B
//     lo | 16 B pad: an odd number of sixteen-byte slots, so the "lane = row" fragment reads of kernel R are conflict-free);
//   * streams B (the f16x2 pack, already in fragment order) through the same two-buffer ring: a slot's 32 KiB pieces, four per wave;
//   * walks the k-steps with no VALU and no address arithmetic: 2 + 8 ds_read_b128 per 12 matrix instructions per wave, the operands of step
//     v + 1 requested before the matrix instructions of step v, one barrier per slot; global loads run two slots ahead of their LDS write
//     (kernel R's ring protocol, now for both operands), ACROSS the boundary between two blocks of the workgroup's list;
//   * is persistent: workgroup -> XCD-contiguous range of the block order, supertiles of `super_rows` row blocks x all column blocks, so that
//     the blocks in flight on one XCD share A rows (forward: the two column halves of a row block) or B columns (data gradient) in its L2.
// Arithmetic: kernel Z's SPLIT = 1 products in kernel Z's order (k-steps ascending; per step hi hi, hi lo, lo hi; f32 accumulate; the same
// epilogues) -- results, mask handling and amax records are kernel Z's BIT FOR BIT (tools/conv_traffic hashes, tests/test_gpu_f16x2.py).
#include "common.h"
#include "f16split.h"

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float g_f32x16 __attribute__((ext_vector_type(16)));

enum { G_BIAS_RELU = 0, G_MASKB = 1 };

constexpr unsigned kGOob = 0xFFFFF000u;               // buffer offset out of range for every tensor < 4 GiB - 4 KiB
constexpr int kGRsrcWord3 = 0x00020000;               // raw buffer, 32-bit elements

struct GGeom {
    static constexpr int NW = 8, RW = 4, NS = 2, NTW = 4, NT = NS * NTW, SS = 2, ROWS = 32 * RW, COLS = 32 * NT, THREADS = 64 * NW;
    static constexpr int APITCH = SS * 64 + 16, ALO = SS * 32, ABUF = ROWS * APITCH;     // A row record of a slot: SS x 32 B hi | SS x 32 B lo | 16 B pad
    static constexpr int STEPB = NT * 2048, SLOTB = SS * STEPB;                       // B of a k-step of the block; of a slot
    static constexpr int AUNITS = ROWS * SS * 4, NIA = AUNITS / THREADS;              // 16-byte units of a slot's A; per thread
    static constexpr int PIECES = SS * NT * 2, SHARE = PIECES / NW;                   // KiB pieces of a slot's B; per wave
    static_assert(AUNITS % THREADS == 0 && PIECES % NW == 0 && (APITCH / 16) % 2 == 1 && 2 * ABUF + 2 * SLOTB <= 160 * 1024, "shape");
};

struct GArgs {
    const float* A;             // (M, lda) f32
    unsigned a_bytes;
    int lda;
    const unsigned char* pack;  // f16x2 pack of B: header | [k-step][32-column tile][hi, lo][lane][8 f16]
    unsigned pack_bytes;
    const float* bias;          // G_BIAS_RELU
    const unsigned* bits_in;    // G_MASKB: bit (element index of C) of the ReLU mask
    float* C;                   // (M, ldc) f32, ldc = N
    unsigned c_bytes;
    int M, N, K;
    int row_blocks, col_blocks, super_rows;
    const unsigned* a_amax;
    unsigned* c_amax;           // or null
};

// lane `l` of w := the wave-uniform value x; x where bit (lane) of {hi, lo} is set, else 0 (gemmz.hip's z_keep_where: the s_nop covers the
// two wait states a VALU read of an SGPR needs behind the VALU write the compiler cannot see inside the asm)
__device__ __forceinline__ float g_keep_where(float x, unsigned lo, unsigned hi) {
    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
    float r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
    return r;
}

template <int EPI>
__global__ __launch_bounds__(GGeom::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void g_kernel(GArgs a) {
    using GG = GGeom;
    constexpr int NTW = GG::NTW, NT = GG::NT, SS = GG::SS, NIA = GG::NIA, SHARE = GG::SHARE;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * GG::ABUF + 2 * GG::SLOTB];
    unsigned char* const ringa = lds;
    unsigned char* const ringb = lds + 2 * GG::ABUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int rw = wave & (GG::RW - 1), jg = wave / GG::RW;        // row-wave; first column tile / NTW (wave-uniform)

    const int ea = f16_scale_exp(amax_load(a.a_amax, lane));
    const int eb = f16_scale_exp(*reinterpret_cast<const unsigned*>(a.pack));
    const float sa = f16_pow2(ea), un = f16_unscale(ea, eb);

    const int ntiles = a.N >> 5, nslot = a.K / (16 * SS);
    const unsigned stepn = (unsigned)ntiles * 2048u;               // one k-step of the pack
    const unsigned lda4 = (unsigned)a.lda * 4u, ldc4 = (unsigned)a.N * 4u;

    // ---- this workgroup's blocks: XCD x (workgroups are dealt to the XCDs round robin) takes a contiguous range of the logical order
    const unsigned items = (unsigned)a.row_blocks * (unsigned)a.col_blocks;
    unsigned it_cur, it_stride, it_end;
    if ((gridDim.x & 7u) == 0u && items >= gridDim.x) {
        const unsigned x = blockIdx.x & 7u, q = items >> 3, rem = items & 7u;
        const unsigned s0 = x * q + (x < rem ? x : rem);
        it_end = s0 + q + (x < rem ? 1u : 0u);
        it_stride = gridDim.x >> 3;
        it_cur = s0 + (blockIdx.x >> 3);
    } else {
        it_cur = blockIdx.x; it_stride = gridDim.x; it_end = items;
    }
    // logical index -> (row block, column block): supertiles of super_rows row blocks x all column blocks, row block fastest
    auto decode = [&](unsigned l, int& rb, int& cb) __attribute__((always_inline)) {
        const unsigned per = (unsigned)a.super_rows * (unsigned)a.col_blocks;
        const unsigned sup = l / per, r = l - sup * per;
        const unsigned left = (unsigned)a.row_blocks - sup * (unsigned)a.super_rows;
        const unsigned rows = left < (unsigned)a.super_rows ? left : (unsigned)a.super_rows;
        cb = (int)(r / rows);
        rb = (int)(sup * (unsigned)a.super_rows + (r - (unsigned)cb * rows));
    };

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)a.a_bytes, kGRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.pack), 0, (int)a.pack_bytes, kGRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, EPI == G_BIAS_RELU ? a.N * 4 : 0, kGRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsrc_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.bits_in), 0, EPI == G_MASKB ? (int)(a.c_bytes >> 5) : 0, kGRsrcWord3);

    // A unit u = it * THREADS + tid of a slot: row u / (4 SS), 16-byte chunk u % (4 SS) of the row's SS * 64 bytes.  Rows past M re-read the last
    // row (kernel Z's clamp: their values are a real row's, their stores fall out of C's range).
    int arow[NIA], aq[NIA];
#pragma unroll
    for (int it = 0; it < NIA; ++it) {
        const int u = it * GG::THREADS + tid;
        arow[it] = u / (4 * SS);
        aq[it] = u - arow[it] * (4 * SS);
    }
    // this wave's B pieces of a slot: piece x = wave SHARE + u = (step h of the slot, tile j of the block, term t), x = (h NT + j) 2 + t
    const unsigned lane16 = 16u * (unsigned)lane;
    struct Block {
        unsigned aoff[NIA];          // byte offset of the thread's A units at slot 0
        unsigned bbase;              // pack offset of the block's first tile at k-step 0
        int t0, m0;                  // first column tile; first row
        bool ok;
    };
    auto describe = [&](unsigned l, Block& b) __attribute__((always_inline)) {
        b.ok = l < it_end;
        int rb = 0, cb = 0;
        if (b.ok) decode(l, rb, cb);
        b.m0 = rb * GG::ROWS;
        b.t0 = b.ok ? cb * NT : ntiles;          // (no block: every tile invalid -> its loads fall out of range)
        b.bbase = (unsigned)kF16PackHeader + (unsigned)(cb * NT) * 2048u;
#pragma unroll
        for (int it = 0; it < NIA; ++it) {
            const int r = b.m0 + arow[it];
            b.aoff[it] = b.ok ? (unsigned)(r < a.M ? r : a.M - 1) * lda4 + (unsigned)aq[it] * 16u : kGOob;
        }
    };
    Block cur, nxt;
    describe(it_cur, cur);
    describe(it_cur + it_stride, nxt);
    if (!cur.ok) return;                                  // (whole workgroup: before any barrier)

    s_u32x4 prea[2][NIA], preb[2][SHARE];                 // the two register sets: slot s travels in set s & 1
    // Slot s of the current block, or slot s - nslot of the next one (nslot >= 4: host-checked).  No branch around a load: behind one the compiler
    // loses count of the outstanding loads and every later wait becomes "all of them" -- the descriptors are selected instead.
    auto load_ahead = [&](int set, int s) __attribute__((always_inline)) {
        const bool nx = s >= nslot;
        const int slot = nx ? s - nslot : s;
        const unsigned bbase = nx ? nxt.bbase : cur.bbase;
        const int t0 = nx ? nxt.t0 : cur.t0;
#pragma unroll
        for (int it = 0; it < NIA; ++it)
            prea[set][it] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, nx ? nxt.aoff[it] : cur.aoff[it], slot * (SS * 64), MI355_AUX_STREAM_LD));
#pragma unroll
        for (int u = 0; u < SHARE; ++u) {
            const int x = wave * SHARE + u, h = x / (NT * 2), jt = x - h * (NT * 2);       // (wave-uniform: scalar arithmetic)
            const bool ok = t0 + (jt >> 1) < ntiles;
            preb[set][u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_p, ok ? lane16 : kGOob,
                                                                                              bbase + (unsigned)(slot * SS + h) * stepn + (unsigned)jt * 1024u, 0));
        }
    };
    auto write_slot = [&](int set, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NIA; ++it) {
            unsigned hi[2], lo[2];
            f16_split4(prea[set][it], sa, hi, lo);
            unsigned char* const d = ringa + buf * GG::ABUF + arow[it] * GG::APITCH + aq[it] * 8;
            *reinterpret_cast<uint2*>(d) = make_uint2(hi[0], hi[1]);
            *reinterpret_cast<uint2*>(d + GG::ALO) = make_uint2(lo[0], lo[1]);
        }
#pragma unroll
        for (int u = 0; u < SHARE; ++u) *reinterpret_cast<s_u32x4*>(ringb + buf * GG::SLOTB + (wave * SHARE + u) * 1024 + 16 * lane) = preb[set][u];
    };
    auto ring_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ---- fragments: lane (li, lh) reads row 32 rw + li of the block, k = 8 lh .. + 7 of the step
    const unsigned char* const fa = ringa + (32 * rw + li) * GG::APITCH + 16 * lh;
    const unsigned char* const fb = ringb + (jg * NTW) * 2048 + 16 * lane;
    s_u32x4 pa[2][2], wb[2][NTW][2];                      // [k-step parity]: A fragment [hi, lo]; B fragments [tile][hi, lo]
    auto read_ab = [&](int par, int buf, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) wb[par][j][t] = *reinterpret_cast<const s_u32x4*>(fb + buf * GG::SLOTB + ((h * NT + j) * 2 + t) * 1024);
        pa[par][0] = *reinterpret_cast<const s_u32x4*>(fa + buf * GG::ABUF + h * 32);
        pa[par][1] = *reinterpret_cast<const s_u32x4*>(fa + buf * GG::ABUF + h * 32 + GG::ALO);
    };
    g_f32x16 acc[NTW];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    };
    // hi hi, hi lo (B), lo hi (A): kernel Z's order of the three term pairs, tiles innermost
    auto mfmas = [&](int q) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pi = 0; pi < 3; ++pi)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, pa[q][pi == 2 ? 1 : 0]),
                                                                __builtin_bit_cast(s_f16x8, wb[q][j][pi == 1 ? 1 : 0]), acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    float cmax = 0.0f;
    unsigned wm[NTW];                                     // G_MASKB: lane L (< 32) holds the mask word of row 32 rw + L of the block, per tile
    float bj[NTW];

    // ---- prologue: slots 0 and 1 in the sets, slot 0 in buffer 0, slot 2 on its way, the operands of step 0 requested
    load_ahead(0, 0);
    load_ahead(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    write_slot(0, 0);
    load_ahead(0, 2);
    ring_barrier();
    read_ab(0, 0, 0);

    for (;;) {
        zero_acc();
        const int t0w = cur.t0 + jg * NTW;                // this wave's first tile
        // bias elements / mask words of the block, for its epilogue: requested here, unconditionally (tiles past N: out of the buffers' ranges)
        if constexpr (EPI == G_BIAS_RELU) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) bj[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_bias, (unsigned)(32 * (t0w + j) + li) * 4u, 0, 0));
        } else {
            const unsigned ro = (unsigned)(cur.m0 + 32 * rw + li) * ldc4;
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                wm[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_m, t0w + j < ntiles ? (ro + (unsigned)(32 * (t0w + j)) * 4u) >> 5 : kGOob, 0, 0);
        }
#pragma clang loop unroll(disable)
        for (int s = 0; s < nslot; s += 2) {
            // slot s (buffer 0): set 1 holds slot s + 1, set 0 slot s + 2
            read_ab(1, 0, 1);
            write_slot(1, 1);
            load_ahead(1, s + 3);
            mfmas(0);
            ring_barrier();                               // slot s + 1 has landed in buffer 1
            read_ab(0, 1, 0);
            mfmas(1);
            // slot s + 1 (buffer 1): set 0 holds slot s + 2, set 1 slot s + 3
            read_ab(1, 1, 1);
            write_slot(0, 0);
            load_ahead(0, s + 4);
            mfmas(0);
            ring_barrier();                               // slot s + 2 (or the next block's slot 0) has landed in buffer 0
            read_ab(0, 0, 0);
            mfmas(1);
        }
        // ---- epilogue (kernel Z's): accumulator e of tile j = row 32 rw + (e & 3) + 8 (e >> 2) + 4 lh of the block, column 32 (t0w + j) + li.
        // Rows past M fall out of C's range (stores dropped, mask words read as zero); tiles past N get an empty range.
        const unsigned rbase = (unsigned)(cur.m0 + 32 * rw + 4 * lh) * ldc4 + 4u * (unsigned)li;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, t0w + j < ntiles ? (int)a.c_bytes : 0, kGRsrcWord3);
            const unsigned cbase = rbase + (unsigned)(32 * (t0w + j)) * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned off = cbase + (unsigned)((e & 3) + 8 * (e >> 2)) * ldc4;
                float v;
                if constexpr (EPI == G_MASKB) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)wm[j], (e & 3) + 8 * (e >> 2));
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)wm[j], (e & 3) + 8 * (e >> 2) + 4);
                    v = g_keep_where(acc[j][e] * un, lo, hi);                // lanes 0..31 (lh = 0): bit li of `lo`, lanes 32..63: of `hi`
                } else {
                    v = acc[j][e] * un + bj[j];
                    v = v < 0.0f ? 0.0f : v;                                  // (a NaN stays a NaN, as kernel Z's SPLIT epilogue)
                }
                cmax = __builtin_fmaxf(cmax, __builtin_fabsf(v));             // (tiles past N: zeros)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_c, off, 0, MI355_AUX_STREAM_ST);
            }
        }
        if (!nxt.ok) break;
        cur = nxt;
        it_cur += it_stride;
        describe(it_cur + it_stride, nxt);
    }
    if (a.c_amax) amax_commit(a.c_amax, __float_as_uint(cmax), blockIdx.x * GG::NW + (unsigned)wave, lane);
}

// MI355PPO_FC_G=0: the FC forward / data gradient stay on kernel Z (A/B runs; the results are bit-identical either way); =min:<n>: from n rows on
// (common.h::kernel_switch).
bool gemmg_on(long long rows, long long min_rows) { return kernel_switch("MI355PPO_FC_G", rows, min_rows); }

// -> 0 launched, 1 not applicable (shape), < 0 error
int gemmg_launch(const char* fn, int epi, const float* A, int lda, const void* pack, const float* bias, const unsigned* bits, float* C, int M, int N,
                 int K, const unsigned* a_amax, unsigned* c_amax, hipStream_t s) {
    using GG = GGeom;
    if (!a_amax || M <= 0 || N % 32 || K % (32 * GG::SS) || K < 64 * GG::SS) return 1;
    const long long a_bytes = (long long)M * lda * 4, c_bytes = ((long long)M + GG::ROWS) * N * 4;
    if (a_bytes >= (1LL << 32) - 8192 || c_bytes >= (1LL << 32) - 8192) return 1;
    GArgs a{};
    a.A = A; a.a_bytes = (unsigned)a_bytes; a.lda = lda; a.pack = static_cast<const unsigned char*>(pack);
    a.pack_bytes = (unsigned)(kF16PackHeader + (size_t)(K / 16) * (size_t)(N / 32) * 2048);
    a.bias = bias; a.bits_in = bits; a.C = C; a.c_bytes = (unsigned)((long long)M * N * 4); a.M = M; a.N = N; a.K = K;
    a.row_blocks = (M + GG::ROWS - 1) / GG::ROWS;
    a.col_blocks = (N + GG::COLS - 1) / GG::COLS;
    // supertiles: forward (two column blocks) 16 row blocks -- the 32 blocks in flight on an XCD are both halves of 16 row blocks; data
    // gradient (13 column blocks) 8 row blocks -- 2 MB of A stay in the L2 while the pack streams through once per supertile
    a.super_rows = a.col_blocks <= 2 ? 16 : 8;
    a.a_amax = a_amax; a.c_amax = c_amax;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        cus = n;
    }
    const long long items = (long long)a.row_blocks * a.col_blocks;
    const int grid = items < cus ? (int)items : cus;
    if (epi == G_MASKB) hipLaunchKernelGGL((g_kernel<G_MASKB>), dim3((unsigned)grid), dim3(GG::THREADS), 0, s, a);
    else hipLaunchKernelGGL((g_kernel<G_BIAS_RELU>), dim3((unsigned)grid), dim3(GG::THREADS), 0, s, a);
    return check_launch(fn);
}

}  // namespace mi355ppo
