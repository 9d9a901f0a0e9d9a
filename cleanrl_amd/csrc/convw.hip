// Kernel V -- weight (+ bias) gradient of Conv2d(32,64,4,stride 2) and Conv2d(64,64,3,stride 1) of the NatureCNN
// (cleanrl/ppo_atari_multigpu.py:139-142; autograd's conv2d weight gradient) on the bf16 matrix pipe, gfx950:
//     dW[co][(ty, tx, ci)] = sum over output pixels p of dz[p][co] * src[pixel of p shifted by tap (ty, tx)][ci]
// the six largest term pairs of the three-term bf16 split of both operands (bf16split.h; MI355PPO_BF16_PAIRS=9: all nine),
// f32 accumulation -- kernel W's scheme (fcw.hip) with a convolution's B operand.
//
// The reduction index (the output pixel) is the slow index of both operands in memory: they are transposed on the way in.
// Per k-step (16 consecutive output pixels, image boundaries included) a wave loads, COALESCED,
//   * its [16 p][64 co] block of dz -- 4 KiB contiguous,
//   * for each of its NT tiles (one tap x 32 input channels) the 16 source pixels of that tap: 128 contiguous bytes per pixel,
// writes them to a wave-private LDS tile as they are and reads fragments back TRANSPOSED (lane (li, lh): column li of its 32,
// rows 8 lh .. 8 lh + 7, eight 4-byte LDS reads), then splits them in registers.  Wave-private, double-buffered LDS: no
// workgroup barrier.  A wave owns 64 (co) x 32 NT columns of dW: layer 2 NT = 4 (four taps; the 16 taps are four wave
// groups), layer 3 NT = 3 (18 tiles = 9 taps x two channel halves: six groups).  The pixels are dealt to S slabs in 16-pixel
// blocks (slab s: blocks s, s + S, ...), a "unit" = (slab, group) = one wave; units are dealt to 4-wave workgroups in order
// (the four waves of a workgroup mostly share a slab, i.e. the same dz and overlapping source pixels in L1).  Every slab
// writes one partial dW (its groups disjoint column ranges) and one partial db (group 0: column sums of the dz fragments it
// reads anyway); conv.hip's two-stage reduce adds the partials in a fixed order.
// Source-pixel addresses: a lane owns two of the 16 pixels (p0 + lane / 8 and + 8); their (image, pixel-in-image) pairs are
// carried from step to step by an add-with-carry, the divisions by the row length are multiplies.
#include "common.h"
#include "bf16split.h"
#include "f16split.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace mi355ppo {

typedef float v_f32x16 __attribute__((ext_vector_type(16)));
constexpr int kVRsrcWord3 = 0x00020000;

template <int H_, int W_, int C_, int KH_, int KW_, int OH_, int OW_, int S_>
struct VGeom {
    static constexpr int H = H_, W = W_, C = C_, KH = KH_, KW = KW_, OH = OH_, OW = OW_, SS = S_;
    static constexpr int PER_IMG = OH * OW, K = KH * KW * C, TILES = K / 32, CT = C / 32;       // CT: 32-channel tiles per tap
};
using VGeom2 = VGeom<20, 20, 32, 4, 4, 9, 9, 2>;
using VGeom3 = VGeom<9, 9, 64, 3, 3, 7, 7, 1>;
constexpr int kVCout = 64;

// schedule of a k-step: NITEMS items dealt evenly behind the step's MFMAs.  kinds: 0 = read fragment `arg` from LDS, 1 = split
// piece `arg` = 6 * fragment + piece, 2 = LDS writes (third `arg`), 3 = global loads (third `arg`), 4 = source addresses of
// the block three steps ahead.  Order: R0 R1 | per fragment f: S_f.0-2, X_f, S_f.3-5, R_{f+2} | the X items left over, with
// X = W0 W1 W2 ADDR L0 L1 L2.
// ppf = split pieces per fragment: 6 (bf16) or 2 (f16, one per half fragment: f16split.h) -- S_f.0 .. S_f.(ppf/2 - 1), X_f, the rest, R_{f+2}
struct VItem { int kind, arg; };
constexpr VItem v_xitem(int x) { return x < 3 ? VItem{2, x} : x == 3 ? VItem{4, 0} : VItem{3, x - 4}; }
constexpr int v_nitems(int nf, int ppf) { return (ppf + 1) * nf + 7; }
constexpr VItem v_item(int t, int nf, int ppf) {
    if (t < 2) return {0, t};
    int u = t - 2;
    for (int f = 0; f < nf; ++f) {
        const int len = ppf + 1 + (f + 2 < nf ? 1 : 0);
        if (u < len) {
            if (u < ppf / 2) return {1, ppf * f + u};
            if (u == ppf / 2) return v_xitem(f);
            if (u < ppf + 1) return {1, ppf * f + u - 1};
            return {0, f + 2};
        }
        u -= len;
    }
    return u < 7 - nf ? v_xitem(nf + u) : VItem{-1, 0};
}

// SPLIT: 0 = three bf16 terms, NP = 6 / 9 pairs; 1 = two f16 terms under the operands' power-of-two scales (amax records of dz and
// src), NP = 3 pairs, the partial dW un-scaled on its way out (f16split.h).  The bias gradient sums the dz fragments as read (f32).
// OCCV: waves per SIMD the kernel is compiled for.  1: NT = 4 / 3 tiles per wave (128 / 96 accumulator registers, round 3).  2 (SPLIT = 1
// only): NT = 2 -- 64 accumulators, two waves per SIMD from two workgroups per CU: with half the matrix instructions per k-step the single
// wave per SIMD was bound by its own instruction issue (profiles/r05_pmc_busy.csv: instruction-active 0.51 of the wave's time, matrix pipe
// 0.28 busy), and a second wave's VALU / LDS work fills the first one's matrix-pipe time.
template <class G, int NT, int NP, int SPLIT, int OCCV = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCCV, OCCV))) void convw_bf16_kernel(
    const float* __restrict__ src, const float* __restrict__ dz, float* __restrict__ part_w, float* __restrict__ part_b, int images,
    int nslabs, unsigned m8, unsigned m16, int xcd_order, const unsigned* __restrict__ dz_amax, const unsigned* __restrict__ src_amax) {
    constexpr int MT = 2, NF = MT + NT, NGROUPS = G::TILES / NT, NL = 4 + 2 * NT;       // fragments; wave groups; loads per block
    constexpr int TERMS = SPLIT ? 2 : 3, PPF = SPLIT ? 2 : 6;
    constexpr int LDSF = 16 * (kVCout + 32 * NT);                                       // floats of one block in LDS
    static_assert(G::TILES % NT == 0, "whole groups of NT tiles");
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * LDSF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    float sz = 1.0f, ss = 1.0f, un = 1.0f;                // SPLIT: the operands' scales and the factor that removes both from the partial
    if constexpr (SPLIT) {                                // (before the early exit below: the wave reduction wants every lane)
        const int ez = f16_scale_exp(amax_load(dz_amax, lane)), es = f16_scale_exp(amax_load(src_amax, lane));
        sz = f16_pow2(ez);
        ss = f16_pow2(es);
        un = f16_unscale(ez, es);
    }
    // Workgroups are dealt to the eight XCDs round robin in launch order; neighbours in the UNIT order share data -- the six wave
    // groups of a layer-3 slab sit in two consecutive workgroups and read the same dz and source pixels, consecutive slabs take
    // adjacent 16-pixel blocks whose windows overlap -- so XCD x takes a contiguous range of the unit order (kernel Z's scheme):
    // what neighbours share is fetched into ONE L2 (layer 3 fetched 1.9 x its algorithmic bytes in launch order, profiles/traffic.json).
    unsigned wg = blockIdx.x;
    if (xcd_order) {
        const unsigned total = gridDim.x, x = wg & 7u, q = total >> 3, rem = total & 7u;
        wg = x * q + (x < rem ? x : rem) + (wg >> 3);
    }
    const int unit = (int)wg * 4 + wave;
    if (unit >= nslabs * NGROUPS) return;                 // (whole wave; no barriers in this kernel)
    const int slab = unit / NGROUPS, grp = unit - slab * NGROUPS;
    const long long P = (long long)images * G::PER_IMG;
    const int nblocks = (int)(P / 16);
    const int nsteps = (nblocks - slab + nslabs - 1) / nslabs;
    float* const wl = lds + wave * (2 * LDSF);
    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, (int)(unsigned)(P * kVCout * 4), kVRsrcWord3);
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(unsigned)((long long)images * G::H * G::W * G::C * 4), kVRsrcWord3);
    // dz block: 4 KiB contiguous, load u covers bytes 1024 u + 16 lane.  Source tile j of the group: tap and channel half of
    // global tile grp * NT + j; its load h covers pixels (lane >> 3) + 8 h of the block, bytes 16 (lane & 7) of the 128.
    unsigned tap_off[NT];                                 // byte offset of tile j's tap (and channel half) from a pixel's window origin
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int T = grp * NT + j, tap = T / G::CT, ty = tap / G::KW, tx = tap - ty * G::KW;
        tap_off[j] = (unsigned)(((ty * G::W + tx) * G::C + 32 * (T % G::CT)) * 4) + 16u * (unsigned)(lane & 7);
    }
    // the lane's two pixels: (image, pixel in image), advanced by 16 * nslabs pixels per step
    const int adv = 16 * nslabs, adv_img = adv / G::PER_IMG, adv_rem = adv - adv_img * G::PER_IMG;
    int pimg[2], prem[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long p = (long long)slab * 16 + (lane >> 3) + 8 * h;
        pimg[h] = (int)(p / G::PER_IMG);
        prem[h] = (int)(p - (long long)pimg[h] * G::PER_IMG);
    }
    auto pix_off = [&](int h) -> unsigned {               // byte offset of the window origin of the lane's pixel h
        const int oy = prem[h] / G::OW, ox = prem[h] - oy * G::OW;
        return (unsigned)(((pimg[h] * G::H + oy * G::SS) * G::W + ox * G::SS) * G::C * 4);
    };
    auto advance = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            prem[h] += adv_rem;
            pimg[h] += adv_img;
            if (prem[h] >= G::PER_IMG) {
                prem[h] -= G::PER_IMG;
                pimg[h] += 1;
            }
        }
    };
    unsigned po[2];                                       // window origins of the block about to be loaded
    float* const wr_dz = wl + (lane >> 4) * kVCout + 4 * (lane & 15);                  // + 4 u rows (u < 4)
    float* const wr_src = wl + 16 * kVCout + (lane >> 3) * (32 * NT) + 4 * (lane & 7);   // + 8 h rows, + 32 j columns
    const float* const rd_dz = wl + (8 * lh) * kVCout + li;                            // + e rows, + 32 i columns
    const float* const rd_src = wl + 16 * kVCout + (8 * lh) * (32 * NT) + li;          // + e rows, + 32 j columns

    v_f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    float bsum[MT] = {0.0f, 0.0f}, pend[MT] = {0.0f, 0.0f};   // bias gradient: column sums of dz; `pend` = of the fragments split last

    s_u32x4 stage[NL];                                    // one block as loaded: 4 x dz, 2 per source tile
    unsigned raw[2][8];                                   // two fragments as read back from LDS (f32, lane = column, 8 rows)
    unsigned tt[2][NF][TERMS][4];                         // split fragments: [k-step parity][fragment: MT x dz, NT x source][term][4 x 2 bf16 / f16]
    auto sclamp = [&](int s) { return s < nsteps ? s : nsteps - 1; };               // past the end: re-read, never multiplied
    constexpr int TH = (NL + 2) / 3;                      // loads / LDS writes per third
    auto load_third = [&](int s, auto tc) {               // third tc of the loads of step s's block (source origins in `po`)
        constexpr int t3 = decltype(tc)::value;
        const unsigned so_dz = (unsigned)(slab + sclamp(s) * nslabs) * 4096u;
#pragma unroll
        for (int u = TH * t3; u < TH * t3 + TH && u < NL; ++u) {
            if (u < 4) stage[u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dz, 16u * (unsigned)lane + 1024u * (unsigned)u, so_dz, 0));
            else stage[u] = __builtin_bit_cast(s_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, po[(u - 4) & 1] + tap_off[(u - 4) >> 1], 0, 0));
        }
    };
    auto write_third = [&](int buf, auto tc) {
        constexpr int t3 = decltype(tc)::value;
#pragma unroll
        for (int u = TH * t3; u < TH * t3 + TH && u < NL; ++u) {
            if (u < 4) *reinterpret_cast<s_u32x4*>(wr_dz + buf * LDSF + 4 * u * kVCout) = stage[u];
            else *reinterpret_cast<s_u32x4*>(wr_src + buf * LDSF + 8 * ((u - 4) & 1) * (32 * NT) + 32 * ((u - 4) >> 1)) = stage[u];
        }
    };
    auto read_frag = [&](int buf, auto fc) {              // fragment f (< MT: dz tile, else source tile f - MT) of buffer `buf` -> raw[f & 1]
        constexpr int f = decltype(fc)::value;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            raw[f & 1][e] = f < MT ? reinterpret_cast<const unsigned*>(rd_dz)[buf * LDSF + e * kVCout + 32 * f]
                                   : reinterpret_cast<const unsigned*>(rd_src)[buf * LDSF + e * (32 * NT) + 32 * (f - MT)];
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
            if constexpr (f < MT) {                        // bias gradient: column sums of the dz fragment (group 0's partial is stored)
                const float h4 = (__uint_as_float(raw[f & 1][4 * hf]) + __uint_as_float(raw[f & 1][4 * hf + 1])) +
                                 (__uint_as_float(raw[f & 1][4 * hf + 2]) + __uint_as_float(raw[f & 1][4 * hf + 3]));
                pend[f] = hf == 0 ? h4 : pend[f] + h4;
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
    auto split_piece_h = [&](int par, auto fc, auto hc) { // SPLIT = 1: half hf (4 elements) of fragment f, 10 VALU (+ the bias gradient's 3 adds)
        constexpr int f = decltype(fc)::value, hf = decltype(hc)::value;
        if constexpr (f < MT) {
            const float h4 = (__uint_as_float(raw[f & 1][4 * hf]) + __uint_as_float(raw[f & 1][4 * hf + 1])) +
                             (__uint_as_float(raw[f & 1][4 * hf + 2]) + __uint_as_float(raw[f & 1][4 * hf + 3]));
            pend[f] = hf == 0 ? h4 : pend[f] + h4;
        }
        unsigned hi[2], lo[2];
        f16_split4((s_u32x4){raw[f & 1][4 * hf], raw[f & 1][4 * hf + 1], raw[f & 1][4 * hf + 2], raw[f & 1][4 * hf + 3]}, f < MT ? sz : ss, hi, lo);
        tt[par][f][0][2 * hf] = hi[0]; tt[par][f][0][2 * hf + 1] = hi[1];
        tt[par][f][TERMS - 1][2 * hf] = lo[0]; tt[par][f][TERMS - 1][2 * hf + 1] = lo[1];
    };
    auto frag_bits = [&](int par, int f, int term) { return (s_u32x4){tt[par][f][term][0], tt[par][f][term][1], tt[par][f][term][2], tt[par][f][term][3]}; };
    auto next_origins = [&]() {                           // `po` <- window origins of the NEXT block of this slab (then advance)
        po[0] = pix_off(0);
        po[1] = pix_off(1);
        advance();
    };
    constexpr int NM = NP * MT * NT, NI = v_nitems(NF, PPF);
    static_assert(SPLIT ? (NP == 3 || NP == 4) : (NP == 6 || NP == 9), "term pairs of the split");
    constexpr int PX[9] = {0, 0, 1, SPLIT ? 1 : 0, 2, 1, 1, 2, 2}, PY[9] = {0, 1, 0, SPLIT ? 1 : 2, 0, 1, 2, 1, 2};
    // One pipeline step of parity q (kernel W's): the MFMAs of step s on tt[q]; meanwhile the fragments of step s + 1 are read
    // from LDS buffer q ^ 1 and split into tt[q ^ 1]; `stage` (the block of step s + 2) goes to LDS buffer q; the block of step
    // s + 3 is loaded into `stage`.  Items t with t * NM / NI == g sit behind MFMA g, each followed by a sched_barrier.
    auto run_item = [&](auto qc, int s, auto tc) {
        constexpr int q = decltype(qc)::value;
        constexpr VItem it = v_item(decltype(tc)::value, NF, PPF);
        if constexpr (it.kind == 0) read_frag(q ^ 1, std::integral_constant<int, it.arg>{});
        else if constexpr (it.kind == 1 && SPLIT) split_piece_h(q ^ 1, std::integral_constant<int, it.arg / 2>{}, std::integral_constant<int, it.arg % 2>{});
        else if constexpr (it.kind == 1) split_piece(q ^ 1, std::integral_constant<int, it.arg / 6>{}, std::integral_constant<int, it.arg % 6>{});
        else if constexpr (it.kind == 2) write_third(q, std::integral_constant<int, it.arg>{});
        else if constexpr (it.kind == 3) load_third(s + 3, std::integral_constant<int, it.arg>{});
        else if constexpr (it.kind == 4) next_origins();
    };
    auto step = [&](auto qc, int s) {
        constexpr int q = decltype(qc)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i) bsum[i] += pend[i];  // the dz fragments split during the previous step belong to THIS step: it exists
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... GI>(std::integer_sequence<int, GI...>) {
            ([&] {
                constexpr int g = GI, pi = g / (MT * NT), i = (g / NT) % MT, j = g % NT;
                if constexpr (SPLIT)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s_f16x8, frag_bits(q, i, PX[pi])),
                                                                       __builtin_bit_cast(s_f16x8, frag_bits(q, MT + j, PY[pi])), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s_bf16x8, frag_bits(q, i, PX[pi])),
                                                                        __builtin_bit_cast(s_bf16x8, frag_bits(q, MT + j, PY[pi])), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int lo = (g * NI + NM - 1) / NM, hi = ((g + 1) * NI + NM - 1) / NM;      // items t with t * NM / NI == g
                static_assert(hi - lo <= 2, "at most two items behind one MFMA");
                if constexpr (lo < hi && lo < NI) {
                    run_item(qc, s, std::integral_constant<int, lo>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (lo + 1 < hi && lo + 1 < NI) {
                    run_item(qc, s, std::integral_constant<int, lo + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }(), ...);
        }(std::make_integer_sequence<int, NM>{});
    };
    auto load_all = [&](int s) { load_third(s, std::integral_constant<int, 0>{}); load_third(s, std::integral_constant<int, 1>{}); load_third(s, std::integral_constant<int, 2>{}); };
    auto write_all = [&](int buf) { write_third(buf, std::integral_constant<int, 0>{}); write_third(buf, std::integral_constant<int, 1>{}); write_third(buf, std::integral_constant<int, 2>{}); };
    auto split_all = [&](int par, int buf) {              // all fragments of LDS buffer `buf` -> tt[par] (prologue; ONE pack expansion, see fcw.hip)
        [&]<int... T>(std::integer_sequence<int, T...>) {
            ([&] {
                if constexpr (T % PPF == 0) read_frag(buf, std::integral_constant<int, T / PPF>{});
                if constexpr (SPLIT) split_piece_h(par, std::integral_constant<int, T / 2>{}, std::integral_constant<int, T % 2>{});
                else split_piece(par, std::integral_constant<int, T / 6>{}, std::integral_constant<int, T % 6>{});
            }(), ...);
        }(std::make_integer_sequence<int, PPF * NF>{});
    };
    if (nsteps > 0) {
        // prologue: step 0 -> LDS buffer 0 -> tt[0]; step 1 -> LDS buffer 1; step 2 in `stage`; origins of step 3 follow in step 0
        next_origins();
        load_all(0);
        write_all(0);
        next_origins();
        load_all(1);
        split_all(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        write_all(1);
        next_origins();
        load_all(2);
        __builtin_amdgcn_sched_barrier(0);
        int s = 0;
        for (; s + 2 <= nsteps; s += 2) {
            step(std::integral_constant<int, 0>{}, s);
            step(std::integral_constant<int, 1>{}, s + 1);
        }
        if (s < nsteps) step(std::integral_constant<int, 0>{}, s);
    }
    // partial [slab][co][K]: accumulator element e of tile (i, j) is row co = 32 i + (e & 3) + 8 (e >> 2) + 4 lh, column (tap, ci) of tile j
    float* out = part_w + (size_t)slab * kVCout * G::K;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int T = grp * NT + j, kcol = (T / G::CT) * G::C + 32 * (T % G::CT) + li;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) out[(size_t)(32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh) * G::K + kcol] = SPLIT ? acc[i][j][e] * un : acc[i][j][e];
    }
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float both = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            if (lh == 0) part_b[(size_t)slab * kVCout + 32 * i + li] = both;
        }
    }
}

// f16 split: two tiles per wave, two waves per SIMD -- 8 / 9 wave groups: layer 3 740 -> 606 us, layer 2 872 -> 850 us at 32,768 images, layer 2
// bit-identical (same slabs), profiles/r05_tile_shape_experiments.txt.
static bool convw_nt2() { return true; }
static int convw_slabs(int layer, bool nt2 = false) {      // x 4 / x 6 wave groups = 1,024 / 1,020 waves (nt2: x 8 / x 9 = 2,048 / 2,043)
    if (nt2) return layer == 2 ? 256 : 227;
    return layer == 2 ? 256 : 170;
}

// Kernel V takes a batch of layer 2 / 3 that is a multiple of 16 images, large enough for every slab to have work, with
// tensors inside the 32-bit buffer range.
bool convw_applies(int64_t images, int layer, bool f16) {
    if ((layer != 2 && layer != 3) || images <= 0) return false;
    const long long P = images * (layer == 2 ? 81 : 49);
    // (the slab count of the shape that would run: the f16 split's two-tile shape has more slabs -- with it asked for the bf16 path too, layer-3
    //  batches of 224 .. 288 images fell to kernel T without reason)
    return images % 16 == 0 && P / 16 >= 4LL * convw_slabs(layer, f16 && convw_nt2()) &&
           images * (layer == 2 ? 20 * 20 * 32 : 9 * 9 * 64) * 4 < (1LL << 32) - 8192 && P * 64 * 4 < (1LL << 32) - 8192;
}

// partials kernel V writes for this batch (0: it does not take it) -- the workspace of mi355ppo_cnn_conv_wgrad_* must hold that many
int convw_parts(int64_t images, int layer) { return convw_applies(images, layer) ? convw_slabs(layer, convw_nt2()) : 0; }      // (the larger count: sizes the workspace)

// Launches kernel V if the batch qualifies; *nparts = partials written (part_w [nparts][64 * K], part_b [nparts][64]).
// Returns 1 if it does not apply.
int convw_launch(const float* src, const float* dz, float* part_w, float* part_b, int64_t images, int layer, int* nparts, hipStream_t s,
                 const unsigned* dz_amax, const unsigned* src_amax) {
    if (!convw_applies(images, layer, dz_amax != nullptr)) return 1;
    const bool nt2 = dz_amax && convw_nt2();
    const int S = convw_slabs(layer, nt2);
    *nparts = S;
    const int np = bf16_term_pairs();
    const int xcd = 1;      // XCD-contiguous unit order: L2-miss reads 3.20 -> 2.38 GB (layer 2), 2.07 -> 1.12 GB (layer 3), times -0 .. 1.5 %
                            // (same-box A/B with a run-time switch, profiles/r03_raster_ab.jsonl, r03_pmc_fetch_raster_{before,after}.csv)
    if (layer == 2) {
        const int grid = (S * (VGeom2::TILES / 4) + 3) / 4;
        if (nt2) hipLaunchKernelGGL((convw_bf16_kernel<VGeom2, 2, 3, 1, 2>), dim3((S * (VGeom2::TILES / 2) + 3) / 4), dim3(256), 0, s, src, dz, part_w, part_b, (int)images, S, 0xffff0000u, 0xffffff00u, xcd, dz_amax, src_amax);
        else if (dz_amax) hipLaunchKernelGGL((convw_bf16_kernel<VGeom2, 4, 3, 1>), dim3(grid), dim3(256), 0, s, src, dz, part_w, part_b, (int)images, S, 0xffff0000u, 0xffffff00u, xcd, dz_amax, src_amax);
        else if (np == 9) hipLaunchKernelGGL((convw_bf16_kernel<VGeom2, 4, 9, 0>), dim3(grid), dim3(256), 0, s, src, dz, part_w, part_b, (int)images, S, 0xffff0000u, 0xffffff00u, xcd, dz_amax, src_amax);
        else hipLaunchKernelGGL((convw_bf16_kernel<VGeom2, 4, 6, 0>), dim3(grid), dim3(256), 0, s, src, dz, part_w, part_b, (int)images, S, 0xffff0000u, 0xffffff00u, xcd, dz_amax, src_amax);
    } else {
        const int grid = (S * (VGeom3::TILES / 3) + 3) / 4;
        if (nt2) hipLaunchKernelGGL((convw_bf16_kernel<VGeom3, 2, 3, 1, 2>), dim3((S * (VGeom3::TILES / 2) + 3) / 4), dim3(256), 0, s, src, dz, part_w, part_b, (int)images, S, 0xffff0000u, 0xffffff00u, xcd, dz_amax, src_amax);
        else if (dz_amax) hipLaunchKernelGGL((convw_bf16_kernel<VGeom3, 3, 3, 1>), dim3(grid), dim3(256), 0, s, src, dz, part_w, part_b, (int)images, S, 0xffff0000u, 0xffffff00u, xcd, dz_amax, src_amax);
        else if (np == 9) hipLaunchKernelGGL((convw_bf16_kernel<VGeom3, 3, 9, 0>), dim3(grid), dim3(256), 0, s, src, dz, part_w, part_b, (int)images, S, 0xffff0000u, 0xffffff00u, xcd, dz_amax, src_amax);
        else hipLaunchKernelGGL((convw_bf16_kernel<VGeom3, 3, 6, 0>), dim3(grid), dim3(256), 0, s, src, dz, part_w, part_b, (int)images, S, 0xffff0000u, 0xffffff00u, xcd, dz_amax, src_amax);
    }
    return check_launch("convw_bf16_kernel");
}

}  // namespace mi355ppo
