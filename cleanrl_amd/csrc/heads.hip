// Policy / value heads of the NatureCNN agent: actor = Linear(512, A), critic = Linear(512, 1)
// (cleanrl/ppo_atari_multigpu.py:148-149,157-159), forward and backward, as two bandwidth-bound kernels (gfx950).
//
// As GEMMs these are degenerate (N = A + 1 <= 8 columns forward, K = A + 1 backward): hipBLASLt spends ~100 us forward and
// ~250 us backward per 32,768-row minibatch on six launches.  What they really are is one pass over the hidden
// activations h (M x 512 f32 = 67 MB at M = 32,768):
//   forward : logits[m][a] = h[m] . Wa[a] + ba[a],  value[m] = h[m] . Wc + bc      (read h once)
//   backward: dh[m][j] = sum_a dlogits[m][a] Wa[a][j] + dvalue[m] Wc[j]            (read h once, write dh once)
//             dWa[a][j] = sum_m dlogits[m][a] h[m][j],  dWc[j] = sum_m dvalue[m] h[m][j],  db = column sums
// A wave owns a row at a time: lane l holds h[m][8l .. 8l+7] (two float4: the 2 KB row is one contiguous read) and
// the matching slices of all NA = A + 1 weight rows in registers.  The weight-gradient accumulators (NA x 8 per lane)
// stay in registers over all rows a wave visits; waves -> workgroup (LDS, wave order) -> grid partials are folded in a
// fixed order (deterministic).  Algorithmic bytes: forward 2048*M + 4*NA*M, backward 4096*M + 4*NA*M (+ partials).
#include "common.h"
#include "f16split.h"
#include "catrow.h"

#pragma clang fp contract(off)

namespace mi355ppo {

constexpr int kHid = 512;
constexpr int kHeadsBlocks = 512;       // persistent workgroups of 4 waves (2 per CU)
// Round 6: heads wider than 7 actions (Linear(512, envs.single_action_space.n), ppo_atari_multigpu.py:148: ALE games have up to 18 actions).
// NA = A + 1 weight rows of 8 floats per lane fit the register file up to NA = 8 (the backward also keeps NA x 8 accumulators); from there on
// the kernels are compiled ONCE for kWideNA = 19 rows with the live count a run-time (wave-uniform) argument, and the weights sit in LDS
// (19 x 2 KB; lane l reads its 32 bytes of a row with two ds_read_b128, conflict-free) instead of registers: same arithmetic, same order.
constexpr int kWideNA = 19;

template <int NA>
__device__ __forceinline__ void load_w(float (&w)[NA][8], const float* __restrict__ Wa, const float* __restrict__ Wc, int A,
                                       int lane) {
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        if (a < A) {
            const float* row = Wa + (size_t)a * kHid;
            const float4 x = *reinterpret_cast<const float4*>(row + lane * 8), y = *reinterpret_cast<const float4*>(row + lane * 8 + 4);
            w[a][0] = x.x; w[a][1] = x.y; w[a][2] = x.z; w[a][3] = x.w; w[a][4] = y.x; w[a][5] = y.y; w[a][6] = y.z; w[a][7] = y.w;
        } else {
            // the critic's weight row: 4-byte loads.  In the flat parameter buffer (the reference's cat order, ppo_atari_multigpu.py:360) it
            // follows actor.bias -- A floats --, so it sits on a 16-byte boundary only when A % 4 == 0 (Breakout's 4 actions; not Pong's 6).
#pragma unroll
            for (int i = 0; i < 8; ++i) w[a][i] = Wc[lane * 8 + i];
        }
    }
}

// The weight rows of a lane: registers (WIDE = false: all NA rows live) or LDS (WIDE: `na` of kWideNA rows live, rows >= na never touched).
template <int NA, bool WIDE>
struct HeadW {
    float w[WIDE ? 1 : NA][8];
    const float* l;                       // WIDE: &lds[0][lane * 8]
    __device__ __forceinline__ void load(float* lds, const float* __restrict__ Wa, const float* __restrict__ Wc, int A, int lane) {
        if constexpr (WIDE) {
            for (int e = threadIdx.x; e < A * (kHid / 4); e += blockDim.x) {
                const int a = e / (kHid / 4), q = e - a * (kHid / 4);
                reinterpret_cast<float4*>(lds + a * kHid)[q] = reinterpret_cast<const float4*>(Wa + (size_t)a * kHid)[q];
            }
            for (int e = threadIdx.x; e < kHid; e += blockDim.x) lds[A * kHid + e] = Wc[e];       // (the critic's row: 4-byte aligned only, see load_w)
            __syncthreads();
            l = lds + lane * 8;
        } else {
            load_w<NA>(w, Wa, Wc, A, lane);
            l = nullptr;
        }
    }
    __device__ __forceinline__ void row(int a, float (&out)[8]) const {
        if constexpr (WIDE) {
            const float4 x = *reinterpret_cast<const float4*>(l + a * kHid), y = *reinterpret_cast<const float4*>(l + a * kHid + 4);
            out[0] = x.x; out[1] = x.y; out[2] = x.z; out[3] = x.w; out[4] = y.x; out[5] = y.y; out[6] = y.z; out[7] = y.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) out[i] = w[a][i];
        }
    }
};

template <int NA, bool WIDE = false>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ h, const float* __restrict__ Wa,
                                                        const float* __restrict__ ba, const float* __restrict__ Wc,
                                                        const float* __restrict__ bc, float* __restrict__ logits,
                                                        float* __restrict__ value, int M, int A) {
    __shared__ __attribute__((aligned(16))) float wl[WIDE ? kWideNA * kHid : 4];
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
    HeadW<NA, WIDE> W;
    W.load(wl, Wa, Wc, A, lane);
    float bias = 0.0f;
    if (lane < A) bias = ba[lane];
    else if (lane == A) bias = bc[0];
    auto row = [&](int m, const float4& x, const float4& y) {
        const float hv[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        float out = 0.0f;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (!WIDE || a <= A) {                          // (wave-uniform; the unrolled loop keeps every row's registers static)
                float wa[8];
                W.row(a, wa);
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += hv[i] * wa[i];
                s = wave_sum(s);
                if (lane == a) out = s;
            }
        }
        out = out + bias;
        if (lane < A) logits[(size_t)m * A + lane] = out;
        else if (lane == A) value[m] = out;
    };
    // Two rows per iteration, and the NEXT iteration's four 16-byte loads requested before this iteration's arithmetic (round 6: without that every
    // iteration of a wave was one trip to memory followed by its five butterflies -- 27 us for 32,768 rows, a 67-MB read at 2.5 TB/s; a larger grid
    // only multiplies the waves' prologues: 1,024 / 2,048 / 4,096 workgroups 27.8 / 37.1 / 62.8 us).  Rows past M re-read row M - 1 (never stored).
    auto ldrow = [&](int m, float4& x, float4& y) {
        const float* hr = h + (size_t)(m < M ? m : M - 1) * kHid + lane * 8;
        x = *reinterpret_cast<const float4*>(hr);
        y = *reinterpret_cast<const float4*>(hr + 4);
    };
    int m = wv;
    float4 x0, y0, x1, y1;
    ldrow(m, x0, y0);
    ldrow(m + nwv, x1, y1);
    for (; m < M; m += 2 * nwv) {
        float4 nx0, ny0, nx1, ny1;
        ldrow(m + 2 * nwv, nx0, ny0);
        ldrow(m + 3 * nwv, nx1, ny1);
        row(m, x0, y0);
        if (m + nwv < M) row(m + nwv, x1, y1);
        x0 = nx0; y0 = ny0; x1 = nx1; y1 = ny1;
    }
}

// Rollout step, fused (round 4): the K-split fold of the FC layer (h = relu(bias + part[0] + part[1] + ...), the order of
// gemmz.hip's zsplit_reduce_kernel), the two heads (heads_fwd_kernel's arithmetic: 8 products per lane, wave butterfly, + bias) and
// the Categorical draw (catrow.h's row functions: K2's) of one row per wave iteration -- action, log-prob and value go straight
// into their rollout-storage rows.  Four launches of a captured env step (fold, heads, sample, the value copy) become one; the
// results are bit-identical to the four (tests/test_gpu_kernels.py).
template <int NA, bool WIDE = false>
__global__ __launch_bounds__(256) void heads_act_kernel(const float* __restrict__ part, int splits, size_t slab, const float* __restrict__ fc_bias,
                                                        const float* __restrict__ Wa, const float* __restrict__ ba,
                                                        const float* __restrict__ Wc, const float* __restrict__ bc,
                                                        const float* __restrict__ noise, uint64_t seed, uint64_t offset,
                                                        const uint64_t* __restrict__ offset_base, int64_t* __restrict__ action_i64,
                                                        float* __restrict__ action_f32, float* __restrict__ logprob,
                                                        float* __restrict__ value, float* __restrict__ hidden, int M, int a_rt) {
    __shared__ __attribute__((aligned(16))) float wl[WIDE ? kWideNA * kHid : 4];
    const int A = WIDE ? a_rt : NA - 1;                 // (narrow heads: a compile-time constant)
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
    HeadW<NA, WIDE> W;
    W.load(wl, Wa, Wc, A, lane);
    float fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fb[i] = fc_bias[lane * 8 + i];
    if (offset_base) offset += *offset_base;
    for (int m = wv; m < M; m += nwv) {
        const float* pr = part + (size_t)m * kHid + lane * 8;
        float4 x = *reinterpret_cast<const float4*>(pr), y = *reinterpret_cast<const float4*>(pr + 4);
        for (int z = 1; z < splits; ++z) {
            const float4 p = *reinterpret_cast<const float4*>(pr + (size_t)z * slab), q = *reinterpret_cast<const float4*>(pr + (size_t)z * slab + 4);
            x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
            y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w;
        }
        float hv[8] = {x.x + fb[0], x.y + fb[1], x.z + fb[2], x.w + fb[3], y.x + fb[4], y.y + fb[5], y.z + fb[6], y.w + fb[7]};
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] = hv[i] > 0.0f ? hv[i] : 0.0f;
        if (hidden) {
            float* hr = hidden + (size_t)m * kHid + lane * 8;
            *reinterpret_cast<float4*>(hr) = make_float4(hv[0], hv[1], hv[2], hv[3]);
            *reinterpret_cast<float4*>(hr + 4) = make_float4(hv[4], hv[5], hv[6], hv[7]);
        }
        float out[NA], vout = 0.0f;                     // every lane ends up with all NA sums (the butterfly is an all-reduce)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            out[a] = 0.0f;
            if (!WIDE || a <= A) {
                float wa[8];
                W.row(a, wa);
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += hv[i] * wa[i];
                out[a] = wave_sum(s) + (a < A ? ba[a] : bc[0]);
                if (a == A) vout = out[a];
            }
        }
        if (lane == 0) {
            constexpr int AMAX = WIDE ? 18 : (NA - 1 <= 4 ? 4 : 8);
            float xl[AMAX];
#pragma unroll
            for (int j = 0; j < AMAX; ++j) xl[j] = j < A ? out[j] : -INFINITY;
            CatRow<AMAX> c;
            categorical_row<AMAX>(xl, A, c);
            float best_lp;
            const int best = categorical_sample_row<AMAX>(c, A, noise ? noise + (size_t)m * A : nullptr, seed, offset, (uint64_t)m, &best_lp);
            if (action_i64) action_i64[m] = best;
            if (action_f32) action_f32[m] = (float)best;
            logprob[m] = best_lp;
            value[m] = vout;
        }
    }
}

// RELU: h is the output of a ReLU (the FC layer's, ppo_atari_multigpu.py:145): the gradient written is the one with
// respect to that layer's PRE-activation, dz = dh * (h > 0), and its column sums (that layer's bias gradient) are
// accumulated as one more partial row -- both for free here (h and dh are in registers), a `threshold_backward` pass over
// 2 x 67 MB and a column reduction over 67 MB when done by the layer itself.
template <int NA, bool RELU, bool WIDE = false>
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ h, const float* __restrict__ Wa,
                                                        const float* __restrict__ Wc, const float* __restrict__ dlogits,
                                                        const float* __restrict__ dvalue, float* __restrict__ dh,
                                                        float* __restrict__ part,      // [grid][NA + 1][512 + 1]
                                                        int M, int A, int lddh, unsigned* __restrict__ dh_amax) {
    __shared__ float red[NA + 1][kHid + 1];
    __shared__ __attribute__((aligned(16))) float wl[WIDE ? kWideNA * kHid : 4];
    float accz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (the wave index in an SGPR)
    const int wv = blockIdx.x * 4 + wave, nwv = gridDim.x * 4;
    const int na = WIDE ? A + 1 : NA;      // live rows (narrow heads: all NA, a compile-time constant)
    float acc[NA][8], accb[NA];
    unsigned dmax = 0u;                    // bits of the largest |dh| this lane stored: dh's amax record (f16split.h), when asked for
    HeadW<NA, WIDE> W;
    W.load(wl, Wa, Wc, A, lane);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        accb[a] = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[a][i] = 0.0f;
    }
    for (int m = wv; m < M; m += nwv) {
        // the row's NA output gradients: wave-uniform addresses, i.e. scalar loads -- every lane holds all of them.  (Round 4: they
        // used to be fetched by NA lanes and broadcast with ds_bpermute; with a second process time-slicing the GPU a few rows
        // per million came out with stale lanes 48-63 of the LAST permute -- tools/gpu/heads_stress.py)
        float gv[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            gv[a] = 0.0f;
            if (a < na - 1) gv[a] = dlogits[(size_t)m * (na - 1) + a];
            else if (a == na - 1) gv[a] = dvalue[m];
        }
        const float* hr = h + (size_t)m * kHid + lane * 8;
        const float4 x = *reinterpret_cast<const float4*>(hr), y = *reinterpret_cast<const float4*>(hr + 4);
        const float hv[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (!WIDE || a < na) {
                const float ga = gv[a];
                float wa[8];
                W.row(a, wa);
                accb[a] += ga;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    d[i] += ga * wa[i];
                    acc[a][i] += ga * hv[i];
                }
            }
        }
        if (RELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                d[i] = hv[i] > 0.0f ? d[i] : 0.0f;
                accz[i] += d[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned b = __float_as_uint(d[i]) & 0x7fffffffu;
            dmax = b > dmax ? b : dmax;
        }
        float* dr = dh + (size_t)m * lddh + lane * 8;
        *reinterpret_cast<float4*>(dr) = make_float4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<float4*>(dr + 4) = make_float4(d[4], d[5], d[6], d[7]);
    }
    if (dh_amax) amax_commit(dh_amax, dmax, (unsigned)wv, lane);        // (uniform; every lane is back from the row loop)
    // waves fold into one LDS accumulator in wave order (fixed), then one partial per workgroup
    for (int wsel = 0; wsel < 4; ++wsel) {
        if (wave == wsel) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                if (!WIDE || a < na) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float* r = &red[a][lane * 8 + i];
                        *r = wsel == 0 ? acc[a][i] : *r + acc[a][i];
                    }
                    if (lane == 0) red[a][kHid] = wsel == 0 ? accb[a] : red[a][kHid] + accb[a];
                }
            }
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float* r = &red[NA][lane * 8 + i];
                    *r = wsel == 0 ? accz[i] : *r + accz[i];
                }
            }
        }
        __syncthreads();
    }
    // partial [na (+ 1 with RELU: the column sums of dz)][513]; the wide kernel's dz row sits in red[NA], behind rows it never filled
    const int rows = RELU ? na + 1 : na;
    float* out = part + (size_t)blockIdx.x * rows * (kHid + 1);
    for (int e = threadIdx.x; e < rows * (kHid + 1); e += 256) {
        const int a = e / (kHid + 1), j = e - a * (kHid + 1);
        out[e] = (j == kHid && a == na) ? 0.0f : red[a == na ? NA : a][j];
    }
}

// dWa (A,512), dba (A), dWc (512), dbc (1) from the workgroup partials, fixed order.
// (with dbh: one more partial row, the column sums of dz -> dbh (512))
__global__ __launch_bounds__(256) void heads_bwd_reduce(const float* __restrict__ part, int nparts, int NA, int A,
                                                        float* __restrict__ dWa, float* __restrict__ dba,
                                                        float* __restrict__ dWc, float* __restrict__ dbc, float* __restrict__ dbh) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int rows = dbh ? NA + 1 : NA;
    if (e >= rows * (kHid + 1)) return;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)rows * (kHid + 1);
    int p = 0;
    for (; p + 32 <= nparts; p += 32) {              // 32 loads in flight (the fold is a latency chain: 13 workgroups), added in the order below
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = part[(size_t)(p + u) * stride + e];
#pragma unroll
        for (int u = 0; u < 32; ++u) s8[u & 7] += v[u];
    }
    for (; p + 8 <= nparts; p += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += part[(size_t)(p + u) * stride + e];
    }
    for (; p < nparts; ++p) s8[p & 7] += part[(size_t)p * stride + e];
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    const int a = e / (kHid + 1), j = e - a * (kHid + 1);
    if (a == NA) {
        if (j < kHid) dbh[j] = s;
    } else if (a < A) {
        if (j < kHid) dWa[(size_t)a * kHid + j] = s;
        else dba[a] = s;
    } else {
        if (j < kHid) dWc[j] = s;
        else dbc[0] = s;
    }
}

static int heads_grid(int M) { return M < kHeadsBlocks * 4 ? (M + 3) / 4 : kHeadsBlocks; }

}  // namespace mi355ppo

using namespace mi355ppo;

static int heads_check(const char* fn, int M, int A, int H) {
    MI355_REQUIRE(M > 0, MI355PPO_EINVAL, "%s: M=%d must be positive", fn, M);
    MI355_REQUIRE(A >= 1 && A <= kWideNA - 1, MI355PPO_EINVAL, "%s: A=%d must be in 1..%d (use a library GEMM for wider heads)", fn, A, kWideNA - 1);
    MI355_REQUIRE(H == kHid, MI355PPO_EINVAL, "%s: hidden width %d is not supported (NatureCNN: 512)", fn, H);
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_heads_fwd_f32(const float* h, const float* Wa, const float* ba, const float* Wc,
                                                   const float* bc, float* logits, float* value, int M, int A, int H,
                                                   void* stream) {
    const char* fn = "mi355ppo_heads_fwd_f32";
    MI355_REQUIRE(h && Wa && ba && Wc && bc && logits && value, MI355PPO_EINVAL, "%s: null pointer", fn);
    int rc = heads_check(fn, M, A, H);
    if (rc) return rc;
    MI355_REQUIRE(aligned(h, 16) && aligned(Wa, 16) && aligned(Wc, 4) && aligned(ba, 4) && aligned(bc, 4) && aligned(logits, 4) &&
                      aligned(value, 4), MI355PPO_EALIGN, "%s: h / Wa must be 16-byte aligned", fn);
    const dim3 grid(heads_grid(M));
    hipStream_t s = as_stream(stream);
#define LAUNCH(NA) hipLaunchKernelGGL((heads_fwd_kernel<NA>), grid, dim3(256), 0, s, h, Wa, ba, Wc, bc, logits, value, M, A)
    switch (A + 1) {
        case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; case 4: LAUNCH(4); break; case 5: LAUNCH(5); break;
        case 6: LAUNCH(6); break; case 7: LAUNCH(7); break; case 8: LAUNCH(8); break;
        default: hipLaunchKernelGGL((heads_fwd_kernel<kWideNA, true>), grid, dim3(256), 0, s, h, Wa, ba, Wc, bc, logits, value, M, A); break;
    }
#undef LAUNCH
    return check_launch("heads_fwd_kernel");
}

static int fc_heads_act_impl(const char* fn, const float* a3, int lda, const void* fc_pack, const float* fc_bias, const float* Wa, const float* ba,
                             const float* Wc, const float* bc, int M, int A, int H, int K, const float* noise_exp1, uint64_t seed, uint64_t offset,
                             const uint64_t* offset_base, int64_t* action_i64, float* action_f32, float* logprob, float* value, float* hidden_out,
                             void* workspace, size_t workspace_bytes, const unsigned* a3_amax, void* stream) {
    MI355_REQUIRE(a3 && fc_pack && fc_bias && Wa && ba && Wc && bc && logprob && value, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(action_i64 || action_f32, MI355PPO_EINVAL, "%s: no action output", fn);
    int rc = heads_check(fn, M, A, H);
    if (rc) return rc;
    MI355_REQUIRE(aligned(Wa, 16) && aligned(Wc, 4) && aligned(fc_bias, 4) && aligned(ba, 4) && aligned(bc, 4) && aligned(noise_exp1, 4) &&
                      aligned(offset_base, 8) && aligned(action_i64, 8) && aligned(action_f32, 4) && aligned(logprob, 4) && aligned(value, 4) &&
                      aligned(hidden_out, 16), MI355PPO_EALIGN, "%s: Wa / hidden_out must be 16-byte aligned", fn);
    int splits = 0;
    rc = z_fc_raw_launch(fn, a3, lda, fc_pack, M, H, K, workspace, workspace_bytes, &splits, as_stream(stream), a3_amax);
    if (rc) return rc;
    const dim3 grid(heads_grid(M));
    const float* part = static_cast<const float*>(workspace);
    const size_t slab = (size_t)M * kHid;
#define LAUNCH(NA) hipLaunchKernelGGL((heads_act_kernel<NA>), grid, dim3(256), 0, as_stream(stream), part, splits, slab, fc_bias, Wa, ba, Wc, bc, \
                                      noise_exp1, seed, offset, offset_base, action_i64, action_f32, logprob, value, hidden_out, M, A)
    switch (A + 1) {
        case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; case 4: LAUNCH(4); break; case 5: LAUNCH(5); break;
        case 6: LAUNCH(6); break; case 7: LAUNCH(7); break; case 8: LAUNCH(8); break;
        default: hipLaunchKernelGGL((heads_act_kernel<kWideNA, true>), grid, dim3(256), 0, as_stream(stream), part, splits, slab, fc_bias, Wa, ba, Wc, bc,
                                    noise_exp1, seed, offset, offset_base, action_i64, action_f32, logprob, value, hidden_out, M, A); break;
    }
#undef LAUNCH
    return check_launch("heads_act_kernel");
}

extern "C" MI355PPO_API int mi355ppo_fc_heads_act_categorical_f32(const float* a3, int lda, const void* fc_pack, const float* fc_bias,
                                                                  const float* Wa, const float* ba, const float* Wc, const float* bc, int M,
                                                                  int A, int H, int K, const float* noise_exp1, uint64_t seed,
                                                                  uint64_t offset, const uint64_t* offset_base, int64_t* action_i64,
                                                                  float* action_f32, float* logprob, float* value, float* hidden_out,
                                                                  void* workspace, size_t workspace_bytes, void* stream) {
    return fc_heads_act_impl("mi355ppo_fc_heads_act_categorical_f32", a3, lda, fc_pack, fc_bias, Wa, ba, Wc, bc, M, A, H, K, noise_exp1, seed,
                             offset, offset_base, action_i64, action_f32, logprob, value, hidden_out, workspace, workspace_bytes, nullptr, stream);
}

extern "C" MI355PPO_API size_t mi355ppo_heads_bwd_workspace_bytes(int M, int A) {
    if (M <= 0 || A < 1 || A > kWideNA - 1) return 0;
    return (size_t)heads_grid(M) * (A + 2) * (kHid + 1) * sizeof(float);      // (A + 1 rows; one more for the ReLU variant's bias gradient)
}

static int heads_bwd(const char* fn, const float* h, const float* Wa, const float* Wc, const float* dlogits, const float* dvalue, float* dh,
                     float* dWa, float* dba, float* dWc, float* dbc, float* dbh, int lddh, int M, int A, int H, void* workspace,
                     size_t workspace_bytes, void* stream, unsigned* dh_amax = nullptr) {
    MI355_REQUIRE(h && Wa && Wc && dlogits && dvalue && dh && dWa && dba && dWc && dbc, MI355PPO_EINVAL, "%s: null pointer", fn);
    int rc = heads_check(fn, M, A, H);
    if (rc) return rc;
    MI355_REQUIRE(lddh >= H && lddh % 4 == 0, MI355PPO_EINVAL, "%s: row pitch %d of the hidden gradient (a multiple of 4, >= %d)", fn, lddh, H);
    const size_t need = mi355ppo_heads_bwd_workspace_bytes(M, A);
    MI355_REQUIRE(workspace && workspace_bytes >= need, MI355PPO_EWORKSPACE, "%s: workspace %zu bytes < required %zu", fn,
                  workspace ? workspace_bytes : (size_t)0, need);
    MI355_REQUIRE(aligned(h, 16) && aligned(Wa, 16) && aligned(Wc, 4) && aligned(dh, 16) && aligned(workspace, 4) &&
                      aligned(dlogits, 4) && aligned(dvalue, 4) && aligned(dbh, 4), MI355PPO_EALIGN, "%s: h / Wa / dh must be 16-byte aligned", fn);
    const int nb = heads_grid(M);
    float* part = static_cast<float*>(workspace);
    hipStream_t s = as_stream(stream);
#define LAUNCH(NA)                                                                                                                   \
    do {                                                                                                                             \
        if (dbh) hipLaunchKernelGGL((heads_bwd_kernel<NA, true>), dim3(nb), dim3(256), 0, s, h, Wa, Wc, dlogits, dvalue, dh, part, M, A, lddh, dh_amax);  \
        else hipLaunchKernelGGL((heads_bwd_kernel<NA, false>), dim3(nb), dim3(256), 0, s, h, Wa, Wc, dlogits, dvalue, dh, part, M, A, lddh, dh_amax);    \
    } while (0)
    switch (A + 1) {
        case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; case 4: LAUNCH(4); break; case 5: LAUNCH(5); break;
        case 6: LAUNCH(6); break; case 7: LAUNCH(7); break; case 8: LAUNCH(8); break;
        default:
            if (dbh) hipLaunchKernelGGL((heads_bwd_kernel<kWideNA, true, true>), dim3(nb), dim3(256), 0, s, h, Wa, Wc, dlogits, dvalue, dh, part, M, A, lddh, dh_amax);
            else hipLaunchKernelGGL((heads_bwd_kernel<kWideNA, false, true>), dim3(nb), dim3(256), 0, s, h, Wa, Wc, dlogits, dvalue, dh, part, M, A, lddh, dh_amax);
            break;
    }
#undef LAUNCH
    rc = check_launch("heads_bwd_kernel");
    if (rc) return rc;
    const int total = (A + 1 + (dbh ? 1 : 0)) * (kHid + 1);
    hipLaunchKernelGGL(heads_bwd_reduce, dim3((total + 255) / 256), dim3(256), 0, s, part, nb, A + 1, A, dWa, dba, dWc, dbc, dbh);
    return check_launch("heads_bwd_reduce");
}

extern "C" MI355PPO_API int mi355ppo_heads_bwd_f32(const float* h, const float* Wa, const float* Wc, const float* dlogits,
                                                   const float* dvalue, float* dh, float* dWa, float* dba, float* dWc,
                                                   float* dbc, int M, int A, int H, void* workspace, size_t workspace_bytes,
                                                   void* stream) {
    return heads_bwd("mi355ppo_heads_bwd_f32", h, Wa, Wc, dlogits, dvalue, dh, dWa, dba, dWc, dbc, nullptr, H, M, A, H, workspace,
                     workspace_bytes, stream);
}

extern "C" MI355PPO_API int mi355ppo_heads_bwd_relu_f32(const float* h, const float* Wa, const float* Wc, const float* dlogits,
                                                        const float* dvalue, float* dz, int lddz, float* dWa, float* dba,
                                                        float* dWc, float* dbc, float* dbh, int M, int A, int H, void* workspace,
                                                        size_t workspace_bytes, void* stream) {
    const char* fn = "mi355ppo_heads_bwd_relu_f32";
    MI355_REQUIRE(dbh, MI355PPO_EINVAL, "%s: null pointer", fn);
    return heads_bwd(fn, h, Wa, Wc, dlogits, dvalue, dz, dWa, dba, dWc, dbc, dbh, lddz, M, A, H, workspace, workspace_bytes, stream);
}

// The same, also folding |dz| into `dz_amax` (dz's amax record, zeroed by the caller): the FC layer's data and weight gradients on the
// f16 split (mi355ppo_fc_dgrad_packed_f16x2_f32, mi355ppo_fc_wgrad_f16x2_f32) scale dz by it.
extern "C" MI355PPO_API int mi355ppo_heads_bwd_relu_amax_f32(const float* h, const float* Wa, const float* Wc, const float* dlogits,
                                                             const float* dvalue, float* dz, int lddz, float* dWa, float* dba, float* dWc,
                                                             float* dbc, float* dbh, int M, int A, int H, void* workspace,
                                                             size_t workspace_bytes, uint32_t* dz_amax, void* stream) {
    const char* fn = "mi355ppo_heads_bwd_relu_amax_f32";
    MI355_REQUIRE(dbh && dz_amax, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(aligned(dz_amax, 64), MI355PPO_EALIGN, "%s: the amax record must be 64-byte aligned", fn);
    return heads_bwd(fn, h, Wa, Wc, dlogits, dvalue, dz, dWa, dba, dWc, dbc, dbh, lddz, M, A, H, workspace, workspace_bytes, stream, dz_amax);
}

// mi355ppo_fc_heads_act_categorical_f32 with the FC layer on the f16 split: `fc_pack` is an f16x2 pack, `a3_amax` a3's amax record
extern "C" MI355PPO_API int mi355ppo_fc_heads_act_categorical_f16x2_f32(const float* a3, int lda, const void* fc_pack, const float* fc_bias,
                                                                        const float* Wa, const float* ba, const float* Wc, const float* bc,
                                                                        int M, int A, int H, int K, const float* noise_exp1, uint64_t seed,
                                                                        uint64_t offset, const uint64_t* offset_base, int64_t* action_i64,
                                                                        float* action_f32, float* logprob, float* value, float* hidden_out,
                                                                        void* workspace, size_t workspace_bytes, const uint32_t* a3_amax,
                                                                        void* stream) {
    const char* fn = "mi355ppo_fc_heads_act_categorical_f16x2_f32";
    MI355_REQUIRE(a3_amax && aligned(a3_amax, 64), MI355PPO_EINVAL, "%s: a3's amax record missing or not 64-byte aligned", fn);
    return fc_heads_act_impl(fn, a3, lda, fc_pack, fc_bias, Wa, ba, Wc, bc, M, A, H, K, noise_exp1, seed, offset, offset_base, action_i64,
                             action_f32, logprob, value, hidden_out, workspace, workspace_bytes, a3_amax, stream);
}
