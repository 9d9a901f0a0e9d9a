// K7 -- the reference's 64-64 tanh MLP agents as ONE kernel family (gfx950): rollout step (both networks' forward + sampling),
// and the whole minibatch update body -- gather, both forwards, the distribution, the fused PPO loss row terms, both backward
// passes and the weight gradients -- in one launch, plus a fold.
//
// Replaces, per minibatch (cleanrl/ppo.py:250-287 / ppo_continuous_action.py:265-302 with the Agents of ppo.py:100-126 /
// ppo_continuous_action.py:112-141): b_obs[mb_inds], six Linear forwards (addmm + tanh: 10 launches), the distribution and
// loss chains (~45 launches, K3 fused them already), and the autograd backward of all of it (~50 launches incl. the
// column-sum reductions for the biases and for the shared actor_logstd) -- ~45 library / elementwise launches of <= 10 us
// each after K3, which made BASELINE configs[4] host-launch-bound (profiles/r03_kernel_stats_cfgE.csv).
// Per rollout step (:205-213 / :221-229): the six Linear forwards + K2 / K2' in one launch.
//
// Mapping.  The two networks (actor, critic) are independent 3-layer MLPs of 5.4-5.7 k parameters: a workgroup is ONE wave that
// owns one network and a block of R <= 64 minibatch rows.
//   hidden layers : lane j = hidden unit j.  The lane keeps row j of W1 / W2 (and, for the backward pass, column j of W2) in
//                   registers; the activations of a row are broadcast through the wave's private LDS (all lanes read one
//                   address: conflict-free): 17 + 64 v_fma per row and layer pair, no cross-lane reduction anywhere.
//   output layer, distribution, loss : lane = row.  After the hidden layers of all R rows are in LDS, lane r computes its
//                   row's n_out <= 8 outputs (weights through scalar loads), then -- with the row in registers, exactly as
//                   K2 / K3 do -- the Categorical / Normal math, ppo_row_terms (the ONE definition shared with K3 and the
//                   host twins) and the closed-form gradient with respect to the outputs.
//   backward      : lane j again; the row's output gradient comes back through v_readlane.  dW accumulators live in
//                   registers across the block's rows (dW2: 64, dW1: <= 32, dW3: <= 8 per lane) and are written once per
//                   wave as a partial; mlp_fold adds the partials in block order (f64, deterministic) into the gradients.
// f32 v_fma throughout (the reference's MLP runs in f32 GEMMs whose summation order is unspecified): results agree with
// torch to f32 round-off, tests/test_gpu_mlp.py.  No MFMA: at 64 x 64 the layer is 8 KFLOP per row -- the launch is bound by
// latency, not by any pipe.
//
// WIDE variant (round 6: Humanoid's 376 observations / 17 actions, Ant-v2's 111 -- every agent of ppo_continuous_action.py): observation width
// up to 512 and up to 20 outputs.  A row of W1 no longer fits the lane's registers, so layer 1 walks the observation in CHUNKS of 32 columns: the
// lane loads 32 elements of its W1 row, adds their products with the chunk of every staged row (LDS broadcast reads) into one accumulator per row
// (blocks of <= 32 rows), and moves on; the weight gradient of W1 does the same per chunk with 32 accumulators against the rows' dz1 (parked in H1's
// LDS rows).  Everything behind layer 1 is the narrow code with AMAX = 20.  Shapes inside the narrow limits keep the narrow kernels.
#include "common.h"
#include "catrow.h"
#include "ppo_rows.h"

namespace mi355ppo {

constexpr int kH = 64;             // hidden width of the reference's MLP agents
constexpr int kH2P = 68;           // LDS row pitch of the second hidden layer (lane = row reads it as 16-byte pieces)
constexpr int kMlpMaxBlocks = 2048;

struct MlpNet {                    // torch layouts: W1 (64,O), b1 (64), W2 (64,64), b2 (64), W3 (n_out,64), b3 (n_out)
    const float *w1, *b1, *w2, *b2, *w3, *b3;
};
struct MlpGrads {
    float *w1, *b1, *w2, *b2, *w3, *b3;
};

#define MI355_LOG_SQRT_2PI 0.91893853320467274178f
#define MI355_HALF_LOG_2PIE 1.4189385332046727418f

__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// ---- staging: rows [row0, row0 + nrows) of the (gathered) observation matrix -> LDS, pitch OMAX, zero padded ----------------
// Lane r first fetches the flat-batch index of row r (ONE round trip for the whole block) and parks it in `scratch` (LDS the caller
// does not need yet: >= 64 int64); the element loop then reads a row's index from there, so all its loads are independent and in
// flight together.  (Round 4 fetched it with a lane permute, ds_bpermute -- the primitive whose broadcast returned stale lanes in the
// heads' backward kernel when a second process time-sliced the GPU; plain LDS writes and reads of one wave
// execute in order.)  Without `inds` the index is arithmetic.
template <int OMAX>
__device__ __forceinline__ void stage_rows(float* xs, const float* __restrict__ obs, const int64_t* __restrict__ inds, int64_t row0,
                                           int nrows, int O, int lane, float* scratch) {
    int64_t* const ids = reinterpret_cast<int64_t*>(scratch);
    if (inds) {                                            // (uniform)
        ids[lane] = lane < nrows ? inds[row0 + lane] : 0;
        __builtin_amdgcn_wave_barrier();
    }
    constexpr int kIter = 64 * OMAX / 64;                  // R <= 64 rows x OMAX columns, 64 lanes
    float v[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int e = lane + 64 * it;
        const int r = e / OMAX, k = e - r * OMAX;
        const int rr = r < nrows ? r : 0;
        const int64_t i = inds ? ids[rr] : row0 + rr;
        v[it] = (r < nrows && k < O) ? obs[i * O + k] : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();                       // (the index reads are done before `scratch` is written again)
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        const int e = lane + 64 * it;
        if (e < nrows * OMAX) xs[e] = v[it];
    }
}

// WIDE: rows of runtime pitch OP (a multiple of 32, zero padded); a plain strided loop, eight loads in flight
__device__ __forceinline__ void stage_rows_wide(float* xs, const float* __restrict__ obs, const int64_t* __restrict__ inds, int64_t row0, int nrows,
                                                int O, int OP, int lane, float* scratch) {
    int64_t* const ids = reinterpret_cast<int64_t*>(scratch);
    if (inds) {
        ids[lane] = lane < nrows ? inds[row0 + lane] : 0;
        __builtin_amdgcn_wave_barrier();
    }
    const int total = nrows * OP;
    for (int e0 = 0; e0 < total; e0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 64 * u + lane;
            const int r = e / OP, k = e - r * OP;
            const int rr = r < nrows ? r : 0;
            const int64_t i = inds ? ids[rr] : row0 + rr;
            v[u] = (e < total && k < O) ? obs[i * O + k] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 64 * u + lane;
            if (e < total) xs[e] = v[u];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- hidden layers, lane = hidden unit ------------------------------------------------------------------------------------------
// layer 1 of every staged row -> H1 (pitch 64); layer 2 of every row -> H2 (pitch kH2P).  Two loops: a row's layer 2 reads what
// layer 1 wrote through LDS, and one wave per SIMD has nobody to hide that round trip behind -- so the round trips of all rows
// are taken together.
template <int OMAX>
struct HiddenW {
    float w1[OMAX], w2[kH], b1, b2;
};

// Row `lane` of W1 and W2 and the two bias elements -> registers.  Called BEFORE the observation rows are staged: one memory round
// trip for everything the forward hidden layers need.
template <int OMAX, bool WIDE = false>
__device__ __forceinline__ void hidden_load(HiddenW<OMAX>& w, const MlpNet& n, int O, int lane) {
    if (!WIDE) {
#pragma unroll
        for (int k = 0; k < OMAX; ++k) w.w1[k] = k < O ? n.w1[lane * O + k] : 0.0f;
    }
    if ((reinterpret_cast<uintptr_t>(n.w2) & 15) == 0) {          // (wave-uniform) rows of W2 are 256 bytes: 16-byte pieces when the base allows
        const float4* wrow = reinterpret_cast<const float4*>(n.w2 + lane * kH);
#pragma unroll
        for (int q = 0; q < kH / 4; ++q) {
            const float4 v = wrow[q];
            w.w2[4 * q] = v.x; w.w2[4 * q + 1] = v.y; w.w2[4 * q + 2] = v.z; w.w2[4 * q + 3] = v.w;
        }
    } else {                                                       // a flat parameter buffer packs tensors without padding
#pragma unroll
        for (int q = 0; q < kH; ++q) w.w2[q] = n.w2[lane * kH + q];
    }
    w.b1 = n.b1[lane];
    w.b2 = n.b2[lane];
}

constexpr int kWideRows = 32;      // WIDE: rows per block (one layer-1 accumulator per row in registers)

// WIDE layer 1: H1[r][lane] = tanh(b1 + sum over chunks c of W1[lane][32 c ..] . xs[r][32 c ..])
__device__ __forceinline__ void layer1_wide(const MlpNet& n, float b1, const float* xs, float* H1, int nrows, int O, int OP, int lane) {
    float z[kWideRows];
#pragma unroll
    for (int r = 0; r < kWideRows; ++r) z[r] = 0.0f;
    const float* wrow = n.w1 + (int64_t)lane * O;
    for (int c0 = 0; c0 < OP; c0 += 32) {
        float w[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) w[k] = c0 + k < O ? wrow[c0 + k] : 0.0f;
#pragma unroll
        for (int r = 0; r < kWideRows; ++r) {
            if (r < nrows) {                                   // (uniform)
                const float4* x4 = reinterpret_cast<const float4*>(xs + r * OP + c0);
                float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 v = x4[q];
                    z0 = fmaf(w[4 * q + 0], v.x, z0);
                    z1 = fmaf(w[4 * q + 1], v.y, z1);
                    z0 = fmaf(w[4 * q + 2], v.z, z0);
                    z1 = fmaf(w[4 * q + 3], v.w, z1);
                }
                z[r] += z0 + z1;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kWideRows; ++r)
        if (r < nrows) H1[r * kH + lane] = tanhf(z[r] + b1);
}

template <int OMAX, bool WIDE = false>
__device__ __forceinline__ void hidden_fwd(const HiddenW<OMAX>& hw, const float* xs, float* H1, float* H2, int nrows, int lane,
                                           const MlpNet* wide_net = nullptr, int O = 0, int OP = 0) {
    if constexpr (WIDE) {
        layer1_wide(*wide_net, hw.b1, xs, H1, nrows, O, OP, lane);
    } else {
        const float (&w1)[OMAX] = hw.w1;
        const float b1 = hw.b1;
#pragma unroll 2
        for (int r = 0; r < nrows; ++r) {
            const float4* x4 = reinterpret_cast<const float4*>(xs + r * OMAX);
            float z0 = b1, z1 = 0.0f;
#pragma unroll
            for (int q = 0; q < OMAX / 4; ++q) {
                const float4 v = x4[q];
                z0 = fmaf(w1[4 * q + 0], v.x, z0);
                z1 = fmaf(w1[4 * q + 1], v.y, z1);
                z0 = fmaf(w1[4 * q + 2], v.z, z0);
                z1 = fmaf(w1[4 * q + 3], v.w, z1);
            }
            H1[r * kH + lane] = tanhf(z0 + z1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    {
        const float (&w2)[kH] = hw.w2;
        const float b2 = hw.b2;
#pragma unroll 2
        for (int r = 0; r < nrows; ++r) {
            const float4* h4 = reinterpret_cast<const float4*>(H1 + r * kH);
            float a0 = b2, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
            for (int q = 0; q < kH / 4; ++q) {
                const float4 v = h4[q];
                a0 = fmaf(w2[4 * q + 0], v.x, a0);
                a1 = fmaf(w2[4 * q + 1], v.y, a1);
                a2 = fmaf(w2[4 * q + 2], v.z, a2);
                a3 = fmaf(w2[4 * q + 3], v.w, a3);
            }
            H2[r * kH2P + lane] = tanhf((a0 + a1) + (a2 + a3));
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- output layer, lane = row: out[o] = b3[o] + sum_j W3[o][j] h2[row][j] (weights are wave-uniform: scalar loads) -------------
template <int AMAX>
__device__ __forceinline__ void output_fwd(const MlpNet& n, const float* H2, int row, int nout, float (&out)[AMAX]) {
#pragma unroll
    for (int o = 0; o < AMAX; ++o) out[o] = n.b3[o < nout ? o : nout - 1];
    const float4* h4 = reinterpret_cast<const float4*>(H2 + row * kH2P);
#pragma unroll
    for (int q = 0; q < kH / 4; ++q) {
        const float4 v = h4[q];
#pragma unroll
        for (int o = 0; o < AMAX; ++o) {
            const float* w = n.w3 + (o < nout ? o : nout - 1) * kH + 4 * q;      // rows past n_out repeat the last one (never used)
            out[o] = fmaf(w[0], v.x, out[o]);
            out[o] = fmaf(w[1], v.y, out[o]);
            out[o] = fmaf(w[2], v.z, out[o]);
            out[o] = fmaf(w[3], v.w, out[o]);
        }
    }
}

// Box-Muller on one Philox block: the stream of distributions.hip's normal_kernel (counter = row * nblk + d / 4)
__device__ __forceinline__ void normal4(const Philox& rng, uint64_t ctr, uint64_t offset, float (&z4)[4]) {
    const uint4 r = rng(ctr, offset);
    const float r0 = sqrtf(-2.0f * logf(u32_to_unit_open(r.x)));
    const float r1 = sqrtf(-2.0f * logf(u32_to_unit_open(r.z)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u32_to_unit_open(r.y), &s0, &c0);
    sincosf(6.283185307179586f * u32_to_unit_open(r.w), &s1, &c1);
    z4[0] = r0 * c0; z4[1] = r0 * s0; z4[2] = r1 * c1; z4[3] = r1 * s1;
}

// =================================================================================================================================
// Rollout step / plain forward.  Workgroup (one wave) = (network, block of R rows).
//   MODE 0: outputs only (actor_out (B,n_out), value (B));  MODE 1: + sample (Categorical or Normal), action / logprob written.
struct MlpActArgs {
    MlpNet net[2];                 // 0 = critic, 1 = actor
    const float* obs;              // (B, O)
    int B, O, nout, R;
    const float* logstd;           // Normal: (D)
    const float* noise;            // (B, n_out) caller-supplied draws (Exp(1) / N(0,1)) or NULL
    uint64_t seed, offset;
    const uint64_t* offset_base;   // device-resident stream position (captured launches) or NULL
    int64_t* action_i64;           // Categorical, may be NULL
    float* action_f32;             // Categorical: (B) f32 ; Normal: (B, D)
    float *logprob, *entropy, *value, *actor_out;      // entropy / actor_out may be NULL
};

template <int OMAX, int AMAX, bool NORMAL, int MODE, bool WIDE = false>
__global__ __launch_bounds__(64) void mlp_act_kernel(const MlpActArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x;
    const int netid = blockIdx.x & 1;
    const int64_t row0 = (int64_t)(blockIdx.x >> 1) * a.R;
    const int nrows = (int)((a.B - row0) < (int64_t)a.R ? (a.B - row0) : (int64_t)a.R);
    const MlpNet n = a.net[netid];
    const int nout = netid ? a.nout : 1;
    const int OP = WIDE ? (a.O + 31) / 32 * 32 : OMAX;      // pitch of a staged row
    float* xs = lds;
    float* H1 = xs + a.R * OP;
    float* H2 = H1 + a.R * kH;
    HiddenW<OMAX> hw;
    hidden_load<OMAX, WIDE>(hw, n, a.O, lane);
    if constexpr (WIDE) stage_rows_wide(xs, a.obs, nullptr, row0, nrows, a.O, OP, lane, nullptr);
    else stage_rows<OMAX>(xs, a.obs, nullptr, row0, nrows, a.O, lane, nullptr);
    __builtin_amdgcn_wave_barrier();
    hidden_fwd<OMAX, WIDE>(hw, xs, H1, H2, nrows, lane, &n, a.O, OP);
    if (lane >= nrows) return;
    const int64_t row = row0 + lane;
    float out[AMAX];
    output_fwd<AMAX>(n, H2, lane, nout, out);
    if (netid == 0) {
        a.value[row] = out[0];
        return;
    }
    if (a.actor_out) {
#pragma unroll
        for (int o = 0; o < AMAX; ++o) if (o < nout) a.actor_out[row * nout + o] = out[o];
    }
    if (MODE == 0) return;
    uint64_t offset = a.offset;
    if (a.offset_base) offset += *a.offset_base;
    if (!NORMAL) {
        const int A = nout;
        float x[AMAX];
#pragma unroll
        for (int j = 0; j < AMAX; ++j) x[j] = j < A ? out[j] : -INFINITY;
        CatRow<AMAX> c;
        categorical_row<AMAX>(x, A, c);
        float best_lp;
        const int best = categorical_sample_row<AMAX>(c, A, a.noise ? a.noise + row * A : nullptr, a.seed, offset, (uint64_t)row, &best_lp);
        if (a.action_i64) a.action_i64[row] = best;
        if (a.action_f32) a.action_f32[row] = (float)best;
        a.logprob[row] = best_lp;
        if (a.entropy) a.entropy[row] = c.H;
    } else {
        const int D = nout;
        const Philox rng(a.seed);
        const int nblk = (D + 3) / 4;
        float lp = 0.0f, ent = 0.0f;
        float z4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < AMAX; ++d) {
            if (d < D) {
                const float mu = out[d];
                const float sd = expf(a.logstd[d]);
                float z;
                if (a.noise) {
                    z = a.noise[row * D + d];
                } else {
                    if ((d & 3) == 0) normal4(rng, (uint64_t)row * nblk + (d >> 2), offset, z4);
                    z = z4[d & 3];
                }
                float act = z * sd;          // torch.normal(mean, std): normal_(0,1).mul_(std).add_(mean)
                act = act + mu;
                a.action_f32[row * D + d] = act;
                const float diff = act - mu;
                const float var = sd * sd;
                const float log_scale = logf(sd);
                float t = -(diff * diff);
                t = t / (2.0f * var);
                t = t - log_scale;
                t = t - MI355_LOG_SQRT_2PI;
                lp += t;
                ent += MI355_HALF_LOG_2PIE + log_scale;
            }
        }
        a.logprob[row] = lp;
        if (a.entropy) a.entropy[row] = ent;
    }
}

// =================================================================================================================================
// Minibatch update body.
struct MlpPpoArgs {
    MlpNet net[2];
    const float* obs;              // (Bflat, O) flat batch
    const int64_t* inds;           // (M) rows of the flat batch, or NULL = identity
    int M, O, nout, R, nblocks;
    const float* b_actions;        // Categorical: (Bflat) f32 ; Normal: (Bflat, D)
    const float *b_logprobs, *b_adv, *b_ret, *b_val;
    const float* logstd;           // Normal: (D)
    const float* mean_shift;       // Normal: (M, D) added to the mean before the loss (RPO), or NULL
    const float* adv_mean_den;     // (2) = mean, std + 1e-8 of the minibatch advantages; NULL when !norm_adv
    LossParams P;
    float* part[2];                // per network: [nblocks][Pnet] weight-gradient partials (internal order, see mlp_part_layout)
    double* sums;                  // [2 * nblocks][kNumSums + AMAXSUM]: rows 2b (critic) and 2b + 1 (actor) of block b
    int sum_stride;
};

// internal order of a network's partial vector: W1 as [k][j], b1, W2 as [k][j], b2, W3 as [o][j], b3
__host__ __device__ inline int mlp_part_size(int O, int nout) { return O * kH + kH + kH * kH + kH + nout * kH + nout; }

template <int OMAX, int AMAX, bool NORMAL, bool WIDE = false>
__global__ __launch_bounds__(64) void mlp_ppo_kernel(const MlpPpoArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x;
    const int netid = blockIdx.x & 1;
    const int blk = blockIdx.x >> 1;
    const int64_t row0 = (int64_t)blk * a.R;
    const int nrows = (int)((a.M - row0) < (int64_t)a.R ? (a.M - row0) : (int64_t)a.R);
    const MlpNet n = a.net[netid];
    const int nout = netid ? a.nout : 1;
    const int O = a.O;
    const int OP = WIDE ? (O + 31) / 32 * 32 : OMAX;        // pitch of a staged row
    float* xs = lds;
    float* H1 = xs + a.R * OP;
    float* H2 = H1 + a.R * kH;
    HiddenW<OMAX> hw;
    hidden_load<OMAX, WIDE>(hw, n, O, lane);
    if constexpr (WIDE) stage_rows_wide(xs, a.obs, a.inds, row0, nrows, O, OP, lane, H1);
    else stage_rows<OMAX>(xs, a.obs, a.inds, row0, nrows, O, lane, H1);
    // this lane's row (lane = row phase): behaviour data, requested before the hidden layers run
    const bool valid = lane < nrows;
    const int64_t m = row0 + (valid ? lane : 0);
    const int64_t i = a.inds ? a.inds[m] : m;
    const float old_lp = a.b_logprobs[i], adv = a.b_adv[i], ret = a.b_ret[i], old_v = a.b_val[i];
    float bact[AMAX];
    if (NORMAL) {
#pragma unroll
        for (int d = 0; d < AMAX; ++d) bact[d] = a.b_actions[i * a.nout + (d < a.nout ? d : 0)];
    } else {
        bact[0] = a.b_actions[i];
    }
    float amean = 0.0f, aden = 1.0f;
    if (a.P.norm_adv) { amean = a.adv_mean_den[0]; aden = a.adv_mean_den[1]; }
    __builtin_amdgcn_wave_barrier();
    hidden_fwd<OMAX, WIDE>(hw, xs, H1, H2, nrows, lane, &n, O, OP);

    // ---- lane = row: outputs, distribution, loss row terms, gradient with respect to the outputs ----------------------------
    float out[AMAX], dout[AMAX];
    output_fwd<AMAX>(n, H2, valid ? lane : 0, nout, out);
    float rs[kNumSums];             // this row's contributions to the six scalar sums
    float cls[AMAX];                // Normal: this row's contributions to d loss / d logstd
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) rs[k] = 0.0f;
#pragma unroll
    for (int o = 0; o < AMAX; ++o) { dout[o] = 0.0f; cls[o] = 0.0f; }
    if (netid == 0) {
        const RowTerms t = ppo_row_terms(0.0f, 0.0f, out[0], 0.0f, 0.0f, ret, old_v, amean, aden, a.P);
        dout[0] = t.dvalue;
        rs[1] = t.sums[1];
    } else if (!NORMAL) {
        const int A = nout;
        float x[AMAX];
#pragma unroll
        for (int j = 0; j < AMAX; ++j) x[j] = j < A ? out[j] : -INFINITY;
        CatRow<AMAX> c;
        categorical_row<AMAX>(x, A, c);
        const int act = (int)bact[0];                      // b_actions.long()[mb_inds]
        float newlp = 0.0f;
#pragma unroll
        for (int j = 0; j < AMAX; ++j) if (j == act) newlp = c.lp[j];
        const RowTerms t = ppo_row_terms(newlp, c.H, 0.0f, old_lp, adv, 0.0f, 0.0f, amean, aden, a.P);
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) rs[k] = k == 1 ? 0.0f : t.sums[k];
        // d loss / d logits_j = g_lp * (1[j == a] - p_j) + (ent_coef / M) * p_j * (lp_j + H)       (loss.hip cat_rows)
        const float ge = a.P.ent_coef / (float)a.P.M;
#pragma unroll
        for (int j = 0; j < AMAX; ++j) {
            const float onehot = (j == act) ? 1.0f : 0.0f;
            const float lpj = fmaxf(c.lp[j], -FLT_MAX);
            dout[j] = j < A ? t.g_lp * (onehot - c.p[j]) + ge * (c.p[j] * (lpj + c.H)) : 0.0f;
        }
    } else {
        const int D = nout;
        float lp = 0.0f, ent = 0.0f;
        float diffv[AMAX], varv[AMAX];
#pragma unroll
        for (int d = 0; d < AMAX; ++d) {
            diffv[d] = 0.0f; varv[d] = 1.0f;
            if (d < D) {                                   // ppo_continuous_action.py:134-141 via torch normal.py (loss.hip loss_normal_main)
                float mu = out[d];
                if (a.mean_shift) mu = mu + a.mean_shift[m * D + d];
                const float sd = expf(a.logstd[d]);
                const float diff = bact[d] - mu;
                const float var = sd * sd;
                const float log_scale = logf(sd);
                float t = -(diff * diff);
                t = t / (2.0f * var);
                t = t - log_scale;
                t = t - MI355_LOG_SQRT_2PI;
                lp += t;
                ent += MI355_HALF_LOG_2PIE + log_scale;
                diffv[d] = diff; varv[d] = var;
            }
        }
        const RowTerms t = ppo_row_terms(lp, ent, 0.0f, old_lp, adv, 0.0f, 0.0f, amean, aden, a.P);
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) rs[k] = k == 1 ? 0.0f : t.sums[k];
        const float g_ent = -(a.P.ent_coef / (float)a.P.M);
#pragma unroll
        for (int d = 0; d < AMAX; ++d) {
            if (d < D) {
                dout[d] = t.g_lp * (diffv[d] / varv[d]);
                cls[d] = t.g_lp * ((diffv[d] * diffv[d]) / varv[d] - 1.0f) + g_ent;
            }
        }
    }
    if (!valid) {
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) rs[k] = 0.0f;
#pragma unroll
        for (int o = 0; o < AMAX; ++o) { dout[o] = 0.0f; cls[o] = 0.0f; }
    }
    // the block's scalar partials (f64, lane order) and the output bias gradient (column sums of dout)
    {
        double* srow = a.sums + (int64_t)blockIdx.x * a.sum_stride;
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) {
            const double w = wave_sum((double)rs[k]);
            if (lane == 0) srow[k] = w;
        }
#pragma unroll
        for (int d = 0; d < AMAX; ++d) {
            const double w = wave_sum((double)cls[d]);
            if (lane == 0 && d < a.nout) srow[kNumSums + d] = (NORMAL && netid) ? w : 0.0;
        }
    }
    float db3[AMAX];
#pragma unroll
    for (int o = 0; o < AMAX; ++o) db3[o] = wave_sum(dout[o]);

    // ---- backward, lane = hidden unit ------------------------------------------------------------------------------------------
    float* part = a.part[netid] + (int64_t)blk * mlp_part_size(O, nout);
    const int base2 = O * kH + kH, base3 = base2 + kH * kH + kH;
    float w2t[kH];                                   // column `lane` of W2: requested now, used by the second backward loop
#pragma unroll
    for (int q = 0; q < kH; ++q) w2t[q] = n.w2[q * kH + lane];
    {
        float w3c[AMAX], dw3[AMAX], dw2[kH];
#pragma unroll
        for (int o = 0; o < AMAX; ++o) { w3c[o] = o < nout ? n.w3[o * kH + lane] : 0.0f; dw3[o] = 0.0f; }
#pragma unroll
        for (int k = 0; k < kH; ++k) dw2[k] = 0.0f;
        float db2 = 0.0f;
#pragma unroll 2
        for (int r = 0; r < nrows; ++r) {
            const float h2 = H2[r * kH2P + lane];
            float dh2 = 0.0f;
#pragma unroll
            for (int o = 0; o < AMAX; ++o) {
                const float g = readlane_f(dout[o], r);
                dh2 = fmaf(w3c[o], g, dh2);
                dw3[o] = fmaf(g, h2, dw3[o]);
            }
            const float dz2 = dh2 * (1.0f - h2 * h2);          // tanh_backward: grad * (1 - y * y)
            db2 += dz2;
            H2[r * kH2P + lane] = dz2;                          // the row's slot now carries dz2 for the second loop
            const float4* h4 = reinterpret_cast<const float4*>(H1 + r * kH);
#pragma unroll
            for (int q = 0; q < kH / 4; ++q) {
                const float4 v = h4[q];
                dw2[4 * q + 0] = fmaf(dz2, v.x, dw2[4 * q + 0]);
                dw2[4 * q + 1] = fmaf(dz2, v.y, dw2[4 * q + 1]);
                dw2[4 * q + 2] = fmaf(dz2, v.z, dw2[4 * q + 2]);
                dw2[4 * q + 3] = fmaf(dz2, v.w, dw2[4 * q + 3]);
            }
        }
#pragma unroll
        for (int k = 0; k < kH; ++k) part[base2 + k * kH + lane] = dw2[k];
        part[base2 + kH * kH + lane] = db2;
#pragma unroll
        for (int o = 0; o < AMAX; ++o) {
            if (o < nout) {
                part[base3 + o * kH + lane] = dw3[o];
                if (lane == o) part[base3 + nout * kH + o] = db3[o];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if constexpr (WIDE) {
        // dz1 of every row first (parked in H1's slot of the row: h1 is not needed again), then dW1 chunk by chunk: 32 accumulators against the rows' dz1
        float db1 = 0.0f;
#pragma unroll 2
        for (int r = 0; r < nrows; ++r) {
            const float4* z4 = reinterpret_cast<const float4*>(H2 + r * kH2P);
            float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
#pragma unroll
            for (int q = 0; q < kH / 4; ++q) {
                const float4 v = z4[q];
                d0 = fmaf(w2t[4 * q + 0], v.x, d0);
                d1 = fmaf(w2t[4 * q + 1], v.y, d1);
                d2 = fmaf(w2t[4 * q + 2], v.z, d2);
                d3 = fmaf(w2t[4 * q + 3], v.w, d3);
            }
            const float h1 = H1[r * kH + lane];
            const float dz1 = ((d0 + d1) + (d2 + d3)) * (1.0f - h1 * h1);
            db1 += dz1;
            H1[r * kH + lane] = dz1;                            // (the lane's own element: no other lane reads H1 from here on)
        }
        part[O * kH + lane] = db1;
        for (int c0 = 0; c0 < OP; c0 += 32) {
            float dw1[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) dw1[k] = 0.0f;
#pragma unroll 2
            for (int r = 0; r < nrows; ++r) {
                const float dz1 = H1[r * kH + lane];
                const float4* x4 = reinterpret_cast<const float4*>(xs + r * OP + c0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 v = x4[q];
                    dw1[4 * q + 0] = fmaf(dz1, v.x, dw1[4 * q + 0]);
                    dw1[4 * q + 1] = fmaf(dz1, v.y, dw1[4 * q + 1]);
                    dw1[4 * q + 2] = fmaf(dz1, v.z, dw1[4 * q + 2]);
                    dw1[4 * q + 3] = fmaf(dz1, v.w, dw1[4 * q + 3]);
                }
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) if (c0 + k < O) part[(c0 + k) * kH + lane] = dw1[k];
        }
    } else {
        float dw1[OMAX];
#pragma unroll
        for (int k = 0; k < OMAX; ++k) dw1[k] = 0.0f;
        float db1 = 0.0f;
#pragma unroll 2
        for (int r = 0; r < nrows; ++r) {
            const float4* z4 = reinterpret_cast<const float4*>(H2 + r * kH2P);
            float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
#pragma unroll
            for (int q = 0; q < kH / 4; ++q) {
                const float4 v = z4[q];
                d0 = fmaf(w2t[4 * q + 0], v.x, d0);
                d1 = fmaf(w2t[4 * q + 1], v.y, d1);
                d2 = fmaf(w2t[4 * q + 2], v.z, d2);
                d3 = fmaf(w2t[4 * q + 3], v.w, d3);
            }
            const float h1 = H1[r * kH + lane];
            const float dz1 = ((d0 + d1) + (d2 + d3)) * (1.0f - h1 * h1);
            db1 += dz1;
            const float4* x4 = reinterpret_cast<const float4*>(xs + r * OMAX);
#pragma unroll
            for (int q = 0; q < OMAX / 4; ++q) {
                const float4 v = x4[q];
                dw1[4 * q + 0] = fmaf(dz1, v.x, dw1[4 * q + 0]);
                dw1[4 * q + 1] = fmaf(dz1, v.y, dw1[4 * q + 1]);
                dw1[4 * q + 2] = fmaf(dz1, v.z, dw1[4 * q + 2]);
                dw1[4 * q + 3] = fmaf(dz1, v.w, dw1[4 * q + 3]);
            }
        }
#pragma unroll
        for (int k = 0; k < OMAX; ++k) if (k < O) part[k * kH + lane] = dw1[k];
        part[O * kH + lane] = db1;
    }
}

// ---- fold: partials of all blocks -> gradients (+=), scalars, dlogstd -----------------------------------------------------------
struct MlpFoldArgs {
    const float* part[2];
    MlpGrads g[2];
    int O, nout, nblocks, M;
    const double* sums;
    int sum_stride;
    float ent_coef, vf_coef;
    float* scalars7;               // (7) loss, pg_loss, v_loss, entropy, old_approx_kl, approx_kl, clipfrac
    float* dlogstd;                // Normal: (D) gradient of actor_logstd (+=), else NULL
    int param_blocks;              // workgroups that fold parameters; the one after them folds the scalars
};

__device__ __forceinline__ float* mlp_grad_addr(const MlpGrads& g, int p, int O, int nout) {
    // internal order -> torch layout
    if (p < O * kH) { const int k = p / kH, j = p - k * kH; return g.w1 + j * O + k; }
    p -= O * kH;
    if (p < kH) return g.b1 + p;
    p -= kH;
    if (p < kH * kH) { const int k = p / kH, j = p - k * kH; return g.w2 + j * kH + k; }
    p -= kH * kH;
    if (p < kH) return g.b2 + p;
    p -= kH;
    if (p < nout * kH) return g.w3 + p;
    p -= nout * kH;
    return g.b3 + p;
}

__global__ __launch_bounds__(256) void mlp_fold_kernel(const MlpFoldArgs a) {
    const int Pc = mlp_part_size(a.O, 1), Pa = mlp_part_size(a.O, a.nout);
    if ((int)blockIdx.x < a.param_blocks) {
        // 64 consecutive parameters x 4 interleaved block groups per workgroup: thread (q, j) adds the partials of blocks
        // b = q, q + 4, ... of parameter 64 * blockIdx + j, eight loads in flight (a partial row is 22 KB away from the next:
        // every load is its own round trip); the four group sums meet in LDS and are added in group order.  Fixed order: deterministic.
        __shared__ double grp[4][64];
        const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
        const int t = blockIdx.x * 64 + j;
        const bool live = t < Pc + Pa;
        const int netid = live && t >= Pc;
        const int p = live ? (netid ? t - Pc : t) : 0;
        const int P = netid ? Pa : Pc;
        const float* src = a.part[netid] + p;
        double s = 0.0;
        int b = q;
        for (; b + 28 < a.nblocks; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + 4 * u) * P];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; b < a.nblocks; b += 4) s += (double)src[(int64_t)b * P];
        grp[q][j] = s;
        __syncthreads();
        if (q == 0 && live) {
            float* dst = mlp_grad_addr(a.g[netid], p, a.O, netid ? a.nout : 1);
            *dst = *dst + (float)(((grp[0][j] + grp[1][j]) + grp[2][j]) + grp[3][j]);
        }
        return;
    }
    // scalars (loss.hip loss_finalize's formulas) and dlogstd: one workgroup, rows of `sums` in index order
    __shared__ double red[4];
    __shared__ double tot[kNumSums + 64];
    const int nrows = 2 * a.nblocks, nextra = a.dlogstd ? a.nout : 0;
    for (int k = 0; k < kNumSums + nextra; ++k) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nrows; b += 256) s += a.sums[(int64_t)b * a.sum_stride + k];
        const double r = block_sum<4>(s, red);
        if (threadIdx.x == 0) tot[k] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0 && a.scalars7) {
        const double nn = (double)a.M;
        const float pg_loss = (float)(tot[0] / nn);
        const float v_loss = 0.5f * (float)(tot[1] / nn);
        const float entropy = (float)(tot[2] / nn);
        float loss = pg_loss - a.ent_coef * entropy;
        loss = loss + v_loss * a.vf_coef;
        a.scalars7[0] = loss;
        a.scalars7[1] = pg_loss;
        a.scalars7[2] = v_loss;
        a.scalars7[3] = entropy;
        a.scalars7[4] = (float)(tot[3] / nn);
        a.scalars7[5] = (float)(tot[4] / nn);
        a.scalars7[6] = (float)(tot[5] / nn);
    }
    if ((int)threadIdx.x < nextra) a.dlogstd[threadIdx.x] = a.dlogstd[threadIdx.x] + (float)tot[kNumSums + threadIdx.x];
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static inline int pick_rows(int64_t rows, int want, int target_blocks, int min_rows = 4) {
    int R = want > 0 ? want : (int)((rows + target_blocks - 1) / target_blocks);
    if (R < min_rows) R = min_rows;
    if (R > 64) R = 64;
    return R;
}
static inline size_t act_lds_bytes(int R, int OMAX) { return (size_t)R * (OMAX + kH + kH2P) * sizeof(float); }
// WIDE (observation width above 32 or more than 8 outputs): rows of pitch OP = O rounded up to 32; at most kWideRows rows per block and 64 KB of LDS
constexpr int kMlpMaxObs = 512, kMlpMaxOut = 20;
static inline bool mlp_wide(int O, int nout) { return O > 32 || nout > 8; }
static inline int wide_pitch(int O) { return (O + 31) / 32 * 32; }
static inline int wide_rows(int O, int R) {
    const int fit = (int)(65536 / ((size_t)(wide_pitch(O) + kH + kH2P) * sizeof(float)));
    const int cap = fit < kWideRows ? fit : kWideRows;
    return R < cap ? R : cap;
}
static inline int sums_extra(int nout) { return nout > 8 ? kMlpMaxOut : 8; }

static int fill_net(const char* fn, const void* const* p, MlpNet& n) {
    MI355_REQUIRE(p, MI355PPO_EINVAL, "%s: null network pointer block", fn);
    for (int k = 0; k < 6; ++k) {
        MI355_REQUIRE(p[k], MI355PPO_EINVAL, "%s: null parameter pointer (entry %d of a network block)", fn, k);
        MI355_REQUIRE(aligned(p[k], 4), MI355PPO_EALIGN, "%s: misaligned parameter pointer", fn);
    }
    n.w1 = (const float*)p[0]; n.b1 = (const float*)p[1]; n.w2 = (const float*)p[2];
    n.b2 = (const float*)p[3]; n.w3 = (const float*)p[4]; n.b3 = (const float*)p[5];
    return MI355PPO_OK;
}

static int check_dims(const char* fn, int O, int nout) {
    MI355_REQUIRE(O > 0 && O <= kMlpMaxObs, MI355PPO_EINVAL, "%s: obs_dim=%d must be in 1..%d", fn, O, kMlpMaxObs);
    MI355_REQUIRE(nout > 0 && nout <= kMlpMaxOut, MI355PPO_EINVAL, "%s: n_out=%d must be in 1..%d", fn, nout, kMlpMaxOut);
    return MI355PPO_OK;
}

#define MI355_MLP_DISPATCH(O, NOUT, FN)                              \
    do {                                                             \
        if ((O) <= 8) { if ((NOUT) <= 4) FN(8, 4); else FN(8, 8); }  \
        else if ((O) <= 20) { if ((NOUT) <= 4) FN(20, 4); else FN(20, 8); } \
        else { if ((NOUT) <= 4) FN(32, 4); else FN(32, 8); }         \
    } while (0)
static inline int omax_for(int O) { return O <= 8 ? 8 : O <= 20 ? 20 : 32; }

template <bool NORMAL, int MODE>
static int act_launch(const char* fn, const MlpActArgs& a0, hipStream_t s) {
    MlpActArgs a = a0;
    if (mlp_wide(a.O, a.nout)) {
        a.R = wide_rows(a.O, a.R);
        const int blocks = (a.B + a.R - 1) / a.R;
        hipLaunchKernelGGL((mlp_act_kernel<32, kMlpMaxOut, NORMAL, MODE, true>), dim3(2 * blocks), dim3(64), act_lds_bytes(a.R, wide_pitch(a.O)), s, a);
        return check_launch(fn);
    }
    const int blocks = (a.B + a.R - 1) / a.R;
    const size_t lds = act_lds_bytes(a.R, omax_for(a.O));
#define LAUNCH(OM, AM) hipLaunchKernelGGL((mlp_act_kernel<OM, AM, NORMAL, MODE>), dim3(2 * blocks), dim3(64), lds, s, a)
    MI355_MLP_DISPATCH(a.O, a.nout, LAUNCH);
#undef LAUNCH
    return check_launch(fn);
}

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API int mi355ppo_mlp_fwd_f32(const float* obs, int B, int O, const void* const* actor, const void* const* critic,
                                                int n_out, float* actor_out, float* value, void* stream) {
    const char* fn = "mi355ppo_mlp_fwd_f32";
    MI355_REQUIRE(obs && actor_out && value, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(B > 0, MI355PPO_EINVAL, "%s: B=%d must be positive", fn, B);
    int rc = check_dims(fn, O, n_out);
    if (rc) return rc;
    MI355_REQUIRE(aligned(obs, 4) && aligned(actor_out, 4) && aligned(value, 4), MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    MlpActArgs a = {};
    if ((rc = fill_net(fn, critic, a.net[0])) || (rc = fill_net(fn, actor, a.net[1]))) return rc;
    a.obs = obs; a.B = B; a.O = O; a.nout = n_out; a.R = pick_rows(B, 0, 256, 1);     // rollout: the launch is a latency chain -- one row per wave when the batch allows
    a.value = value; a.actor_out = actor_out;
    return act_launch<false, 0>(fn, a, as_stream(stream));
}

extern "C" MI355PPO_API int mi355ppo_mlp_act_categorical_f32(const float* obs, int B, int O, const void* const* actor,
                                                            const void* const* critic, int A, const float* noise_exp1, uint64_t seed,
                                                            uint64_t offset, const uint64_t* offset_base, int64_t* action_i64,
                                                            float* action_f32, float* logprob, float* entropy, float* value,
                                                            float* logits_out, void* stream) {
    const char* fn = "mi355ppo_mlp_act_categorical_f32";
    MI355_REQUIRE(obs && logprob && value, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(action_i64 || action_f32, MI355PPO_EINVAL, "%s: no action output", fn);
    MI355_REQUIRE(B > 0, MI355PPO_EINVAL, "%s: B=%d must be positive", fn, B);
    int rc = check_dims(fn, O, A);
    if (rc) return rc;
    MI355_REQUIRE(aligned(obs, 4) && aligned(noise_exp1, 4) && aligned(offset_base, 8) && aligned(action_i64, 8) && aligned(action_f32, 4) &&
                      aligned(logprob, 4) && aligned(entropy, 4) && aligned(value, 4) && aligned(logits_out, 4),
                  MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    MlpActArgs a = {};
    if ((rc = fill_net(fn, critic, a.net[0])) || (rc = fill_net(fn, actor, a.net[1]))) return rc;
    a.obs = obs; a.B = B; a.O = O; a.nout = A; a.R = pick_rows(B, 0, 256, 1);     // rollout: the launch is a latency chain -- one row per wave when the batch allows
    a.noise = noise_exp1; a.seed = seed; a.offset = offset; a.offset_base = offset_base;
    a.action_i64 = action_i64; a.action_f32 = action_f32; a.logprob = logprob; a.entropy = entropy; a.value = value;
    a.actor_out = logits_out;
    return act_launch<false, 1>(fn, a, as_stream(stream));
}

extern "C" MI355PPO_API int mi355ppo_mlp_act_normal_f32(const float* obs, int B, int O, const void* const* actor_mean,
                                                       const void* const* critic, const float* logstd, int D,
                                                       const float* noise_std_normal, uint64_t seed, uint64_t offset,
                                                       const uint64_t* offset_base, float* action, float* logprob_sum,
                                                       float* entropy_sum, float* value, float* mean_out, void* stream) {
    const char* fn = "mi355ppo_mlp_act_normal_f32";
    MI355_REQUIRE(obs && logstd && action && logprob_sum && value, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(B > 0, MI355PPO_EINVAL, "%s: B=%d must be positive", fn, B);
    int rc = check_dims(fn, O, D);
    if (rc) return rc;
    MI355_REQUIRE(aligned(obs, 4) && aligned(logstd, 4) && aligned(noise_std_normal, 4) && aligned(offset_base, 8) && aligned(action, 4) &&
                      aligned(logprob_sum, 4) && aligned(entropy_sum, 4) && aligned(value, 4) && aligned(mean_out, 4),
                  MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    MlpActArgs a = {};
    if ((rc = fill_net(fn, critic, a.net[0])) || (rc = fill_net(fn, actor_mean, a.net[1]))) return rc;
    a.obs = obs; a.B = B; a.O = O; a.nout = D; a.R = pick_rows(B, 0, 256, 1);     // rollout: the launch is a latency chain -- one row per wave when the batch allows
    a.logstd = logstd; a.noise = noise_std_normal; a.seed = seed; a.offset = offset; a.offset_base = offset_base;
    a.action_f32 = action; a.logprob = logprob_sum; a.entropy = entropy_sum; a.value = value; a.actor_out = mean_out;
    return act_launch<true, 1>(fn, a, as_stream(stream));
}

extern "C" MI355PPO_API size_t mi355ppo_mlp_ppo_workspace_bytes(int M, int O, int n_out, int rows_per_block) {
    if (M <= 0 || O <= 0 || O > kMlpMaxObs || n_out <= 0 || n_out > kMlpMaxOut) return 0;
    int R = pick_rows(M, rows_per_block, 256);
    if (mlp_wide(O, n_out)) R = wide_rows(O, R);
    const size_t nb = (size_t)((M + R - 1) / R);
    return nb * 2 * (kNumSums + sums_extra(n_out)) * sizeof(double) + nb * ((size_t)mlp_part_size(O, 1) + (size_t)mlp_part_size(O, n_out)) * sizeof(float);
}

static int mlp_ppo_common(const char* fn, bool normal, const float* b_obs, const int64_t* mb_inds, int M, int O, const void* const* actor,
                          const void* const* critic, int nout, const float* logstd, const float* mean_shift, const float* b_actions,
                          const float* b_logprobs, const float* b_advantages, const float* b_returns, const float* b_values,
                          double clip_coef, double ent_coef, double vf_coef, int norm_adv, int clip_vloss, const float* adv_mean_den,
                          void* const* actor_grads, void* const* critic_grads, float* dlogstd, float* scalars7, int rows_per_block,
                          void* workspace, size_t workspace_bytes, void* stream) {
    MI355_REQUIRE(b_obs && b_actions && b_logprobs && b_advantages && b_returns && b_values && scalars7, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(M > 0, MI355PPO_EINVAL, "%s: M=%d must be positive", fn, M);
    int rc = check_dims(fn, O, nout);
    if (rc) return rc;
    MI355_REQUIRE(!normal || (logstd && dlogstd), MI355PPO_EINVAL, "%s: null pointer (logstd / dlogstd)", fn);
    MI355_REQUIRE(!norm_adv || adv_mean_den, MI355PPO_EINVAL,
                  "%s: norm_adv needs adv_mean_den (mi355ppo_adv_stats_f32: one launch per epoch for all its minibatches)", fn);
    MI355_REQUIRE(aligned(b_obs, 4) && aligned(mb_inds, 8) && aligned(b_actions, 4) && aligned(b_logprobs, 4) && aligned(b_advantages, 4) &&
                      aligned(b_returns, 4) && aligned(b_values, 4) && aligned(logstd, 4) && aligned(mean_shift, 4) &&
                      aligned(adv_mean_den, 4) && aligned(dlogstd, 4) && aligned(scalars7, 4),
                  MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    const size_t need = mi355ppo_mlp_ppo_workspace_bytes(M, O, nout, rows_per_block);
    MI355_REQUIRE(workspace && workspace_bytes >= need, MI355PPO_EWORKSPACE, "%s: workspace %zu bytes < required %zu", fn,
                  workspace ? workspace_bytes : (size_t)0, need);
    MI355_REQUIRE(aligned(workspace, 8), MI355PPO_EALIGN, "%s: workspace must be 8-byte aligned", fn);
    MlpPpoArgs a = {};
    if ((rc = fill_net(fn, critic, a.net[0])) || (rc = fill_net(fn, actor, a.net[1]))) return rc;
    MlpFoldArgs f = {};
    MI355_REQUIRE(actor_grads && critic_grads, MI355PPO_EINVAL, "%s: null gradient pointer block", fn);
    for (int k = 0; k < 6; ++k) {
        MI355_REQUIRE(actor_grads[k] && critic_grads[k], MI355PPO_EINVAL, "%s: null gradient pointer", fn);
        MI355_REQUIRE(aligned(actor_grads[k], 4) && aligned(critic_grads[k], 4), MI355PPO_EALIGN, "%s: misaligned gradient pointer", fn);
    }
    void* const* gp[2] = {critic_grads, actor_grads};
    for (int t = 0; t < 2; ++t) {
        f.g[t].w1 = (float*)gp[t][0]; f.g[t].b1 = (float*)gp[t][1]; f.g[t].w2 = (float*)gp[t][2];
        f.g[t].b2 = (float*)gp[t][3]; f.g[t].w3 = (float*)gp[t][4]; f.g[t].b3 = (float*)gp[t][5];
    }
    const bool wide = mlp_wide(O, nout);
    const int R = wide ? wide_rows(O, pick_rows(M, rows_per_block, 256)) : pick_rows(M, rows_per_block, 256);
    const int nb = (M + R - 1) / R;
    a.obs = b_obs; a.inds = mb_inds; a.M = M; a.O = O; a.nout = nout; a.R = R; a.nblocks = nb;
    a.b_actions = b_actions; a.b_logprobs = b_logprobs; a.b_adv = b_advantages; a.b_ret = b_returns; a.b_val = b_values;
    a.logstd = logstd; a.mean_shift = mean_shift; a.adv_mean_den = adv_mean_den;
    a.P.lo = (float)(1.0 - clip_coef); a.P.hi = (float)(1.0 + clip_coef); a.P.clip = (float)clip_coef;
    a.P.ent_coef = (float)ent_coef; a.P.vf_coef = (float)vf_coef; a.P.norm_adv = norm_adv ? 1 : 0; a.P.clip_vloss = clip_vloss ? 1 : 0;
    a.P.M = M; a.P.stats_blocks = 0;
    a.sums = static_cast<double*>(workspace);
    a.sum_stride = kNumSums + sums_extra(nout);
    float* pbase = reinterpret_cast<float*>(a.sums + (size_t)nb * 2 * a.sum_stride);
    a.part[0] = pbase;
    a.part[1] = pbase + (size_t)nb * mlp_part_size(O, 1);
    hipStream_t s = as_stream(stream);
    const size_t lds = act_lds_bytes(R, wide ? wide_pitch(O) : omax_for(O));
    if (wide) {
        if (normal) hipLaunchKernelGGL((mlp_ppo_kernel<32, kMlpMaxOut, true, true>), dim3(2 * nb), dim3(64), lds, s, a);
        else hipLaunchKernelGGL((mlp_ppo_kernel<32, kMlpMaxOut, false, true>), dim3(2 * nb), dim3(64), lds, s, a);
    } else if (normal) {
#define LAUNCH(OM, AM) hipLaunchKernelGGL((mlp_ppo_kernel<OM, AM, true>), dim3(2 * nb), dim3(64), lds, s, a)
        MI355_MLP_DISPATCH(O, nout, LAUNCH);
#undef LAUNCH
    } else {
#define LAUNCH(OM, AM) hipLaunchKernelGGL((mlp_ppo_kernel<OM, AM, false>), dim3(2 * nb), dim3(64), lds, s, a)
        MI355_MLP_DISPATCH(O, nout, LAUNCH);
#undef LAUNCH
    }
    rc = check_launch("mlp_ppo_kernel");
    if (rc) return rc;
    f.part[0] = a.part[0]; f.part[1] = a.part[1];
    f.O = O; f.nout = nout; f.nblocks = nb; f.M = M; f.sums = a.sums; f.sum_stride = a.sum_stride;
    f.ent_coef = a.P.ent_coef; f.vf_coef = a.P.vf_coef; f.scalars7 = scalars7; f.dlogstd = normal ? dlogstd : nullptr;
    f.param_blocks = (mlp_part_size(O, 1) + mlp_part_size(O, nout) + 63) / 64;
    hipLaunchKernelGGL(mlp_fold_kernel, dim3(f.param_blocks + 1), dim3(256), 0, s, f);
    return check_launch(fn);
}

extern "C" MI355PPO_API int mi355ppo_mlp_ppo_categorical_fwd_bwd_f32(
    const float* b_obs, const int64_t* mb_inds, int M, int O, const void* const* actor, const void* const* critic, int A,
    const float* b_actions_f32, const float* b_logprobs, const float* b_advantages, const float* b_returns, const float* b_values,
    double clip_coef, double ent_coef, double vf_coef, int norm_adv, int clip_vloss, const float* adv_mean_den, void* const* actor_grads,
    void* const* critic_grads, float* scalars7, int rows_per_block, void* workspace, size_t workspace_bytes, void* stream) {
    return mlp_ppo_common("mi355ppo_mlp_ppo_categorical_fwd_bwd_f32", false, b_obs, mb_inds, M, O, actor, critic, A, nullptr, nullptr,
                          b_actions_f32, b_logprobs, b_advantages, b_returns, b_values, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss,
                          adv_mean_den, actor_grads, critic_grads, nullptr, scalars7, rows_per_block, workspace, workspace_bytes, stream);
}

extern "C" MI355PPO_API int mi355ppo_mlp_ppo_normal_fwd_bwd_f32(
    const float* b_obs, const int64_t* mb_inds, int M, int O, const void* const* actor_mean, const void* const* critic, const float* logstd,
    int D, const float* mean_shift, const float* b_actions, const float* b_logprobs, const float* b_advantages, const float* b_returns,
    const float* b_values, double clip_coef, double ent_coef, double vf_coef, int norm_adv, int clip_vloss, const float* adv_mean_den,
    void* const* actor_grads, void* const* critic_grads, float* dlogstd, float* scalars7, int rows_per_block, void* workspace,
    size_t workspace_bytes, void* stream) {
    return mlp_ppo_common("mi355ppo_mlp_ppo_normal_fwd_bwd_f32", true, b_obs, mb_inds, M, O, actor_mean, critic, D, logstd, mean_shift,
                          b_actions, b_logprobs, b_advantages, b_returns, b_values, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss,
                          adv_mean_den, actor_grads, critic_grads, dlogstd, scalars7, rows_per_block, workspace, workspace_bytes, stream);
}
