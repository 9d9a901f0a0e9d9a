// K3 -- PPO minibatch loss: clipped surrogate + (clipped) value loss + entropy bonus, forward AND
// backward in one pass over the minibatch (gfx950).
//
// Replaces cleanrl/ppo_atari_multigpu.py:320-355 (5 index-gathers, exp, 3 KL statistics, advantage
// mean/std normalisation, clamp/max/mean chains: ~45 small launches) plus the autograd backward of
// all of them down to the network outputs (~as many again) and the `.item()` sync of :328.
//
// The math is reduce -> map -> reduce; on MI355X a kernel boundary (~3 us end to end for a kernel of one memory round
// trip) is CHEAPER than any in-kernel substitute: a persistent single-launch version with a ticket barrier between
// the phases and a last-workgroup finish measured 12.8 us at 16 workgroups, 18 us at 128 and 59 us at 512
// (profiles/r02_k3_grid_barrier_attempt_events.jsonl) -- every agent-scope round trip (atomics and loads that must
// bypass the per-XCD L2s) costs 1.4-2 us and the polled word serialises.  So the phases are launches, and the two
// reductions can be hoisted out of the per-minibatch call by the caller:
//   1. loss_adv_stats   : f64 sum / sum of squares of b_advantages[mb_inds] -> <= kStatsMaxBlocks partials.
//                         SKIPPED when the caller passes `adv_mean_den` (mi355ppo_adv_stats_f32 computes the pairs of all
//                         minibatches of an epoch in two launches as soon as the permutation is on the device: they
//                         depend on nothing the network produces).
//   2. loss_*_main      : persistent grid (<= kMaxGrid workgroups), one row per lane and sweep.  A lane loads its
//                         row's index, then all seven operand loads go out together (branch-free: 16-byte logits
//                         rows at A = 4); the distribution, the three loss terms and their closed-form gradients;
//                         dlogits (16-byte stores) and dvalue written; 6 (+D) f64 partials per workgroup.
//   3. loss_finalize    : one workgroup folds the partials in index order into the 7 scalars (+ dlogstd).  DEFERRED when
//                         the caller passes scalars7 == NULL (categorical family): the partials stay in the call's
//                         workspace slot and mi355ppo_loss_scalars_f32 folds the slots of a whole update in one launch
//                         (the scalars are diagnostics; nothing on the device waits for them).
// No host sync, fixed-order (deterministic) reductions, capturable.
//
// HBM traffic == algorithmic bytes: logits are read once and dlogits written once ((8A+28)*M bytes,
// +8*M for mb_inds); the five (Bflat) arrays are gathered 4 bytes at a time: L2-resident at the PPO configs
// (5 x 512 KB at 1024 envs x 128 steps); for Bflat >> L2 every gather moves a whole 128-byte line and THAT
// traffic (5 lines per row), not the algorithmic bytes, bounds the kernel.
//
// Gradient of torch.max(a, b) at a == b is split 1/2 + 1/2 (derivatives.yaml `maximum`), and
// torch.clamp passes gradient on the closed interval; both are reproduced.
#include "common.h"
#include "catrow.h"
#include "ppo_rows.h"

#pragma clang fp contract(off)

namespace mi355ppo {

constexpr int kStatsMaxBlocks = 1024;   // partial pairs of the advantage statistics (256 lanes x 4 rows per sweep each)
constexpr int kMaxGrid = 2048;          // workgroups of the persistent row pass
constexpr int kMaxD = 64;
// Packed behaviour rows: one 32-byte row per flat batch index, {action (f32 storage), old log-prob, advantage, return, old
// value, 0, 0, 0}.  A minibatch row then costs ONE 32-byte gather instead of five 4-byte gathers out of five arrays (five
// 128-byte lines per row once the flat batch outgrows the caches: DESIGN.md section 3.1).
constexpr int kPackFloats = 8, kPackAdv = 2;

struct LossSlot {          // 64-byte head of a workspace: what the scalar fold needs to know about the call that filled it
    int M, nblocks, D, pad;
    float ent_coef, vf_coef;
    int pad2[10];
};

// ---- 1. advantage statistics ---------------------------------------------------------------------
// UN independent rows per lane and sweep (index loads, then gathers, all in flight together).
// `b_adv` element i lives at b_adv[i * STRIDE]: STRIDE = 1 for the (B) advantage array, kPackFloats for packed behaviour
// rows (the caller passes pack + kPackAdv).
template <int UN, int STRIDE = 1>
__device__ __forceinline__ void adv_sums(const int64_t* __restrict__ inds, const float* __restrict__ b_adv, int64_t M,
                                         int64_t first, int64_t step, double& s, double& ss) {
    for (int64_t m = first; m < M; m += UN * step) {
        int64_t i[UN];
        float a[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t mu = m + u * step;
            const int64_t mc = mu < M ? mu : M - 1;
            i[u] = inds ? inds[mc] : mc;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) a[u] = b_adv[i[u] * STRIDE];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const double d = (m + u * step < M) ? (double)a[u] : 0.0;
            s += d;
            ss += d * d;
        }
    }
}

__global__ __launch_bounds__(256) void loss_adv_stats(const int64_t* __restrict__ inds,
                                                      const float* __restrict__ b_adv, int M,
                                                      double* __restrict__ partials) {
    __shared__ double red[4];
    double s = 0.0, ss = 0.0;
    adv_sums<4>(inds, b_adv, M, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256, s, ss);
    const double bs = block_sum<4>(s, red);
    const double bss = block_sum<4>(ss, red);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = bs;
        partials[2 * blockIdx.x + 1] = bss;
    }
}

// (mean, std + 1e-8) pairs of `nseg` consecutive minibatches of `M` rows each (the last may be shorter: `total` rows in
// all) of one permutation: workgroup (x, y) sums rows x*256 + lane + k*gridDim.x*256 of segment y, four in flight per lane;
// adv_stats_fold folds a segment's gridDim.x partial pairs in index order.
template <int STRIDE>
__global__ __launch_bounds__(256) void adv_stats_partials(const int64_t* __restrict__ inds, const float* __restrict__ b_adv,
                                                          int M, int64_t total, double* __restrict__ partials) {
    __shared__ double red[4];
    const int64_t lo = (int64_t)blockIdx.y * M;
    const int64_t n = (total - lo) < (int64_t)M ? (total - lo) : (int64_t)M;
    double s = 0.0, ss = 0.0;
    adv_sums<4, STRIDE>(inds ? inds + lo : nullptr, inds ? b_adv : b_adv + lo * STRIDE, n, (int64_t)blockIdx.x * 256 + threadIdx.x,
                        (int64_t)gridDim.x * 256, s, ss);
    const double bs = block_sum<4>(s, red);
    const double bss = block_sum<4>(ss, red);
    if (threadIdx.x == 0) {
        double* p = partials + 2 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
        p[0] = bs;
        p[1] = bss;
    }
}

__global__ __launch_bounds__(256) void adv_stats_fold(const double* __restrict__ partials, int per_seg, int M, int64_t total,
                                                      float* __restrict__ out) {
    __shared__ double red[4];
    const int64_t lo = (int64_t)blockIdx.x * M;
    const int64_t n = (total - lo) < (int64_t)M ? (total - lo) : (int64_t)M;
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < per_seg; b += 256) {
        s += partials[2 * ((int64_t)blockIdx.x * per_seg + b)];
        ss += partials[2 * ((int64_t)blockIdx.x * per_seg + b) + 1];
    }
    const double fs = block_sum<4>(s, red);
    const double fss = block_sum<4>(ss, red);
    if (threadIdx.x == 0) mean_den_from_sums(fs, fss, (double)n, out + 2 * blockIdx.x, out + 2 * blockIdx.x + 1);
}

// mean and (std + 1e-8) of the minibatch advantages, identical in every workgroup: either the caller's pair or the fold
// of the <= kStatsMaxBlocks partials of loss_adv_stats (<= 16 KiB from L2 per workgroup of the persistent grid).
__device__ __forceinline__ void fold_adv_stats(const double* __restrict__ partials, const float* __restrict__ given,
                                               const LossParams& P, double* lds4, float* s_pub, float& mean, float& den) {
    if (given) {
        mean = given[0];
        den = given[1];
        return;
    }
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < P.stats_blocks; b += 256) {
        s += partials[2 * b];
        ss += partials[2 * b + 1];
    }
    const double fs = block_sum<4>(s, lds4);
    const double fss = block_sum<4>(ss, lds4);
    if (threadIdx.x == 0) mean_den_from_sums(fs, fss, (double)P.M, s_pub, s_pub + 1);
    __syncthreads();
    mean = s_pub[0];
    den = s_pub[1];
}

// Block-reduce the lane sums into this workgroup's partial (fixed order).  `extra` = D more per-wave f64 sums already in
// red[wave][kNumSums + d] (normal family), or 0.
__device__ __forceinline__ void emit_block_sums(const double (&sums)[kNumSums], double* __restrict__ part, int stride,
                                                int extra, double (*red)[kNumSums + kMaxD]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) {
        const double w = wave_sum(sums[k]);
        if (lane == 0) red[wave][k] = w;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kNumSums + extra; k += 256)
        part[(int64_t)blockIdx.x * stride + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
}

__device__ __forceinline__ void write_slot_head(LossSlot* slot, const LossParams& P, int D) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        slot->M = P.M;
        slot->nblocks = (int)gridDim.x;
        slot->D = D;
        slot->ent_coef = P.ent_coef;
        slot->vf_coef = P.vf_coef;
    }
}

// ---- 2a. categorical ----------------------------------------------------------------------------------------------
// U rows of one lane, everything phase 2 needs, in registers.  Loads are branch-free: rows past M read row M-1 (their
// results are discarded), so all U x 7 loads of a lane are issued back to back behind the U index loads.
template <int AMAX, int U>
struct CatRows {
    float x[U][AMAX];
    float v[U], old_lp[U], adv[U], ret[U], old_v[U], act[U];
};

// PACKED: `b_actions` points at the packed behaviour rows (kPackFloats per flat index) and the other four pointers are unused.
template <int AMAX, int U, bool VEC, bool PACKED = false>
__device__ __forceinline__ void cat_load(CatRows<AMAX, U>& r, int64_t mfirst, int64_t S, const float* __restrict__ logits,
                                         const float* __restrict__ value, const int64_t* __restrict__ inds,
                                         const float* __restrict__ b_actions, const float* __restrict__ b_logprobs,
                                         const float* __restrict__ b_adv, const float* __restrict__ b_ret,
                                         const float* __restrict__ b_val, int A, int M) {
    int64_t mc[U], i[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t m = mfirst + u * S;
        mc[u] = m < (int64_t)M ? m : (int64_t)M - 1;
    }
    if (inds) {
#pragma unroll
        for (int u = 0; u < U; ++u) i[u] = inds[mc[u]];
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) i[u] = mc[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (PACKED) {                         // one 32-byte row: a 16-byte and a 4-byte load of the same 32-byte sector
            const float* prow = b_actions + i[u] * kPackFloats;
            const float4 q = *reinterpret_cast<const float4*>(prow);
            r.act[u] = q.x; r.old_lp[u] = q.y; r.adv[u] = q.z; r.ret[u] = q.w;
            r.old_v[u] = prow[4];
        } else {
            r.act[u] = b_actions[i[u]];       // b_actions.long()[mb_inds]  (:320), truncated in cat_rows
            r.old_lp[u] = b_logprobs[i[u]];
            r.adv[u] = b_adv[i[u]];
            r.ret[u] = b_ret[i[u]];
            r.old_v[u] = b_val[i[u]];
        }
        r.v[u] = value[mc[u]];
        const float* row = logits + mc[u] * A;
        if (VEC) {                            // A == AMAX == 4, 16-byte aligned base (host-checked)
            const float4 q = *reinterpret_cast<const float4*>(row);
            r.x[u][0] = q.x; r.x[u][1] = q.y; r.x[u][2] = q.z; r.x[u][3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < AMAX; ++j) {
                const float t = row[j < A ? j : A - 1];
                r.x[u][j] = j < A ? t : -INFINITY;
            }
        }
    }
}

template <int AMAX, int U, bool VEC>
__device__ __forceinline__ void cat_rows(const CatRows<AMAX, U>& r, int64_t mfirst, int64_t S, int A, float mean, float den,
                                         const LossParams& P, float* __restrict__ dlogits, float* __restrict__ dvalue,
                                         double (&sums)[kNumSums]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t m = mfirst + u * S;
        const bool valid = m < (int64_t)P.M;
        const int a = (int)r.act[u];
        CatRow<AMAX> c;
        categorical_row<AMAX>(r.x[u], A, c);
        float newlp = 0.0f;
#pragma unroll
        for (int j = 0; j < AMAX; ++j) if (j == a) newlp = c.lp[j];
        const RowTerms t = ppo_row_terms(newlp, c.H, r.v[u], r.old_lp[u], r.adv[u], r.ret[u], r.old_v[u], mean, den, P);
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) sums[k] += valid ? (double)t.sums[k] : 0.0;
        // d loss/d logits_j = g_lp*(1[j==a] - p_j) + (ent_coef/M) * p_j * (lp_j + H)
        const float ge = P.ent_coef / (float)P.M;
        float g[AMAX];
#pragma unroll
        for (int j = 0; j < AMAX; ++j) {
            const float onehot = (j == a) ? 1.0f : 0.0f;
            const float lpj = fmaxf(c.lp[j], -FLT_MAX);
            g[j] = t.g_lp * (onehot - c.p[j]) + ge * (c.p[j] * (lpj + c.H));
        }
        if (valid) {
            dvalue[m] = t.dvalue;
            float* out = dlogits + m * A;
            if (VEC) {
                *reinterpret_cast<float4*>(out) = make_float4(g[0], g[1], g[2], g[3]);
            } else {
#pragma unroll
                for (int j = 0; j < AMAX; ++j) if (j < A) out[j] = g[j];
            }
        }
    }
}

// Lane's rows: m = blockIdx*256 + thread + k * (grid*256).
template <int AMAX, bool VEC, bool PACKED = false>
__global__ __launch_bounds__(256) void loss_categorical_main(
    const float* __restrict__ logits, const float* __restrict__ value, const int64_t* __restrict__ inds,
    const float* __restrict__ b_actions, const float* __restrict__ b_logprobs, const float* __restrict__ b_adv,
    const float* __restrict__ b_ret, const float* __restrict__ b_val, int A, LossParams P,
    const double* __restrict__ stats_partials, const float* __restrict__ adv_mean_den, LossSlot* __restrict__ slot,
    double* __restrict__ block_partials, float* __restrict__ dlogits, float* __restrict__ dvalue) {
    __shared__ double red[4][kNumSums + kMaxD];
    __shared__ double lds4[4];
    __shared__ float s_pub[2];
    const int64_t S = (int64_t)gridDim.x * 256;
    const int64_t m0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double sums[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) sums[k] = 0.0;
    CatRows<AMAX, 1> r;
    // the first row's loads are issued before the statistics are folded: the fold's round trip hides behind them
    cat_load<AMAX, 1, VEC, PACKED>(r, m0, S, logits, value, inds, b_actions, b_logprobs, b_adv, b_ret, b_val, A, P.M);
    float mean = 0.0f, den = 1.0f;
    if (P.norm_adv) fold_adv_stats(stats_partials, adv_mean_den, P, lds4, s_pub, mean, den);
    cat_rows<AMAX, 1, VEC>(r, m0, S, A, mean, den, P, dlogits, dvalue, sums);
    for (int64_t m = m0 + S; m < (int64_t)P.M; m += S) {
        cat_load<AMAX, 1, VEC, PACKED>(r, m, S, logits, value, inds, b_actions, b_logprobs, b_adv, b_ret, b_val, A, P.M);
        cat_rows<AMAX, 1, VEC>(r, m, S, A, mean, den, P, dlogits, dvalue, sums);
    }
    emit_block_sums(sums, block_partials, kNumSums, 0, red);
    write_slot_head(slot, P, 0);
}

// ---- 2a'. packed behaviour rows ------------------------------------------------------------------------------------
// pack[i] = {actions[i], logprobs[i], advantages[i], returns[i], values[i], 0, 0, 0}: five coalesced 4-byte loads and two
// 16-byte stores per lane; 52 bytes per row.  Once per iteration, after GAE (the five arrays are final then).
__global__ __launch_bounds__(256) void batch_pack_kernel(const float* __restrict__ b_actions, const float* __restrict__ b_logprobs,
                                                         const float* __restrict__ b_adv, const float* __restrict__ b_ret,
                                                         const float* __restrict__ b_val, float* __restrict__ pack, int64_t B) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) {
        const float4 q = make_float4(b_actions[i], b_logprobs[i], b_adv[i], b_ret[i]);
        const float v = b_val[i];
        float4* row = reinterpret_cast<float4*>(pack + i * kPackFloats);
        row[0] = q;
        row[1] = make_float4(v, 0.0f, 0.0f, 0.0f);
    }
}

// ---- 2b. normal main -----------------------------------------------------------------------------
#define MI355_LOG_SQRT_2PI 0.91893853320467274178f
#define MI355_HALF_LOG_2PIE 1.4189385332046727418f

__global__ __launch_bounds__(256) void loss_normal_main(
    const float* __restrict__ mean_in, const float* __restrict__ logstd, const float* __restrict__ value,
    const int64_t* __restrict__ inds, const float* __restrict__ b_actions, const float* __restrict__ b_logprobs,
    const float* __restrict__ b_adv, const float* __restrict__ b_ret, const float* __restrict__ b_val, int D,
    LossParams P, const double* __restrict__ stats_partials, const float* __restrict__ adv_mean_den, LossSlot* __restrict__ slot,
    double* __restrict__ block_partials, float* __restrict__ dmean, float* __restrict__ dvalue) {
    __shared__ double red[4][kNumSums + kMaxD];
    __shared__ double lds4[4];
    __shared__ float s_pub[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = kNumSums + D;
    const int64_t S = (int64_t)gridDim.x * 256;
    const int64_t m0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int R = (int)(((int64_t)P.M + S - 1) / S);
    float amean = 0.0f, den = 1.0f;
    if (P.norm_adv) fold_adv_stats(stats_partials, adv_mean_den, P, lds4, s_pub, amean, den);
    double sums[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) sums[k] = 0.0;
    if (lane == 0)
        for (int d = 0; d < D; ++d) red[wave][kNumSums + d] = 0.0;   // per-wave dlogstd accumulators (only lane 0 touches them)
    const float g_ent = -(P.ent_coef / (float)P.M);   // d loss / d entropy_row ; d entropy_row / d logstd_d = 1
    for (int k = 0; k < R; ++k) {                      // uniform trip count: wave_sum needs every lane
        const int64_t m = m0 + k * S;
        const bool active = m < (int64_t)P.M;
        int64_t i = 0;
        float g_lp = 0.0f;
        if (active) {
            i = inds ? inds[m] : m;
            float lp = 0.0f, ent = 0.0f;
            for (int d = 0; d < D; ++d) {       // ppo_continuous_action.py:134-141 via torch normal.py
                const float mu = mean_in[m * D + d];
                const float sd = expf(logstd[d]);
                const float a = b_actions[i * D + d];
                const float diff = a - mu;
                const float var = sd * sd;
                const float log_scale = logf(sd);
                float t = -(diff * diff);
                t = t / (2.0f * var);
                t = t - log_scale;
                t = t - MI355_LOG_SQRT_2PI;
                lp += t;
                ent += MI355_HALF_LOG_2PIE + log_scale;
            }
            const RowTerms t = ppo_row_terms(lp, ent, value[m], b_logprobs[i], b_adv[i], b_ret[i], b_val[i], amean, den, P);
#pragma unroll
            for (int q = 0; q < kNumSums; ++q) sums[q] += (double)t.sums[q];
            dvalue[m] = t.dvalue;
            g_lp = t.g_lp;
        }
        for (int d = 0; d < D; ++d) {
            float contrib = 0.0f;
            if (active) {
                const float mu = mean_in[m * D + d];
                const float sd = expf(logstd[d]);
                const float a = b_actions[i * D + d];
                const float diff = a - mu;
                const float var = sd * sd;
                dmean[m * D + d] = g_lp * (diff / var);
                contrib = g_lp * ((diff * diff) / var - 1.0f) + g_ent;
            }
            const double w = wave_sum((double)contrib);
            if (lane == 0) red[wave][kNumSums + d] += w;
        }
    }
    emit_block_sums(sums, block_partials, stride, D, red);   // the barrier inside also publishes red[*][6+d]
    write_slot_head(slot, P, D);
}

// ---- 3. finalize ---------------------------------------------------------------------------------
// Workgroup j folds the partials of workspace slot j (index order) into row j of `scalars` (+ dlogstd).  One slot per
// call in the immediate mode; the slots of all minibatches of an update in the deferred mode (one launch per iteration
// instead of one per minibatch).  Each lane loads all the sums of its workgroup rows before anything is reduced.
__global__ __launch_bounds__(256) void loss_finalize(const unsigned char* __restrict__ ws, size_t slot_stride,
                                                     float* __restrict__ scalars, float* __restrict__ dlogstd) {
    __shared__ double red[4];
    __shared__ double tot[kNumSums + kMaxD];
    const unsigned char* base = ws + (size_t)blockIdx.x * slot_stride;
    const LossSlot* slot = reinterpret_cast<const LossSlot*>(base);
    const double* part = reinterpret_cast<const double*>(base + sizeof(LossSlot)) + 2 * kStatsMaxBlocks;
    const int nblocks = slot->nblocks, D = slot->D;
    const int stride = kNumSums + D;
    double acc[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) acc[k] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
#pragma unroll
        for (int k = 0; k < kNumSums; ++k) acc[k] += part[(int64_t)b * stride + k];
    }
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) {
        const double r = block_sum<4>(acc[k], red);
        if (threadIdx.x == 0) tot[k] = r;
    }
    for (int d = 0; d < D; ++d) {
        double sd = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 256) sd += part[(int64_t)b * stride + kNumSums + d];
        const double r = block_sum<4>(sd, red);
        if (threadIdx.x == 0) tot[kNumSums + d] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* scalars7 = scalars + 7 * (size_t)blockIdx.x;
        const double n = (double)slot->M;
        const float pg_loss = (float)(tot[0] / n);
        const float v_loss = 0.5f * (float)(tot[1] / n);
        const float entropy = (float)(tot[2] / n);
        float loss = pg_loss - slot->ent_coef * entropy;     // :355  pg_loss - ent_coef*entropy + v_loss*vf_coef
        loss = loss + v_loss * slot->vf_coef;
        scalars7[0] = loss;
        scalars7[1] = pg_loss;
        scalars7[2] = v_loss;
        scalars7[3] = entropy;
        scalars7[4] = (float)(tot[3] / n);
        scalars7[5] = (float)(tot[4] / n);
        scalars7[6] = (float)(tot[5] / n);
    }
    if (dlogstd && threadIdx.x < D) dlogstd[(size_t)blockIdx.x * D + threadIdx.x] = (float)tot[kNumSums + threadIdx.x];
}

static inline int stats_blocks_for(int M) {
    const int b = (M + 1023) / 1024;
    return b < kStatsMaxBlocks ? b : kStatsMaxBlocks;
}
static inline int main_blocks_for(int M) {
    const int b = (M + 255) / 256;
    return b < kMaxGrid ? b : kMaxGrid;
}
static inline size_t ws_bytes(int D) {
    return sizeof(LossSlot) + (2 * (size_t)kStatsMaxBlocks + (size_t)kMaxGrid * (size_t)(kNumSums + D)) * sizeof(double);
}
static LossParams make_params(int M, double clip, double ent, double vf, int norm_adv, int clip_vloss, bool stats_given) {
    LossParams P;
    P.lo = (float)(1.0 - clip);
    P.hi = (float)(1.0 + clip);
    P.clip = (float)clip;
    P.ent_coef = (float)ent;
    P.vf_coef = (float)vf;
    P.norm_adv = norm_adv ? 1 : 0;
    P.clip_vloss = clip_vloss ? 1 : 0;
    P.M = M;
    P.stats_blocks = stats_given ? 0 : stats_blocks_for(M);
    return P;
}

}  // namespace mi355ppo

using namespace mi355ppo;

extern "C" MI355PPO_API size_t mi355ppo_loss_workspace_bytes(int M, int D) {
    if (M <= 0 || D < 0 || D > kMaxD) return 0;
    return ws_bytes(D);
}

static inline int adv_stats_per_seg(int M) {
    const int b = (M + 1023) / 1024;
    return b < kStatsMaxBlocks ? b : kStatsMaxBlocks;
}

extern "C" MI355PPO_API size_t mi355ppo_adv_stats_workspace_bytes(int64_t total, int M) {
    if (M <= 0 || total <= 0) return 0;
    return (size_t)((total + M - 1) / M) * (size_t)adv_stats_per_seg(M) * 2 * sizeof(double);
}

extern "C" MI355PPO_API int mi355ppo_adv_stats_f32(const float* b_advantages, const int64_t* inds, int64_t total, int M,
                                                  float* mean_den, void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "mi355ppo_adv_stats_f32";
    MI355_REQUIRE(b_advantages && mean_den, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(M > 0 && total > 0, MI355PPO_EINVAL, "%s: M=%d, total=%lld must be positive", fn, M, (long long)total);
    MI355_REQUIRE(aligned(b_advantages, 4) && aligned(inds, 8) && aligned(mean_den, 4), MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    const int64_t nseg = (total + M - 1) / M;
    MI355_REQUIRE(nseg <= 65535, MI355PPO_EINVAL, "%s: %lld minibatches exceed one launch", fn, (long long)nseg);
    const size_t need = mi355ppo_adv_stats_workspace_bytes(total, M);
    MI355_REQUIRE(workspace && workspace_bytes >= need, MI355PPO_EWORKSPACE, "%s: workspace %zu bytes < required %zu", fn,
                  workspace ? workspace_bytes : (size_t)0, need);
    MI355_REQUIRE(aligned(workspace, 8), MI355PPO_EALIGN, "%s: workspace must be 8-byte aligned", fn);
    const int per_seg = adv_stats_per_seg(M);
    double* partials = static_cast<double*>(workspace);
    hipLaunchKernelGGL(adv_stats_partials<1>, dim3(per_seg, (unsigned)nseg), dim3(256), 0, as_stream(stream), inds, b_advantages, M,
                       total, partials);
    int rc = check_launch("adv_stats_partials");
    if (rc) return rc;
    hipLaunchKernelGGL(adv_stats_fold, dim3((unsigned)nseg), dim3(256), 0, as_stream(stream), partials, per_seg, M, total, mean_den);
    return check_launch(fn);
}

extern "C" MI355PPO_API int mi355ppo_batch_pack_f32(const float* b_actions_f32, const float* b_logprobs, const float* b_advantages,
                                                   const float* b_returns, const float* b_values, float* pack, int64_t B,
                                                   void* stream) {
    const char* fn = "mi355ppo_batch_pack_f32";
    MI355_REQUIRE(b_actions_f32 && b_logprobs && b_advantages && b_returns && b_values && pack, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(B > 0, MI355PPO_EINVAL, "%s: B=%lld must be positive", fn, (long long)B);
    MI355_REQUIRE(aligned(b_actions_f32, 4) && aligned(b_logprobs, 4) && aligned(b_advantages, 4) && aligned(b_returns, 4) &&
                      aligned(b_values, 4) && aligned(pack, 32), MI355PPO_EALIGN, "%s: misaligned pointer (pack rows are 32-byte aligned)", fn);
    const int64_t blocks = (B + 255) / 256;
    hipLaunchKernelGGL(batch_pack_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), b_actions_f32,
                       b_logprobs, b_advantages, b_returns, b_values, pack, B);
    return check_launch(fn);
}

extern "C" MI355PPO_API int mi355ppo_adv_stats_packed_f32(const float* pack, const int64_t* inds, int64_t total, int M, float* mean_den,
                                                         void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "mi355ppo_adv_stats_packed_f32";
    MI355_REQUIRE(pack && mean_den, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(M > 0 && total > 0, MI355PPO_EINVAL, "%s: M=%d, total=%lld must be positive", fn, M, (long long)total);
    MI355_REQUIRE(aligned(pack, 32) && aligned(inds, 8) && aligned(mean_den, 4), MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    const int64_t nseg = (total + M - 1) / M;
    MI355_REQUIRE(nseg <= 65535, MI355PPO_EINVAL, "%s: %lld minibatches exceed one launch", fn, (long long)nseg);
    const size_t need = mi355ppo_adv_stats_workspace_bytes(total, M);
    MI355_REQUIRE(workspace && workspace_bytes >= need, MI355PPO_EWORKSPACE, "%s: workspace %zu bytes < required %zu", fn,
                  workspace ? workspace_bytes : (size_t)0, need);
    MI355_REQUIRE(aligned(workspace, 8), MI355PPO_EALIGN, "%s: workspace must be 8-byte aligned", fn);
    const int per_seg = adv_stats_per_seg(M);
    double* partials = static_cast<double*>(workspace);
    hipLaunchKernelGGL(adv_stats_partials<kPackFloats>, dim3(per_seg, (unsigned)nseg), dim3(256), 0, as_stream(stream), inds,
                       pack + kPackAdv, M, total, partials);
    int rc = check_launch("adv_stats_partials");
    if (rc) return rc;
    hipLaunchKernelGGL(adv_stats_fold, dim3((unsigned)nseg), dim3(256), 0, as_stream(stream), partials, per_seg, M, total, mean_den);
    return check_launch(fn);
}

static int check_common(const char* fn, const void* a, const void* b, const float* b_actions, const float* b_logprobs,
                        const float* b_adv, const float* b_ret, const float* b_val, int M, const void* scalars,
                        const void* g1, const void* g2, void* ws, size_t ws_bytes_given, int D) {
    MI355_REQUIRE(a && b && b_actions && b_logprobs && b_adv && b_ret && b_val && g1 && g2, MI355PPO_EINVAL,
                  "%s: null pointer", fn);
    MI355_REQUIRE(M > 0, MI355PPO_EINVAL, "%s: M=%d must be positive", fn, M);
    MI355_REQUIRE(ws && ws_bytes_given >= ws_bytes(D), MI355PPO_EWORKSPACE,
                  "%s: workspace %zu bytes < required %zu", fn, ws ? ws_bytes_given : (size_t)0, ws_bytes(D));
    MI355_REQUIRE(aligned(ws, 8), MI355PPO_EALIGN, "%s: workspace must be 8-byte aligned", fn);
    MI355_REQUIRE(aligned(a, 4) && aligned(b, 4) && aligned(b_actions, 4) && aligned(b_logprobs, 4) && aligned(b_adv, 4) &&
                      aligned(b_ret, 4) && aligned(b_val, 4) && aligned(scalars, 4) && aligned(g1, 4) && aligned(g2, 4),
                  MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    return MI355PPO_OK;
}

extern "C" MI355PPO_API int mi355ppo_loss_categorical_fwd_bwd_f32(const float* new_logits, const float* new_value,
                                                     const int64_t* mb_inds, const float* b_actions_f32,
                                                     const float* b_logprobs, const float* b_advantages,
                                                     const float* b_returns, const float* b_values, int M, int A,
                                                     double clip_coef, double ent_coef, double vf_coef, int norm_adv,
                                                     int clip_vloss, const float* adv_mean_den, float* scalars7,
                                                     float* dlogits, float* dvalue, void* workspace, size_t workspace_bytes,
                                                     void* stream) {
    const char* fn = "mi355ppo_loss_categorical_fwd_bwd_f32";
    int rc = check_common(fn, new_logits, new_value, b_actions_f32, b_logprobs, b_advantages, b_returns, b_values, M,
                          scalars7, dlogits, dvalue, workspace, workspace_bytes, 0);
    if (rc) return rc;
    MI355_REQUIRE(A > 0 && A <= 64, MI355PPO_EINVAL, "%s: A=%d must be in 1..64", fn, A);
    MI355_REQUIRE(aligned(mb_inds, 8) && aligned(adv_mean_den, 4), MI355PPO_EALIGN, "%s: misaligned pointer", fn);
    hipStream_t s = as_stream(stream);
    const LossParams P = make_params(M, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss, adv_mean_den != nullptr);
    LossSlot* slot = static_cast<LossSlot*>(workspace);
    double* stats = reinterpret_cast<double*>(slot + 1);
    double* partials = stats + 2 * kStatsMaxBlocks;
    const int blocks = main_blocks_for(M);
    if (P.norm_adv && !adv_mean_den) {
        hipLaunchKernelGGL(loss_adv_stats, dim3(P.stats_blocks), dim3(256), 0, s, mb_inds, b_advantages, M, stats);
        rc = check_launch("loss_adv_stats");
        if (rc) return rc;
    }
#define LAUNCH(AMAX, VEC)                                                                                              \
    hipLaunchKernelGGL((loss_categorical_main<AMAX, VEC>), dim3(blocks), dim3(256), 0, s, new_logits, new_value, mb_inds, \
                       b_actions_f32, b_logprobs, b_advantages, b_returns, b_values, A, P, stats, adv_mean_den, slot,      \
                       partials, dlogits, dvalue)
    if (A == 4 && aligned(new_logits, 16) && aligned(dlogits, 16)) { LAUNCH(4, true); }
    else if (A <= 4) { LAUNCH(4, false); }
    else if (A <= 8) { LAUNCH(8, false); }
    else if (A <= 18) { LAUNCH(18, false); }
    else { LAUNCH(64, false); }
#undef LAUNCH
    rc = check_launch("loss_categorical_main");
    if (rc || !scalars7) return rc;                  // scalars7 == NULL: deferred, mi355ppo_loss_scalars_f32 folds the slot later
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, s, static_cast<const unsigned char*>(workspace), (size_t)0, scalars7,
                       (float*)nullptr);
    return check_launch("loss_finalize");
}

extern "C" MI355PPO_API int mi355ppo_loss_categorical_packed_fwd_bwd_f32(const float* new_logits, const float* new_value,
                                                                        const int64_t* mb_inds, const float* pack, int M, int A,
                                                                        double clip_coef, double ent_coef, double vf_coef,
                                                                        int norm_adv, int clip_vloss, const float* adv_mean_den,
                                                                        float* scalars7, float* dlogits, float* dvalue,
                                                                        void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "mi355ppo_loss_categorical_packed_fwd_bwd_f32";
    int rc = check_common(fn, new_logits, new_value, pack, pack, pack, pack, pack, M, scalars7, dlogits, dvalue, workspace,
                          workspace_bytes, 0);
    if (rc) return rc;
    MI355_REQUIRE(A > 0 && A <= 64, MI355PPO_EINVAL, "%s: A=%d must be in 1..64", fn, A);
    MI355_REQUIRE(aligned(pack, 32) && aligned(mb_inds, 8) && aligned(adv_mean_den, 4), MI355PPO_EALIGN,
                  "%s: misaligned pointer (pack rows are 32-byte aligned)", fn);
    MI355_REQUIRE(!norm_adv || adv_mean_den, MI355PPO_EINVAL,
                  "%s: norm_adv needs adv_mean_den (mi355ppo_adv_stats_packed_f32); the packed path has no statistics launch of its own", fn);
    hipStream_t s = as_stream(stream);
    const LossParams P = make_params(M, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss, true);
    LossSlot* slot = static_cast<LossSlot*>(workspace);
    double* stats = reinterpret_cast<double*>(slot + 1);
    double* partials = stats + 2 * kStatsMaxBlocks;
    const int blocks = main_blocks_for(M);
#define LAUNCH(AMAX, VEC)                                                                                                      \
    hipLaunchKernelGGL((loss_categorical_main<AMAX, VEC, true>), dim3(blocks), dim3(256), 0, s, new_logits, new_value, mb_inds, \
                       pack, pack, pack, pack, pack, A, P, stats, adv_mean_den, slot, partials, dlogits, dvalue)
    if (A == 4 && aligned(new_logits, 16) && aligned(dlogits, 16)) { LAUNCH(4, true); }
    else if (A <= 4) { LAUNCH(4, false); }
    else if (A <= 8) { LAUNCH(8, false); }
    else if (A <= 18) { LAUNCH(18, false); }
    else { LAUNCH(64, false); }
#undef LAUNCH
    rc = check_launch("loss_categorical_main (packed)");
    if (rc || !scalars7) return rc;                  // scalars7 == NULL: deferred, mi355ppo_loss_scalars_f32 folds the slot later
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, s, static_cast<const unsigned char*>(workspace), (size_t)0, scalars7,
                       (float*)nullptr);
    return check_launch("loss_finalize");
}

extern "C" MI355PPO_API int mi355ppo_loss_normal_fwd_bwd_f32(const float* new_mean, const float* logstd, const float* new_value,
                                                const int64_t* mb_inds, const float* b_actions,
                                                const float* b_logprobs, const float* b_advantages,
                                                const float* b_returns, const float* b_values, int M, int D,
                                                double clip_coef, double ent_coef, double vf_coef, int norm_adv,
                                                int clip_vloss, const float* adv_mean_den, float* scalars7, float* dmean,
                                                float* dlogstd, float* dvalue, void* workspace, size_t workspace_bytes,
                                                void* stream) {
    const char* fn = "mi355ppo_loss_normal_fwd_bwd_f32";
    MI355_REQUIRE(D > 0 && D <= kMaxD, MI355PPO_EINVAL, "%s: D=%d must be in 1..%d", fn, D, kMaxD);
    MI355_REQUIRE(logstd && dlogstd, MI355PPO_EINVAL, "%s: null pointer", fn);
    int rc = check_common(fn, new_mean, new_value, b_actions, b_logprobs, b_advantages, b_returns, b_values, M, scalars7,
                          dmean, dvalue, workspace, workspace_bytes, D);
    if (rc) return rc;
    MI355_REQUIRE(aligned(mb_inds, 8) && aligned(logstd, 4) && aligned(dlogstd, 4) && aligned(adv_mean_den, 4), MI355PPO_EALIGN,
                  "%s: misaligned pointer", fn);
    hipStream_t s = as_stream(stream);
    const LossParams P = make_params(M, clip_coef, ent_coef, vf_coef, norm_adv, clip_vloss, adv_mean_den != nullptr);
    MI355_REQUIRE(scalars7, MI355PPO_EINVAL, "%s: null pointer", fn);   // dlogstd is a gradient: this family always folds at once
    LossSlot* slot = static_cast<LossSlot*>(workspace);
    double* stats = reinterpret_cast<double*>(slot + 1);
    double* partials = stats + 2 * kStatsMaxBlocks;
    const int blocks = main_blocks_for(M);
    if (P.norm_adv && !adv_mean_den) {
        hipLaunchKernelGGL(loss_adv_stats, dim3(P.stats_blocks), dim3(256), 0, s, mb_inds, b_advantages, M, stats);
        rc = check_launch("loss_adv_stats");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(loss_normal_main, dim3(blocks), dim3(256), 0, s, new_mean, logstd, new_value, mb_inds, b_actions,
                       b_logprobs, b_advantages, b_returns, b_values, D, P, stats, adv_mean_den, slot, partials, dmean, dvalue);
    rc = check_launch("loss_normal_main");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, s, static_cast<const unsigned char*>(workspace), (size_t)0, scalars7,
                       dlogstd);
    return check_launch("loss_finalize");
}

extern "C" MI355PPO_API int mi355ppo_loss_scalars_f32(const void* workspaces, size_t slot_stride_bytes, int nslots, float* scalars,
                                                     void* stream) {
    const char* fn = "mi355ppo_loss_scalars_f32";
    MI355_REQUIRE(workspaces && scalars, MI355PPO_EINVAL, "%s: null pointer", fn);
    MI355_REQUIRE(nslots > 0 && nslots <= 65535, MI355PPO_EINVAL, "%s: nslots=%d must be in 1..65535", fn, nslots);
    MI355_REQUIRE(aligned(workspaces, 8) && slot_stride_bytes % 8 == 0 && aligned(scalars, 4), MI355PPO_EALIGN,
                  "%s: misaligned pointer / stride", fn);
    MI355_REQUIRE(nslots == 1 || slot_stride_bytes >= ws_bytes(0), MI355PPO_EWORKSPACE, "%s: slot stride %zu < %zu", fn,
                  slot_stride_bytes, ws_bytes(0));
    hipLaunchKernelGGL(loss_finalize, dim3(nslots), dim3(256), 0, as_stream(stream), static_cast<const unsigned char*>(workspaces),
                       slot_stride_bytes, scalars, (float*)nullptr);
    return check_launch(fn);
}
