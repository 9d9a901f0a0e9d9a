/*
 * mi355ppo.h -- C ABI of libmi355ppo.so: the MI355X (gfx950 / CDNA4) PPO hot path.
 *
 * The reference (vwxyzjn/cleanrl) has no FFI, plugin or operator boundary: its PPO scripts are
 * executed, never imported, and the hot path is inline tensor code.  This header therefore DEFINES
 * the drop-in boundary at the tensor seams of that inline code; every entry point cites the
 * reference lines (cleanrl/<file>:<lines>) whose op chain it replaces.  A maintainer binds it with
 * ctypes (the reference is Python) -- see INTEGRATION.md for the stub.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor .data_ptr()),
 *     contiguous, row-major, never retained past the call;
 *   - no allocation, no hipMalloc/hipFree, no host<->device copy, no device synchronisation inside:
 *     work is only enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream),
 *     so every call is legal inside a hipGraph stream capture;
 *   - scratch memory is caller-provided (`workspace`, sized by the *_workspace_bytes query) and is
 *     written before it is read in every call: it needs no initialisation and may be shared by
 *     calls that are ordered on one stream;
 *   - hyper-parameters are `double` because the reference holds them as Python floats and torch
 *     rounds them to f32 at each scalar-tensor op; the kernels reproduce that rounding;
 *   - return 0 on success, a negative MI355PPO_E* code otherwise; mi355ppo_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - re-entrant and thread-safe: no global mutable state.
 */
#ifndef MI355PPO_H_
#define MI355PPO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355PPO_VERSION 210 /* major*100 + minor*10 + patch.  The minor moves whenever an exported signature changes or an entry
                                  point is added (1.1: adv_mean_den / conv1_variant arguments of round 2; 1.2, 1.3: round 3;
                                  1.4: the *_cpu host-pointer twins; 1.5: mi355ppo_init; 1.6: round 4 -- the fused MLP family K7,
                                  mi355ppo_clip_adam_sched_f32; 1.7: mi355ppo_fc_heads_act_categorical_f32, mi355ppo_nature_packs_f32,
                                  mi355ppo_synth_atari_step_hwc_ctr_u8; 1.8: round 5 -- the *_f16x2 / *_amax entry points and mi355ppo_absmax_f32;
                                  1.9: round 6 -- the kernel queries mi355ppo_fc_packed_kernel_f16x2, mi355ppo_fc_wgrad_kernel_f16x2, mi355ppo_cnn_conv_wgrad_kernel_f16x2;
                                  heads of up to 18 actions, a 4-byte-aligned critic row; 2.0: the peer-memory gradient exchange mi355ppo_dp_*; 2.1: the fused MLP family takes obs_dim <= 512, n_out <= 20);
                                  a binding must check major AND minor (cleanrl_amd/_lib.py does) */

#if defined(__GNUC__)
#define MI355PPO_API __attribute__((visibility("default")))
#else
#define MI355PPO_API
#endif

/* An "amax record": the bit pattern of max |x| of an f32 tensor, kept as 16 uint32 slots 64 bytes apart (1 KiB; the record's value is
 * the maximum over the slots).  The *_f16x2 entry points read the record of every tensor they split into two f16 terms (the tensor's
 * power-of-two scale derives from it) and fold the values they store into the record of the tensor they produce; the caller zeroes
 * records before their producers run (one memset for all records of a pass) -- see mi355ppo_absmax_f32 and csrc/f16split.h. */
#define MI355PPO_AMAX_WORDS 256

#define MI355PPO_OK 0
#define MI355PPO_EINVAL (-1)     /* null pointer, non-positive or unsupported shape            */
#define MI355PPO_EALIGN (-2)     /* a pointer is not aligned to its element size               */
#define MI355PPO_EHIP (-3)       /* a HIP runtime call / kernel launch failed                  */
#define MI355PPO_EWORKSPACE (-4) /* workspace NULL or smaller than *_workspace_bytes()         */
#define MI355PPO_ETIMEOUT (-5)   /* mi355ppo_dp_comm_status: a wait on a peer gave up          */

MI355PPO_API int mi355ppo_version(void);
MI355PPO_API const char* mi355ppo_last_error(void);
/* Capability check: 0 if `device` (a HIP device ordinal) can run this library's kernels (gfx950), MI355PPO_EHIP / EINVAL with a
 * message otherwise (no device, ordinal out of range, another architecture).  Optional -- the library keeps no per-device state
 * and every entry point works without it --; cleanrl_amd/_lib.py calls it once per device, so that a wrong GPU fails at load
 * time with a sentence instead of at the first launch with "invalid device function". */
MI355PPO_API int mi355ppo_init(int device);

/* ---------------------------------------------------------------------------------------------
 * K1  Generalised Advantage Estimation, fused reverse scan.
 * Replaces cleanrl/ppo_atari_multigpu.py:290-301 (== ppo.py:220-231, ppo_atari.py:237-248,
 * ppo_atari_envpool.py:252-263, ppo_continuous_action.py:235-246): ~10 elementwise torch kernels
 * per time step x T steps, plus `returns = advantages + values`.
 *   rewards, dones, values : (T,N) f32, time-major (N contiguous)      [in]
 *   next_done, next_value  : (N)   f32                                  [in]
 *   advantages, returns    : (T,N) f32                                  [out]
 * Bit-exact with the reference's f32 op order: ((g*nv)*nnt), ((r+.)-v), (((g*l)*nnt)*last), where
 * g*l is formed in double and then rounded to f32 (no FMA contraction).
 * `variant`: 0 = auto, 1 / 3 = column-streaming kernels (large N), 6 = staged-scan kernel (small N: off-chain math by all threads, chain
 * out of LDS) (tuning/testing; 2 / 4 / 5 -- round 1's single-scanner tile kernels -- were retired in ABI 2.1: MI355PPO_EINVAL).
 */
MI355PPO_API int mi355ppo_gae_f32(const float* rewards, const float* dones, const float* values,
                     const float* next_done, const float* next_value,
                     float* advantages, float* returns,
                     int T, int N, double gamma, double gae_lambda, void* stream);
MI355PPO_API int mi355ppo_gae_f32_variant(const float* rewards, const float* dones, const float* values,
                             const float* next_done, const float* next_value,
                             float* advantages, float* returns,
                             int T, int N, double gamma, double gae_lambda, int variant, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2  Categorical(logits): sample + log_prob + entropy in one launch.
 * Replaces Agent.get_action_and_value's distribution ops, cleanrl/ppo_atari_multigpu.py:156-159
 * (== ppo.py:122-126, ppo_atari_envpool.py:134-139): logits - logsumexp, softmax,
 * multinomial(probs,1) [= argmax(probs / Exp(1)) in ATen], gather, entropy.
 *   logits      : (B,A) f32                                             [in]
 *   noise_exp1  : (B,A) f32 Exponential(1) draws, or NULL               [in]
 *                 NULL => draws come from an internal Philox4x32-10 stream keyed by (seed, offset);
 *                 counter = row*A + column, so results do not depend on the launch geometry.
 *   action_i64  : (B) int64, may be NULL                                [out]
 *   action_f32  : (B) f32,   may be NULL  (the reference stores Discrete actions in an f32 rollout
 *                 tensor, ppo_atari_multigpu.py:236,265)                [out]
 *   logprob     : (B) f32                                               [out]
 *   entropy     : (B) f32, may be NULL                                  [out]
 */
MI355PPO_API int mi355ppo_categorical_sample_f32(const float* logits, const float* noise_exp1,
                                    uint64_t seed, uint64_t offset,
                                    int64_t* action_i64, float* action_f32,
                                    float* logprob, float* entropy,
                                    int B, int A, void* stream);
/* The same with the Philox stream position in device memory: offset_eff = offset + *offset_base (offset_base may be
 * NULL).  A launch captured into a hipGraph can then be replayed with a new position (the rollout-step graphs of
 * PPOLearner.capture_rollout advance the base once per rollout). */
MI355PPO_API int mi355ppo_categorical_sample_ctr_f32(const float* logits, const float* noise_exp1, uint64_t seed,
                                        uint64_t offset, const uint64_t* offset_base, int64_t* action_i64,
                                        float* action_f32, float* logprob, float* entropy, int B, int A, void* stream);

/* log_prob / entropy of GIVEN actions (the `action is not None` branch, :157-159).
 * Exactly one of action_i64 / action_f32 must be non-NULL. */
MI355PPO_API int mi355ppo_categorical_logprob_entropy_f32(const float* logits,
                                             const int64_t* action_i64, const float* action_f32,
                                             float* logprob, float* entropy,
                                             int B, int A, void* stream);

/* Backward of the two outputs above w.r.t. logits, for code that differentiates
 * Agent.get_action_and_value(x, action) itself (autograd of torch categorical.py log_prob/entropy):
 *   dlogits[b,j] = g_logprob[b]*(1[j==a_b] - p_bj) - g_entropy[b]*p_bj*(logp_bj + H_b).
 * g_logprob / g_entropy (B) may be NULL (= zeros). */
MI355PPO_API int mi355ppo_categorical_logprob_entropy_bwd_f32(const float* logits,
                                             const int64_t* action_i64, const float* action_f32,
                                             const float* g_logprob, const float* g_entropy,
                                             float* dlogits, int B, int A, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2' Normal(mean, exp(logstd)): sample + summed log_prob + summed entropy.
 * Replaces cleanrl/ppo_continuous_action.py:134-141.
 *   mean (B,D), logstd (D), noise_std_normal (B,D) or NULL (=> Philox + Box-Muller),
 *   action (B,D) out, logprob_sum (B) out, entropy_sum (B) out or NULL.
 * action = noise * exp(logstd) + mean (mul then add, as torch.normal); log_prob uses
 * log(exp(logstd)) for log_scale exactly as torch.distributions.Normal does.
 */
MI355PPO_API int mi355ppo_normal_sample_f32(const float* mean, const float* logstd, const float* noise_std_normal,
                               uint64_t seed, uint64_t offset,
                               float* action, float* logprob_sum, float* entropy_sum,
                               int B, int D, void* stream);
MI355PPO_API int mi355ppo_normal_logprob_entropy_f32(const float* mean, const float* logstd, const float* action,
                                        float* logprob_sum, float* entropy_sum,
                                        int B, int D, void* stream);

/* Backward of (logprob_sum, entropy_sum): dmean (B,D) and the per-row contributions to dlogstd
 * (B,D), which the caller sums over rows. */
MI355PPO_API int mi355ppo_normal_logprob_entropy_bwd_f32(const float* mean, const float* logstd, const float* action,
                                        const float* g_logprob, const float* g_entropy,
                                        float* dmean, float* dlogstd_rows, int B, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3  Minibatch PPO loss, forward + backward fused (clipped surrogate + value loss + entropy).
 * Replaces cleanrl/ppo_atari_multigpu.py:320(b_actions.long()[mb_inds])-355 and the autograd
 * backward of those ops down to the network outputs (== ppo.py:250-285, ppo_atari.py:267-302,
 * ppo_atari_envpool.py:282-317).
 *   new_logits (M,A), new_value (M)        : network outputs for the minibatch rows  [in]
 *   mb_inds (M) int64 or NULL (= identity) : rows of the flat batch                  [in]
 *   b_actions_f32, b_logprobs, b_advantages, b_returns, b_values : (Bflat) f32       [in]
 *   scalars7 : loss, pg_loss, v_loss, entropy, old_approx_kl, approx_kl, clipfrac    [out, 7 f32]
 *   dlogits (M,A), dvalue (M) : d loss / d new_logits, d loss / d new_value          [out]
 *   adv_mean_den : NULL, or 2 f32 = (mean, unbiased std + 1e-8) of b_advantages[mb_inds] from
 *                  mi355ppo_adv_stats_f32 (then no statistics launch happens here)        [in]
 * Reductions are computed in a fixed order (deterministic); advantage mean / unbiased std are
 * accumulated in f64.  Tolerances vs the reference are stated in tests/test_gpu_kernels.py.
 * Launches per call: statistics (only if norm_adv and adv_mean_den == NULL), row pass, scalar fold
 * (only if scalars7 != NULL).  scalars7 == NULL defers the fold: the call's partial sums stay in
 * its workspace, which then must be a slot no other call overwrites until
 * mi355ppo_loss_scalars_f32 has folded it (categorical family only).
 */
MI355PPO_API size_t mi355ppo_loss_workspace_bytes(int M, int D);
MI355PPO_API int mi355ppo_loss_categorical_fwd_bwd_f32(const float* new_logits, const float* new_value,
                                          const int64_t* mb_inds, const float* b_actions_f32,
                                          const float* b_logprobs, const float* b_advantages,
                                          const float* b_returns, const float* b_values,
                                          int M, int A,
                                          double clip_coef, double ent_coef, double vf_coef,
                                          int norm_adv, int clip_vloss, const float* adv_mean_den,
                                          float* scalars7, float* dlogits, float* dvalue,
                                          void* workspace, size_t workspace_bytes, void* stream);

/* Deferred scalar fold: row j of scalars (nslots,7) <- the partial sums that a categorical call
 * with scalars7 == NULL left in the workspace at workspaces + j*slot_stride_bytes.  One launch for
 * all minibatches of an update (the scalars are diagnostics: nothing on the device waits for them). */
MI355PPO_API int mi355ppo_loss_scalars_f32(const void* workspaces, size_t slot_stride_bytes, int nslots,
                                           float* scalars, void* stream);

/* Advantage statistics of ALL minibatches of one epoch in one call / two launches (`mb_advantages.mean()`,
 * `mb_advantages.std() + 1e-8`, cleanrl/ppo_atari_multigpu.py:337-338): they depend only on the
 * epoch's permutation and the GAE output, not on anything the network produces, so the learner
 * computes them when it uploads the permutation and K3 runs without its statistics launch.
 *   inds (total) int64 or NULL (= identity); minibatch j = rows [j*M, min((j+1)*M, total));
 *   mean_den : (ceil(total / M), 2) f32                                                  [out] */
MI355PPO_API size_t mi355ppo_adv_stats_workspace_bytes(int64_t total, int M);
MI355PPO_API int mi355ppo_adv_stats_f32(const float* b_advantages, const int64_t* inds, int64_t total, int M,
                                        float* mean_den, void* workspace, size_t workspace_bytes, void* stream);

/* Packed behaviour rows (round 3): the five per-row behaviour scalars the loss gathers through mb_inds --
 * `b_actions.long()[mb_inds]`, `b_logprobs[mb_inds]`, `b_advantages[mb_inds]`, `b_returns[mb_inds]`, `b_values[mb_inds]`
 * (cleanrl/ppo_atari_multigpu.py:320-352) -- stored as ONE 32-byte row per flat batch index,
 *   pack[i] = {action (f32 storage), old log-prob, advantage, return, old value, 0, 0, 0},
 * so that a minibatch row costs one 32-byte gather instead of five 4-byte gathers from five arrays (five 128-byte lines
 * per row once the flat batch outgrows the caches).  mi355ppo_batch_pack_f32 builds the rows once per iteration, after
 * GAE; the *_packed_* entry points are the K3 calls above reading them: same arithmetic, bit-identical results.
 *   pack (B, 8) f32, 32-byte aligned.  The packed loss call takes its advantage statistics from the caller
 *   (adv_mean_den from mi355ppo_adv_stats_packed_f32) whenever norm_adv is set: one launch per minibatch. */
MI355PPO_API int mi355ppo_batch_pack_f32(const float* b_actions_f32, const float* b_logprobs, const float* b_advantages,
                                         const float* b_returns, const float* b_values, float* pack, int64_t B, void* stream);
MI355PPO_API int mi355ppo_adv_stats_packed_f32(const float* pack, const int64_t* inds, int64_t total, int M,
                                               float* mean_den, void* workspace, size_t workspace_bytes, void* stream);
MI355PPO_API int mi355ppo_loss_categorical_packed_fwd_bwd_f32(const float* new_logits, const float* new_value,
                                          const int64_t* mb_inds, const float* pack, int M, int A,
                                          double clip_coef, double ent_coef, double vf_coef,
                                          int norm_adv, int clip_vloss, const float* adv_mean_den,
                                          float* scalars7, float* dlogits, float* dvalue,
                                          void* workspace, size_t workspace_bytes, void* stream);

/* Continuous-action variant, cleanrl/ppo_continuous_action.py:265-300:
 *   new_mean (M,D), logstd (D), b_actions (Bflat,D) -> dmean (M,D), dlogstd (D), dvalue (M). */
MI355PPO_API int mi355ppo_loss_normal_fwd_bwd_f32(const float* new_mean, const float* logstd, const float* new_value,
                                     const int64_t* mb_inds, const float* b_actions,
                                     const float* b_logprobs, const float* b_advantages,
                                     const float* b_returns, const float* b_values,
                                     int M, int D,
                                     double clip_coef, double ent_coef, double vf_coef,
                                     int norm_adv, int clip_vloss, const float* adv_mean_den,
                                     float* scalars7, float* dmean, float* dlogstd, float* dvalue,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5  Observation path: uint8 rollout storage -> f32 network input, gather + convert fused.
 * Replaces `b_obs[mb_inds]` followed by `x / 255.0` (cleanrl/ppo_atari_multigpu.py:320,154) and,
 * with inds == NULL, the per-step `x / 255.0` of :151,154.  The reference stores observations as
 * f32 (:235); uint8 storage is exact because frames are integers 0..255.
 *   src_u8 : (rows_total, row_bytes) u8;  inds : (rows) int64 or NULL;  dst_f32 : (rows, row_bytes)
 *   scale_255 != 0 => dst = (float)src / 255.0f, correctly rounded (bit-equal to torch's division);
 *   scale_255 == 0 => dst = (float)src.
 * row_bytes must be a multiple of 4; src rows must be 4-byte aligned, dst 16-byte aligned.
 */
MI355PPO_API int mi355ppo_obs_u8_to_f32(const uint8_t* src_u8, const int64_t* inds, float* dst_f32,
                           int64_t rows, int64_t row_bytes, int scale_255, void* stream);

/* Store-time relayout of incoming frames: channel-planar (rows, C, HW) uint8 -> pixel-interleaved
 * (rows, HW, C) uint8 ("NHWC").  Replaces nothing in the reference (it stores f32 NCHW, :235,258); it lets
 * the rollout buffer feed the conv stack channels-last so that no layout transposes run on f32 data.
 * C == 4 with HW % 4 == 0 (FrameStack(4) of 84x84) takes the coalesced register-transpose path. */
MI355PPO_API int mi355ppo_obs_nchw_to_nhwc_u8(const uint8_t* src, uint8_t* dst, int64_t rows, int C, int HW,
                                              void* stream);

/* FrameStack(4) delta store: dst_rows[r] = prev_rows[r] with channels 1..3 moved to 0..2 and channel 3 taken from
 * newest_planes[r] (rows, HW) -- the next observation of every env that was not reset, from ONE new plane.  Lets the
 * host send 1/4 of the frame bytes per step (envs flagged done send their full stack through
 * mi355ppo_obs_nchw_to_nhwc_u8).  prev_rows may equal dst_rows. */
MI355PPO_API int mi355ppo_obs_shift_append_u8_c4(const uint8_t* prev_rows, const uint8_t* newest_planes, uint8_t* dst_rows,
                                                 int64_t rows, int HW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a8/a9  Flat-buffer optimiser step: (grad * grad_scale) -> global-norm clip -> Adam, fused.
 * Replaces the unpack-and-divide of cleanrl/ppo_atari_multigpu.py:368-374 (grad_scale =
 * 1/world_size), nn.utils.clip_grad_norm_ (:376) and optim.Adam(eps=1e-5).step() (:377) on a
 * persistent flat parameter/gradient buffer.
 *   params, grads, exp_avg, exp_avg_sq : (n) f32 ; grads is scaled+clipped in place
 *   step : 1-based Adam step count ; total_norm_out : (1) f32 pre-clip norm, may be NULL
 */
MI355PPO_API size_t mi355ppo_clip_adam_workspace_bytes(int64_t n);
MI355PPO_API int mi355ppo_clip_adam_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                           double grad_scale, double max_grad_norm, double lr,
                           double beta1, double beta2, double eps, int64_t step,
                           float* total_norm_out, void* workspace, size_t workspace_bytes, void* stream);

/* The two schedule-dependent constants of an Adam step, as the kernels consume them (HOST function, host pointer):
 * out2_host = {(float)(-(lr / (1 - beta1^step))), (float)sqrt(1 - beta2^step)}  (torch adam.py: step_size, bias_correction2_sqrt). */
MI355PPO_API int mi355ppo_adam_schedule_f32(double lr, double beta1, double beta2, int64_t step, float* out2_host);
/* mi355ppo_clip_adam_f32 with those two constants read from DEVICE memory (`sched2`, 2 floats the caller copied there): a launch
 * captured into a hipGraph is replayed with the next step's learning rate / bias corrections without re-capturing
 * (PPOLearner.capture_update).  Bit-identical to mi355ppo_clip_adam_f32 for the same (lr, step). */
MI355PPO_API int mi355ppo_clip_adam_sched_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                              double grad_scale, double max_grad_norm, double beta1, double beta2, double eps,
                                              const float* sched2, float* total_norm_out, void* workspace, size_t workspace_bytes,
                                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * a9/e  The data-parallel gradient exchange over peer memory (csrc/dpcomm.hip): all_reduce(SUM) of the persistent flat gradient
 * buffer across the ranks of ONE node without a collective library -- replaces dist.all_reduce(all_grads_list, op=SUM) of
 * cleanrl/ppo_atari_multigpu.py:360-367 (the division by world_size of :368-374 stays grad_scale of mi355ppo_clip_adam_*).
 * Every rank owns one device segment which its peers map through HIP IPC (xGMI between GPUs); a call is five small launches on
 * `stream` -- publish, reduce slice `rank` of every rank's copy in rank order and push it to every rank, collect -- synchronised by
 * flags in the segments.  No host round trip, no second stream: legal inside a hipGraph capture (the round counter lives on the
 * device).  Every element is summed by one rank in one fixed order: all ranks receive the same bits.
 *
 * The communicator is the ONE object of this header that owns memory, so the conventions at the top hold for
 * mi355ppo_dp_allreduce_sum_f32 only; create / handle / connect / destroy are set-up calls (they allocate, synchronise the device
 * and must not run during a capture).  Set-up, on every rank, with the current HIP device = the rank's GPU:
 *   mi355ppo_dp_comm_create(world, rank, max_floats, timeout_ms, &comm)     world <= MI355PPO_DP_MAX_WORLD
 *   mi355ppo_dp_comm_handle(comm, handle)                                    MI355PPO_DP_HANDLE_BYTES bytes to send to every rank
 *   ... the ranks exchange their handles (any host channel: the reference's process group, a file, a pipe) ...
 *   mi355ppo_dp_comm_connect(comm, handles)                                  world x MI355PPO_DP_HANDLE_BYTES, in rank order
 *   ... a host barrier: every rank has connected before the first exchange ...
 * Every rank must then issue the same sequence of mi355ppo_dp_allreduce_sum_f32 calls (same n).  grads: (n) f32, 16-byte aligned,
 * n <= max_floats; summed in place.  A wait on a peer gives up after timeout_ms (the peer died or skipped a call): the round, the
 * peer and the phase (1 = reduce, 2 = collect) are recorded, every later wait of this communicator returns at once, and
 * mi355ppo_dp_comm_status -- a HOST read, no synchronisation -- returns MI355PPO_ETIMEOUT from then on; grads then hold garbage and
 * the communicator must be destroyed.  Ranks are processes: one communicator per process and rank. */
#define MI355PPO_DP_MAX_WORLD 8
#define MI355PPO_DP_HANDLE_BYTES 64
typedef struct mi355ppo_dp_comm mi355ppo_dp_comm;
MI355PPO_API int mi355ppo_dp_comm_create(int world, int rank, int64_t max_floats, double timeout_ms, mi355ppo_dp_comm** comm_out);
MI355PPO_API int mi355ppo_dp_comm_handle(mi355ppo_dp_comm* comm, unsigned char* handle_out);
MI355PPO_API int mi355ppo_dp_comm_connect(mi355ppo_dp_comm* comm, const unsigned char* handles);
MI355PPO_API int mi355ppo_dp_allreduce_sum_f32(mi355ppo_dp_comm* comm, float* grads, int64_t n, void* stream);
MI355PPO_API int mi355ppo_dp_comm_status(mi355ppo_dp_comm* comm, int* round_out, int* peer_out, int* phase_out);   /* outputs may be NULL */
MI355PPO_API int mi355ppo_dp_comm_destroy(mi355ppo_dp_comm* comm);

/* ---------------------------------------------------------------------------------------------
 * K7  The reference's MLP agents (two independent 64-64 tanh networks: actor and critic) as one kernel family (csrc/mlp.hip).
 * Agents: cleanrl/ppo.py:100-126 (Categorical head), cleanrl/ppo_continuous_action.py:112-141 (Normal head with the
 * state-independent actor_logstd).  A network is passed as a HOST array of six DEVICE pointers in torch's own layouts:
 *   {W1 (64, O), b1 (64), W2 (64, 64), b2 (64), W3 (n_out, 64), b3 (n_out)}   (nn.Linear.weight is (out, in))
 * obs_dim O <= 512 and n_out <= 20 (ABI 2.1; CartPole 4 / 2, HalfCheetah 17 / 6, Ant 27 / 8, Humanoid 376 / 17 ...): beyond that the entry points
 * return MI355PPO_EINVAL and the caller keeps the networks on library GEMMs.  Up to O = 32 and n_out = 8 a lane keeps its row of W1 in registers;
 * wider shapes run the WIDE kernels (layer 1 and its weight gradient walk the observation in chunks of 32 columns, blocks of <= 32 rows).
 *
 * mi355ppo_mlp_fwd_f32: both forwards -- actor_out (B, n_out) = logits or mean, value (B) -- e.g. the bootstrap value of
 *   ppo.py:218-219.
 * mi355ppo_mlp_act_*: the rollout step of Agent.get_action_and_value(next_obs) (ppo.py:205-210,
 *   ppo_continuous_action.py:221-226): both forwards + K2 / K2' (same Philox streams as mi355ppo_categorical_sample_ctr_f32 /
 *   mi355ppo_normal_sample_f32: counter = row * ceil(n_out / 4) + column / 4; offset_eff = offset + *offset_base) in ONE launch.
 *   logits_out / mean_out / entropy may be NULL.
 * mi355ppo_mlp_ppo_*_fwd_bwd_f32: one minibatch of the update (ppo.py:250-287 / ppo_continuous_action.py:265-302 up to and
 *   including loss.backward()): gather b_obs[mb_inds], both forwards, the distribution, the PPO loss terms of K3 (same row
 *   function), both backward passes; the parameter gradients are ADDED to `actor_grads` / `critic_grads` (host arrays of six
 *   device pointers, layouts as the parameters; `dlogstd` (D) likewise), the seven scalars of K3 written to scalars7.
 *   norm_adv needs adv_mean_den (mi355ppo_adv_stats_f32).  mean_shift (M, D) or NULL is added to the mean before the loss
 *   (rpo_continuous_action.py:138-142).  rows_per_block: 0 = auto.  Two launches; deterministic (fixed-order f64 folds).
 */
MI355PPO_API int mi355ppo_mlp_fwd_f32(const float* obs, int B, int O, const void* const* actor, const void* const* critic, int n_out,
                                      float* actor_out, float* value, void* stream);
MI355PPO_API int mi355ppo_mlp_act_categorical_f32(const float* obs, int B, int O, const void* const* actor, const void* const* critic,
                                                  int A, const float* noise_exp1, uint64_t seed, uint64_t offset,
                                                  const uint64_t* offset_base, int64_t* action_i64, float* action_f32, float* logprob,
                                                  float* entropy, float* value, float* logits_out, void* stream);
MI355PPO_API int mi355ppo_mlp_act_normal_f32(const float* obs, int B, int O, const void* const* actor_mean, const void* const* critic,
                                             const float* logstd, int D, const float* noise_std_normal, uint64_t seed, uint64_t offset,
                                             const uint64_t* offset_base, float* action, float* logprob_sum, float* entropy_sum,
                                             float* value, float* mean_out, void* stream);
MI355PPO_API size_t mi355ppo_mlp_ppo_workspace_bytes(int M, int O, int n_out, int rows_per_block);
MI355PPO_API int mi355ppo_mlp_ppo_categorical_fwd_bwd_f32(
    const float* b_obs, const int64_t* mb_inds, int M, int O, const void* const* actor, const void* const* critic, int A,
    const float* b_actions_f32, const float* b_logprobs, const float* b_advantages, const float* b_returns, const float* b_values,
    double clip_coef, double ent_coef, double vf_coef, int norm_adv, int clip_vloss, const float* adv_mean_den, void* const* actor_grads,
    void* const* critic_grads, float* scalars7, int rows_per_block, void* workspace, size_t workspace_bytes, void* stream);
MI355PPO_API int mi355ppo_mlp_ppo_normal_fwd_bwd_f32(
    const float* b_obs, const int64_t* mb_inds, int M, int O, const void* const* actor_mean, const void* const* critic, const float* logstd,
    int D, const float* mean_shift, const float* b_actions, const float* b_logprobs, const float* b_advantages, const float* b_returns,
    const float* b_values, double clip_coef, double ent_coef, double vf_coef, int norm_adv, int clip_vloss, const float* adv_mean_den,
    void* const* actor_grads, void* const* critic_grads, float* dlogstd, float* scalars7, int rows_per_block, void* workspace,
    size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Host-pointer twins (csrc/host_twins.hip) of the PPO-path entry points above: the same arguments minus `stream` and
 * `workspace`, every pointer a HOST pointer, the call returns when the result is written.  Same math by construction: the
 * row / element functions (GAE step, Categorical row, loss row terms, advantage statistics fold, Adam element, Philox stream)
 * are the device kernels' own, compiled for the host from the same headers (csrc/ppo_rows.h, catrow.h, common.h) without FMA
 * contraction; results differ from the device's only by libm vs the device math library (expf / logf: a few ulp) and by the
 * order of the f64 reductions (row order here).  Serial.  They serve BASELINE config A -- cleanrl/ppo.py on CPU (`--no-cuda`:
 * CartPole, num_envs = 4) -- and the world_size-2 gloo tests through cleanrl_amd/host_ops.py.  They are NOT a fallback: nothing
 * routes a device pointer here, and a CUDA device without the HIP kernels raises (cleanrl_amd/_lib.py).
 * Reference lines: as for the device entry point of the same name.
 */
MI355PPO_API int mi355ppo_gae_f32_cpu(const float* rewards, const float* dones, const float* values, const float* next_done,
                                      const float* next_value, float* advantages, float* returns, int T, int N, double gamma,
                                      double gae_lambda);
MI355PPO_API int mi355ppo_categorical_sample_f32_cpu(const float* logits, const float* noise_exp1, uint64_t seed, uint64_t offset,
                                                     int64_t* action_i64, float* action_f32, float* logprob, float* entropy,
                                                     int B, int A);
MI355PPO_API int mi355ppo_categorical_logprob_entropy_f32_cpu(const float* logits, const int64_t* action_i64,
                                                              const float* action_f32, float* logprob, float* entropy, int B,
                                                              int A);
MI355PPO_API int mi355ppo_categorical_logprob_entropy_bwd_f32_cpu(const float* logits, const int64_t* action_i64,
                                                                  const float* action_f32, const float* g_logprob,
                                                                  const float* g_entropy, float* dlogits, int B, int A);
MI355PPO_API int mi355ppo_normal_sample_f32_cpu(const float* mean, const float* logstd, const float* noise_std_normal, uint64_t seed,
                                                uint64_t offset, float* action, float* logprob_sum, float* entropy_sum, int B,
                                                int D);
MI355PPO_API int mi355ppo_normal_logprob_entropy_f32_cpu(const float* mean, const float* logstd, const float* action,
                                                         float* logprob_sum, float* entropy_sum, int B, int D);
MI355PPO_API int mi355ppo_normal_logprob_entropy_bwd_f32_cpu(const float* mean, const float* logstd, const float* action,
                                                             const float* g_logprob, const float* g_entropy, float* dmean,
                                                             float* dlogstd_rows, int B, int D);
MI355PPO_API int mi355ppo_loss_categorical_fwd_bwd_f32_cpu(const float* new_logits, const float* new_value, const int64_t* mb_inds,
                                                           const float* b_actions_f32, const float* b_logprobs,
                                                           const float* b_advantages, const float* b_returns, const float* b_values,
                                                           int M, int A, double clip_coef, double ent_coef, double vf_coef,
                                                           int norm_adv, int clip_vloss, const float* adv_mean_den, float* scalars7,
                                                           float* dlogits, float* dvalue);
MI355PPO_API int mi355ppo_loss_normal_fwd_bwd_f32_cpu(const float* new_mean, const float* logstd, const float* new_value,
                                                      const int64_t* mb_inds, const float* b_actions, const float* b_logprobs,
                                                      const float* b_advantages, const float* b_returns, const float* b_values,
                                                      int M, int D, double clip_coef, double ent_coef, double vf_coef, int norm_adv,
                                                      int clip_vloss, const float* adv_mean_den, float* scalars7, float* dmean,
                                                      float* dlogstd, float* dvalue);
MI355PPO_API int mi355ppo_clip_adam_f32_cpu(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                            double grad_scale, double max_grad_norm, double lr, double beta1, double beta2,
                                            double eps, int64_t step, float* total_norm_out);
MI355PPO_API int mi355ppo_obs_u8_to_f32_cpu(const uint8_t* src_u8, const int64_t* inds, float* dst_f32, int64_t rows,
                                            int64_t row_bytes, int scale_255);

/* ---------------------------------------------------------------------------------------------
 * FC   Linear(3136, 512) + ReLU of the NatureCNN (cleanrl/ppo_atari_multigpu.py:144-145) on the bf16 matrix pipe
 * (kernel Z, csrc/gemmz.hip): every f32 operand is split into three bf16 terms that sum to it exactly, so products of terms
 * are exact in f32 and bf16 MFMAs with f32 accumulation do the work of f32 MFMAs in a fraction of their matrix-pipe time.
 * The six largest of the nine term pairs are multiplied (MI355PPO_BF16_PAIRS=9: all nine, every f32 product exact); the
 * three dropped pairs are together below the rounding of one f32 multiply.  The weight matrix is split AHEAD (once per
 * optimizer step) into MFMA fragment order (`pack`); the activations are loaded coalesced, transposed through wave-private
 * LDS and split in registers.  (The first generation -- lane = row gathers of both operands, round 2's kernel X -- kept the
 * vector-memory front end 83 % busy and the matrix pipe 37 % busy: profiles/r03_pmc_busy_kernels_x_c.csv.)
 *   pack  : B (N,K) f32 with leading dimension ldb -> mi355ppo_fc_pack_bytes(N, K) bytes, 16-byte aligned
 *   fwd   : h (M,N) = relu(a (M,K; lda) @ B^T + bias)         pack = pack(W  (N = 512,  K = 3136))
 *   dgrad : da (M,N) = (dz (M,K; lddz) @ B^T) * (act_in > 0)  pack = pack(Wt (N = 3136, K = 512)): the ReLU backward of the
 *           layer that produced act_in (conv3) is applied where the gradient is produced
 * All matrices row-major; h / da / act_in dense.  K % 16 == 0; a / dz 16-byte aligned with leading dimensions that are
 * multiples of 4 floats, below 4 GiB in all. */
MI355PPO_API size_t mi355ppo_fc_pack_bytes(int N, int K);
MI355PPO_API int mi355ppo_fc_pack_f32(const float* B, int ldb, int N, int K, void* pack, void* stream);
MI355PPO_API int mi355ppo_fc_fwd_relu_packed_f32(const float* a, int lda, const void* pack, const float* bias, float* h,
                                                 int M, int N, int K, void* stream);
MI355PPO_API int mi355ppo_fc_dgrad_mask_packed_f32(const float* dz, int lddz, const void* pack, const float* act_in, float* da,
                                                   int M, int N, int K, void* stream);
/* Forward for rollout-sized and small minibatches (M < 8192: a rollout step's 1,024 envs, config B's 4,096-row minibatch).  M / 64 x N / 64 wave tiles alone cannot fill the chip,
 * so K is split over the grid: raw f32 partials go to `ws` (mi355ppo_fc_fwd_workspace_bytes(M, N, K) bytes, 16-byte aligned; 0 bytes =
 * no split needed), one more pass adds them in a fixed order, then bias and ReLU -- Agent.network[7:9] of the rollout's policy forward
 * (cleanrl/ppo_atari_multigpu.py:144-145,262-264) without a library GEMM.  With ws = NULL: mi355ppo_fc_fwd_relu_packed_f32. */
MI355PPO_API size_t mi355ppo_fc_fwd_workspace_bytes(int M, int N, int K);
MI355PPO_API int mi355ppo_fc_fwd_relu_packed_ws_f32(const float* a, int lda, const void* pack, const float* bias, float* h,
                                                    int M, int N, int K, void* ws, size_t ws_bytes, void* stream);

/* The same kernel family on the convolutions of layers 2 and 3 (cleanrl/ppo_atari_multigpu.py:139-142): a row of the GEMM is an
 * output pixel (forward) or a pixel of the data gradient's grid, a k-step 64 contiguous bytes of a tap row of its window
 * (channels-last activations; zero padding through the buffer range check).  `pack` = mi355ppo_fc_pack_f32 of the layer's
 * (N, K) f32 matrix from mi355ppo_cnn_repack_weights_f32 -- mode 0 (forward: N = 64, K = 512 / 576), mode 1 (layer-3 data
 * gradient: N = 64, K = 576), mode 2 (layer-2 data gradient: N = 128 = 4 stride-parity classes x 32 channels, K = 256).
 *   fwd  : dst (images, Hout, Hout, 64) = relu(conv(src (images, Hin, Hin, Cin)) + bias)
 *   dgrad: dsrc (images, Hin, Hin, Cin) = conv_transpose(dz (images, Hout, Hout, 64)) * (act_in > 0)
 * Tensors channels-last f32, 16-byte aligned, sources below 4 GiB. */
MI355PPO_API int mi355ppo_cnn_conv_fwd_packed_f32(const float* src, const void* pack, const float* bias, float* dst,
                                                  int64_t images, int layer, void* stream);
MI355PPO_API int mi355ppo_cnn_conv_dgrad_packed_f32(const float* dz, const void* pack, const float* act_in, float* dsrc,
                                                    int64_t images, int layer, void* stream);

/* ReLU masks as bits.  The data gradients read the forward activation of the layer below only for its sign -- the reference's
 * autograd keeps the whole f32 tensor for `threshold_backward` (cleanrl/ppo_atari_multigpu.py:137-145: nn.ReLU after every layer).
 * The *_bits forwards write, beside the f32 activation, ONE BIT per element: word w, bit b <-> element 32 w + b of the flat
 * channels-last tensor, set where the activation is > 0 (numel / 32 uint32 words: 400 / 162 / 98 per image for a1 / a2 / a3);
 * the *_bits / maskbits data gradients take those words instead of `act_in` -- 1/32 of the mask bytes, identical results.
 *   conv1q_fwd_bits            : kernel Q (layer 1), also a1's mask       -> conv_dgrad_packed_bits(layer 2)
 *   conv_fwd_packed_bits(2, 3) : kernel Z forward, also a2's / a3's mask  -> conv_dgrad_packed_bits(layer 3) / fc_dgrad_maskbits
 * dst / dsrc / da must sit on a 128-byte boundary (a 32-element tile of the tensor = one word). */
MI355PPO_API int mi355ppo_cnn_conv1q_fwd_bits(const void* src_u8, const int64_t* inds, const void* pack, const float* bias,
                                              float* dst, uint32_t* mask_bits, int64_t images, void* stream);
MI355PPO_API int mi355ppo_cnn_conv_fwd_packed_bits_f32(const float* src, const void* pack, const float* bias, float* dst,
                                                       uint32_t* mask_bits, int64_t images, int layer, void* stream);
MI355PPO_API int mi355ppo_cnn_conv_dgrad_packed_bits_f32(const float* dz, const void* pack, const uint32_t* mask_bits, float* dsrc,
                                                         int64_t images, int layer, void* stream);
MI355PPO_API int mi355ppo_fc_dgrad_maskbits_packed_f32(const float* dz, int lddz, const void* pack, const uint32_t* mask_bits,
                                                       float* da, int M, int N, int K, void* stream);

/* FC weight gradient (csrc/fcw.hip): dW (N,K) = dz (M,N)^T @ a (M,K), the batch cut into slabs whose partials are added in a
 * fixed order (deterministic).  Kernel W (bf16 pipe, both operands transposed through LDS and split in registers) when
 * N == 512, K % 64 == 0, M % 16 == 0 and M >= 1024; kernel Y (f32 matrix pipe) otherwise.  dz takes a leading dimension (even); a is dense.  N % 64 == 0,
 * K % 224 == 0 (whole 64 x 224 wave tiles: 512 x 3136 = 8 x 14 of them); dz 8-byte, a and the workspace 16-byte aligned.
 * hwc_channels = C > 0: the columns of a are features in (h, w, c) order with C channels (the trunk's layout) and dW is
 * written in the reference's (c, h, w) order, i.e. directly as the gradient of Linear(3136,512).weight; 0: as computed. */
MI355PPO_API size_t mi355ppo_fc_wgrad_workspace_bytes(int M, int N, int K);
MI355PPO_API int mi355ppo_fc_wgrad_kernel(int M, int N, int K);      /* 'W' or 'Y': the kernel a call of this shape runs (profiling aid) */
/* 'H', 'W' or 'Y': the kernel mi355ppo_fc_wgrad_f16x2_f32 runs for this shape with a dense dz (ABI 1.9).  Kernel H (csrc/gemmh.hip, round 6)
 * streams both operands through a workgroup-wide LDS ring, split once, fragments by LDS transpose reads; from 4,096 rows on. */
MI355PPO_API int mi355ppo_fc_wgrad_kernel_f16x2(int M, int N, int K);
MI355PPO_API int mi355ppo_fc_wgrad_f32(const float* dz, int lddz, const float* a, float* dW, int M, int N, int K, int hwc_channels,
                                       void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CNN  NatureCNN convolution stack (Agent.network convs, cleanrl/ppo_atari_multigpu.py:136-142) as
 * f32-MFMA implicit GEMMs on channels-last tensors: forward with fused uint8 gather + /255 + bias +
 * ReLU, data gradient with the ReLU-backward mask fused, weight + bias gradient.  f32 in, f32
 * accumulate (v_mfma_f32_32x32x2_f32): same precision class as the reference's f32 convolutions;
 * results differ from a CPU/MIOpen convolution only by summation order (tolerance in
 * tests/test_gpu_cnn.py).
 *   layer 1: Conv2d(4,32,8,stride 4)  on (84,84,4)  -> (20,20,32)
 *   layer 2: Conv2d(32,64,4,stride 2) on (20,20,32) -> (9,9,64)
 *   layer 3: Conv2d(64,64,3,stride 1) on (9,9,64)   -> (7,7,64)
 * Activations are (images, H, W, C) row-major ("NHWC").  Weights enter the kernels as repacked
 * matrices Bt (produced by mi355ppo_cnn_repack_weights_f32 from torch's (Cout,Cin,KH,KW) tensor):
 *   mode 0  forward / weight-gradient order  Bt[Cout][(kh,kw,cin)]
 *   mode 1  layer-3 data gradient            Bt[Cin][(r,c,cout)]   (taps flipped)
 *   mode 2  layer-2 data gradient            Bt[4][Cin][(r,c,cout)] (one 2x2-tap matrix per parity class)
 *   mode 3  layer-3 data gradient, per border class: 25 matrices [Cin][(r',c',cout)] for the 5x5 (row class,
 *           column class) tap windows (mi355ppo_cnn_conv_dgrad_f32_variant(..., variant 5)); 81*4096 floats
 *   mode 5  layer-2 data gradient, per border class: 9 matrices [4*Cin][(r',c',cout)] for the 3x3 (row class, column
 *           class) tap windows of the 10x10 class grid (variant 6); 16*128*64 floats
 *   mode 4  layer 1 only: the integer-digit pack of kernel Q (csrc/conv1q.hip): every weight as a 31-bit
 *           fixed-point number on its output channel's scale, four signed radix-256 digits in the operand
 *           layout of v_mfma_i32_32x32x32_i8, + accumulator start values + per-channel scales;
 *           mi355ppo_cnn_conv1q_pack_bytes() = 33,408 bytes.  Consumed by forward variant 6.
 * modes 0-2 have Cout*Cin*KH*KW floats.
 */
MI355PPO_API int mi355ppo_cnn_repack_weights_f32(const float* W, float* Bt, int layer, int mode, void* stream);

/* Kernel Q: layer-1 forward on the integer matrix pipe.  The uint8 frame bytes are exact int8 operands
 * (v - 128), the weights four int8 digits; int32 accumulation is exact, the only roundings are the
 * weight's one-time rounding to 2^-30 of its channel's largest weight and five f32 roundings per output
 * (Horner over the digits, scale, bias) -- closer to the float64 convolution than an f32 fma chain of
 * 256 terms, at 1/8 of its matrix-pipe time.  Same contract as mi355ppo_cnn_conv_fwd_f32(layer 1). */
MI355PPO_API size_t mi355ppo_cnn_conv1q_pack_bytes(void);
MI355PPO_API int mi355ppo_cnn_conv1q_pack(const float* W, void* pack, void* stream);
MI355PPO_API int mi355ppo_cnn_conv1q_fwd(const void* src_u8, const int64_t* inds, const void* pack, const float* bias,
                                         float* dst, int64_t images, void* stream);

/* dst = relu(conv(src) + bias).  layer 1: src is the uint8 rollout buffer (rows_total, 84,84,4) and
 * `inds` (images) int64 optionally gathers rows (b_obs[mb_inds]); the /255 is applied in registers,
 * correctly rounded.  layers 2,3: src is f32, inds must be NULL. */
MI355PPO_API int mi355ppo_cnn_conv_fwd_f32(const void* src, const int64_t* inds, const float* Bt, const float* bias,
                                           float* dst, int64_t images, int layer, void* stream);
/* `variant` (tuning/testing): 0 = auto (= 2; 4 for tensors beyond 4 GiB),
 * 2 = fixed-geometry streaming kernel F (weights resident in LDS, A fragments fetched straight into a register ring,
 *     taps as compile-time immediates, buffer loads/stores; tensors must be < 4 GiB), 4 = its run-time-geometry
 *     predecessor S; data gradient only: 3 = one launch per stride-parity class (layer 2), 5 = layer 3 split into
 *     its 25 border classes so that no padding zeros are multiplied (needs the mode-3 repack), 6 = layer 2 split
 *     into the 9 border classes of its class grid (needs the mode-5 repack);
 *     forward of layer 1 only: 6 = kernel Q (Bt = the mode-4 pack). */
MI355PPO_API int mi355ppo_cnn_conv_fwd_f32_variant(const void* src, const int64_t* inds, const float* Bt, const float* bias,
                                                   float* dst, int64_t images, int layer, int variant, void* stream);

/* The three forward layers back to back (a1 (images,20,20,32), a2 (images,9,9,64), a3 (images,7,7,64) out): one call
 * for inference-sized batches, where the host-side cost of a launch matters.  conv1_variant: 0 (bt1 = mode-0 matrix)
 * or 6 (bt1 = mode-4 pack, kernel Q). */
MI355PPO_API int mi355ppo_cnn_trunk_fwd_f32(const void* obs_u8, const int64_t* inds, const float* bt1, const float* b1,
                                            const float* bt2, const float* b2, const float* bt3, const float* b3,
                                            float* a1, float* a2, float* a3, int64_t images, int conv1_variant,
                                            void* stream);

/* dsrc = conv_transpose(dz) * (act_in > 0): gradient w.r.t. the layer's INPUT activation act_in
 * (itself a ReLU output), i.e. the pre-activation gradient of the previous layer.  layer = 2 or 3;
 * Bt in mode 2 / mode 1. */
MI355PPO_API int mi355ppo_cnn_conv_dgrad_f32(const float* dz, const float* Bt, const float* act_in, float* dsrc,
                                             int64_t images, int layer, void* stream);
MI355PPO_API int mi355ppo_cnn_conv_dgrad_f32_variant(const float* dz, const float* Bt, const float* act_in, float* dsrc,
                                                     int64_t images, int layer, int variant, void* stream);

/* dW (torch layout (Cout,Cin,KH,KW)) and db (Cout) from the layer input `src` (layer 1: uint8 + inds
 * as above) and the pre-activation gradient dz (images, Hout, Wout, Cout).  Overwrites dW / db.
 * Deterministic: per-workgroup partials in `workspace`, summed in a fixed order.
 * Layer 1: kernel P (csrc/conv1p.hip; bf16 pipe: the uint8 taps are exact bf16 operands, dz is split into three bf16 terms).
 * Layers 2, 3: kernel V (csrc/convw.hip; bf16 pipe, both operands transposed through LDS and split in registers, the batch
 * cut into slabs) for batches of a multiple of 16 images with every tensor below 4 GiB; kernel T (f32 pipe, csrc/conv.hip)
 * otherwise.  mi355ppo_cnn_conv_wgrad_kernel: 'P', 'V' or 'T', the kernel a call of this size runs (profiling aid). */
MI355PPO_API size_t mi355ppo_cnn_conv_wgrad_workspace_bytes(int64_t images, int layer);
MI355PPO_API int mi355ppo_cnn_conv_wgrad_kernel(int64_t images, int layer);
MI355PPO_API int mi355ppo_cnn_conv_wgrad_f32(const void* src, const int64_t* inds, const float* dz, float* dW, float* db,
                                             int64_t images, int layer, void* workspace, size_t workspace_bytes,
                                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Heads  actor = Linear(H, A) and critic = Linear(H, 1) of Agent (cleanrl/ppo_atari_multigpu.py:148-149,
 * used at :151,157-159), forward and backward.  Degenerate as GEMMs (A + 1 <= 8 columns); here one
 * bandwidth-bound pass over the hidden activations each.  H must be 512 (NatureCNN), 1 <= A <= 18 (every ALE action set; from A = 8 on the weight rows sit in LDS instead of registers -- round 6).
 *   h (M,H); Wa (A,H), ba (A); Wc (H) [= critic.weight (1,H)], bc (1)   ->   logits (M,A), value (M)
 *   backward: dlogits (M,A), dvalue (M)  ->  dh (M,H), dWa (A,H), dba (A), dWc (H), dbc (1)
 * Weight gradients are summed in a fixed order (deterministic); all outputs are overwritten.
 */
MI355PPO_API int mi355ppo_heads_fwd_f32(const float* h, const float* Wa, const float* ba, const float* Wc, const float* bc,
                                        float* logits, float* value, int M, int A, int H, void* stream);
MI355PPO_API size_t mi355ppo_heads_bwd_workspace_bytes(int M, int A);
MI355PPO_API int mi355ppo_heads_bwd_f32(const float* h, const float* Wa, const float* Wc, const float* dlogits,
                                        const float* dvalue, float* dh, float* dWa, float* dba, float* dWc, float* dbc,
                                        int M, int A, int H, void* workspace, size_t workspace_bytes, void* stream);
/* The same for an h that is the output of a ReLU (the FC layer's, :145): writes dz = dh * (h > 0), the gradient with respect
 * to that layer's pre-activation, and dbh (H) = its column sums (that layer's bias gradient) -- autograd's
 * threshold_backward pass and the bias reduction of the Linear below, done where h and dh are in registers anyway.
 * dz takes a row pitch lddz (floats, a multiple of 4, >= H). */
MI355PPO_API int mi355ppo_heads_bwd_relu_f32(const float* h, const float* Wa, const float* Wc, const float* dlogits,
                                             const float* dvalue, float* dz, int lddz, float* dWa, float* dba, float* dWc,
                                             float* dbc, float* dbh, int M, int A, int H, void* workspace,
                                             size_t workspace_bytes, void* stream);

/* Every weight pack of the NatureCNN agent in one launch, from the parameters in torch's layouts (round 4): W1 (32,4,8,8) -> kernel
 * Q's pack (mi355ppo_cnn_conv1q_pack_bytes()); W2 (64,32,4,4) -> kernel Z's layer-2 forward pack [mi355ppo_fc_pack_bytes(64, 512)] and
 * data-gradient pack [(128, 256)]; W3 (64,64,3,3) -> layer-3 forward and data-gradient packs [(64, 576) each]; Wfc (512, 3136) in the
 * reference's (c, h, w) feature order -> the FC forward pack [(512, 3136), features re-ordered (h, w, c)] and data-gradient pack
 * [(3136, 512)].  Bit-identical to mi355ppo_cnn_repack_weights_f32 (modes 4 / 0 / 1 / 2) + mi355ppo_fc_pack_f32 per matrix -- what the
 * learner re-derives after each optimizer step (cleanrl/ppo_atari_multigpu.py:377).  A null output skips its piece. */
MI355PPO_API int mi355ppo_nature_packs_f32(const float* W1, const float* W2, const float* W3, const float* Wfc, void* qpack,
                                           void* conv2_fwd, void* conv3_fwd, void* conv3_dgrad, void* conv2_dgrad, void* fc_fwd,
                                           void* fc_dgrad, void* stream);

/* Rollout step of the NatureCNN agent, fused behind the trunk (round 4): Linear(3136,512) with K split over the grid (raw partials in
 * `workspace`, mi355ppo_fc_fwd_workspace_bytes(M, 512, 3136) bytes; M < 8192), then ONE kernel for the partial fold + bias + ReLU,
 * the two heads and the Categorical draw -- Agent.get_action_and_value(next_obs) from conv3's output on
 * (cleanrl/ppo_atari_multigpu.py:144-149,155-159) -- writing action / log-prob / value (and optionally the hidden activations) in
 * place.  Bit-identical to mi355ppo_fc_fwd_relu_packed_ws_f32 -> mi355ppo_heads_fwd_f32 -> mi355ppo_categorical_sample_ctr_f32. */
MI355PPO_API int mi355ppo_fc_heads_act_categorical_f32(const float* a3, int lda, const void* fc_pack, const float* fc_bias,
                                                       const float* Wa, const float* ba, const float* Wc, const float* bc, int M, int A,
                                                       int H, int K, const float* noise_exp1, uint64_t seed, uint64_t offset,
                                                       const uint64_t* offset_base, int64_t* action_i64, float* action_f32,
                                                       float* logprob, float* value, float* hidden_out, void* workspace,
                                                       size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Round 5 -- the NatureCNN GEMMs on a TWO-TERM f16 split (csrc/f16split.h): three v_mfma_f32_32x32x16_f16 per f32 product where the
 * three-term bf16 split of the entry points above needs six, f32 accumulation, products of terms exact.  Every split tensor is scaled
 * by a power of two derived from its amax record (MI355PPO_AMAX_WORDS uint32, 64-byte aligned, see above): an entry point READS the
 * record of each f32 operand it splits (`*_amax` inputs; required) and FOLDS the values it stores into the record of its result
 * (`*_amax` outputs; null = nobody will split the result); the caller zeroes the output records of a pass before its first launch.
 * Weights travel as "f16x2 packs": [64-byte header: max |B|][k-step][32-column tile][hi, lo][64 lanes][8 f16] of s B.
 * The reference arithmetic is the same as for the bf16 entry points they mirror (cleanrl/ppo_atari_multigpu.py:136-148 forward, :358
 * backward: f32 Conv2d / Linear); error against float64 at or below theirs (tools/err_f16x2.py, tests/test_gpu_f16x2.py).
 */
MI355PPO_API int mi355ppo_absmax_f32(const float* x, int64_t n, uint32_t* amax, void* stream);   /* amax = max(amax, max |x|) */
MI355PPO_API size_t mi355ppo_fc_pack_f16x2_bytes(int N, int K);
MI355PPO_API int mi355ppo_fc_pack_f16x2_f32(const float* B, int ldb, int N, int K, const uint32_t* b_amax, void* pack, void* stream);
/* mi355ppo_nature_packs_f32 with the six kernel-Z packs as f16x2 packs [mi355ppo_fc_pack_f16x2_bytes of the same shapes]; `w_amax` =
 * three records (3 * MI355PPO_AMAX_WORDS) that receive max |W2|, |W3|, |Wfc| -- every slot written by this call (no zeroing needed). */
MI355PPO_API int mi355ppo_nature_packs_f16x2_f32(const float* W1, const float* W2, const float* W3, const float* Wfc, void* qpack,
                                                 void* conv2_fwd, void* conv3_fwd, void* conv3_dgrad, void* conv2_dgrad, void* fc_fwd,
                                                 void* fc_dgrad, uint32_t* w_amax, void* stream);
/* mi355ppo_cnn_conv1q_fwd / _bits (mask_bits may be null) that also records max(dst) -- the layer-2 forward's A-operand scale. */
MI355PPO_API int mi355ppo_cnn_conv1q_fwd_amax(const void* src_u8, const int64_t* inds, const void* pack, const float* bias, float* dst,
                                              uint32_t* mask_bits, int64_t images, uint32_t* dst_amax, void* stream);
/* mi355ppo_cnn_conv_fwd_packed_f32 / _bits_f32 (mask_bits may be null) */
MI355PPO_API int mi355ppo_cnn_conv_fwd_packed_f16x2_f32(const float* src, const void* pack, const float* bias, float* dst, uint32_t* mask_bits,
                                                        int64_t images, int layer, const uint32_t* src_amax, uint32_t* dst_amax, void* stream);
/* mi355ppo_cnn_conv_dgrad_packed_f32 / _bits_f32: the ReLU mask from mask_bits if given, else from act_in */
MI355PPO_API int mi355ppo_cnn_conv_dgrad_packed_f16x2_f32(const float* dz, const void* pack, const float* act_in, const uint32_t* mask_bits,
                                                          float* dsrc, int64_t images, int layer, const uint32_t* dz_amax, uint32_t* dsrc_amax,
                                                          void* stream);
/* 'R', 'B' or 'Z': the kernel a forward (dgrad = 0) / bit-masked data-gradient (dgrad = 1) call of this size and layer runs (profiling aid).
 * Kernel R (csrc/convr.hip) holds the source of a group of images in LDS, split once, and takes both forwards and the layer-3 data
 * gradient at every size and the layer-2 data gradient from 512 images on; its results are kernel Z's bit for bit (same products, same order).
 * 'B' = kernel RB (csrc/convrb.hip): kernel R's layer-2 data gradient from 3,072 images on with three images per group and the rows dealt to
 * tiles by border class (the products with a zero-border operand are not issued: two thirds of the matrix instructions, the same bits). */
MI355PPO_API int mi355ppo_cnn_conv_packed_kernel_f16x2(int64_t images, int layer, int dgrad);
/* 'G' or 'Z': the kernel mi355ppo_fc_fwd_relu_packed_f16x2_f32 without a K split (dgrad = 0) / mi355ppo_fc_dgrad_packed_f16x2_f32 with mask
 * bits (dgrad = 1) runs for this shape (ABI 1.9).  Kernel G (csrc/gemmg.hip, round 6) streams both operands through workgroup-wide LDS
 * rings, A split once per workgroup; its results are kernel Z's bit for bit (same products, same order). */
MI355PPO_API int mi355ppo_fc_packed_kernel_f16x2(int M, int N, int K, int dgrad);
/* mi355ppo_cnn_conv_wgrad_f32 for layers 2 / 3 (kernel V); other batches fall to the f32-pipe kernel and ignore the records */
MI355PPO_API int mi355ppo_cnn_conv_wgrad_f16x2_f32(const float* src, const float* dz, float* dW, float* db, int64_t images, int layer,
                                                   void* workspace, size_t workspace_bytes, const uint32_t* src_amax, const uint32_t* dz_amax,
                                                   void* stream);
/* 'U', 'V', 'P' or 'T': the kernel the two f16x2 weight-gradient entry points run for this batch and layer (1..3) -- kernel U (csrc/convu.hip) while the
 * tensors stay inside the 32-bit buffer range, else kernel V / P, else the f32-pipe kernel T (ABI 1.9; profiling aid: bench.py labels its rows with it). */
MI355PPO_API int mi355ppo_cnn_conv_wgrad_kernel_f16x2(int64_t images, int layer);
/* mi355ppo_cnn_conv_wgrad_f32 for layer 1 (kernel P) with dz in two f16 terms; the uint8 frames are exact f16 operands: only dz's record */
MI355PPO_API int mi355ppo_cnn_conv1_wgrad_f16x2(const void* src_u8, const int64_t* inds, const float* dz, float* dW, float* db, int64_t images,
                                                void* workspace, size_t workspace_bytes, const uint32_t* dz_amax, void* stream);
/* mi355ppo_fc_fwd_relu_packed_ws_f32 (ws may be null / 0: whole-K wave tiles); the K-split route records no h_amax (pass null) */
MI355PPO_API int mi355ppo_fc_fwd_relu_packed_f16x2_f32(const float* a, int lda, const void* pack, const float* bias, float* h, int M, int N,
                                                       int K, void* ws, size_t ws_bytes, const uint32_t* a_amax, uint32_t* h_amax, void* stream);
/* mi355ppo_fc_dgrad_mask_packed_f32 / _maskbits_ */
MI355PPO_API int mi355ppo_fc_dgrad_packed_f16x2_f32(const float* dz, int lddz, const void* pack, const float* act_in, const uint32_t* mask_bits,
                                                    float* da, int M, int N, int K, const uint32_t* dz_amax, uint32_t* da_amax, void* stream);
/* mi355ppo_fc_wgrad_f32 (kernel W; other shapes fall to the f32-pipe kernel Y and ignore the records) */
MI355PPO_API int mi355ppo_fc_wgrad_f16x2_f32(const float* dz, int lddz, const float* a, float* dW, int M, int N, int K, int hwc_channels,
                                             void* workspace, size_t workspace_bytes, const uint32_t* dz_amax, const uint32_t* a_amax,
                                             void* stream);
/* mi355ppo_heads_bwd_relu_f32 that also records max |dz| (the FC layer's data / weight gradients scale dz by it) */
MI355PPO_API int mi355ppo_heads_bwd_relu_amax_f32(const float* h, const float* Wa, const float* Wc, const float* dlogits, const float* dvalue,
                                                  float* dz, int lddz, float* dWa, float* dba, float* dWc, float* dbc, float* dbh, int M, int A,
                                                  int H, void* workspace, size_t workspace_bytes, uint32_t* dz_amax, void* stream);
/* mi355ppo_fc_heads_act_categorical_f32 with an f16x2 `fc_pack` and a3's record */
MI355PPO_API int mi355ppo_fc_heads_act_categorical_f16x2_f32(const float* a3, int lda, const void* fc_pack, const float* fc_bias,
                                                             const float* Wa, const float* ba, const float* Wc, const float* bc, int M, int A,
                                                             int H, int K, const float* noise_exp1, uint64_t seed, uint64_t offset,
                                                             const uint64_t* offset_base, int64_t* action_i64, float* action_f32,
                                                             float* logprob, float* value, float* hidden_out, void* workspace,
                                                             size_t workspace_bytes, const uint32_t* a3_amax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Test / benchmark support -- NOT part of the reference path.  One step of the device-resident synthetic Atari
 * vector env (cleanrl_amd/envs.py::DeviceSyntheticAtariVecEnv) that stands in for envpool / ALE, which this image
 * does not have: obs[n] = planes[cursor_n .. cursor_n+3] (mod pool) as (N,4,84,84) uint8; with advance != 0 first
 * reward in {-1,0,+1} (P = .05,.9,.05), done ~ Bernoulli(done_p), cursor += 1 or jumps on done (Philox(seed; n, step)).
 */
MI355PPO_API int mi355ppo_synth_atari_step_u8(const uint8_t* planes, int pool, int64_t* cursor, uint64_t seed, uint64_t step,
                                              uint8_t* obs, float* reward, float* done, int N, double done_p, int advance,
                                              void* stream);
MI355PPO_API int mi355ppo_synth_atari_step_ctr_u8(const uint8_t* planes, int pool, int64_t* cursor, uint64_t seed, uint64_t step,
                                                  const uint64_t* step_base, uint8_t* obs, float* reward, float* done, int N,
                                                  double done_p, int advance, void* stream);   /* step_eff = step + *step_base */
/* The same step with obs written pixel-interleaved, (N,84,84,4): the rollout rows' layout (gather + relayout in one launch). */
MI355PPO_API int mi355ppo_synth_atari_step_hwc_ctr_u8(const uint8_t* planes, int pool, int64_t* cursor, uint64_t seed, uint64_t step,
                                                      const uint64_t* step_base, uint8_t* obs, float* reward, float* done, int N,
                                                      double done_p, int advance, void* stream);

/* The continuous-control stand-in (cleanrl_amd/envs.py::DeviceSyntheticContinuousVecEnv; bench.py --config E), one launch per env
 * step: a = clip(action, -1, 1); next = noise[(k + *k_base) % bank][n] + state[n] @ At + a @ Bm; reward = next . w - 0.1 |a|^2;
 * truncation after `horizon` steps (state := reset_state).  state (N, O) is updated in place and copied to obs_out. */
MI355PPO_API int mi355ppo_synth_continuous_step_f32(float* state, const float* reset_state, const float* At, const float* Bm, const float* w,
                                                    const float* noise, int bank, uint64_t k, const uint64_t* k_base, float* steps,
                                                    double horizon, const float* action, float* obs_out, float* reward, float* done, int N,
                                                    int O, int D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355PPO_H_ */
