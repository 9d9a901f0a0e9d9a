// Torch-free driver of the conv entry points for HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE).
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/conv_traffic.cpp -o tools/conv_traffic \
//         -Lcleanrl_amd/csrc -lmi355ppo -Wl,-rpath,'$ORIGIN/../cleanrl_amd/csrc'
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o fetch -- tools/conv_traffic [images] [reps] [dump.bin]
// Also prints the mean duration of every entry point (HIP events on the launch stream, one JSON line) and, with a third
// argument, dumps dW/db of the three layers and strided samples of the activations / gradients to a file, so that two
// builds or two tuning-switch settings can be compared with `cmp` (address-only changes must stay bit-identical).
//
// Launches, at the minibatch size of config C (32,768 images gathered from a 131,072-row uint8 rollout buffer), the
// eight conv kernels of one minibatch update in their order of use, `reps` times each, preceded by calibration
// copies of known byte counts (1 GiB read + 1 GiB written, 16 B and 4 B per lane; sized past the 256 MiB Infinity
// Cache) -- MI355X_MICROARCH.md asks for FETCH_SIZE / WRITE_SIZE to be calibrated per access width on gfx950.
// Starts in about a second (no Python, no torch), so a whole pass costs a few seconds of GPU-box time.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "mi355ppo.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } } while (0)
#define ABI(x) do { int rc_ = (x); if (rc_) { std::fprintf(stderr, "%s:%d rc=%d %s\n", __FILE__, __LINE__, rc_, mi355ppo_last_error()); std::exit(3); } } while (0)

__global__ void calib_copy_b128(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void calib_copy_b32(const uint32_t* __restrict__ s, uint32_t* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void fill_f32(float* p, size_t n, uint32_t salt) {      // signed pseudo-random values in (-1, 1), ~half positive
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + salt; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int32_t)h * (1.0f / 2147483648.0f);
    }
}
__global__ void fill_u8(uint32_t* p, size_t n, uint32_t salt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + salt; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = h;
    }
}
__global__ void fill_inds(int64_t* p, int64_t n, int64_t rows) {    // a slice of a permutation of the rollout rows
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = (i * 40503 + 17) % rows;
}

// order-independent 64-bit hash of a tensor's bit patterns: sum_i bits_i * (2 i + 1) mod 2^64 (integer adds commute: the value does
// not depend on how the grid is scheduled) -- two runs print equal hashes iff (up to collisions) every element is bit-identical
__global__ void hash_u32(const uint32_t* __restrict__ p, size_t n, unsigned long long* out) {
    unsigned long long h = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        h += (unsigned long long)p[i] * (2ull * i + 1ull);
    atomicAdd(out, h);
}

template <class T> static T* dalloc(size_t n) { void* p; CHECK(hipMalloc(&p, n * sizeof(T))); return (T*)p; }

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 32768;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const int64_t rows = 4 * M;                                     // T*N = 131,072 at config C
    const size_t obs_b = (size_t)rows * 28224, a1n = (size_t)M * 20 * 20 * 32, a2n = (size_t)M * 81 * 64, a3n = (size_t)M * 49 * 64;
    hipStream_t st; CHECK(hipStreamCreate(&st));
    uint8_t* obs = dalloc<uint8_t>(obs_b);
    int64_t* inds = dalloc<int64_t>(M);
    float *a1 = dalloc<float>(a1n), *a2 = dalloc<float>(a2n), *a3 = dalloc<float>(a3n);
    float *dz1 = dalloc<float>(a1n), *dz2 = dalloc<float>(a2n), *dz3 = dalloc<float>(a3n);
    float *W1 = dalloc<float>(8192), *W2 = dalloc<float>(32768), *W3 = dalloc<float>(36864);
    float *bt1 = dalloc<float>(8192), *bt2 = dalloc<float>(32768), *bt3 = dalloc<float>(36864);
    float *bt2d = dalloc<float>(32768), *bt3d = dalloc<float>(36864), *bt3c = dalloc<float>(81 * 4096);
    float *bt1q = dalloc<float>(mi355ppo_cnn_conv1q_pack_bytes() / 4), *bt2c = dalloc<float>(16 * 128 * 64);     // round-2 defaults
    // FC layer (kernel X): a3 flat (M, 3136) -> h (M, 512); data gradient back into dz3 with the (a3 > 0) mask
    float *Wfc = dalloc<float>(512 * 3136), *Wfct = dalloc<float>(3136 * 516), *hfc = dalloc<float>((size_t)M * 512), *dzfc = dalloc<float>((size_t)M * 516);
    const size_t wsfcb = mi355ppo_fc_wgrad_workspace_bytes((int)M, 512, 3136);
    float *dWfc = dalloc<float>(512 * 3136); void* wsfc; CHECK(hipMalloc(&wsfc, wsfcb));
    const size_t wsfwdb = mi355ppo_fc_fwd_workspace_bytes((int)M, 512, 3136);
    void* wsfwd = nullptr; if (wsfwdb) CHECK(hipMalloc(&wsfwd, wsfwdb));
    float *bias = dalloc<float>(512), *dW = dalloc<float>(36864), *db = dalloc<float>(64);
    size_t wsb = 0;
    for (int l = 1; l <= 3; l++) { size_t b = mi355ppo_cnn_conv_wgrad_workspace_bytes(M, l); if (b > wsb) wsb = b; }
    void* ws; CHECK(hipMalloc(&ws, wsb));
    const size_t cal_b = (size_t)1 << 30;
    uint4 *c0 = dalloc<uint4>(cal_b / 16), *c1 = dalloc<uint4>(cal_b / 16);

    fill_u8<<<4096, 256, 0, st>>>((uint32_t*)obs, obs_b / 4, 1u);
    fill_inds<<<(unsigned)((M + 255) / 256), 256, 0, st>>>(inds, M, rows);
    if (getenv("CONV_TRAFFIC_NOINDS")) inds = nullptr;              // layer 1 then reads rows 0 .. M-1 in order
    fill_f32<<<4096, 256, 0, st>>>(dz1, a1n, 2u); fill_f32<<<4096, 256, 0, st>>>(dz2, a2n, 3u); fill_f32<<<4096, 256, 0, st>>>(dz3, a3n, 4u);
    fill_f32<<<32, 256, 0, st>>>(W1, 8192, 5u); fill_f32<<<128, 256, 0, st>>>(W2, 32768, 6u); fill_f32<<<144, 256, 0, st>>>(W3, 36864, 7u);
    fill_f32<<<2, 256, 0, st>>>(bias, 512, 8u);
    fill_u8<<<4096, 256, 0, st>>>((uint32_t*)c0, cal_b / 4, 9u);
    ABI(mi355ppo_cnn_repack_weights_f32(W1, bt1, 1, 0, st)); ABI(mi355ppo_cnn_repack_weights_f32(W2, bt2, 2, 0, st));
    ABI(mi355ppo_cnn_repack_weights_f32(W3, bt3, 3, 0, st)); ABI(mi355ppo_cnn_repack_weights_f32(W2, bt2d, 2, 2, st));
    ABI(mi355ppo_cnn_repack_weights_f32(W3, bt3d, 3, 1, st)); ABI(mi355ppo_cnn_repack_weights_f32(W3, bt3c, 3, 3, st));
    ABI(mi355ppo_cnn_repack_weights_f32(W1, bt1q, 1, 4, st)); ABI(mi355ppo_cnn_repack_weights_f32(W2, bt2c, 2, 5, st));
    // kernel Z (round 3): the FC weight and its transpose pre-split into fragment order
    void *pk_fwd, *pk_dg;
    CHECK(hipMalloc(&pk_fwd, mi355ppo_fc_pack_bytes(512, 3136))); CHECK(hipMalloc(&pk_dg, mi355ppo_fc_pack_bytes(3136, 512)));
    const bool bits = getenv("CONV_TRAFFIC_NOBITS") == nullptr;        // ReLU masks as bits (the learner's default) or as the f32 activations
    uint32_t *mb1 = dalloc<uint32_t>(a1n / 32), *mb2 = dalloc<uint32_t>(a2n / 32), *mb3 = dalloc<uint32_t>(a3n / 32);
    const bool conv_z = getenv("CONV_TRAFFIC_CONV_F") == nullptr;      // layers 2 / 3 forward + data gradients: kernel Z (the learner's default) or, for A/B runs, kernel F
    void *pz2, *pz3, *pzd2, *pzd3;
    CHECK(hipMalloc(&pz2, mi355ppo_fc_pack_bytes(64, 512))); CHECK(hipMalloc(&pz3, mi355ppo_fc_pack_bytes(64, 576)));
    CHECK(hipMalloc(&pzd2, mi355ppo_fc_pack_bytes(128, 256))); CHECK(hipMalloc(&pzd3, mi355ppo_fc_pack_bytes(64, 576)));
    fill_f32<<<4096, 256, 0, st>>>(Wfc, 512 * 3136, 10u); fill_f32<<<4096, 256, 0, st>>>(Wfct, 3136 * 516, 11u);
    fill_f32<<<4096, 256, 0, st>>>(dzfc, (size_t)M * 516, 12u);
    ABI(mi355ppo_fc_pack_f32(bt2, 512, 64, 512, pz2, st)); ABI(mi355ppo_fc_pack_f32(bt3, 576, 64, 576, pz3, st));
    ABI(mi355ppo_fc_pack_f32(bt2d, 256, 128, 256, pzd2, st)); ABI(mi355ppo_fc_pack_f32(bt3d, 576, 64, 576, pzd3, st));
    ABI(mi355ppo_fc_pack_f32(Wfc, 3136, 512, 3136, pk_fwd, st)); ABI(mi355ppo_fc_pack_f32(Wfct, 516, 3136, 512, pk_dg, st));
    // CONV_TRAFFIC_F16=1: the same launches on the two-term f16 split (round 5; csrc/f16split.h): f16x2 packs, amax records.  Records:
    // 0 a1, 1 a2, 2 a3, 3 dz3, 4 dz2, 5 dz1 (zeroed at the top of every repetition, filled by the producers' epilogues), 8 dzfc (an input
    // here: mi355ppo_absmax_f32 once), 9 .. 14 the six weight matrices
    const bool f16 = getenv("CONV_TRAFFIC_F16") != nullptr;
    uint32_t* rec = dalloc<uint32_t>(16 * MI355PPO_AMAX_WORDS);
    auto R = [&](int i) { return rec + (size_t)i * MI355PPO_AMAX_WORDS; };
    void *hz2 = nullptr, *hz3 = nullptr, *hzd2 = nullptr, *hzd3 = nullptr, *hk_fwd = nullptr, *hk_dg = nullptr;
    if (f16) {
        CHECK(hipMemsetAsync(rec, 0, 16 * MI355PPO_AMAX_WORDS * 4, st));
        CHECK(hipMalloc(&hz2, mi355ppo_fc_pack_f16x2_bytes(64, 512))); CHECK(hipMalloc(&hz3, mi355ppo_fc_pack_f16x2_bytes(64, 576)));
        CHECK(hipMalloc(&hzd2, mi355ppo_fc_pack_f16x2_bytes(128, 256))); CHECK(hipMalloc(&hzd3, mi355ppo_fc_pack_f16x2_bytes(64, 576)));
        CHECK(hipMalloc(&hk_fwd, mi355ppo_fc_pack_f16x2_bytes(512, 3136))); CHECK(hipMalloc(&hk_dg, mi355ppo_fc_pack_f16x2_bytes(3136, 512)));
        ABI(mi355ppo_absmax_f32(dzfc, (int64_t)M * 516, R(8), st));
        ABI(mi355ppo_absmax_f32(bt2, 32768, R(9), st)); ABI(mi355ppo_absmax_f32(bt3, 36864, R(10), st));
        ABI(mi355ppo_absmax_f32(bt2d, 32768, R(11), st)); ABI(mi355ppo_absmax_f32(bt3d, 36864, R(12), st));
        ABI(mi355ppo_absmax_f32(Wfc, 512 * 3136, R(13), st)); ABI(mi355ppo_absmax_f32(Wfct, 3136 * 516, R(14), st));
        ABI(mi355ppo_fc_pack_f16x2_f32(bt2, 512, 64, 512, R(9), hz2, st)); ABI(mi355ppo_fc_pack_f16x2_f32(bt3, 576, 64, 576, R(10), hz3, st));
        ABI(mi355ppo_fc_pack_f16x2_f32(bt2d, 256, 128, 256, R(11), hzd2, st)); ABI(mi355ppo_fc_pack_f16x2_f32(bt3d, 576, 64, 576, R(12), hzd3, st));
        ABI(mi355ppo_fc_pack_f16x2_f32(Wfc, 3136, 512, 3136, R(13), hk_fwd, st)); ABI(mi355ppo_fc_pack_f16x2_f32(Wfct, 516, 3136, 512, R(14), hk_dg, st));
    }
    CHECK(hipStreamSynchronize(st));

    if (getenv("CONV_TRAFFIC_CALIB"))
        for (int r = 0; r < reps; r++) {
            calib_copy_b128<<<8192, 256, 0, st>>>(c0, c1, cal_b / 16);
            calib_copy_b32<<<8192, 256, 0, st>>>((const uint32_t*)c0, (uint32_t*)c1, cal_b / 4);
        }
    hipEvent_t ev[12][2];
    float tot[11] = {0};
    for (auto& e : ev) { CHECK(hipEventCreate(&e[0])); CHECK(hipEventCreate(&e[1])); }
    float *dW1 = dalloc<float>(8192), *dW2 = dalloc<float>(32768), *dW3 = dalloc<float>(36864), *db1 = dalloc<float>(64), *db2 = dalloc<float>(64), *db3 = dalloc<float>(64);
#define TIMED(i, call) do { CHECK(hipEventRecord(ev[i][0], st)); ABI(call); CHECK(hipEventRecord(ev[i][1], st)); } while (0)
    for (int r = 0; r < reps; r++) {                                // one minibatch update's conv launches, in order
        if (f16) {
            CHECK(hipMemsetAsync(rec, 0, 8 * MI355PPO_AMAX_WORDS * 4, st));      // the records the producers fill (what the learner's one fill per pass does)
            TIMED(0, mi355ppo_cnn_conv1q_fwd_amax(obs, inds, bt1q, bias, a1, mb1, M, R(0), st));
            TIMED(1, mi355ppo_cnn_conv_fwd_packed_f16x2_f32(a1, hz2, bias, a2, bits ? mb2 : nullptr, M, 2, R(0), R(1), st));      // (CONV_TRAFFIC_NOBITS: the forwards without their mask words)
            TIMED(2, mi355ppo_cnn_conv_fwd_packed_f16x2_f32(a2, hz3, bias, a3, bits ? mb3 : nullptr, M, 3, R(1), R(2), st));
            TIMED(8, mi355ppo_fc_fwd_relu_packed_f16x2_f32(a3, 3136, hk_fwd, bias, hfc, (int)M, 512, 3136, wsfwd, wsfwdb, R(2), nullptr, st));
            TIMED(9, mi355ppo_fc_dgrad_packed_f16x2_f32(dzfc, 516, hk_dg, nullptr, mb3, dz3, (int)M, 3136, 512, R(8), R(3), st));
            TIMED(10, mi355ppo_fc_wgrad_f16x2_f32(dzfc, 516, a3, dWfc, (int)M, 512, 3136, 64, wsfc, wsfcb, R(8), R(2), st));
            TIMED(3, mi355ppo_cnn_conv_wgrad_f16x2_f32(a2, dz3, dW3, db3, M, 3, ws, wsb, R(1), R(3), st));
            TIMED(4, mi355ppo_cnn_conv_dgrad_packed_f16x2_f32(dz3, hzd3, nullptr, mb2, dz2, M, 3, R(3), R(4), st));
            TIMED(5, mi355ppo_cnn_conv_wgrad_f16x2_f32(a1, dz2, dW2, db2, M, 2, ws, wsb, R(0), R(4), st));
            TIMED(6, mi355ppo_cnn_conv_dgrad_packed_f16x2_f32(dz2, hzd2, nullptr, mb1, dz1, M, 2, R(4), R(5), st));
            TIMED(7, mi355ppo_cnn_conv1_wgrad_f16x2(obs, inds, dz1, dW1, db1, M, ws, wsb, R(5), st));
        } else {
        if (conv_z && bits) {
            TIMED(0, mi355ppo_cnn_conv1q_fwd_bits(obs, inds, bt1q, bias, a1, mb1, M, st));                          // kernel Q (+ a1's mask bits)
            TIMED(1, mi355ppo_cnn_conv_fwd_packed_bits_f32(a1, pz2, bias, a2, mb2, M, 2, st));                     // kernel Z (+ mask bits)
            TIMED(2, mi355ppo_cnn_conv_fwd_packed_bits_f32(a2, pz3, bias, a3, mb3, M, 3, st));
        } else
        TIMED(0, mi355ppo_cnn_conv_fwd_f32_variant(obs, inds, bt1q, bias, a1, M, 1, 6, st));        // kernel Q
        if (conv_z && bits) {
        } else if (conv_z) {
            TIMED(1, mi355ppo_cnn_conv_fwd_packed_f32(a1, pz2, bias, a2, M, 2, st));                  // kernel Z
            TIMED(2, mi355ppo_cnn_conv_fwd_packed_f32(a2, pz3, bias, a3, M, 3, st));
        } else {
            TIMED(1, mi355ppo_cnn_conv_fwd_f32(a1, nullptr, bt2, bias, a2, M, 2, st));                // kernel F
            TIMED(2, mi355ppo_cnn_conv_fwd_f32(a2, nullptr, bt3, bias, a3, M, 3, st));
        }
        TIMED(8, mi355ppo_fc_fwd_relu_packed_ws_f32(a3, 3136, pk_fwd, bias, hfc, (int)M, 512, 3136, wsfwd, wsfwdb, st));   // kernel Z forward (K split below 4,096 rows)
        if (conv_z && bits) TIMED(9, mi355ppo_fc_dgrad_maskbits_packed_f32(dzfc, 516, pk_dg, mb3, dz3, (int)M, 3136, 512, st));
        else TIMED(9, mi355ppo_fc_dgrad_mask_packed_f32(dzfc, 516, pk_dg, a3, dz3, (int)M, 3136, 512, st));        // kernel Z data gradient + (a3 > 0)
        TIMED(10, mi355ppo_fc_wgrad_f32(dzfc, 516, a3, dWfc, (int)M, 512, 3136, 64, wsfc, wsfcb, st));  // kernel W + its slab reduction
        TIMED(3, mi355ppo_cnn_conv_wgrad_f32(a2, nullptr, dz3, dW3, db3, M, 3, ws, wsb, st));
        if (conv_z && bits) TIMED(4, mi355ppo_cnn_conv_dgrad_packed_bits_f32(dz3, pzd3, mb2, dz2, M, 3, st));
        else if (conv_z) TIMED(4, mi355ppo_cnn_conv_dgrad_packed_f32(dz3, pzd3, a2, dz2, M, 3, st));
        else TIMED(4, mi355ppo_cnn_conv_dgrad_f32_variant(dz3, bt3c, a2, dz2, M, 3, 5, st));
        TIMED(5, mi355ppo_cnn_conv_wgrad_f32(a1, nullptr, dz2, dW2, db2, M, 2, ws, wsb, st));
        if (conv_z && bits) TIMED(6, mi355ppo_cnn_conv_dgrad_packed_bits_f32(dz2, pzd2, mb1, dz1, M, 2, st));
        else if (conv_z) TIMED(6, mi355ppo_cnn_conv_dgrad_packed_f32(dz2, pzd2, a1, dz1, M, 2, st));
        else TIMED(6, mi355ppo_cnn_conv_dgrad_f32_variant(dz2, bt2c, a1, dz1, M, 2, 6, st));           // border classes
        TIMED(7, mi355ppo_cnn_conv_wgrad_f32(obs, inds, dz1, dW1, db1, M, 1, ws, wsb, st));
        }
        CHECK(hipStreamSynchronize(st));
        if (r > 0 || reps == 1)
            for (int i = 0; i < 11; i++) { float ms; CHECK(hipEventElapsedTime(&ms, ev[i][0], ev[i][1])); tot[i] += ms; }
        if (r + 1 < reps) {   // the gradients the next repetition starts from stay a bounded signal
            fill_f32<<<4096, 256, 0, st>>>(dz1, a1n, 2u); fill_f32<<<4096, 256, 0, st>>>(dz2, a2n, 3u);
        }
    }
    CHECK(hipStreamSynchronize(st));
    if (getenv("CONV_TRAFFIC_PAIRS")) {
        // two-stream experiment: kernels with complementary bottlenecks side by side.  Pairs that are independent in the
        // backward pass: (layer-1 wgrad = kernel P: VALU-bound) with (layer-2 wgrad: matrix-pipe-bound), both need only dz1 / dz2.
        hipStream_t st2; CHECK(hipStreamCreate(&st2));
        void* ws2; CHECK(hipMalloc(&ws2, wsb));
        hipEvent_t e0, e1, e2, f0;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2)); CHECK(hipEventCreate(&f0));
        float serial = 0, both = 0;
        for (int r = 0; r < 4; r++) {
            CHECK(hipEventRecord(e0, st));
            ABI(mi355ppo_cnn_conv_wgrad_f32(a1, nullptr, dz2, dW2, db2, M, 2, ws, wsb, st));
            ABI(mi355ppo_cnn_conv_wgrad_f32(obs, inds, dz1, dW1, db1, M, 1, ws, wsb, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (r) serial += ms;
            CHECK(hipEventRecord(e0, st));
            CHECK(hipStreamWaitEvent(st2, e0, 0));
            ABI(mi355ppo_cnn_conv_wgrad_f32(a1, nullptr, dz2, dW2, db2, M, 2, ws2, wsb, st2));
            ABI(mi355ppo_cnn_conv_wgrad_f32(obs, inds, dz1, dW1, db1, M, 1, ws, wsb, st));
            CHECK(hipEventRecord(f0, st2));
            CHECK(hipStreamWaitEvent(st, f0, 0));
            CHECK(hipEventRecord(e2, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventElapsedTime(&ms, e0, e2)); if (r) both += ms;
        }
        std::printf("{\"pair\": \"wgrad2 || wgrad1\", \"serial_us\": %.1f, \"two_streams_us\": %.1f}\n", serial / 3 * 1e3f, both / 3 * 1e3f);
    }
    {
        const char* nm[11] = {"fwd1", "fwd2", "fwd3", "wgrad3", "dgrad3", "wgrad2", "dgrad2", "wgrad1", "fc_fwd", "fc_dgrad", "fc_wgrad"};
        const int nt = reps > 1 ? reps - 1 : 1;
        float sum = 0;
        std::printf("{\"images\": %lld, \"timed_reps\": %d", (long long)M, nt);
        for (int i = 0; i < 11; i++) { std::printf(", \"%s_us\": %.1f", nm[i], tot[i] / nt * 1e3f); sum += tot[i] / nt; }
        std::printf(", \"sum_ms\": %.3f}\n", sum);
    }
    if (argc > 3) {
        FILE* f = std::fopen(argv[3], "wb");
        if (!f) { std::perror(argv[3]); return 4; }
        auto dump = [&](const float* d, size_t n, size_t stride) {
            const size_t cnt = (n + stride - 1) / stride;
            float* hbuf = (float*)std::malloc(cnt * 4);
            CHECK(hipMemcpy2D(hbuf, 4, d, stride * 4, 4, cnt, hipMemcpyDeviceToHost));
            std::fwrite(hbuf, 4, cnt, f); std::free(hbuf);
        };
        dump(dW1, 8192, 1); dump(db1, 32, 1); dump(dW2, 32768, 1); dump(db2, 64, 1); dump(dW3, 36864, 1); dump(db3, 64, 1);
        dump(a1, a1n, 4099); dump(a2, a2n, 1031); dump(a3, a3n, 1031); dump(dz2, a2n, 1031); dump(dz1, a1n, 4099);
        dump(hfc, (size_t)M * 512, 1021); dump(dz3, a3n, 1031); dump(dWfc, 512 * 3136, 211);
        std::fclose(f);
    }
    if (std::getenv("CONV_TRAFFIC_HASH")) {
        unsigned long long* dh = dalloc<unsigned long long>(1);
        auto hash = [&](const char* name, const float* d, size_t n) {
            CHECK(hipMemsetAsync(dh, 0, 8, st));
            hipLaunchKernelGGL(hash_u32, dim3(2048), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(d), n, dh);
            unsigned long long hv = 0;
            CHECK(hipMemcpyAsync(&hv, dh, 8, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
            std::printf("hash %s %016llx\n", name, hv);
        };
        hash("a1", a1, a1n); hash("a2", a2, a2n); hash("a3", a3, a3n); hash("hfc", hfc, (size_t)M * 512);
        hash("dz3", dz3, a3n); hash("dz2", dz2, a2n); hash("dz1", dz1, a1n);
        hash("dW1", dW1, 8192); hash("db1", db1, 32); hash("dW2", dW2, 32768); hash("db2", db2, 64); hash("dW3", dW3, 36864); hash("db3", db3, 64);
        hash("dWfc", dWfc, (size_t)512 * 3136);
    }
    dW = dW1;
    float h[4]; CHECK(hipMemcpy(h, dW, sizeof h, hipMemcpyDeviceToHost));
    std::printf("conv_traffic: images=%lld reps=%d dW1[0..3]=%g %g %g %g\n", (long long)M, reps, h[0], h[1], h[2], h[3]);
    return 0;
}
