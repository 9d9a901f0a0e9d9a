// Matrix-pipe floor measurements for the exact-product bf16 scheme: how many cycles one
// `v_mfma_f32_32x32x16_bf16` costs on this chip for a stream shaped like kernels X / C -- one wave per SIMD, NACC independent
// accumulators round-robin -- on RANDOM operand bits (the chip clocks to its power budget: zero operands run faster), with
// FILL plain VALU instructions issued between consecutive MFMAs, and the same for `v_mfma_f32_32x32x2_f32`.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_floor.cpp -o tools/mfma_floor && tools/mfma_floor > profiles/r03_mfma_floor.jsonl
//
// One JSON line per configuration: nominal cycles per MFMA = wall time x 2.4 GHz / MFMAs per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t h) {
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}

// two bf16 values in [1, 2) with random mantissas and random signs, packed
__device__ __forceinline__ uint32_t rnd_bf16x2(uint32_t h, bool zero) {
    if (zero) return 0u;
    return (h & 0x807f807fu) | 0x3f803f80u;
}

template <int NACC, int FILL, bool F32, int WAVES, int CHAINS = 16>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void floor_kernel(
    float* __restrict__ out, int iters, uint32_t seed, int zero) {
    const uint32_t id = (blockIdx.x * blockDim.x + threadIdx.x) * 16u + seed;
    // two operand sets: the MFMAs of iteration i read set (i & 1) while the fillers rewrite the words of the other set (as the
    // split of k-step s + 1 does beside the MFMAs of k-step s in kernels X / C), so operand bits toggle like a real stream's
    u32x4 aw[2][2], bw[2][2];
    float fa[2], fb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t o = id + 1000u * q + 4 * i;
            aw[q][i] = (u32x4){rnd_bf16x2(hash32(o), zero), rnd_bf16x2(hash32(o + 1), zero), rnd_bf16x2(hash32(o + 2), zero), rnd_bf16x2(hash32(o + 3), zero)};
            bw[q][i] = (u32x4){rnd_bf16x2(hash32(~o), zero), rnd_bf16x2(hash32(~o + 1), zero), rnd_bf16x2(hash32(~o + 2), zero), rnd_bf16x2(hash32(~o + 3), zero)};
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = zero ? 0.0f : __uint_as_float((hash32(id + 77 + i) & 0x807fffffu) | 0x3f800000u);
        fb[i] = zero ? 0.0f : __uint_as_float((hash32(id + 99 + i) & 0x807fffffu) | 0x3f800000u);
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    // fillers: v_perm_b32 (a plain one-slot VALU instruction with no packed form -- hipcc SLP-packs adjacent f32 fma / mul into
    // v_pk_*, which MI355X_MICROARCH.md lists as an anti-lever beside MFMAs).  Selector 0x07040100 moves the two low bytes of
    // each bf16 pair around and keeps sign / exponent bytes in place: values stay finite, mantissa bits keep toggling.
    const uint32_t c1 = zero ? 0u : (hash32(id + 300) & 0x007f007fu);
    auto body = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if constexpr (F32)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 1], fb[(j >> 1) & 1], acc[j], 0, 0, 0);
            else
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[q][j & 1]), __builtin_bit_cast(bf16x8, bw[q][(j >> 1) & 1]),
                                                                 acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < FILL * NACC; ++t) {
            const int w = t % CHAINS;      // CHAINS independent dependency chains: a filler reads what the filler CHAINS slots earlier wrote
            u32x4& v = (w & 8) ? bw[q ^ 1][(w >> 2) & 1] : aw[q ^ 1][(w >> 2) & 1];
            // first visit of a word in this body: swap its two mantissa-carrying bytes (sign / exponent bytes stay: finite values,
            // toggling operand bits); later visits: the identity selection (same instruction, same issue cost)
            v[w & 3] = __builtin_amdgcn_perm(v[w & 3], c1, t < CHAINS ? 0x07040506u : 0x07060504u);
        }
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (FILL) __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int it = 0; it < iters; it += 2) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) s += (float)((aw[q][i].x ^ bw[q][i].y) & 0xffu);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int FILL, bool F32, int WAVES, int CHAINS = 16>
static void run(const char* what, int iters, int zero, float* out, hipStream_t st) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const int reps = 4;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((floor_kernel<NACC, FILL, F32, WAVES, CHAINS>), dim3(256), dim3(64 * WAVES), 0, st, out, iters, 1234u + r, zero);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r) { sum += ms; if (ms < best) best = ms; }
    }
    const double per_simd = (double)iters * NACC * (WAVES / 4);
    const double mean_ms = sum / (reps - 1);
    const double flop = F32 ? 4096.0 : 32768.0;
    std::printf("{\"what\": \"%s\", \"mfma\": \"%s\", \"nacc\": %d, \"valu_per_mfma\": %d, \"waves_per_simd\": %d, \"valu_chains\": %d, \"operands\": \"%s\", \"ms\": %.3f, "
                "\"nominal_cycles_per_mfma\": %.2f, \"TFLOPs\": %.1f}\n",
                what, F32 ? "f32_32x32x2" : "bf16_32x32x16", NACC, FILL, WAVES / 4, FILL ? CHAINS : 0, zero ? "zero" : "random", mean_ms,
                mean_ms * 1e-3 * 2.4e9 / per_simd, per_simd * 1024.0 * flop / (mean_ms * 1e-3) / 1e12);
    std::fflush(stdout);
}

int main() {
    hipStream_t st; CHECK(hipStreamCreate(&st));
    float* out; CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    const int it = 20000;
    run<8, 0, false, 4>("bare stream", it, 0, out, st);
    run<8, 0, false, 4>("bare stream, zero operands", it, 1, out, st);
    run<8, 1, false, 4>("fillers", it, 0, out, st);
    run<8, 2, false, 4>("fillers", it, 0, out, st);
    run<8, 3, false, 4>("fillers", it, 0, out, st);
    run<8, 4, false, 4>("fillers", it, 0, out, st);
    run<8, 5, false, 4>("fillers", it, 0, out, st);
    run<8, 6, false, 4>("fillers", it, 0, out, st);
    run<8, 8, false, 4>("fillers", it, 0, out, st);
    // dependent VALU chains: a filler reads the result of the filler CHAINS slots before it (kernels X / C as compiled: 1 - 2)
    run<8, 4, false, 4, 1>("fillers in ONE dependency chain", it, 0, out, st);
    run<8, 4, false, 4, 2>("fillers in two chains", it, 0, out, st);
    run<8, 4, false, 4, 3>("fillers in three chains", it, 0, out, st);
    run<8, 4, false, 4, 4>("fillers in four chains", it, 0, out, st);
    run<8, 4, false, 4, 8>("fillers in eight chains", it, 0, out, st);
    run<8, 3, false, 4, 2>("3 fillers in two chains", it, 0, out, st);
    run<8, 2, false, 4, 1>("2 fillers in one chain", it, 0, out, st);
    run<8, 4, false, 8, 1>("two waves per SIMD, fillers in ONE chain", it / 2, 0, out, st);
    run<8, 4, false, 8, 2>("two waves per SIMD, fillers in two chains", it / 2, 0, out, st);
    run<1, 0, false, 4>("one accumulator (dependent chain)", it * 4, 0, out, st);
    run<2, 0, false, 4>("two accumulators", it * 2, 0, out, st);
    run<4, 0, false, 4>("four accumulators", it * 2, 0, out, st);
    run<8, 0, false, 8>("two waves per SIMD", it / 2, 0, out, st);
    run<8, 4, false, 8>("two waves per SIMD + fillers", it / 2, 0, out, st);
    run<8, 0, true, 4>("f32 bare stream", it / 2, 0, out, st);
    run<8, 0, true, 4>("f32 bare stream, zero operands", it / 2, 1, out, st);
    run<8, 4, true, 4>("f32 + fillers", it / 2, 0, out, st);
    run<8, 8, true, 4>("f32 + fillers", it / 2, 0, out, st);
    run<8, 0, true, 8>("f32 two waves per SIMD", it / 4, 0, out, st);
    return 0;
}
