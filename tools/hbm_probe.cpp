// What a stream reaches from HBM on THIS box (the HBM-bound launches' roof as measured, not as specified): grid-stride kernels over
// buffers past the 256 MiB Infinity Cache -- copy, read-only, write-only and a 1 : 2 read : write mix (kernel Q's ratio) -- with
// 16-byte and 4-byte accesses per lane, default cache policy against non-temporal (`nt`) loads / stores.  HIP events, alternating order.
//
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.cpp -o tools/hbm_probe && tools/hbm_probe > profiles/r06_hbm_probe.jsonl
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld16(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st16(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void st4(unsigned* p, unsigned v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// copy: n 16-byte units
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void copy16(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld16<NTL>(s + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) st16<NTS>(d + i + u * stride, v[u]);
    }
    for (; i < n; i += stride) st16<NTS>(d + i, ld16<NTL>(s + i));
}
template <bool NTL, int U>
__global__ __launch_bounds__(256) void read16(const u32x4* __restrict__ s, unsigned* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld16<NTL>(s + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool NTS>
__global__ __launch_bounds__(256) void write16(u32x4* __restrict__ d, size_t n, unsigned salt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) st16<NTS>(d + i, (u32x4){salt, (unsigned)i, salt, salt});
}
// 4 bytes per lane, a wave's store instruction = two whole 128-byte lines (the resident kernels' epilogue shape)
template <bool NTS>
__global__ __launch_bounds__(256) void write4(unsigned* __restrict__ d, size_t n, unsigned salt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) st4<NTS>(d + i, salt + (unsigned)i);
}
// kernel Q's ratio: one 16-byte read per two 16-byte-equivalents written as 4-byte stores (8 store instructions per load)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void mix_1r2w(const u32x4* __restrict__ s, unsigned* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const u32x4 v = ld16<NTL>(s + i);
        const size_t wave0 = (i & ~(size_t)63) * 8, l = i & 63;        // the wave's 8 x 256 bytes of output
#pragma unroll
        for (int k = 0; k < 8; ++k) st4<NTS>(d + wave0 + 64 * k + l, v[k & 3] + k);
    }
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && std::string(argv[1]) == "quick";   // bench.py's leg: one configuration per access pattern, one pass (~0.1 s of GPU time)
    const size_t GiB = 1ull << 30, B = 1 * GiB;                     // 1 GiB in, up to 2 GiB out
    void *s, *d;
    CHECK(hipMalloc(&s, B)); CHECK(hipMalloc(&d, 2 * B));
    CHECK(hipMemset(s, 1, B)); CHECK(hipMemset(d, 2, 2 * B));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](auto launch, double bytes, const char* name, int grid) {
        float best = 1e30f, sum = 0;
        const int reps = 6;
        launch(grid);
        for (int r = 0; r < reps; ++r) {
            CHECK(hipEventRecord(e0)); launch(grid); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; if (ms < best) best = ms;
        }
        CHECK(hipGetLastError());
        std::printf("{\"kernel\": \"%s\", \"grid\": %d, \"bytes\": %.0f, \"avg_us\": %.1f, \"min_us\": %.1f, \"GBps_avg\": %.0f, \"GBps_best\": %.0f}\n", name, grid, bytes,
                    sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
        std::fflush(stdout);
    };
    const size_t n16 = B / 16;
    if (quick) {
        const int grid = 8192;
        time([&](int g) { read16<true, 4><<<g, 256>>>((const u32x4*)s, (unsigned*)d, n16); }, 1.0 * B, "read16 nt U4", grid);
        time([&](int g) { write4<false><<<g, 256>>>((unsigned*)d, B / 4, 7u); }, 1.0 * B, "write4 plain", grid);
        time([&](int g) { copy16<true, true, 4><<<g, 256>>>((const u32x4*)s, (u32x4*)d, n16); }, 2.0 * B, "copy16 ntLS U4", grid);
        time([&](int g) { mix_1r2w<false, false><<<g, 256>>>((const u32x4*)s, (unsigned*)d, n16); }, 3.0 * B, "mix 1r:2w(4B) plain", grid);
        return 0;
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (int grid : {2048, 8192, 32768}) {
            time([&](int g) { copy16<false, false, 1><<<g, 256>>>((const u32x4*)s, (u32x4*)d, n16); }, 2.0 * B, "copy16 plain U1", grid);
            time([&](int g) { copy16<false, false, 4><<<g, 256>>>((const u32x4*)s, (u32x4*)d, n16); }, 2.0 * B, "copy16 plain U4", grid);
            time([&](int g) { copy16<true, false, 4><<<g, 256>>>((const u32x4*)s, (u32x4*)d, n16); }, 2.0 * B, "copy16 ntL U4", grid);
            time([&](int g) { copy16<false, true, 4><<<g, 256>>>((const u32x4*)s, (u32x4*)d, n16); }, 2.0 * B, "copy16 ntS U4", grid);
            time([&](int g) { copy16<true, true, 4><<<g, 256>>>((const u32x4*)s, (u32x4*)d, n16); }, 2.0 * B, "copy16 ntLS U4", grid);
        }
        const int grid = 8192;
        time([&](int g) { read16<false, 4><<<g, 256>>>((const u32x4*)s, (unsigned*)d, n16); }, 1.0 * B, "read16 plain U4", grid);
        time([&](int g) { read16<true, 4><<<g, 256>>>((const u32x4*)s, (unsigned*)d, n16); }, 1.0 * B, "read16 nt U4", grid);
        time([&](int g) { write16<false><<<g, 256>>>((u32x4*)d, n16, 7u); }, 1.0 * B, "write16 plain", grid);
        time([&](int g) { write16<true><<<g, 256>>>((u32x4*)d, n16, 7u); }, 1.0 * B, "write16 nt", grid);
        time([&](int g) { write4<false><<<g, 256>>>((unsigned*)d, B / 4, 7u); }, 1.0 * B, "write4 plain", grid);
        time([&](int g) { write4<true><<<g, 256>>>((unsigned*)d, B / 4, 7u); }, 1.0 * B, "write4 nt", grid);
        time([&](int g) { mix_1r2w<false, false><<<g, 256>>>((const u32x4*)s, (unsigned*)d, n16); }, 3.0 * B, "mix 1r:2w(4B) plain", grid);
        time([&](int g) { mix_1r2w<false, true><<<g, 256>>>((const u32x4*)s, (unsigned*)d, n16); }, 3.0 * B, "mix 1r:2w(4B) ntS", grid);
        time([&](int g) { mix_1r2w<true, true><<<g, 256>>>((const u32x4*)s, (unsigned*)d, n16); }, 3.0 * B, "mix 1r:2w(4B) ntLS", grid);
    }
    return 0;
}
