// Does v_mfma_f32_32x32x16_bf16 on gfx950 honour bf16 SUBNORMAL inputs?  (MI200's matrix instructions flush them; PyTorch's numerical-accuracy
// notes.)  The 16-bit pattern 0x00vv read as bf16 is v * 2^-133 for every byte v -- subnormal below 128, exponent field 1 from 128 on -- so a
// uint8 operand could enter the bf16 pipe by ZERO-EXTENSION (one v_perm per two values) instead of v_cvt_f32_ubyte + pack, if (and only if)
// the pipe multiplies subnormals exactly.  This program multiplies A[i][k] = a small integer times 2^96 (bf16-exact) with B[k][j] = pattern
// 0x00vv, and compares D with the exact integer result scaled by 2^-37.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_denorm.cpp -o tools/mfma_denorm && tools/mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ D) {
    // A operand: lane (i = lane & 31, h = lane >> 5) holds A[i][8h .. 8h+7]; B operand: lane (j, h) holds B[8h .. 8h+7][j]
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    u16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[i * 16 + 8 * h + e]; b[e] = B[(8 * h + e) * 32 + i]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + i] = acc[e];      // row, column i
}

// Round 5: the same question for v_mfma_f32_32x32x16_f16 (the two-term f16 split of kernels Z / V / W, csrc/f16split.h: the `lo` terms of
// small elements are f16 SUBNORMALS, pattern 0x0vvv = v * 2^-24).  A[i][k] = an 8-bit integer (f16-exact), B[k][j] = pattern 0x0vvv, v < 1024.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void kh(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ D) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    u16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[i * 16 + 8 * h + e]; b[e] = B[(8 * h + e) * 32 + i]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + i] = acc[e];
}

static unsigned short bf16_of(float x) { uint32_t u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }

int main() {
    unsigned short hA[32 * 16], hB[16 * 32];
    double ia[32 * 16];
    int vb[16 * 32];
    uint32_t s = 12345;
    for (int n = 0; n < 32 * 16; ++n) {
        s = s * 1664525u + 1013904223u;
        const int m = (int)((s >> 20) % 255) - 127;                 // 8-bit signed integer: exact in bf16
        ia[n] = m;
        hA[n] = bf16_of(ldexpf((float)m, 96));
    }
    for (int n = 0; n < 16 * 32; ++n) {
        s = s * 1664525u + 1013904223u;
        vb[n] = (int)((s >> 16) & 255);
        if (n < 8) vb[n] = n == 0 ? 0 : n == 1 ? 1 : n == 2 ? 127 : n == 3 ? 128 : n == 4 ? 255 : vb[n];
        hB[n] = (unsigned short)vb[n];                              // ZERO-EXTENDED byte = bf16 v * 2^-133
    }
    unsigned short *dA, *dB; float* dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, 32 * 32 * 4);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    float hD[32 * 32];
    if (hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost) != hipSuccess) { printf("{\"error\": \"hip\"}\n"); return 2; }
    int bad = 0; double worst = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int kk = 0; kk < 16; ++kk) ref += ia[i * 16 + kk] * vb[kk * 32 + j];
            const double got = ldexp((double)hD[i * 32 + j], 37);  // D = 2^96 * 2^-133 * sum
            if (got != ref) { ++bad; if (fabs(got - ref) > worst) worst = fabs(got - ref); }
        }
    printf("{\"test\": \"v_mfma_f32_32x32x16_bf16 with subnormal bf16 inputs (pattern 0x00vv = v * 2^-133)\", \"mismatches\": %d, \"of\": 1024, \"worst_abs_err\": %g, "
           "\"verdict\": \"%s\"}\n", bad, worst, bad == 0 ? "subnormal inputs are multiplied exactly" : "subnormal inputs are NOT exact (flushed?)");
    {
        // f16: A = integers m in [-127, 127] as f16 (exact), B = subnormal patterns 0x0vvv (v * 2^-24, v < 1024) and a few normals
        unsigned short gA[32 * 16], gB[16 * 32];
        double ja[32 * 16], jb[16 * 32];
        for (int n = 0; n < 32 * 16; ++n) {
            s = s * 1664525u + 1013904223u;
            const int m = (int)((s >> 20) % 255) - 127;
            ja[n] = m;
            const _Float16 hm = (_Float16)(float)m;
            memcpy(&gA[n], &hm, 2);
        }
        for (int n = 0; n < 16 * 32; ++n) {
            s = s * 1664525u + 1013904223u;
            int v = (int)((s >> 16) & 1023);
            if (n < 6) v = n == 0 ? 0 : n == 1 ? 1 : n == 2 ? 1023 : n == 3 ? 512 : n == 4 ? 3 : v;
            gB[n] = (unsigned short)v;                               // exponent field 0: subnormal f16, value v * 2^-24
            jb[n] = v;
        }
        (void)hipMemcpy(dA, gA, sizeof gA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, gB, sizeof gB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kh, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        if (hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost) != hipSuccess) { printf("{\"error\": \"hip\"}\n"); return 2; }
        int badh = 0; double worsth = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int kk = 0; kk < 16; ++kk) ref += ja[i * 16 + kk] * jb[kk * 32 + j];
                const double got = ldexp((double)hD[i * 32 + j], 24);
                if (got != ref) { ++badh; if (fabs(got - ref) > worsth) worsth = fabs(got - ref); }
            }
        printf("{\"test\": \"v_mfma_f32_32x32x16_f16 with subnormal f16 inputs (pattern 0x0vvv = v * 2^-24)\", \"mismatches\": %d, \"of\": 1024, \"worst_abs_err\": %g, "
               "\"verdict\": \"%s\"}\n", badh, worsth, badh == 0 ? "subnormal inputs are multiplied exactly" : "subnormal inputs are NOT exact (flushed?)");
    }
    return 0;
}
