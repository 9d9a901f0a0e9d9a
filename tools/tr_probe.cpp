// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): which 16-bit elements does lane l receive, given per-lane addresses?
// The LDS holds the element's own index (u16) at every position; every lane passes the address of "its" 4 contiguous elements of a
// [row][col] image with a row pitch of PITCH bytes: lane l of group g = l >> 4, i = l & 15: row = i >> 2, column chunk = i & 3 -> address
// of element (row, 4 * (i & 3)) of the group's block.  Prints, per lane, the four elements it got (as row * 256 + col of the block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned short* out, int pitch) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int e = threadIdx.x; e < 16384; e += 64) lds[e] = 0xffff;
    __syncthreads();
    // four blocks (one per 16-lane group) of 4 rows x 16 columns, block g at byte offset g * 4096, rows `pitch` bytes apart
    for (int g = 0; g < 4; ++g)
        for (int r = 0; r < 4; ++r)
            for (int c = threadIdx.x; c < 16; c += 64) lds[(g * 4096 + r * pitch) / 2 + c] = (unsigned short)(g * 4096 + r * 256 + c);
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    const unsigned addr = (unsigned)(g * 4096 + (i >> 2) * pitch + (i & 3) * 8);
    typedef short __attribute__((address_space(3))) lds_short;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)((lds_short*)lds + addr / 2));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof h);
    for (int pitch : {32, 272}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pitch);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("pitch %d\n", pitch);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" g%d r%d c%2d", h[l * 4 + j] >> 12, (h[l * 4 + j] >> 8) & 15, h[l * 4 + j] & 255);
            printf("\n");
        }
    }
    return 0;
}
