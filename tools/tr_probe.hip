// ds_read_b64_tr_b16 lane map on this chip (cdna_hip_programming.md T10): per group of 16 lanes, lane 4q+p supplies the address of
// row q, columns 4p..4p+3 of a 4 x 16 block of 16-bit elements; lane i receives column i, row q in element q.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short sm[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) sm[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    const short* addr = sm + (4 * g + q) * 64 + 4 * p + 16 * (g & 1);     // block g: rows 4g..4g+3, columns 16(g&1)..+15
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)addr);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, want = (4 * g + j) * 64 + 16 * (g & 1) + (l & 15);
        if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, h[l * 4 + j], want); ++bad; }
    }
    printf("ds_read_b64_tr_b16 lane map: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as documented", bad);
    return bad != 0;
}
