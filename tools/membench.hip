// Streaming-pattern microbenchmark behind the K1 design: how fast can "4 uint8 planes in, P fp32 planes out"
// go on this GPU with trivial compute, as a function of the access shape?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1);} } while (0)

__global__ void copy4(const float4* a, float4* b, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
// 4 px per thread: four dword loads, NP float4 stores (K1's shape)
template <int NP>
__global__ void planes4(const uint8_t* in, float* out, long P, long nquads) {
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nquads; q += (long)gridDim.x * blockDim.x) {
        const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
        const uint8_t* pb = in + b * 4 * P + p4;
        uint32_t w0 = *(const uint32_t*)pb, w1 = *(const uint32_t*)(pb + P), w2 = *(const uint32_t*)(pb + 2 * P), w3 = *(const uint32_t*)(pb + 3 * P);
        float4 v = make_float4((float)((w0 + w1) & 255), (float)(((w0 >> 8) + (w2 >> 8)) & 255), (float)(((w1 >> 16) ^ (w3 >> 16)) & 255), (float)((w2 >> 24) + (w3 >> 24)));
        float* o = out + b * NP * P + p4;
#pragma unroll
        for (int c = 0; c < NP; ++c) { *(float4*)(o + c * P) = v; v.x += 1.f; }
    }
}
// 16 px per thread: four 16-byte loads, NP x 4 float4 stores (lane-strided 64 B)
template <int NP>
__global__ void planes16(const uint8_t* in, float* out, long P, long n16) {
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < n16; q += (long)gridDim.x * blockDim.x) {
        const long b = q / (P / 16), p16 = (q - b * (P / 16)) * 16;
        const uint8_t* pb = in + b * 4 * P + p16;
        uint4 a0 = *(const uint4*)pb, a1 = *(const uint4*)(pb + P), a2 = *(const uint4*)(pb + 2 * P), a3 = *(const uint4*)(pb + 3 * P);
        const uint32_t w[4] = {a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z, a0.w + a1.w + a2.w + a3.w};
        float* o = out + b * NP * P + p16;
#pragma unroll
        for (int c = 0; c < NP; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(float4*)(o + c * P + 4 * j) = make_float4((float)(w[j] & 255) + c, (float)((w[j] >> 8) & 255), (float)((w[j] >> 16) & 255), (float)(w[j] >> 24));
    }
}
template <typename F> float timeit(F f, int it = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / it;
}
int main() {
    const long B = 128, P = 512L * 612;
    uint8_t* in; float* out; float4 *ca, *cb;
    CK(hipMalloc(&in, B * 4 * P)); CK(hipMalloc(&out, B * 11 * P * 4)); CK(hipMemset(in, 7, B * 4 * P));
    const long ncopy = B * P * 11 / 4; CK(hipMalloc(&ca, ncopy * 16)); CK(hipMalloc(&cb, ncopy * 16));
    float ms = timeit([&] { hipLaunchKernelGGL(copy4, dim3(4096), dim3(256), 0, 0, ca, cb, ncopy); });
    printf("copy float4 %.1f MB: %.3f ms  %.0f GB/s\n", ncopy * 32 / 1e6, ms, ncopy * 32 / ms / 1e6);
    for (int grid : {1024, 4096, 0}) for (int bs : {256, 512, 1024}) {
        const long nq = B * P / 4; const unsigned g = grid ? grid : (unsigned)((nq + bs - 1) / bs);
        ms = timeit([&] { hipLaunchKernelGGL(planes4<2>, dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("planes4<2>  grid %6u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 12 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(planes4<11>, dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("planes4<11> grid %6u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        const long n16 = B * P / 16; const unsigned g16 = grid ? grid : (unsigned)((n16 + bs - 1) / bs);
        ms = timeit([&] { hipLaunchKernelGGL(planes16<2>, dim3(g16), dim3(bs), 0, 0, in, out, P, n16); });
        printf("planes16<2> grid %6u bs %4d: %.3f ms  %.0f GB/s\n", g16, bs, ms, B * P * 12 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(planes16<11>, dim3(g16), dim3(bs), 0, 0, in, out, P, n16); });
        printf("planes16<11> grid %6u bs %4d: %.3f ms  %.0f GB/s\n", g16, bs, ms, B * P * 48 / ms / 1e6);
    }
    return 0;
}
