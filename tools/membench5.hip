// K1's HBM ceiling on THIS box: the guide quotes 6.29 TB/s for a float4 copy (1:1 read:write); K1 moves 4 B in and
// 44 B out per pixel.  What do a pure read, a pure write and a copy reach here, and does any LAYOUT of K1's 48 B/px stream faster
// than the eleven planar fp32 planes it writes?  Trivial compute, rotating buffer sets beyond the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/membench5.hip -o tools/bin/membench5 && tools/bin/membench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st16(float* p, float4 v) {
    if (NT) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, (f32x4*)p); else *(float4*)p = v;
}
template <bool NT> __global__ __launch_bounds__(1024) void copy4(const float4* a, float4* b, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) st16<NT>((float*)(b + i), a[i]);
}
template <bool NT> __global__ __launch_bounds__(1024) void fill4(float4* b, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        st16<NT>((float*)(b + i), make_float4((float)i, 1.f, 2.f, 3.f));
}
__global__ __launch_bounds__(1024) void read4(const float4* a, float* sink, long n) {
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) *sink = s;
}
// K1's byte mix: a thread = 4 consecutive pixels: four dword loads (one per polarizer plane), 44 B/px of fp32 out.
//  LAYOUT 0: 11 planar planes, one float4 store each (K1's layout)
//  LAYOUT 1: xolp as 2 planes + normals as THREE planes of float3-interleaved pixels (12 B/px: a lane stores 48 contiguous bytes = 3 float4)
//  LAYOUT 2: xolp as one float2-interleaved plane (8 B/px: 2 float4 per lane) + normals as one 36 B/px record plane (9 float4 per lane)
//  LAYOUT 3: everything in one 44 B/px record (11 float4 per lane, lane stride 176 B)
template <int LAYOUT, bool NT>
__global__ __launch_bounds__(1024) void k1_shape(const uint8_t* __restrict__ in, float* __restrict__ out, long P, long nquads) {
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nquads; q += (long)gridDim.x * blockDim.x) {
        const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
        const uint8_t* pb = in + b * 4 * P + p4;
        const uint32_t w0 = *(const uint32_t*)pb, w1 = *(const uint32_t*)(pb + P), w2 = *(const uint32_t*)(pb + 2 * P), w3 = *(const uint32_t*)(pb + 3 * P);
        float4 v = make_float4((float)((w0 + w1) & 255), (float)(((w0 >> 8) + (w2 >> 8)) & 255), (float)(((w1 >> 16) ^ (w3 >> 16)) & 255), (float)((w2 >> 24) + (w3 >> 24)));
        float* ob = out + b * 11 * P;
        if (LAYOUT == 0) {
#pragma unroll
            for (int c = 0; c < 11; ++c) { st16<NT>(ob + c * P + p4, v); v.x += 1.f; }
        } else if (LAYOUT == 1) {
            st16<NT>(ob + p4, v); v.x += 1.f; st16<NT>(ob + P + p4, v);
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j) { v.x += 1.f; st16<NT>(ob + (2 + 3 * g) * P + 3 * p4 + 4 * j, v); }
        } else if (LAYOUT == 2) {
            st16<NT>(ob + 2 * p4, v); v.x += 1.f; st16<NT>(ob + 2 * p4 + 4, v);
#pragma unroll
            for (int j = 0; j < 9; ++j) { v.x += 1.f; st16<NT>(ob + 2 * P + 9 * p4 + 4 * j, v); }
        } else {
#pragma unroll
            for (int j = 0; j < 11; ++j) { st16<NT>(ob + 11 * p4 + 4 * j, v); v.x += 1.f; }
        }
    }
}

template <typename F> float timeit(F f, int it) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < it; ++i) f(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / it;
}

int main() {
    const long P = 512L * 612;
    for (long B : {16L, 128L}) {
        const long bytes = B * P * 48;                     // one K1 launch worth of traffic
        const int sets = B == 16 ? 4 : 2;                  // 0.96 GB / 3.8 GB of rotating buffers
        const int it = B == 16 ? 24 : 8;
        std::vector<uint8_t*> ins(sets); std::vector<float*> outs(sets); std::vector<float4*> ca(sets), cb(sets);
        for (int s = 0; s < sets; ++s) {
            CK(hipMalloc(&ins[s], B * 4 * P)); CK(hipMemset(ins[s], 7 + s, B * 4 * P));
            CK(hipMalloc(&outs[s], B * 11 * P * 4));
            CK(hipMalloc(&ca[s], bytes / 2)); CK(hipMalloc(&cb[s], bytes / 2)); CK(hipMemset(ca[s], 1, bytes / 2));
        }
        float* sink; CK(hipMalloc(&sink, 4));
        const long n16 = bytes / 2 / 16;                   // copy: bytes/2 read + bytes/2 written
        printf("---- B = %ld: %.1f MB per launch, %d rotating buffer sets\n", B, bytes / 1e6, sets);
        for (int bs : {256, 1024}) for (int grid : {1024, 4096}) {
            float ms = timeit([&](int i) { hipLaunchKernelGGL(copy4<false>, dim3(grid), dim3(bs), 0, 0, ca[i % sets], cb[i % sets], n16); }, it);
            printf("copy  1:1  plain  grid %5d bs %4d: %7.1f us %6.0f GB/s\n", grid, bs, ms * 1e3, bytes / ms / 1e6);
            ms = timeit([&](int i) { hipLaunchKernelGGL(copy4<true>, dim3(grid), dim3(bs), 0, 0, ca[i % sets], cb[i % sets], n16); }, it);
            printf("copy  1:1  nt     grid %5d bs %4d: %7.1f us %6.0f GB/s\n", grid, bs, ms * 1e3, bytes / ms / 1e6);
            ms = timeit([&](int i) { hipLaunchKernelGGL(read4, dim3(grid), dim3(bs), 0, 0, ca[i % sets], sink, n16); }, it);
            printf("read  only        grid %5d bs %4d: %7.1f us %6.0f GB/s\n", grid, bs, ms * 1e3, bytes / 2 / ms / 1e6);
            ms = timeit([&](int i) { hipLaunchKernelGGL(fill4<false>, dim3(grid), dim3(bs), 0, 0, cb[i % sets], n16); }, it);
            printf("write only plain  grid %5d bs %4d: %7.1f us %6.0f GB/s\n", grid, bs, ms * 1e3, bytes / 2 / ms / 1e6);
            ms = timeit([&](int i) { hipLaunchKernelGGL(fill4<true>, dim3(grid), dim3(bs), 0, 0, cb[i % sets], n16); }, it);
            printf("write only nt     grid %5d bs %4d: %7.1f us %6.0f GB/s\n", grid, bs, ms * 1e3, bytes / 2 / ms / 1e6);
        }
        const long nq = B * P / 4;
#define RUN(L, NT, name) for (int bs : {256, 1024}) for (int grid : {256, 1024, 4096}) { \
            float ms = timeit([&](int i) { hipLaunchKernelGGL((k1_shape<L, NT>), dim3(grid), dim3(bs), 0, 0, ins[i % sets], outs[i % sets], P, nq); }, it); \
            printf("K1 mix 4:44 %-34s grid %5d bs %4d: %7.1f us %6.0f GB/s\n", name, grid, bs, ms * 1e3, bytes / ms / 1e6); }
        RUN(0, true, "11 planar planes, nt")
        RUN(0, false, "11 planar planes, plain")
        RUN(1, true, "2 planes + 3 float3 planes, nt")
        RUN(1, false, "2 planes + 3 float3 planes, plain")
        RUN(2, true, "float2 plane + 36 B records, nt")
        RUN(2, false, "float2 plane + 36 B records, plain")
        RUN(3, true, "44 B records, nt")
        RUN(3, false, "44 B records, plain")
        for (int s = 0; s < sets; ++s) { CK(hipFree(ins[s])); CK(hipFree(outs[s])); CK(hipFree(ca[s])); CK(hipFree(cb[s])); }
        CK(hipFree(sink));
    }
    return 0;
}
