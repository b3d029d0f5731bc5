// Can K1 write the stems' space-to-depth NHWC layout directly?  A lane that owns one 2x2 pixel block owns one 36-float
// record of the normals tensor [B][H/2][W/2][36] (144 B) and one 8-float record of the standardised XOLP tensor (32 B):
// nine 16-byte stores at a lane stride of 144 B -- every wave instruction touches 64 different 64-byte sectors.
// Measures that store shape (nontemporal and plain) against the planar shape of the same byte count.
//   hipcc --offload-arch=gfx950 -O3 tools/membench4.hip -o /tmp/membench4 && /tmp/membench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st(float* p, f4 v) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p)); else *reinterpret_cast<f4*>(p) = v; }

// records: lane = one record of R floats (R % 4 == 0), persistent grid
template <int R, bool NT>
__global__ void records(const uint8_t* in, float* out, long nrec) {
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < nrec; r += (long)gridDim.x * blockDim.x) {
        const uint32_t w = reinterpret_cast<const uint32_t*>(in)[r];      // 4 input bytes per 2x2 block and plane (x4 planes folded)
        f4 v = {(float)(w & 255), (float)((w >> 8) & 255), (float)((w >> 16) & 255), (float)(w >> 24)};
        float* o = out + r * R;
#pragma unroll
        for (int k = 0; k < R / 4; ++k) { st<NT>(o + 4 * k, v); v.x += 1.f; }
    }
}
// planar: lane = 4 consecutive pixels of each of NP planes
template <int NP, bool NT>
__global__ void planar(const uint8_t* in, float* out, long P, long nquads) {
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nquads; q += (long)gridDim.x * blockDim.x) {
        const uint32_t w = reinterpret_cast<const uint32_t*>(in)[q];
        f4 v = {(float)(w & 255), (float)((w >> 8) & 255), (float)((w >> 16) & 255), (float)(w >> 24)};
        const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
        float* o = out + b * NP * P + p4;
#pragma unroll
        for (int c = 0; c < NP; ++c) { st<NT>(o + c * P, v); v.x += 1.f; }
    }
}
__global__ void fill_plain(f4* b, long n) {
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = v;
}
int main() {
    const int B = 16; const long P = 512L * 640;
    const long npix = B * P, nrec = npix / 4, nquads = npix / 4;
    uint8_t* in; CK(hipMalloc(&in, npix)); CK(hipMemset(in, 7, npix));
    std::vector<float*> outs(3);
    for (auto& o : outs) CK(hipMalloc(&o, npix * 9 * 4));
    f4* big; const long nbig = (1L << 30) / 16; CK(hipMalloc(&big, nbig * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto bench = [&](const char* name, auto launch) {
        for (int cold = 0; cold < 2; ++cold) {
            std::vector<float> ts;
            for (int rep = 0; rep < 12; ++rep) {
                if (cold) hipLaunchKernelGGL(fill_plain, dim3(2048), dim3(256), 0, 0, big, nbig);
                CK(hipEventRecord(e0)); launch(outs[rep % 3]); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf("%-44s %-22s %6.1f us  %5.0f GB/s (36 B/px written)\n", name, cold ? "after 1 GiB plain fill" : "back to back",
                   ts[ts.size() / 2] * 1e3, npix * 36.0 / ts[ts.size() / 2] / 1e6);
        }
    };
    for (int nth : {1024, 512, 256}) {
        char nm[96];
        snprintf(nm, sizeof nm, "planar 9 planes, nt, 256 x %d", nth);
        bench(nm, [&](float* o) { hipLaunchKernelGGL((planar<9, true>), dim3(256), dim3(nth), 0, 0, in, o, P, nquads); });
        snprintf(nm, sizeof nm, "36-float records (144 B / lane), nt, 256 x %d", nth);
        bench(nm, [&](float* o) { hipLaunchKernelGGL((records<36, true>), dim3(256), dim3(nth), 0, 0, in, o, nrec); });
        snprintf(nm, sizeof nm, "36-float records (144 B / lane), plain, 256 x %d", nth);
        bench(nm, [&](float* o) { hipLaunchKernelGGL((records<36, false>), dim3(256), dim3(nth), 0, 0, in, o, nrec); });
    }
    return 0;
}
