// Does K1 lose time to read/write mixing on HBM when its inputs are not cache-resident?
// K1-shaped stream (4 uint8 planes in, 11 fp32 planes out, nontemporal stores, persistent 256 x NTH grid, B frames of
// 512x612) in two read schedules:
//   interleaved: the planes of iteration i+1 are requested while iteration i is stored (K1 today)
//   upfront:     ALL planes a thread will need are requested before its first store (a read burst, then pure writes)
// each measured back to back (inputs Infinity-Cache resident) and after a predecessor that replaces the cache contents
// (plain 1 GiB fill), which is the state the training step leaves behind.
//   hipcc --offload-arch=gfx950 -O3 tools/membench3.hip -o /tmp/membench3 && /tmp/membench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NP = 11;

__device__ __forceinline__ void st4nt(float* p, f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p)); }
template <bool NTL> __device__ __forceinline__ uint32_t ldw(const uint8_t* p) {
    if (NTL) return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
    return *reinterpret_cast<const uint32_t*>(p);
}
__device__ __forceinline__ f4 mix(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return f4{(float)((w0 + w1) & 255), (float)(((w0 >> 8) + (w2 >> 8)) & 255), (float)(((w1 >> 16) ^ (w3 >> 16)) & 255), (float)((w2 >> 24) + (w3 >> 24))};
}

template <bool NTL>
__global__ void interleaved(const uint8_t* in, float* out, long P, long nquads) {
    const long step = (long)gridDim.x * blockDim.x;
    long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= nquads) return;
    long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
    const uint8_t* pb = in + b * 4 * P + p4;
    uint32_t w0 = ldw<NTL>(pb), w1 = ldw<NTL>(pb + P), w2 = ldw<NTL>(pb + 2 * P), w3 = ldw<NTL>(pb + 3 * P);
    while (true) {
        const long qn = q + step;
        uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0; long bn = 0, pn = 0;
        if (qn < nquads) {
            bn = qn / (P / 4); pn = (qn - bn * (P / 4)) * 4;
            const uint8_t* pq = in + bn * 4 * P + pn;
            n0 = ldw<NTL>(pq); n1 = ldw<NTL>(pq + P); n2 = ldw<NTL>(pq + 2 * P); n3 = ldw<NTL>(pq + 3 * P);
        }
        f4 v = mix(w0, w1, w2, w3);
        float* o = out + b * NP * P + p4;
#pragma unroll
        for (int c = 0; c < NP; ++c) { st4nt(o + c * P, v); v.x += 1.f; }
        if (qn >= nquads) break;
        q = qn; b = bn; p4 = pn; w0 = n0; w1 = n1; w2 = n2; w3 = n3;
    }
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_void_t;

// prefetch: at kernel start every thread requests the planes of its first PF iterations through direct-to-LDS loads into a
// junk LDS slot (no registers, nothing waits for them): the lines land in the L2 on the way; then the interleaved loop
template <bool NTL>
__global__ void prefetch_l2(const uint8_t* in, float* out, long P, long nquads, int PF) {
    __shared__ uint32_t junk[1024 * 4];
    const long step = (long)gridDim.x * blockDim.x;
    long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (int i = 0; i < PF; ++i) {
        const long qq = q + i * step;
        if (qq < nquads) {
            const long b = qq / (P / 4), p4 = (qq - b * (P / 4)) * 4;
            const uint8_t* pb = in + b * 4 * P + p4;
            const int wv = threadIdx.x >> 6;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_amdgcn_global_load_lds((global_void_t*)(pb + c * P), (lds_void_t*)(junk + (wv * 4 + c) * 64), 4, 0, 0);
        }
    }
    if (q >= nquads) return;
    long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
    const uint8_t* pb = in + b * 4 * P + p4;
    uint32_t w0 = ldw<NTL>(pb), w1 = ldw<NTL>(pb + P), w2 = ldw<NTL>(pb + 2 * P), w3 = ldw<NTL>(pb + 3 * P);
    while (true) {
        const long qn = q + step;
        uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0; long bn = 0, pn = 0;
        if (qn < nquads) {
            bn = qn / (P / 4); pn = (qn - bn * (P / 4)) * 4;
            const uint8_t* pq = in + bn * 4 * P + pn;
            n0 = ldw<NTL>(pq); n1 = ldw<NTL>(pq + P); n2 = ldw<NTL>(pq + 2 * P); n3 = ldw<NTL>(pq + 3 * P);
        }
        f4 v = mix(w0, w1, w2, w3);
        float* o = out + b * NP * P + p4;
#pragma unroll
        for (int c = 0; c < NP; ++c) { st4nt(o + c * P, v); v.x += 1.f; }
        if (qn >= nquads) break;
        q = qn; b = bn; p4 = pn; w0 = n0; w1 = n1; w2 = n2; w3 = n3;
    }
}

// one quad per thread, hardware-scheduled grid (no persistent loop)
template <bool NTL>
__global__ void onequad(const uint8_t* in, float* out, long P, long nquads) {
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= nquads) return;
    const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
    const uint8_t* pb = in + b * 4 * P + p4;
    f4 v = mix(ldw<NTL>(pb), ldw<NTL>(pb + P), ldw<NTL>(pb + 2 * P), ldw<NTL>(pb + 3 * P));
    float* o = out + b * NP * P + p4;
#pragma unroll
    for (int c = 0; c < NP; ++c) { st4nt(o + c * P, v); v.x += 1.f; }
}

// persistent, but every workgroup owns one contiguous range of quads (256 sequential streams per plane instead of one
// window that all workgroups share)
template <bool NTL>
__global__ void wg_contiguous(const uint8_t* in, float* out, long P, long nquads) {
    const long per = (nquads + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = lo + per < nquads ? lo + per : nquads;
    for (long q = lo + threadIdx.x; q < hi; q += blockDim.x) {
        const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
        const uint8_t* pb = in + b * 4 * P + p4;
        f4 v = mix(ldw<NTL>(pb), ldw<NTL>(pb + P), ldw<NTL>(pb + 2 * P), ldw<NTL>(pb + 3 * P));
        float* o = out + b * NP * P + p4;
#pragma unroll
        for (int c = 0; c < NP; ++c) { st4nt(o + c * P, v); v.x += 1.f; }
    }
}

// IT = iterations per thread (compile time: the words live in registers)
template <int IT, bool NTL>
__global__ void upfront(const uint8_t* in, float* out, long P, long nquads) {
    const long step = (long)gridDim.x * blockDim.x;
    const long q0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    uint32_t w[IT][4];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const long q = q0 + i * step;
        w[i][0] = w[i][1] = w[i][2] = w[i][3] = 0;
        if (q < nquads) {
            const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
            const uint8_t* pb = in + b * 4 * P + p4;
            w[i][0] = ldw<NTL>(pb); w[i][1] = ldw<NTL>(pb + P); w[i][2] = ldw<NTL>(pb + 2 * P); w[i][3] = ldw<NTL>(pb + 3 * P);
        }
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const long q = q0 + i * step;
        if (q < nquads) {
            const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
            f4 v = mix(w[i][0], w[i][1], w[i][2], w[i][3]);
            float* o = out + b * NP * P + p4;
#pragma unroll
            for (int c = 0; c < NP; ++c) { st4nt(o + c * P, v); v.x += 1.f; }
        }
    }
}

__global__ void fill_plain(f4* b, long n) {
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = v;
}

int main() {
    const long P = 512L * 612;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f4* big; const long nbig = (1L << 30) / 16; CK(hipMalloc(&big, nbig * 16));
    for (int B : {16, 32}) {
        const long nquads = B * P / 4;
        const int SETS = 3;
        std::vector<uint8_t*> in(SETS); std::vector<float*> out(SETS);
        for (int s = 0; s < SETS; ++s) { CK(hipMalloc(&in[s], B * 4 * P)); CK(hipMemset(in[s], 37 + s, B * 4 * P)); CK(hipMalloc(&out[s], B * NP * P * 4)); }
        for (int nth : {1024, 512, 256}) {
            const int iters = (int)((nquads + 256L * nth - 1) / (256L * nth));
            auto bench = [&](const char* name, auto launch) {
                for (int cold = 0; cold < 2; ++cold) {
                    std::vector<float> ts;
                    for (int rep = 0; rep < 12; ++rep) {
                        if (cold) hipLaunchKernelGGL(fill_plain, dim3(2048), dim3(256), 0, 0, big, nbig);
                        CK(hipEventRecord(e0));
                        launch(in[rep % SETS], out[rep % SETS]);
                        CK(hipEventRecord(e1));
                        CK(hipDeviceSynchronize());
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep >= 2) ts.push_back(ms);
                    }
                    std::sort(ts.begin(), ts.end());
                    const float ms = ts[ts.size() / 2];
                    printf("B=%3d wg=%4d it=%d %-26s %-22s %6.1f us  %5.0f GB/s\n", B, nth, iters, name,
                           cold ? "after 1 GiB plain fill" : "back to back", ms * 1e3, B * P * 48 / ms / 1e6);
                }
            };
            bench("interleaved", [&](uint8_t* i, float* o) { hipLaunchKernelGGL((interleaved<false>), dim3(256), dim3(nth), 0, 0, i, o, P, nquads); });
            bench("interleaved nt-loads", [&](uint8_t* i, float* o) { hipLaunchKernelGGL((interleaved<true>), dim3(256), dim3(nth), 0, 0, i, o, P, nquads); });
            bench("onequad nt-loads", [&](uint8_t* i, float* o) { hipLaunchKernelGGL((onequad<true>), dim3((unsigned)((nquads + nth - 1) / nth)), dim3(nth), 0, 0, i, o, P, nquads); });
            bench("onequad", [&](uint8_t* i, float* o) { hipLaunchKernelGGL((onequad<false>), dim3((unsigned)((nquads + nth - 1) / nth)), dim3(nth), 0, 0, i, o, P, nquads); });
            bench("wg_contiguous nt-loads", [&](uint8_t* i, float* o) { hipLaunchKernelGGL((wg_contiguous<true>), dim3(256), dim3(nth), 0, 0, i, o, P, nquads); });
            for (int pf : {2})
                if (pf <= iters) {
                    char nm[64]; snprintf(nm, sizeof nm, "prefetch_l2 x%d", pf);
                    bench(nm, [&](uint8_t* i, float* o) { hipLaunchKernelGGL((prefetch_l2<false>), dim3(256), dim3(nth), 0, 0, i, o, P, nquads, pf); });
                }
#define UP(IT) if (iters == IT) { \
            bench("upfront", [&](uint8_t* i, float* o) { hipLaunchKernelGGL((upfront<IT, false>), dim3(256), dim3(nth), 0, 0, i, o, P, nquads); }); \
            bench("upfront nt-loads", [&](uint8_t* i, float* o) { hipLaunchKernelGGL((upfront<IT, true>), dim3(256), dim3(nth), 0, 0, i, o, P, nquads); }); }
            UP(5) UP(10) UP(20) UP(40)
        }
        for (int s = 0; s < SETS; ++s) { CK(hipFree(in[s])); CK(hipFree(out[s])); }
    }
    return 0;
}
