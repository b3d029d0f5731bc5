// Streaming ceiling of K1's access shape on MI355X, beyond the Infinity Cache (B = 128 frames of 512x612:
// 160 MB in, 1.76 GB out).  "4 uint8 planes in, NP fp32 planes out" with trivial compute, as a function of
// store policy (plain / nontemporal), grid shape (one quad per thread / persistent), pixels per thread.
//   hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o /tmp/membench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ void st4(float* p, f4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
    else *reinterpret_cast<f4*>(p) = v;
}
template <bool NT> __device__ __forceinline__ uint32_t ld1(const uint8_t* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
    return *reinterpret_cast<const uint32_t*>(p);
}

template <bool NT>
__global__ void copy4(const f4* a, f4* b, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
template <bool NT>
__global__ void fill4(f4* b, long n) {
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
// 4 px per thread: four dword loads, NP float4 stores (K1's shape)
template <int NP, bool NT, bool NTL>
__global__ void planes4(const uint8_t* in, float* out, long P, long nquads) {
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nquads; q += (long)gridDim.x * blockDim.x) {
        const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
        const uint8_t* pb = in + b * 4 * P + p4;
        uint32_t w0 = ld1<NTL>(pb), w1 = ld1<NTL>(pb + P), w2 = ld1<NTL>(pb + 2 * P), w3 = ld1<NTL>(pb + 3 * P);
        f4 v = {(float)((w0 + w1) & 255), (float)(((w0 >> 8) + (w2 >> 8)) & 255), (float)(((w1 >> 16) ^ (w3 >> 16)) & 255), (float)((w2 >> 24) + (w3 >> 24))};
        float* o = out + b * NP * P + p4;
#pragma unroll
        for (int c = 0; c < NP; ++c) { st4<NT>(o + c * P, v); v.x += 1.f; }
    }
}
// software-pipelined: the loads of the next quad are issued before the stores of the current one
template <int NP, bool NT>
__global__ void planes4_pf(const uint8_t* in, float* out, long P, long nquads) {
    const long step = (long)gridDim.x * blockDim.x;
    long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= nquads) return;
    long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
    const uint8_t* pb = in + b * 4 * P + p4;
    uint32_t w0 = ld1<false>(pb), w1 = ld1<false>(pb + P), w2 = ld1<false>(pb + 2 * P), w3 = ld1<false>(pb + 3 * P);
    while (true) {
        const long qn = q + step;
        uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0; long bn = 0, pn = 0;
        if (qn < nquads) {
            bn = qn / (P / 4); pn = (qn - bn * (P / 4)) * 4;
            const uint8_t* pq = in + bn * 4 * P + pn;
            n0 = ld1<false>(pq); n1 = ld1<false>(pq + P); n2 = ld1<false>(pq + 2 * P); n3 = ld1<false>(pq + 3 * P);
        }
        f4 v = {(float)((w0 + w1) & 255), (float)(((w0 >> 8) + (w2 >> 8)) & 255), (float)(((w1 >> 16) ^ (w3 >> 16)) & 255), (float)((w2 >> 24) + (w3 >> 24))};
        float* o = out + b * NP * P + p4;
#pragma unroll
        for (int c = 0; c < NP; ++c) { st4<NT>(o + c * P, v); v.x += 1.f; }
        if (qn >= nquads) break;
        q = qn; b = bn; p4 = pn; w0 = n0; w1 = n1; w2 = n2; w3 = n3;
    }
}
// 8 px per thread: four 8-byte loads, NP x 2 float4 stores (32 contiguous bytes per lane)
template <int NP, bool NT>
__global__ void planes8(const uint8_t* in, float* out, long P, long noct) {
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < noct; q += (long)gridDim.x * blockDim.x) {
        const long b = q / (P / 8), p8 = (q - b * (P / 8)) * 8;
        const uint8_t* pb = in + b * 4 * P + p8;
        uint2 a0 = *(const uint2*)pb, a1 = *(const uint2*)(pb + P), a2 = *(const uint2*)(pb + 2 * P), a3 = *(const uint2*)(pb + 3 * P);
        const uint32_t w[2] = {a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y};
        float* o = out + b * NP * P + p8;
#pragma unroll
        for (int c = 0; c < NP; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f4 v = {(float)(w[j] & 255) + c, (float)((w[j] >> 8) & 255), (float)((w[j] >> 16) & 255), (float)(w[j] >> 24)};
                st4<NT>(o + c * P + 4 * j, v);
            }
    }
}
// two quads per thread, 256 B apart... i.e. a wave handles two consecutive 1-KiB segments of every plane
template <int NP, bool NT>
__global__ void planes4x2(const uint8_t* in, float* out, long P, long nquads) {
    // thread t of wave w handles quads base + lane and base + 64 + lane, base = (global wave id) * 128
    const long gw = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long base = gw * 128; base < nquads; base += nw * 128) {
        uint32_t w[2][4]; long bb[2], pp[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long q = base + 64 * k + lane;
            bb[k] = q / (P / 4); pp[k] = (q - bb[k] * (P / 4)) * 4;
            const uint8_t* pb = in + bb[k] * 4 * P + pp[k];
            if (q < nquads) { w[k][0] = ld1<false>(pb); w[k][1] = ld1<false>(pb + P); w[k][2] = ld1<false>(pb + 2 * P); w[k][3] = ld1<false>(pb + 3 * P); }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long q = base + 64 * k + lane;
            if (q >= nquads) continue;
            f4 v = {(float)((w[k][0] + w[k][1]) & 255), (float)(((w[k][0] >> 8) + (w[k][2] >> 8)) & 255), (float)(((w[k][1] >> 16) ^ (w[k][3] >> 16)) & 255), (float)((w[k][2] >> 24) + (w[k][3] >> 24))};
            float* o = out + bb[k] * NP * P + pp[k];
#pragma unroll
            for (int c = 0; c < NP; ++c) { st4<NT>(o + c * P, v); v.x += 1.f; }
        }
    }
}


// persistent, each wave owns CH consecutive 1-KiB segments per plane per visit (chunked walk)
template <int NP, bool NT, int CH>
__global__ void planes4_chunk(const uint8_t* in, float* out, long P, long nquads) {
    const long gw = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long base = gw * (64 * CH); base < nquads; base += nw * (64 * CH)) {
#pragma unroll 1
        for (int k = 0; k < CH; ++k) {
            const long q = base + 64 * k + lane;
            if (q >= nquads) break;
            const long b = q / (P / 4), p4 = (q - b * (P / 4)) * 4;
            const uint8_t* pb = in + b * 4 * P + p4;
            uint32_t w0 = ld1<false>(pb), w1 = ld1<false>(pb + P), w2 = ld1<false>(pb + 2 * P), w3 = ld1<false>(pb + 3 * P);
            f4 v = {(float)((w0 + w1) & 255), (float)(((w0 >> 8) + (w2 >> 8)) & 255), (float)(((w1 >> 16) ^ (w3 >> 16)) & 255), (float)((w2 >> 24) + (w3 >> 24))};
            float* o = out + b * NP * P + p4;
#pragma unroll
            for (int c = 0; c < NP; ++c) { st4<NT>(o + c * P, v); v.x += 1.f; }
        }
    }
}
// pitched output (W -> Wout): INSPACE walks input quads (wave segments unaligned in the output),
// otherwise output quads (1-KiB aligned segments, padding lanes store zeros)
template <int NP, bool NT, bool INSPACE>
__global__ void planes4_pitch(const uint8_t* in, float* out, int B, int H, int W, int Wout) {
    const long wq_in = W / 4, wq_out = Wout / 4, wq = INSPACE ? wq_in : wq_out;
    const long P = (long)H * W, Pout = (long)H * Wout, total = (long)B * H * wq;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long r = q / wq, cq = q - r * wq, b = r / H, row = r - b * H;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (cq < wq_in) {
            const uint8_t* pb = in + b * 4 * P + row * W + 4 * cq;
            uint32_t w0 = ld1<false>(pb), w1 = ld1<false>(pb + P), w2 = ld1<false>(pb + 2 * P), w3 = ld1<false>(pb + 3 * P);
            v = (f4){(float)((w0 + w1) & 255), (float)(((w0 >> 8) + (w2 >> 8)) & 255), (float)(((w1 >> 16) ^ (w3 >> 16)) & 255), (float)((w2 >> 24) + (w3 >> 24))};
        }
        float* o = out + b * NP * Pout + row * Wout + 4 * cq;
#pragma unroll
        for (int c = 0; c < NP; ++c) st4<NT>(o + c * Pout, v);
    }
}


__global__ void sum4(const f4* a, float* out, long n) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
__global__ void axpy4(f4* a, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] = a[i] + 1.f;
}


// K1-shaped stream + one data-dependent 4-byte gather per pixel from a 1 MB table (the AoLP LUT's access shape)
template <int NP, bool NT>
__global__ void planes4_pitch_lut(const uint8_t* in, float* out, const float* lut, int B, int H, int W, int Wout) {
    const long wq_in = W / 4, wq = Wout / 4;
    const long P = (long)H * W, Pout = (long)H * Wout, total = (long)B * H * wq;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long r = q / wq, cq = q - r * wq, b = r / H, row = r - b * H;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (cq < wq_in) {
            const uint8_t* pb = in + b * 4 * P + row * W + 4 * cq;
            uint32_t w0 = ld1<false>(pb), w1 = ld1<false>(pb + P), w2 = ld1<false>(pb + 2 * P), w3 = ld1<false>(pb + 3 * P);
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d1 = (int)((w0 >> (8 * j)) & 255) - (int)((w2 >> (8 * j)) & 255), d2 = (int)((w1 >> (8 * j)) & 255) - (int)((w3 >> (8 * j)) & 255);
                g[j] = lut[(d2 + 255) * 511 + d1 + 255];
            }
            v = (f4){g[0], g[1], g[2], g[3]};
        }
        float* o = out + b * NP * Pout + row * Wout + 4 * cq;
#pragma unroll
        for (int c = 0; c < NP; ++c) st4<NT>(o + c * Pout, v);
    }
}
__global__ void fill_pattern(uint8_t* p, long n) {   // smooth planes + noise: |d1|, |d2| within ~ +-60 like real frames
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15;
        p[i] = (uint8_t)(100 + (int)(40.f * __sinf((float)(i % 612) * 0.02f + (float)(i / 313344) * 1.3f)) + (int)(h & 7));
    }
}

template <typename F> float timeit(F f, int it = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / it;
}
int main(int argc, char** argv) {
    const long B = 128, P = 512L * 612;
    uint8_t* in; float* out; f4 *ca, *cb;
    CK(hipMalloc(&in, B * 4 * P)); CK(hipMalloc(&out, B * 11 * 512L * 640 * 4)); CK(hipMemset(in, 7, B * 4 * P));
    const long ncopy = B * P * 11 / 4 / 2; CK(hipMalloc(&ca, ncopy * 16)); CK(hipMalloc(&cb, ncopy * 16));
    CK(hipMemset(ca, 1, ncopy * 16));
    float ms;
    const long nq = B * P / 4;

    if (argc > 1 && argv[1][0] == 'd') {
        // Who pays for a predecessor's dirty cache lines?  Time the K1-shaped stream (B = 16) right after a kernel
        // that wrote `mb` MB with plain or nontemporal stores.
        printf("---- K1-shaped stream (B=16, 612->640, nt stores, grid 256 x 1024) after a dirtying predecessor\n");
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mb : {0, 256, 800}) for (int kind = 0; kind < 4; ++kind) {
            if (mb == 0 && kind) continue;
            float tot = 0.f; const int it = 12;
            const char* names[4] = {"plain fill", "nt fill   ", "read only ", "read+write"};
            for (int i = 0; i < it + 2; ++i) {
                const long n16 = (long)mb * 1000000 / 16;
                if (mb) {
                    if (kind == 0) hipLaunchKernelGGL(fill4<false>, dim3(4096), dim3(256), 0, 0, ca, n16);
                    if (kind == 1) hipLaunchKernelGGL(fill4<true>, dim3(4096), dim3(256), 0, 0, ca, n16);
                    if (kind == 2) hipLaunchKernelGGL(sum4, dim3(4096), dim3(256), 0, 0, ca, (float*)cb, n16);
                    if (kind == 3) hipLaunchKernelGGL(axpy4, dim3(4096), dim3(256), 0, 0, ca, n16);
                }
                float* o = out + (long)(i % 3) * 16 * 11 * 512 * 640;          // rotate three output sets
                const uint8_t* ii = in + (long)(i % 3) * 16 * 4 * P;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL((planes4_pitch<11, true, false>), dim3(256), dim3(1024), 0, 0, ii, o, 16, 512, 612, 640);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 2) tot += ms;
            }
            printf("predecessor %s %3d MB: stream kernel %.1f us  (%.0f GB/s algorithmic)\n", names[kind], mb, tot / it * 1e3, 16 * P * 48 / (tot / it) / 1e6);
        }
        float* lut; CK(hipMalloc(&lut, 511 * 511 * 4)); CK(hipMemset(lut, 0, 511 * 511 * 4));
        hipLaunchKernelGGL(fill_pattern, dim3(4096), dim3(256), 0, 0, in, B * 4 * P); CK(hipDeviceSynchronize());
        for (int nb : {16}) for (int kind = 0; kind < 5; ++kind) for (int mb : {0, 300, 800}) {
            if ((mb == 0) != (kind == 0)) continue;
            float tot = 0.f; const int it = 12;
            const char* names[5] = {"nothing", "plain fill", "nt fill", "read only", "read + nt write"};
            for (int i = 0; i < it + 2; ++i) {
                const long n16 = (long)mb * 1000000 / 16;
                if (kind == 1) hipLaunchKernelGGL(fill4<false>, dim3(4096), dim3(256), 0, 0, ca, n16);
                if (kind == 2) hipLaunchKernelGGL(fill4<true>, dim3(4096), dim3(256), 0, 0, ca, n16);
                if (kind == 3) hipLaunchKernelGGL(sum4, dim3(4096), dim3(256), 0, 0, ca, (float*)cb, n16);
                if (kind == 4) hipLaunchKernelGGL(copy4<true>, dim3(4096), dim3(256), 0, 0, ca, cb, n16 / 2);
                float* o = out + (long)(i % 2) * 64 * 11 * 512 * 640;
                const uint8_t* ii = in + (long)(i % 2) * 64 * 4 * P;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL((planes4_pitch_lut<11, true>), dim3(256), dim3(1024), 0, 0, ii, o, lut, nb, 512, 612, 640);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 2) tot += ms;
            }
            printf("stream + LUT gather, B=%d, after %s of %3d MB: %.1f us  (%.0f GB/s algorithmic)\n", nb, names[kind], mb, tot / it * 1e3, nb * P * 48 / (tot / it) / 1e6);
        }
        return 0;
    }
    if (argc > 1) {
    printf("---- chunked / pitched variants\n");
    for (int bs : {256, 512, 1024}) {
        const unsigned g = 256;
        ms = timeit([&] { hipLaunchKernelGGL((planes4_chunk<11, true, 1>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("chunk1  nt grid %u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4_chunk<11, true, 4>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("chunk4  nt grid %u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4_chunk<11, true, 16>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("chunk16 nt grid %u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4_chunk<11, true, 64>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("chunk64 nt grid %u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4_pitch<11, true, true>), dim3(g), dim3(bs), 0, 0, in, out, (int)B, 512, 612, 640); });
        printf("pitch in-space  nt grid %u bs %4d: %.3f ms  %.0f GB/s (algorithmic 48 B/px)\n", g, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4_pitch<11, true, false>), dim3(g), dim3(bs), 0, 0, in, out, (int)B, 512, 612, 640); });
        printf("pitch out-space nt grid %u bs %4d: %.3f ms  %.0f GB/s (algorithmic 48 B/px)\n", g, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4_pitch<11, false, false>), dim3(g), dim3(bs), 0, 0, in, out, (int)B, 512, 612, 640); });
        printf("pitch out-space plain grid %u bs %4d: %.3f ms  %.0f GB/s (algorithmic 48 B/px)\n", g, bs, ms, B * P * 48 / ms / 1e6);
        fflush(stdout);
    }
    return 0;
    }
    for (int grid : {1024, 2048, 4096, 8192, 0}) for (int bs : {256, 512, 1024}) {
        const unsigned g = grid ? grid : (unsigned)((ncopy + bs - 1) / bs);
        ms = timeit([&] { hipLaunchKernelGGL(copy4<false>, dim3(g), dim3(bs), 0, 0, ca, cb, ncopy); });
        printf("copy4     grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, ncopy * 32 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(copy4<true>, dim3(g), dim3(bs), 0, 0, ca, cb, ncopy); });
        printf("copy4 nt  grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, ncopy * 32 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(fill4<false>, dim3(g), dim3(bs), 0, 0, cb, ncopy); });
        printf("fill4     grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, ncopy * 16 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(fill4<true>, dim3(g), dim3(bs), 0, 0, cb, ncopy); });
        printf("fill4 nt  grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, ncopy * 16 / ms / 1e6);
        fflush(stdout);
    }
    for (int grid : {256, 512, 1024, 2048, 4096, 0}) for (int bs : {256, 512, 1024}) {
        const unsigned g = grid ? grid : (unsigned)((nq + bs - 1) / bs);
        if ((long)g * bs > nq) continue;
        ms = timeit([&] { hipLaunchKernelGGL((planes4<11, false, false>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("planes4<11>        grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4<11, true, false>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("planes4<11> nt-st  grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4<11, true, true>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("planes4<11> nt-all grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        if (grid) {
            ms = timeit([&] { hipLaunchKernelGGL((planes4_pf<11, false>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
            printf("planes4_pf<11>     grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((planes4_pf<11, true>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
            printf("planes4_pf<11> nt  grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((planes4x2<11, false>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
            printf("planes4x2<11>      grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((planes4x2<11, true>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
            printf("planes4x2<11> nt   grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 48 / ms / 1e6);
        }
        const long n8 = B * P / 8; const unsigned g8 = grid ? grid : (unsigned)((n8 + bs - 1) / bs);
        ms = timeit([&] { hipLaunchKernelGGL((planes8<11, false>), dim3(g8), dim3(bs), 0, 0, in, out, P, n8); });
        printf("planes8<11>        grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g8, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes8<11, true>), dim3(g8), dim3(bs), 0, 0, in, out, P, n8); });
        printf("planes8<11> nt     grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g8, bs, ms, B * P * 48 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4<2, false, false>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("planes4<2>         grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 12 / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((planes4<2, true, false>), dim3(g), dim3(bs), 0, 0, in, out, P, nq); });
        printf("planes4<2> nt-st   grid %7u bs %4d: %.3f ms  %.0f GB/s\n", g, bs, ms, B * P * 12 / ms / 1e6);
        fflush(stdout);
    }
    return 0;
}
