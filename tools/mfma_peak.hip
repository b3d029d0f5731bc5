// What the fp32 matrix pipe of gfx950 delivers to a loop of v_mfma_f32_32x32x2_f32 alone, with LDS fragment reads, and
// with a workgroup barrier every 32 MFMAs -- the ceiling the implicit-GEMM conv kernel is measured against in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { M_ONLY = 0, M_READ = 1, M_BARRIER = 2, M_PREFETCH = 4, M_VALU = 8, M_B64 = 16, M_ONE = 32, M_TWO = 64, M_B32 = 128 };

// NACC independent accumulators; per group of 8 MFMAs (the conv kernel issues 3 ds_read_b128 per 8 MFMAs):
//   M_READ     3 x ds_read_b128 consumed by the group's own MFMAs
//   M_PREFETCH the reads of group g+1 are issued before the MFMAs of group g (register double buffer)
//   M_B64      the same bytes as 6 x ds_read_b64
//   M_ONE/TWO  1 or 2 ds_read_b128 per group instead of 3
//   M_VALU     no LDS reads, ~30 dependent VALU ops per group instead
//   M_B32      12 x ds_read_b32 per group (the weight-gradient kernel's fragment reads: 3 per 2 MFMAs)
//   M_BARRIER  s_barrier every 32 MFMAs
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[12288];   // 48 KB: three workgroups per CU like the conv kernel
    const int tid = threadIdx.x;
    for (int i = tid; i < 12288; i += 256) lds[i] = a0 * (float)(i & 15);
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 f[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 3; ++j) f[s][j] = make_float4(a0 + j, a0 + 1, b0 + 2, b0 + s);
    constexpr int NREAD = (MODE & M_ONE) ? 1 : (MODE & M_TWO) ? 2 : 3;
    auto read = [&](float4* dst, int grp) {
        const int o = ((grp & 3) * 1024 + tid * 4);           // 0 .. 4 K floats
        if (MODE & M_B32) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int w = (grp & 3) * 1024 + 4096 * j + (tid & 63) + 256 * (tid >> 6);
                dst[j].x = lds[w]; dst[j].y = lds[w + 64]; dst[j].z = lds[w + 128]; dst[j].w = lds[w + 192];
            }
        } else if (MODE & M_B64) {
#pragma unroll
            for (int j = 0; j < NREAD; ++j) {
                const float2 lo = *reinterpret_cast<const float2*>(&lds[o + 4096 * j]);
                const float2 hi = *reinterpret_cast<const float2*>(&lds[o + 4096 * j + 2]);
                asm volatile("" ::: "memory");
                dst[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NREAD; ++j) dst[j] = *reinterpret_cast<const float4*>(&lds[o + 4096 * j]);
        }
    };
    float vx = a0;
    if (MODE & M_PREFETCH) read(f[0], 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {        // 4 groups of 8 MFMAs = one K-chunk of the conv kernel
            const int cur = (MODE & M_PREFETCH) ? (g & 1) : 0;
            if (MODE & M_PREFETCH) read(f[cur ^ 1], it * 4 + g + 1);
            else if (MODE & M_READ) read(f[0], it * 4 + g);
            if (MODE & M_VALU) {
#pragma unroll
                for (int k = 0; k < 30; ++k) vx = __builtin_fmaf(vx, 1.0001f, 0.5f);
            }
            const float4 fa = f[cur][0], fb = f[cur][NREAD > 1 ? 1 : 0], fc = f[cur][NREAD > 2 ? 2 : 0];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float av = (m & 4) ? ((m & 1) ? fc.y : fc.x) : ((m & 1) ? fa.y : fa.x);
                const float bv = (m & 2) ? fb.z : fb.w;
                acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m % NACC], 0, 0, 0);
            }
        }
        if (MODE & M_BARRIER) __builtin_amdgcn_s_barrier();
    }
    float s = vx;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[tid] = s;
}

template <int NACC, int MODE>
void run(const char* name, int wg_per_cu, float* out) {
    const int iters = 2000;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 32 * 4096.0;
    printf("%-40s acc=%d waves/SIMD=%d  %.3f ms  %.1f TF (%.3f of 157.3)\n", name, NACC, wg_per_cu, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4096);
    for (int w = 1; w <= 3; ++w) {
        run<2, M_ONLY>("mfma only", w, out);
        run<2, M_BARRIER>("mfma + barrier/32", w, out);
        run<2, M_VALU>("mfma + 30 VALU / 8 mfma", w, out);
        run<2, M_READ>("mfma + 3 ds_read_b128 / 8", w, out);
        run<2, M_READ | M_TWO>("mfma + 2 ds_read_b128 / 8", w, out);
        run<2, M_READ | M_ONE>("mfma + 1 ds_read_b128 / 8", w, out);
        run<2, M_READ | M_B64>("mfma + 6 ds_read_b64 / 8", w, out);
        run<2, M_PREFETCH>("mfma + 3 ds_read_b128 / 8, prefetched", w, out);
        run<2, M_PREFETCH | M_B64>("mfma + 6 ds_read_b64 / 8, prefetched", w, out);
        run<2, M_PREFETCH | M_BARRIER>("prefetched + barrier/32", w, out);
        run<2, M_READ | M_B32>("mfma + 12 ds_read_b32 / 8", w, out);
        run<2, M_PREFETCH | M_B32>("mfma + 12 ds_read_b32 / 8, prefetched", w, out);
    }
    return 0;
}
