// What an fp32 product emulated on the bf16 matrix cores of gfx950 would deliver: every fp32 operand is split into
// three bf16 terms (x = hi + mid + lo, 8 significant bits each) and a 32x32x16 block is six v_mfma_f32_32x32x16_bf16
// (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the dropped terms are <= 2^-23 of the product).  The question this
// probe answers: does the split (vector instructions) overlap with the bf16 MFMAs of other waves, or do the two pipes
// run one after the other, as the fp32 MFMA and the VALU do (tools/mfma_peak.hip)?
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x3_peak.hip -o tools/bin/bf16x3_peak && tools/bin/bf16x3_peak
// The loop models the 64x32 wave tile of the conv kernel over one K-chunk of 32: fragments of two A blocks and one B
// block, 2 x 16 k values.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { S_NONE = 0, S_A = 1, S_AB = 2, S_A_RNE = 3, S_AB_RNE = 4, S_VALU_ONLY = 5 };

struct Split { u32x4 hi, mid, lo; };

// eight fp32 -> three packs of eight bf16 (truncation split: exact, hi + mid + lo == x up to 2^-24)
__device__ __forceinline__ Split split8(const float4 a, const float4 b) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    Split s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2 v = {x[2 * p], x[2 * p + 1]};
        const f32x2 h = {__uint_as_float(__float_as_uint(v.x) & 0xffff0000u), __uint_as_float(__float_as_uint(v.y) & 0xffff0000u)};
        const f32x2 r1 = v - h;
        const f32x2 m = {__uint_as_float(__float_as_uint(r1.x) & 0xffff0000u), __uint_as_float(__float_as_uint(r1.y) & 0xffff0000u)};
        const f32x2 r2 = r1 - m;
        s.hi[p] = __builtin_amdgcn_perm(__float_as_uint(v.y), __float_as_uint(v.x), 0x07060302u);
        s.mid[p] = __builtin_amdgcn_perm(__float_as_uint(r1.y), __float_as_uint(r1.x), 0x07060302u);
        s.lo[p] = __builtin_amdgcn_perm(__float_as_uint(r2.y), __float_as_uint(r2.x), 0x07060302u);
    }
    return s;
}

// the same with round-to-nearest conversions (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ Split split8_rne(const float4 a, const float4 b) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    Split s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2 v = {x[2 * p], x[2 * p + 1]};
        const unsigned hp = cvt_pk(v.x, v.y);
        const f32x2 h = {__uint_as_float(hp << 16), __uint_as_float(hp & 0xffff0000u)};
        const f32x2 r1 = v - h;
        const unsigned mp = cvt_pk(r1.x, r1.y);
        const f32x2 m = {__uint_as_float(mp << 16), __uint_as_float(mp & 0xffff0000u)};
        const f32x2 r2 = r1 - m;
        s.hi[p] = hp; s.mid[p] = mp; s.lo[p] = cvt_pk(r2.x, r2.y);
    }
    return s;
}

template <int MODE>
__global__ __launch_bounds__(256) void loop(float* out, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float lds[12288];   // 48 KB
    const int tid = threadIdx.x;
    for (int i = tid; i < 12288; i += 256) lds[i] = a0 * (float)((i * 7) & 63) + 0.37f;
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto bf = [](u32x4 v) { return __builtin_bit_cast(bf16x8, v); };
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {                 // two 16-wide k groups of the chunk
            const int o = ((it * 2 + g) & 3) * 1024 + tid * 4;
            float4 fa[2][2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i][0] = *reinterpret_cast<const float4*>(&lds[o + 2048 * i]);
                fa[i][1] = *reinterpret_cast<const float4*>(&lds[(o + 2048 * i) ^ 4]);
            }
            fb[0] = *reinterpret_cast<const float4*>(&lds[o + 8192]);
            fb[1] = *reinterpret_cast<const float4*>(&lds[(o + 8192) ^ 4]);
            Split sa[2], sb;
            if (MODE == S_NONE) {                      // operands used as they are (three "terms" = the same bits)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    sa[i].hi = __builtin_bit_cast(u32x4, fa[i][0]); sa[i].mid = __builtin_bit_cast(u32x4, fa[i][1]);
                    sa[i].lo = sa[i].hi;
                }
                sb.hi = __builtin_bit_cast(u32x4, fb[0]); sb.mid = __builtin_bit_cast(u32x4, fb[1]); sb.lo = sb.hi;
            } else {
                constexpr bool RNE = MODE == S_A_RNE || MODE == S_AB_RNE;
#pragma unroll
                for (int i = 0; i < 2; ++i) sa[i] = RNE ? split8_rne(fa[i][0], fa[i][1]) : split8(fa[i][0], fa[i][1]);
                if (MODE == S_AB || MODE == S_AB_RNE || MODE == S_VALU_ONLY) sb = RNE ? split8_rne(fb[0], fb[1]) : split8(fb[0], fb[1]);
                else { sb.hi = __builtin_bit_cast(u32x4, fb[0]); sb.mid = __builtin_bit_cast(u32x4, fb[1]); sb.lo = sb.hi; }
            }
            if (MODE == S_VALU_ONLY) {
                sink += __uint_as_float(sa[0].hi[0] ^ sa[0].mid[1] ^ sa[0].lo[2] ^ sa[1].hi[3] ^ sa[1].mid[0] ^ sa[1].lo[1] ^ sb.hi[2] ^ sb.mid[3] ^ sb.lo[0]);
                sink += __uint_as_float(sa[0].hi[1] ^ sa[0].mid[2] ^ sa[0].lo[3] ^ sa[1].hi[0] ^ sa[1].mid[1] ^ sa[1].lo[2] ^ sb.hi[3] ^ sb.mid[0] ^ sb.lo[1]);
                sink += __uint_as_float(sa[0].hi[2] ^ sa[0].mid[3] ^ sa[0].lo[0] ^ sa[1].hi[1] ^ sa[1].mid[2] ^ sa[1].lo[3] ^ sb.hi[0] ^ sb.mid[1] ^ sb.lo[2]);
                sink += __uint_as_float(sa[0].hi[3] ^ sa[0].mid[0] ^ sa[0].lo[1] ^ sa[1].hi[2] ^ sa[1].mid[3] ^ sa[1].lo[0] ^ sb.hi[1] ^ sb.mid[2] ^ sb.lo[3]);
                continue;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa[i].lo), bf(sb.hi), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa[i].hi), bf(sb.lo), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa[i].mid), bf(sb.mid), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa[i].mid), bf(sb.hi), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa[i].hi), bf(sb.mid), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa[i].hi), bf(sb.hi), acc[i], 0, 0, 0);
            }
        }
    }
    float s = sink;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[tid] = s;
}

template <int MODE>
void run(const char* name, int wg_per_cu, float* out) {
    const int iters = 4000;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((loop<MODE>), dim3(grid), dim3(256), 0, 0, out, 10, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop<MODE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // fp32-equivalent work: per wave and iteration a 64x32 tile over k = 32
    const double flops = (double)grid * 4 * iters * 2.0 * 64 * 32 * 32;
    printf("%-44s waves/SIMD=%d  %.3f ms  %.1f fp32-equivalent TF (%.2f x 157.3); bf16 MFMA rate %.0f TF\n", name, wg_per_cu, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3, MODE == S_VALU_ONLY ? 0.0 : 6 * flops / ms / 1e9);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4096);
    for (int w = 1; w <= 3; ++w) {
        run<S_NONE>("6 mfma / block, no split", w, out);
        run<S_A>("split A (trunc), B pre-split", w, out);
        run<S_AB>("split A and B (trunc)", w, out);
        run<S_A_RNE>("split A (rne), B pre-split", w, out);
        run<S_AB_RNE>("split A and B (rne)", w, out);
        run<S_VALU_ONLY>("split A and B (trunc), no mfma", w, out);
    }
    return 0;
}
