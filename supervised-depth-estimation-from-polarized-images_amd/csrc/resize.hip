// Pillow-compatible 8-bit LANCZOS resampling of the polarizer planes on the device
// (Image.resize((W,H), Image.ANTIALIAS) in indoor_dataset.py:335-349 = ImagingResample, 8bpc branch):
// a horizontal pass and a vertical pass with fixed-point (22-bit) coefficients,
//   out = clip8((2^21 + sum_x in[xmin + x] * k[x]) >> 22),
// the intermediate image rounded to uint8 like Pillow.  The coefficient / bound tables are built on the host
// (polardepth/resize.py, Pillow's precompute_coeffs + normalize_coeffs_8bpc in double precision) -- with them the
// result is bit-identical to PIL (tests/test_resize_gpu.py).  Used when the loader hands over the raw frames
// (decode on the host, resize + K1 on the device).
#include "pd_common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int acc) {
    const int v = acc >> kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src [P][H][Ws] -> dst [P][H][Wd]
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                       const int* __restrict__ kk, const int* __restrict__ bounds,
                                                       int ksize, long rows, int Ws, int Wd) {
    const long total = rows * Wd;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / Wd;
        const int xx = (int)(i - r * Wd);
        const int xmin = bounds[2 * xx], xn = bounds[2 * xx + 1];
        const uint8_t* s = src + r * Ws + xmin;
        const int* k = kk + (long)xx * ksize;
        int acc = 1 << (kPrecisionBits - 1);
        for (int x = 0; x < xn; ++x) acc += (int)s[x] * k[x];
        dst[i] = clip8(acc);
    }
}

// src [P][Hs][W] -> dst [P][Hd][W]
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                       const int* __restrict__ kk, const int* __restrict__ bounds,
                                                       int ksize, int P, int Hs, int Hd, int W) {
    const long total = (long)P * Hd * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        const long t = i / W;
        const int yy = (int)(t % Hd);
        const long p = t / Hd;
        const int ymin = bounds[2 * yy], yn = bounds[2 * yy + 1];
        const uint8_t* s = src + (p * Hs + ymin) * W + x;
        const int* k = kk + (long)yy * ksize;
        int acc = 1 << (kPrecisionBits - 1);
        for (int y = 0; y < yn; ++y) acc += (int)s[(long)y * W] * k[y];
        dst[i] = clip8(acc);
    }
}

inline unsigned grid_for(long n) {
    const long b = (n + 255) / 256;
    return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int pd_resize_u8_pass(const void* src, void* dst, const void* coeffs, const void* bounds, int ksize,
                                 int P, int Hs, int Ws, int out_size, int vertical, void* stream) {
    PD_REQUIRE(P >= 0 && Hs > 0 && Ws > 0 && out_size > 0 && ksize > 0, "pd_resize_u8_pass: bad shape");
    if (P == 0) return PD_OK;
    PD_REQUIRE(src && dst && coeffs && bounds, "pd_resize_u8_pass: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (vertical)
        hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((long)P * out_size * Ws)), dim3(256), 0, st, (const uint8_t*)src,
                           (uint8_t*)dst, (const int*)coeffs, (const int*)bounds, ksize, P, Hs, out_size, Ws);
    else
        hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((long)P * Hs * out_size)), dim3(256), 0, st, (const uint8_t*)src,
                           (uint8_t*)dst, (const int*)coeffs, (const int*)bounds, ksize, (long)P * Hs, Ws, out_size);
    return pd::check_launch("pd_resize_u8_pass");
}
