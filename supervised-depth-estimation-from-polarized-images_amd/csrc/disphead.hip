// Disparity heads of the depth decoder: sigmoid(Conv3x3(x)) with ONE output channel
// (depth_decoder.py:52-53,69-71: Conv3x3 = ReflectionPad2d(1) + Conv2d(C, 1, 3); layers.py:364-380).
//
// As an implicit GEMM these layers waste 31/32 of a 32-wide MFMA tile (2-3 TFLOP/s, 3.4 ms of the
// train step for 0.02 % of its flops).  They are memory-bound reductions over the channel vector, so they
// get direct kernels: C/4 lanes share a pixel (one float4 of channels each), a wave covers 64/(C/4)
// consecutive pixels, every tap is a fully coalesced read of the NHWC tensor and the eight re-reads of a
// pixel by its neighbours hit the cache.
//   forward          y  = sigmoid(b + sum_t x[refl(p + t - 1)] . w[t])
//   data gradient    dx[p] = sum_{q in Q(p)} sum_t dz[q - t + 1] w[t],  dz = dy * y * (1 - y),
//                    Q(p) = virtual (padded-grid) positions that reflect onto p: p itself, -1 if p == 1,
//                    H if p == H-2 (per axis) -- the fold of the reflection padding is built in
//   weight gradient  dw[t] = sum_p dz[p] x[refl(p + t - 1)],  db = sum_p dz[p]   (deterministic two-stage sum)
#include "pd_common.h"

namespace {

__device__ __forceinline__ int refl(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float dot4(float4 a, float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc);
    return fmaf(a.w, b.w, acc);
}

struct Pix {
    int n, h, w;
    bool valid;
};
__device__ __forceinline__ Pix decode(int p, int npix, int H, int W) {
    Pix r;
    r.valid = p < npix;
    const int pc = r.valid ? p : npix - 1;
    const int hw = H * W;
    r.n = pc / hw;
    const int rem = pc - r.n * hw;
    r.h = rem / W;
    r.w = rem - r.h * W;
    return r;
}

// Forward, direct form (64 and 128 channels: the lane groups are wide, the cross-lane sums of the tile form cost more
// than the cached re-reads).
template <int L>   // lanes per pixel, C = 4 L
__global__ __launch_bounds__(256) void disphead_fwd_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int npix, int H, int W) {
    constexpr int C = 4 * L, PPW = 64 / L;
    const int lane = threadIdx.x & 63, c4 = lane % L, pl = lane / L;
    float4 wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t] = ld4(w + t * C + 4 * c4);
    const float b = bias ? bias[0] : 0.f;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int p0 = wave * PPW; p0 < npix; p0 += nwaves * PPW) {
        const int p = p0 + pl;
        const Pix q = decode(p, npix, H, W);
        const float* xn = x + (long)q.n * H * W * C + 4 * c4;
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = refl(q.h + kh - 1, H);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = refl(q.w + kw - 1, W);
                acc = dot4(ld4(xn + ((long)hh * W + ww) * C), wr[kh * 3 + kw], acc);
            }
        }
#pragma unroll
        for (int m = L / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
        if (c4 == 0 && q.valid) y[p] = 1.f / (1.f + expf(-(acc + b)));
    }
}

constexpr int TR = 8, TW = 64, HR = TR + 2, HC = TW + 2, HP = HR * HC;   // output tile and its one-pixel halo

__device__ __forceinline__ int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

// Forward, tile form: every pixel of the (reflected) halo is read ONCE; its nine tap products x[p].w[t] go to LDS and
// an output pixel is the sum of nine LDS words.  (One pixel per lane group and nine global reads per pixel, the first
// version, ran at 1.6 TB/s: the texture path, not HBM, was the limit.)
template <int L>   // lanes per pixel, C = 4 L
__global__ __launch_bounds__(256) void disphead_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int H, int W, int tiles_h, int tiles_w, int ntiles) {
    constexpr int C = 4 * L, PPR = 256 / L;          // halo pixels per staging round
    __shared__ float D[9][HR][HC];
    const int tid = threadIdx.x, c4 = tid % L, ps = tid / L;
    float4 wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t] = ld4(w + t * C + 4 * c4);
    const float b = bias ? bias[0] : 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, n = tile / (tiles_w * tiles_h);
        const int h0 = th * TR, w0 = tw * TW;
        const float* xn = x + (long)n * H * W * C + 4 * c4;
        __syncthreads();                              // the previous tile's outputs have read D
        for (int base = 0; base < HP; base += PPR) {
            const int s = base + ps;
            const bool in = s < HP;
            const int sc = in ? s : HP - 1;
            const int r = sc / HC, c = sc - r * HC;
            // slot (r, c) stands for the virtual pixel (h0 - 1 + r, w0 - 1 + c) of the reflection-padded image
            const int hh = clampi(refl(h0 - 1 + r, H), 0, H - 1), ww = clampi(refl(w0 - 1 + c, W), 0, W - 1);
            const float4 v = ld4(xn + ((long)hh * W + ww) * C);
            float p[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) p[t] = dot4(v, wr[t], 0.f);
#pragma unroll
            for (int m = 1; m < L; m <<= 1) {
#pragma unroll
                for (int t = 0; t < 9; ++t) p[t] += __shfl_xor(p[t], m);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if (in && (t % L) == c4) D[t][r][c] = p[t];
        }
        __syncthreads();
#pragma unroll
        for (int o = tid; o < TR * TW; o += 256) {
            const int r = o / TW, c = o - r * TW;
            float acc = b;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc += D[kh * 3 + kw][r + kh][c + kw];
            const int h = h0 + r, wq = w0 + c;
            if (h < H && wq < W) y[((long)n * H + h) * W + wq] = 1.f / (1.f + expf(-acc));
        }
    }
}

// dz at (n, h, w) or 0 outside the image
__device__ __forceinline__ float dz_at(const float* __restrict__ dy, const float* __restrict__ y, int n, int h, int w,
                                       int H, int W) {
    if ((unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) return 0.f;
    const long i = ((long)n * H + h) * W + w;
    const float s = y[i];
    return dy[i] * s * (1.f - s);
}

template <int L>
__global__ __launch_bounds__(256) void disphead_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                const float* __restrict__ w, float* __restrict__ dx,
                                                                int npix, int H, int W) {
    constexpr int C = 4 * L, PPW = 64 / L;
    const int lane = threadIdx.x & 63, c4 = lane % L, pl = lane / L;
    float4 wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t] = ld4(w + t * C + 4 * c4);
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int p0 = wave * PPW; p0 < npix; p0 += nwaves * PPW) {
        const int p = p0 + pl;
        const Pix q = decode(p, npix, H, W);
        float g[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t] = 0.f;
        // virtual rows / columns whose reflection is (h, w): the pixel itself plus, next to a border, the padded line
        const int qh[3] = {q.h, -1, H}, qw[3] = {q.w, -1, W};
        const bool uh[3] = {true, q.h == 1, q.h == H - 2}, uw[3] = {true, q.w == 1, q.w == W - 2};
        for (int a = 0; a < 3; ++a) {
            if (!uh[a]) continue;
            for (int bq = 0; bq < 3; ++bq) {
                if (!uw[bq]) continue;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        g[kh * 3 + kw] += dz_at(dy, y, q.n, qh[a] - kh + 1, qw[bq] - kw + 1, H, W);
            }
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            o.x = fmaf(g[t], wr[t].x, o.x); o.y = fmaf(g[t], wr[t].y, o.y);
            o.z = fmaf(g[t], wr[t].z, o.z); o.w = fmaf(g[t], wr[t].w, o.w);
        }
        if (q.valid) *reinterpret_cast<float4*>(dx + (long)p * C + 4 * c4) = o;
    }
}

template <int L>
__global__ __launch_bounds__(256) void disphead_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  const float* __restrict__ x, float* __restrict__ part,
                                                                  float* __restrict__ bpart, int npix, int H, int W) {
    constexpr int C = 4 * L, PPW = 64 / L;
    __shared__ float4 red[4][9][L];
    __shared__ float bred[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c4 = lane % L, pl = lane / L;
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float bsum = 0.f;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    for (int p0 = wave * PPW; p0 < npix; p0 += nwaves * PPW) {
        const int p = p0 + pl;
        const Pix q = decode(p, npix, H, W);
        float g = 0.f;
        if (q.valid) { const float s = y[p]; g = dy[p] * s * (1.f - s); }
        bsum += g;
        const float* xn = x + (long)q.n * H * W * C + 4 * c4;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = refl(q.h + kh - 1, H);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = refl(q.w + kw - 1, W);
                const float4 v = ld4(xn + ((long)hh * W + ww) * C);
                float4& a = acc[kh * 3 + kw];
                a.x = fmaf(g, v.x, a.x); a.y = fmaf(g, v.y, a.y); a.z = fmaf(g, v.z, a.z); a.w = fmaf(g, v.w, a.w);
            }
        }
    }
    // pixels of the wave (lanes with equal c4), then the four waves, in a fixed order
#pragma unroll
    for (int m = L; m < 64; m <<= 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            acc[t].x += __shfl_xor(acc[t].x, m); acc[t].y += __shfl_xor(acc[t].y, m);
            acc[t].z += __shfl_xor(acc[t].z, m); acc[t].w += __shfl_xor(acc[t].w, m);
        }
        bsum += __shfl_xor(bsum, m);
    }
    if (pl == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) red[wv][t][c4] = acc[t];
        if (c4 == 0) bred[wv] = bsum;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * L; i += 256) {
        const int t = i / L, c = i - t * L;
        float4 s = red[0][t][c];
        for (int k = 1; k < 4; ++k) { s.x += red[k][t][c].x; s.y += red[k][t][c].y; s.z += red[k][t][c].z; s.w += red[k][t][c].w; }
        *reinterpret_cast<float4*>(part + (long)blockIdx.x * 9 * C + t * C + 4 * c) = s;
    }
    if (threadIdx.x == 0) bpart[blockIdx.x] = (bred[0] + bred[1]) + (bred[2] + bred[3]);
}

// Data and weight gradient in one pass: both are built from the same nine folded neighbour sums g_t[p]
// (dx[p] = sum_t g_t[p] w[t], dw[t] = sum_p g_t[p] x[p]), so x is read once, dx written once, and dz = dy y (1 - y)
// is evaluated once per pixel into an LDS halo tile instead of eighteen scalar loads per lane.
template <int L>
__global__ __launch_bounds__(256, 4) void disphead_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ add, int elu,
                                                           float* __restrict__ dx, float* __restrict__ part,
                                                           float* __restrict__ bpart, int H, int W, int tiles_h,
                                                           int tiles_w, int ntiles) {
    constexpr int C = 4 * L, PPW = 64 / L;
    __shared__ float dzs[HR][HC];
    __shared__ float4 red[4][9][L];
    __shared__ float bred[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c4 = lane % L, pl = lane / L;
    float4 wr[9], acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { wr[t] = ld4(w + t * C + 4 * c4); acc[t] = make_float4(0.f, 0.f, 0.f, 0.f); }
    float bsum = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, n = tile / (tiles_w * tiles_h);
        const int h0 = th * TR, w0 = tw * TW;
        __syncthreads();
        for (int i = tid; i < HP; i += 256) {
            const int r = i / HC, c = i - r * HC;
            dzs[r][c] = dz_at(dy, y, n, h0 - 1 + r, w0 - 1 + c, H, W);
        }
        __syncthreads();
        // a wave owns two rows of the tile
        for (int j = 0; j < 2 * TW / PPW; ++j) {
            const int pi = j * PPW + pl, r = 2 * wv + pi / TW, c = pi % TW;
            const int h = h0 + r, wq = w0 + c;
            const bool valid = h < H && wq < W;
            float nb[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bq = 0; bq < 3; ++bq) nb[a][bq] = dzs[r + a][c + bq];      // dz at (h - 1 + a, wq - 1 + bq)
            // rows: tap kh reads dz row h - kh + 1; next to a border the padded line folds onto the pixel too
            float rs[3][3];
            const bool top = h == 1, bot = h == H - 2, lef = wq == 1, rig = wq == W - 2;
#pragma unroll
            for (int bq = 0; bq < 3; ++bq) {
                rs[0][bq] = nb[2][bq] + (top ? nb[0][bq] : 0.f);
                rs[1][bq] = nb[1][bq];
                rs[2][bq] = nb[0][bq] + (bot ? nb[2][bq] : 0.f);
            }
            float g[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                g[kh * 3 + 0] = rs[kh][2] + (lef ? rs[kh][0] : 0.f);
                g[kh * 3 + 1] = rs[kh][1];
                g[kh * 3 + 2] = rs[kh][0] + (rig ? rs[kh][2] : 0.f);
            }
            if (valid) {
                const long off = (((long)n * H + h) * W + wq) * C + 4 * c4;
                const float4 v = ld4(x + off);
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    o.x = fmaf(g[t], wr[t].x, o.x); o.y = fmaf(g[t], wr[t].y, o.y);
                    o.z = fmaf(g[t], wr[t].z, o.z); o.w = fmaf(g[t], wr[t].w, o.w);
                    acc[t].x = fmaf(g[t], v.x, acc[t].x); acc[t].y = fmaf(g[t], v.y, acc[t].y);
                    acc[t].z = fmaf(g[t], v.z, acc[t].z); acc[t].w = fmaf(g[t], v.w, acc[t].w);
                }
                if (dx) {
                    if (add) {                       // the other consumer's gradient of x (autograd's sum, done here)
                        const float4 a = ld4(add + off);
                        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                    }
                    if (elu) {                       // x is an ELU output: hand back the gradient of its pre-activation
                        o.x = v.x > 0.f ? o.x : o.x * (v.x + 1.f); o.y = v.y > 0.f ? o.y : o.y * (v.y + 1.f);
                        o.z = v.z > 0.f ? o.z : o.z * (v.z + 1.f); o.w = v.w > 0.f ? o.w : o.w * (v.w + 1.f);
                    }
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    f4v ov = {o.x, o.y, o.z, o.w};
                    __builtin_nontemporal_store(ov, reinterpret_cast<f4v*>(dx + off));
                }
                bsum += nb[1][1];
            }
        }
    }
#pragma unroll
    for (int m = L; m < 64; m <<= 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            acc[t].x += __shfl_xor(acc[t].x, m); acc[t].y += __shfl_xor(acc[t].y, m);
            acc[t].z += __shfl_xor(acc[t].z, m); acc[t].w += __shfl_xor(acc[t].w, m);
        }
        bsum += __shfl_xor(bsum, m);
    }
    if (pl == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) red[wv][t][c4] = acc[t];
        if (c4 == 0) bred[wv] = bsum;
    }
    __syncthreads();
    for (int i = tid; i < 9 * L; i += 256) {
        const int t = i / L, c = i - t * L;
        float4 s = red[0][t][c];
        for (int k = 1; k < 4; ++k) { s.x += red[k][t][c].x; s.y += red[k][t][c].y; s.z += red[k][t][c].z; s.w += red[k][t][c].w; }
        *reinterpret_cast<float4*>(part + (long)blockIdx.x * 9 * C + t * C + 4 * c) = s;
    }
    if (tid == 0) bpart[blockIdx.x] = (bred[0] + bred[1]) + (bred[2] + bred[3]);
}

// out[i] (+)= sum_s part[s][i]: 16 columns x 16 slice lanes per workgroup, lane partials combined in a fixed order
__global__ __launch_bounds__(256) void disphead_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int S,
                                                              int n, int accumulate) {
    __shared__ float red[16][16];
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + col;
    float a0 = 0.f, a1 = 0.f;
    if (i < n) {
        int s = sl;
        for (; s + 16 < S; s += 32) { a0 += part[(long)s * n + i]; a1 += part[(long)(s + 16) * n + i]; }
        if (s < S) a0 += part[(long)s * n + i];
    }
    red[sl][col] = a0 + a1;
    __syncthreads();
    if (sl == 0 && i < n) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][col];
        out[i] = accumulate ? out[i] + v : v;
    }
}

constexpr int kWgradBlocks = 1024;

int check_dims(const char* what, int N, int H, int W, int C) {
    if (!(N >= 0 && H >= 2 && W >= 2)) return pd::fail(PD_EINVAL, "%s: bad shape N=%d H=%d W=%d (reflection needs H, W >= 2)", what, N, H, W);
    if (!(C == 16 || C == 32 || C == 64 || C == 128)) return pd::fail(PD_EINVAL, "%s: C=%d not in {16,32,64,128}", what, C);
    if ((long)N * H * W >= (1L << 31) - 64 * 4096) return pd::fail(PD_EINVAL, "%s: too many pixels for 32-bit indexing", what);
    return PD_OK;
}

inline unsigned grid_for(long npix, int C) {
    const int ppw = 64 / (C / 4);
    long blocks = (npix + 4L * ppw - 1) / (4L * ppw);
    return (unsigned)(blocks > 16384 ? 16384 : blocks);
}

}  // namespace

#define PD_DISPATCH_L(C, CALL)                 \
    switch (C) {                               \
        case 16: { constexpr int L = 4; CALL; } break;   \
        case 32: { constexpr int L = 8; CALL; } break;   \
        case 64: { constexpr int L = 16; CALL; } break;  \
        default: { constexpr int L = 32; CALL; } break;  \
    }

extern "C" int pd_disphead_fwd(const void* x, const void* w, const void* bias, void* y, int N, int H, int W, int C,
                               void* stream) {
    int rc = check_dims("pd_disphead_fwd", N, H, W, C);
    if (rc) return rc;
    if (N == 0) return PD_OK;
    PD_REQUIRE(x && w && y, "pd_disphead_fwd: null tensor");
    PD_REQUIRE(pd::aligned16(x) && pd::aligned16(w), "pd_disphead_fwd: x and w must be 16-byte aligned");
    if (C >= 64) {
        const int npix = N * H * W;
        PD_DISPATCH_L(C, hipLaunchKernelGGL(disphead_fwd_direct_kernel<L>, dim3(grid_for(npix, C)), dim3(256), 0,
                                            (hipStream_t)stream, (const float*)x, (const float*)w, (const float*)bias,
                                            (float*)y, npix, H, W));
        return pd::check_launch("pd_disphead_fwd");
    }
    const int tiles_h = (H + TR - 1) / TR, tiles_w = (W + TW - 1) / TW;
    const long ntiles = (long)N * tiles_h * tiles_w;
    PD_DISPATCH_L(C, hipLaunchKernelGGL(disphead_fwd_kernel<L>, dim3((unsigned)(ntiles > 4096 ? 4096 : ntiles)), dim3(256), 0,
                                        (hipStream_t)stream, (const float*)x, (const float*)w, (const float*)bias, (float*)y,
                                        H, W, tiles_h, tiles_w, (int)ntiles));
    return pd::check_launch("pd_disphead_fwd");
}

extern "C" int pd_disphead_bwd_data(const void* dy, const void* y, const void* w, void* dx, int N, int H, int W, int C,
                                    void* stream) {
    int rc = check_dims("pd_disphead_bwd_data", N, H, W, C);
    if (rc) return rc;
    if (N == 0) return PD_OK;
    PD_REQUIRE(dy && y && w && dx, "pd_disphead_bwd_data: null tensor");
    PD_REQUIRE(pd::aligned16(dx) && pd::aligned16(w), "pd_disphead_bwd_data: dx and w must be 16-byte aligned");
    const int npix = N * H * W;
    PD_DISPATCH_L(C, hipLaunchKernelGGL(disphead_bwd_data_kernel<L>, dim3(grid_for(npix, C)), dim3(256), 0,
                                        (hipStream_t)stream, (const float*)dy, (const float*)y, (const float*)w, (float*)dx,
                                        npix, H, W));
    return pd::check_launch("pd_disphead_bwd_data");
}

extern "C" size_t pd_disphead_workspace(int C) { return (size_t)kWgradBlocks * (9 * (size_t)C + 1) * sizeof(float); }

extern "C" int pd_disphead_bwd_weight(const void* dy, const void* y, const void* x, void* dw, void* dbias, void* workspace,
                                      size_t ws_bytes, int N, int H, int W, int C, int accumulate, void* stream) {
    int rc = check_dims("pd_disphead_bwd_weight", N, H, W, C);
    if (rc) return rc;
    if (N == 0) return PD_OK;
    PD_REQUIRE(dy && y && x && dw && workspace, "pd_disphead_bwd_weight: null tensor");
    PD_REQUIRE(ws_bytes >= pd_disphead_workspace(C), "pd_disphead_bwd_weight: workspace too small");
    PD_REQUIRE(pd::aligned16(x) && pd::aligned16(workspace), "pd_disphead_bwd_weight: x and workspace must be 16-byte aligned");
    const int npix = N * H * W;
    unsigned blocks = grid_for(npix, C);
    if (blocks > (unsigned)kWgradBlocks) blocks = kWgradBlocks;
    float* part = (float*)workspace;
    float* bpart = part + (size_t)kWgradBlocks * 9 * C;
    hipStream_t st = (hipStream_t)stream;
    PD_DISPATCH_L(C, hipLaunchKernelGGL(disphead_bwd_weight_kernel<L>, dim3(blocks), dim3(256), 0, st, (const float*)dy,
                                        (const float*)y, (const float*)x, part, bpart, npix, H, W));
    rc = pd::check_launch("pd_disphead_bwd_weight");
    if (rc) return rc;
    hipLaunchKernelGGL(disphead_reduce_kernel, dim3((9 * C + 15) / 16), dim3(256), 0, st, part, (float*)dw, (int)blocks,
                       9 * C, accumulate);
    if (dbias) hipLaunchKernelGGL(disphead_reduce_kernel, dim3(1), dim3(256), 0, st, bpart, (float*)dbias, (int)blocks, 1, accumulate);
    return pd::check_launch("pd_disphead_bwd_weight/reduce");
}

extern "C" int pd_disphead_bwd(const void* dy, const void* y, const void* x, const void* w, const void* add, int elu, void* dx,
                               void* dw, void* dbias, void* workspace, size_t ws_bytes, int N, int H, int W, int C,
                               int accumulate, void* stream) {
    int rc = check_dims("pd_disphead_bwd", N, H, W, C);
    if (rc) return rc;
    if (N == 0) return PD_OK;
    PD_REQUIRE(dy && y && x && w && dw && workspace, "pd_disphead_bwd: null tensor");
    PD_REQUIRE(ws_bytes >= pd_disphead_workspace(C), "pd_disphead_bwd: workspace too small");
    PD_REQUIRE(pd::aligned16(x) && pd::aligned16(w) && pd::aligned16(dx) && pd::aligned16(workspace) && pd::aligned16(add),
               "pd_disphead_bwd: x, w, add, dx and workspace must be 16-byte aligned");
    PD_REQUIRE(dx || (!add && !elu), "pd_disphead_bwd: add / elu modify dx, which is NULL");
    const int tiles_h = (H + TR - 1) / TR, tiles_w = (W + TW - 1) / TW;
    const long ntiles = (long)N * tiles_h * tiles_w;
    const unsigned blocks = (unsigned)(ntiles > kWgradBlocks ? kWgradBlocks : ntiles);
    float* part = (float*)workspace;
    float* bpart = part + (size_t)kWgradBlocks * 9 * C;
    hipStream_t st = (hipStream_t)stream;
    PD_DISPATCH_L(C, hipLaunchKernelGGL(disphead_bwd_kernel<L>, dim3(blocks), dim3(256), 0, st, (const float*)dy,
                                        (const float*)y, (const float*)x, (const float*)w, (const float*)add, elu, (float*)dx,
                                        part, bpart, H, W, tiles_h, tiles_w, (int)ntiles));
    rc = pd::check_launch("pd_disphead_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(disphead_reduce_kernel, dim3((9 * C + 15) / 16), dim3(256), 0, st, part, (float*)dw, (int)blocks,
                       9 * C, accumulate);
    if (dbias) hipLaunchKernelGGL(disphead_reduce_kernel, dim3(1), dim3(256), 0, st, bpart, (float*)dbias, (int)blocks, 1, accumulate);
    return pd::check_launch("pd_disphead_bwd/reduce");
}
