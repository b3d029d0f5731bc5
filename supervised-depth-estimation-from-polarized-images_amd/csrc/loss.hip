// K5 -- multi-scale supervised loss (forward + analytic backward), wavefront-reduced.
//
// Reference restated (per scale s of opt.scales):
//   trainer.py:538-543   disp_s -> F.interpolate(H x W, bilinear, align_corners=False) -> disp_to_depth
//   layers.py:62-71      disp_to_depth
//   trainer.py:1241-1251 mask = (gt >= min) & (gt <= max);  L1 = sum|gt - depth|*mask / sum(mask)
//   trainer.py:1298-1309 LN = sum((2 - cos(n_gt, n_pred)) * mask) / sum(mask), normals via
//                        kornia.geometry.depth.depth_to_normals (kornia 0.5.11: unproject with K,
//                        Sobel/8 with replicate padding, cross product, L2 normalise)
//   trainer.py:1256-1260 + layers.py:452-465  edge-aware smoothness of the mean-normalised disparity
//   trainer.py:1262-1265 loss_s = L1 + w_N * LN + w_sm * smooth / 2^s ;  loss = sum_s loss_s / S
// All sums are reduced per wavefront (shuffles), then per workgroup, written as fp32 partials and
// finished in fp64 by a one-block kernel -- no float atomics, run-to-run deterministic.
#include "pd_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int LT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum of K values; result valid in thread 0
template <int K>
__device__ __forceinline__ void block_sum(float (&v)[K], float* smem /* [4][K] */) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) smem[wave * K + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = smem[k] + smem[K + k] + smem[2 * K + k] + smem[3 * K + k];
}

inline unsigned lgrid(long n) {
    long b = (n + LT - 1) / LT;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// i -> (i / d, i % d) for a positive divisor: 32-bit unsigned arithmetic whenever the index fits; a 64-bit division is ~5x
// the instructions and every per-pixel kernel below decodes (n, y, x) with two of them
__device__ __forceinline__ long divmod(long i, int d, int& rem) {
    if (i >> 32) { const long q = i / d; rem = (int)(i - q * d); return q; }
    const unsigned u = (unsigned)i, q = u / (unsigned)d;
    rem = (int)(u - q * (unsigned)d);
    return (long)q;
}

// torch upsample_bilinear2d source index (align_corners=False)
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    const float s = fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1);
    l1 = s - i0;
}

// ------------------------------------------------------------------ disp -> full-res depth
__device__ __forceinline__ void disp_to_depth_body(const float* __restrict__ disp, float* __restrict__ depth,
                                                   float* __restrict__ updisp, int N, int hs, int ws, int H,
                                                   int W, float min_disp, float max_disp, unsigned bid, unsigned nb) {
    const long total = (long)N * H * W;
    const float sh = (float)hs / H, sw = (float)ws / W;
    for (long i = bid * (long)LT + threadIdx.x; i < total; i += (long)nb * LT) {
        int x, y;
        const long t = divmod(i, W, x);
        const long n = divmod(t, H, y);
        int y0, y1, x0, x1; float ly1, lx1;
        src_index(y, sh, hs, y0, y1, ly1);
        src_index(x, sw, ws, x0, x1, lx1);
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* d = disp + n * hs * ws;
        const float up = ly0 * (lx0 * d[y0 * ws + x0] + lx1 * d[y0 * ws + x1]) +
                         ly1 * (lx0 * d[y1 * ws + x0] + lx1 * d[y1 * ws + x1]);
        const float scaled = min_disp + (max_disp - min_disp) * up;
        depth[i] = 1.f / scaled;
        if (updisp) updisp[i] = up;
    }
}
__global__ __launch_bounds__(LT) void disp_to_depth_kernel(const float* __restrict__ disp, float* __restrict__ depth,
                                                           float* __restrict__ updisp, int N, int hs, int ws, int H,
                                                           int W, float min_disp, float max_disp) {
    disp_to_depth_body(disp, depth, updisp, N, hs, ws, H, W, min_disp, max_disp, blockIdx.x, gridDim.x);
}

// d(disp_s) = bilinear^T( g_up ), gather form: every low-res pixel sums its footprint
__global__ __launch_bounds__(LT) void up_gather_bwd_kernel(const float* __restrict__ gup, float* __restrict__ gdisp,
                                                           int N, int hs, int ws, int H, int W, int accumulate) {
    const long total = (long)N * hs * ws;
    const float sh = (float)hs / H, sw = (float)ws / W;
    const int fh = H / hs, fw = W / ws;   // integer zoom factors (checked on the host)
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int xs, ys;
        const long t = divmod(i, ws, xs);
        const long n = divmod(t, hs, ys);
        float acc = 0.f;
        const int ylo = max(0, (ys - 1) * fh), yhi = min(H - 1, (ys + 2) * fh - 1);
        const int xlo = max(0, (xs - 1) * fw), xhi = min(W - 1, (xs + 2) * fw - 1);
        for (int y = ylo; y <= yhi; ++y) {
            int y0, y1; float ly1;
            src_index(y, sh, hs, y0, y1, ly1);
            const float wy = (y0 == ys ? 1.f - ly1 : 0.f) + (y1 == ys ? ly1 : 0.f);
            if (wy == 0.f) continue;
            const float* g = gup + (n * H + y) * (long)W;
            for (int x = xlo; x <= xhi; ++x) {
                int x0, x1; float lx1;
                src_index(x, sw, ws, x0, x1, lx1);
                const float wx = (x0 == xs ? 1.f - lx1 : 0.f) + (x1 == xs ? lx1 : 0.f);
                if (wx != 0.f) acc += wy * wx * g[x];
            }
        }
        gdisp[i] = accumulate ? gdisp[i] + acc : acc;
    }
}

// Same sum for an isotropic zoom F in {1, 2, 4, 8}: the 3F column weights of a thread do not depend on the row, so they are
// evaluated once (same expressions, hence the same bits) and the 3F x 3F footprint is unrolled; the generic kernel
// re-derived both source indices for every element of the footprint (~25 instructions per element).
template <int F>
__device__ __forceinline__ void up_gather_bwd_f_body(const float* __restrict__ gup, float* __restrict__ gdisp,
                                                     int N, int hs, int ws, int H, int W, int accumulate, unsigned bid,
                                                     unsigned nb) {
    const long total = (long)N * hs * ws;
    const float sh = (float)hs / H, sw = (float)ws / W;
    for (long i = bid * (long)LT + threadIdx.x; i < total; i += (long)nb * LT) {
        int xs, ys;
        const long t = divmod(i, ws, xs);
        const long n = divmod(t, hs, ys);
        float wx[3 * F];
#pragma unroll
        for (int b = 0; b < 3 * F; ++b) {
            const int x = (xs - 1) * F + b;
            int x0, x1; float lx1;
            src_index(x, sw, ws, x0, x1, lx1);
            const float w = (x0 == xs ? 1.f - lx1 : 0.f) + (x1 == xs ? lx1 : 0.f);
            wx[b] = (x >= 0 && x < W) ? w : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 3 * F; ++a) {
            const int y = (ys - 1) * F + a;
            if (y < 0 || y >= H) continue;
            int y0, y1; float ly1;
            src_index(y, sh, hs, y0, y1, ly1);
            const float wy = (y0 == ys ? 1.f - ly1 : 0.f) + (y1 == ys ? ly1 : 0.f);
            if (wy == 0.f) continue;
            const float* g = gup + (n * H + y) * (long)W + (long)(xs - 1) * F;
#pragma unroll
            for (int b = 0; b < 3 * F; ++b)
                if (wx[b] != 0.f) acc += wy * wx[b] * g[b];
        }
        gdisp[i] = accumulate ? gdisp[i] + acc : acc;
    }
}
template <int F>
__global__ __launch_bounds__(LT) void up_gather_bwd_f_kernel(const float* __restrict__ gup, float* __restrict__ gdisp,
                                                             int N, int hs, int ws, int H, int W, int accumulate) {
    up_gather_bwd_f_body<F>(gup, gdisp, N, hs, ws, H, W, accumulate, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ supervised depth + normals terms
struct Cam { float fx, fy, cx, cy; };
__device__ __forceinline__ Cam load_cam(const float* K, long n) {
    const float* k = K + n * 16;  // [4][4]
    return Cam{k[0], k[5], k[2], k[6]};
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Sobel/8 gradients of the unprojected points around (x,y) with replicate padding
__device__ __forceinline__ void sobel_xyz(const float* __restrict__ D, int H, int W, int x, int y, Cam c, V3& A, V3& B) {
    A = V3{0, 0, 0}; B = V3{0, 0, 0};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = min(max(y + dy, 0), H - 1);
        const float sy = dy == 0 ? 2.f : 1.f, ddy = (float)dy;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = min(max(x + dx, 0), W - 1);
            const float sx = dx == 0 ? 2.f : 1.f, ddx = (float)dx;
            const float d = D[(long)yy * W + xx];
            const float X = (xx - c.cx) / c.fx * d, Y = (yy - c.cy) / c.fy * d;
            const float kx = sy * ddx * 0.125f, ky = ddy * sx * 0.125f;
            A.x += kx * X; A.y += kx * Y; A.z += kx * d;
            B.x += ky * X; B.y += ky * Y; B.z += ky * d;
        }
    }
}

__device__ __forceinline__ V3 normalize12(V3 v, float& nv) {
    nv = sqrtf(dot(v, v));
    const float inv = 1.f / fmaxf(nv, 1e-12f);
    return V3{v.x * inv, v.y * inv, v.z * inv};
}

// gtn[p] = normalised (A x B) of the ground-truth depth at every masked pixel (xyz, w unused): the same value for all
// scales and for the forward and backward kernels of a step, which otherwise each rebuild it from nine depths
__global__ __launch_bounds__(LT) void gt_normals_kernel(const float* __restrict__ gt, const float* __restrict__ K,
                                                        float4* __restrict__ gtn, int N, int H, int W, float min_d,
                                                        float max_d) {
    const long total = (long)N * H * W;
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int x, y;
        const long t = divmod(i, W, x);
        const long n = divmod(t, H, y);
        const float g = gt[i];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g >= min_d && g <= max_d) {
            const Cam c = load_cam(K, n);
            V3 A, B; float nv;
            sobel_xyz(gt + n * H * W, H, W, x, y, c, A, B);
            const V3 ng = normalize12(cross(A, B), nv);
            o = make_float4(ng.x, ng.y, ng.z, 0.f);
        }
        gtn[i] = o;
    }
}

// partial[block][3] = (sum |gt - d| m, sum (2 - cos) m, sum m)
__global__ __launch_bounds__(LT) void sup_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                     const float* __restrict__ K, const float4* __restrict__ gtn,
                                                     float* __restrict__ partial, int N,
                                                     int H, int W, float min_d, float max_d, int with_normals) {
    __shared__ float sm[4 * 3];
    const long total = (long)N * H * W;
    float acc[3] = {0.f, 0.f, 0.f};
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int x, y;
        const long t = divmod(i, W, x);
        const long n = divmod(t, H, y);
        const float g = gt[i];
        const float m = (g >= min_d && g <= max_d) ? 1.f : 0.f;
        if (m == 0.f) continue;   // every term is multiplied by the mask
        acc[0] += fabsf(g - pred[i]);
        acc[2] += 1.f;
        if (with_normals) {
            const Cam c = load_cam(K, n);
            V3 A, B; float nv;
            sobel_xyz(pred + n * H * W, H, W, x, y, c, A, B);
            const V3 np_ = normalize12(cross(A, B), nv);
            V3 ng;
            if (gtn) { const float4 q = gtn[i]; ng = V3{q.x, q.y, q.z}; }
            else { sobel_xyz(gt + n * H * W, H, W, x, y, c, A, B); ng = normalize12(cross(A, B), nv); }
            const float n1 = fmaxf(sqrtf(dot(ng, ng)), 1e-8f), n2 = fmaxf(sqrtf(dot(np_, np_)), 1e-8f);
            const float cs = (ng.x / n1) * (np_.x / n2) + (ng.y / n1) * (np_.y / n2) + (ng.z / n1) * (np_.z / n2);
            acc[1] += 2.f - cs;
        }
    }
    block_sum<3>(acc, sm);
    if (threadIdx.x == 0) { partial[blockIdx.x * 3] = acc[0]; partial[blockIdx.x * 3 + 1] = acc[1]; partial[blockIdx.x * 3 + 2] = acc[2]; }
}

// compute_supervised_normals_losses with an arbitrary mask (the facade method; the training step uses sup_fwd_kernel,
// whose mask is the depth-range mask the trainer always passes): partial[block][2] = (sum (2 - cos) m, sum m)
__global__ __launch_bounds__(LT) void normals_loss_masked_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                 const float* __restrict__ K, const float* __restrict__ mask,
                                                                 float* __restrict__ partial, int N, int H, int W) {
    __shared__ float sm[4 * 2];
    const long total = (long)N * H * W;
    float acc[2] = {0.f, 0.f};
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        const float m = mask[i];
        if (m == 0.f) continue;
        int x, y;
        const long t = divmod(i, W, x);
        const long n = divmod(t, H, y);
        const Cam c = load_cam(K, n);
        V3 A, B; float nv;
        sobel_xyz(pred + n * H * W, H, W, x, y, c, A, B);
        const V3 np_ = normalize12(cross(A, B), nv);
        sobel_xyz(gt + n * H * W, H, W, x, y, c, A, B);
        const V3 ng = normalize12(cross(A, B), nv);
        const float n1 = fmaxf(sqrtf(dot(ng, ng)), 1e-8f), n2 = fmaxf(sqrtf(dot(np_, np_)), 1e-8f);
        const float cs = (ng.x / n1) * (np_.x / n2) + (ng.y / n1) * (np_.y / n2) + (ng.z / n1) * (np_.z / n2);
        acc[0] += (2.f - cs) * m;
        acc[1] += m;
    }
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = acc[0]; partial[blockIdx.x * 2 + 1] = acc[1]; }
}

__global__ __launch_bounds__(256) void ratio_of_partials_kernel(const float* __restrict__ partial, int rows, float* __restrict__ out) {
    __shared__ double s0[256], s1[256];
    double a = 0.0, b = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) { a += partial[2 * r]; b += partial[2 * r + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { s0[threadIdx.x] += s0[threadIdx.x + k]; s1[threadIdx.x] += s1[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(s0[0] / s1[0]);
}

// ---- loss of PREDICTED normals (the `arch1++_separate_normals_dec` variant, README.md:54: "the decoder directly predicts
// normals; these are compared with normals calculated from ground truth"): sum((2 - cos(n_pred, n_gt)) m) / sum(m), the
// formula of trainer.py:1298-1309 with the network's own 3-channel output in place of the normals of the predicted depth.
// pred: pixel-major (NHWC) with pixel stride ld >= 3; gtn: pd_gt_normals; m = ground-truth depth inside [min, max].
__global__ __launch_bounds__(LT) void normals_pred_fwd_kernel(const float* __restrict__ pred, long ld, const float4* __restrict__ gtn,
                                                              const float* __restrict__ gt, float* __restrict__ partial,
                                                              long total, float min_d, float max_d) {
    __shared__ float sm[4 * 2];
    float acc[2] = {0.f, 0.f};
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        const float g = gt[i];
        if (!(g >= min_d && g <= max_d)) continue;
        const float4 q = gtn[i];
        const V3 ng{q.x, q.y, q.z}, np_{pred[i * ld], pred[i * ld + 1], pred[i * ld + 2]};
        const float n1 = fmaxf(sqrtf(dot(ng, ng)), 1e-8f), n2 = fmaxf(sqrtf(dot(np_, np_)), 1e-8f);
        acc[0] += 2.f - ((ng.x / n1) * (np_.x / n2) + (ng.y / n1) * (np_.y / n2) + (ng.z / n1) * (np_.z / n2));
        acc[1] += 1.f;
    }
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = acc[0]; partial[blockIdx.x * 2 + 1] = acc[1]; }
}

// out[0] = sum(p0) / sum(p1), out[1] = sum(p1): ordered fp64 sums of the partial rows
__global__ __launch_bounds__(256) void ratio_and_count_kernel(const float* __restrict__ partial, int rows, float* __restrict__ out) {
    __shared__ double s0[256], s1[256];
    double a = 0.0, b = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) { a += partial[2 * r]; b += partial[2 * r + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { s0[threadIdx.x] += s0[threadIdx.x + k]; s1[threadIdx.x] += s1[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)(s0[0] / s1[0]); out[1] = (float)s1[0]; }
}

// d loss / d pred = gout / count * m * -(d cos / d pred);  cos = <gh, p> / |p|  =>  d cos / d p = gh / |p| - <gh, p> p / |p|^3
__global__ __launch_bounds__(LT) void normals_pred_bwd_kernel(const float* __restrict__ pred, long ld, const float4* __restrict__ gtn,
                                                              const float* __restrict__ gt, const float* __restrict__ gout,
                                                              const float* __restrict__ loss_count, float* __restrict__ dpred,
                                                              long ldd, long total, float min_d, float max_d) {
    const float scale = gout[0] / loss_count[1];
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        const float g = gt[i];
        V3 d{0.f, 0.f, 0.f};
        if (g >= min_d && g <= max_d) {
            const float4 q = gtn[i];
            const V3 ng{q.x, q.y, q.z}, p{pred[i * ld], pred[i * ld + 1], pred[i * ld + 2]};
            const float n1 = fmaxf(sqrtf(dot(ng, ng)), 1e-8f);
            const V3 gh{ng.x / n1, ng.y / n1, ng.z / n1};
            const float pn = sqrtf(dot(p, p));
            if (pn > 1e-8f) {              // (below the clamp the cosine is linear in p: d cos / d p = gh / 1e-8; never met in training)
                const float inv = 1.f / pn, c = dot(gh, p) * inv * inv * inv;
                d = V3{-scale * (gh.x * inv - c * p.x), -scale * (gh.y * inv - c * p.y), -scale * (gh.z * inv - c * p.z)};
            } else {
                d = V3{-scale * gh.x * 1e8f, -scale * gh.y * 1e8f, -scale * gh.z * 1e8f};
            }
        }
        dpred[i * ldd] = d.x; dpred[i * ldd + 1] = d.y; dpred[i * ldd + 2] = d.z;
    }
}

// (dL/dA_p, dL/dB_p) of the normals term at pixel p = (x, y) of one image (pass A of the backward pass); zero outside the
// depth-range mask.  i = flat pixel index of p (for the cached ground-truth normal).
__device__ __forceinline__ void normals_grad_ab(const float* __restrict__ pred_n, const float* __restrict__ gt_n,
                                                const float4* __restrict__ gtn, long i, float g, Cam c, int H, int W, int x,
                                                int y, float wln, float min_d, float max_d, V3& dA, V3& dB) {
    dA = V3{0, 0, 0}; dB = V3{0, 0, 0};
    if (g >= min_d && g <= max_d) {
        V3 A, B, Ag, Bg; float nv, nvg;
        sobel_xyz(pred_n, H, W, x, y, c, A, B);
        const V3 v = cross(A, B);
        const V3 nn = normalize12(v, nv);
        V3 ng;
        if (gtn) { const float4 q = gtn[i]; ng = V3{q.x, q.y, q.z}; }
        else { sobel_xyz(gt_n, H, W, x, y, c, Ag, Bg); ng = normalize12(cross(Ag, Bg), nvg); }
        const float n1 = fmaxf(sqrtf(dot(ng, ng)), 1e-8f);
        const V3 gh{ng.x / n1, ng.y / n1, ng.z / n1};
        // cos = gh . (nn / max(|nn|, eps));  loss = -wln * cos (+ const)
        const float nl = sqrtf(dot(nn, nn));
        V3 wn;   // dL/d nn
        if (nl > 1e-8f) {
            const float inv = 1.f / nl, pr = dot(gh, nn) * inv * inv * inv;
            wn = V3{-wln * (gh.x * inv - pr * nn.x), -wln * (gh.y * inv - pr * nn.y), -wln * (gh.z * inv - pr * nn.z)};
        } else {
            wn = V3{-wln * gh.x * 1e8f, -wln * gh.y * 1e8f, -wln * gh.z * 1e8f};
        }
        V3 wv;   // dL/d v, v -> nn = v / max(|v|, 1e-12)
        if (nv > 1e-12f) {
            const float inv = 1.f / nv, pr = dot(wn, nn);
            wv = V3{(wn.x - pr * nn.x) * inv, (wn.y - pr * nn.y) * inv, (wn.z - pr * nn.z) * inv};
        } else {
            wv = V3{wn.x * 1e12f, wn.y * 1e12f, wn.z * 1e12f};
        }
        dA = cross(B, wv);   // d(A x B).w / dA = B x w
        dB = cross(wv, A);   // d(A x B).w / dB = w x A
    }
}

// pass A: per pixel p, (dL/dA_p, dL/dB_p) of the normals term  -> ab [N,H,W,6]
// wts = (w_L1, w_LN, w_sm) for this scale (device), sums = (.., .., sum mask) of the forward
__global__ __launch_bounds__(LT) void sup_bwd_a_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                       const float* __restrict__ K, const float4* __restrict__ gtn,
                                                       const float* __restrict__ wts,
                                                       const double* __restrict__ sums, float* __restrict__ ab, int N,
                                                       int H, int W, float min_d, float max_d) {
    const long total = (long)N * H * W;
    const float wln = (float)((double)wts[1] / sums[2]);
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int x, y;
        const long t = divmod(i, W, x);
        const long n = divmod(t, H, y);
        const float g = gt[i];
        V3 dA, dB;
        normals_grad_ab(pred + n * H * W, gt + n * H * W, gtn, i, g, load_cam(K, n), H, W, x, y, wln, min_d, max_d, dA, dB);
        float* o = ab + i * 6;
        o[0] = dA.x; o[1] = dA.y; o[2] = dA.z; o[3] = dB.x; o[4] = dB.y; o[5] = dB.z;
    }
}

// separable replicate-padding weights: sum of taps delta in {-1,0,1} of p that land on q
__device__ __forceinline__ void tap_weights(int q, int p, int n, float& wd, float& ws) {
    wd = 0.f; ws = 0.f;
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
        const int t = min(max(p + d, 0), n - 1);
        if (t == q) { wd += (float)d; ws += d == 0 ? 2.f : 1.f; }
    }
}

// pass B: d loss / d depth(q) = L1 term + r(q) . sum_p (...), then through depth = 1/(a + b*up)
__global__ __launch_bounds__(LT) void sup_bwd_b_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                       const float* __restrict__ K, const float* __restrict__ wts,
                                                       const double* __restrict__ sums, const float* __restrict__ ab,
                                                       float* __restrict__ gout, int N, int H, int W, float min_d,
                                                       float max_d, float disp_range, int with_normals, int to_disp) {
    const long total = (long)N * H * W;
    const float wl1 = (float)((double)wts[0] / sums[2]);
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int x, y;
        const long t = divmod(i, W, x);
        const long n = divmod(t, H, y);
        const float g = gt[i], d = pred[i];
        float gd = 0.f;
        if (g >= min_d && g <= max_d) gd = wl1 * (d > g ? 1.f : (d < g ? -1.f : 0.f));
        if (with_normals) {
            const Cam c = load_cam(K, n);
            V3 s{0, 0, 0};
            for (int py = max(y - 1, 0); py <= min(y + 1, H - 1); ++py) {
                float dyw, syw;
                tap_weights(y, py, H, dyw, syw);
                for (int px = max(x - 1, 0); px <= min(x + 1, W - 1); ++px) {
                    float dxw, sxw;
                    tap_weights(x, px, W, dxw, sxw);
                    const float ka = syw * dxw * 0.125f, kb = dyw * sxw * 0.125f;
                    const float* o = ab + ((n * H + py) * (long)W + px) * 6;
                    s.x += ka * o[0] + kb * o[3]; s.y += ka * o[1] + kb * o[4]; s.z += ka * o[2] + kb * o[5];
                }
            }
            gd += (x - c.cx) / c.fx * s.x + (y - c.cy) / c.fy * s.y + s.z;
        }
        // depth = 1 / (min_disp + range * up)  ->  d depth / d up = -range * depth^2
        gout[i] = to_disp ? gd * (-disp_range * d * d) : gd;
    }
}

// Passes A and B in one kernel: a workgroup evaluates (dA, dB) for the 10 x 66 halo of an 8 x 64 pixel tile into LDS
// (1.29x the evaluations) and gathers from there -- the [N,H,W,6] intermediate (126 MB written and re-read per scale)
// never reaches memory.  Same per-pixel arithmetic and summation order as the two-pass form.
constexpr int SB_TR = 8, SB_TW = 64, SB_HR = SB_TR + 2, SB_HC = SB_TW + 2, SB_HP = SB_HR * SB_HC;
__device__ __forceinline__ void sup_bwd_fused_body(float (&abl)[6][SB_HP], const float* __restrict__ pred,
                                                   const float* __restrict__ gt,
                                                   const float* __restrict__ K, const float4* __restrict__ gtn,
                                                   const float* __restrict__ wts, const double* __restrict__ sums,
                                                   float* __restrict__ gout, int N, int H, int W, float min_d,
                                                   float max_d, float disp_range, int to_disp, int tiles_h,
                                                   int tiles_w, int ntiles, unsigned bid, unsigned nb) {
    const float wln = (float)((double)wts[1] / sums[2]);
    const float wl1 = (float)((double)wts[0] / sums[2]);
    for (int tile = (int)bid; tile < ntiles; tile += (int)nb) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h;
        const long n = tile / (tiles_w * tiles_h);
        const int h0 = th * SB_TR, w0 = tw * SB_TW;
        const Cam c = load_cam(K, n);
        const float* pn = pred + n * H * W;
        const float* gn = gt + n * H * W;
        __syncthreads();
        for (int s = threadIdx.x; s < SB_HP; s += LT) {
            const int r = s / SB_HC, cc = s - r * SB_HC;
            const int y = h0 - 1 + r, x = w0 - 1 + cc;
            V3 dA{0, 0, 0}, dB{0, 0, 0};
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const long i = (n * H + y) * (long)W + x;
                normals_grad_ab(pn, gn, gtn, i, gt[i], c, H, W, x, y, wln, min_d, max_d, dA, dB);
            }
            abl[0][s] = dA.x; abl[1][s] = dA.y; abl[2][s] = dA.z; abl[3][s] = dB.x; abl[4][s] = dB.y; abl[5][s] = dB.z;
        }
        __syncthreads();
        for (int o = threadIdx.x; o < SB_TR * SB_TW; o += LT) {
            const int r = o / SB_TW, cc = o - r * SB_TW;
            const int y = h0 + r, x = w0 + cc;
            if (y >= H || x >= W) continue;
            const long i = (n * H + y) * (long)W + x;
            const float g = gt[i], d = pred[i];
            float gd = 0.f;
            if (g >= min_d && g <= max_d) gd = wl1 * (d > g ? 1.f : (d < g ? -1.f : 0.f));
            V3 s{0, 0, 0};
            for (int py = max(y - 1, 0); py <= min(y + 1, H - 1); ++py) {
                float dyw, syw;
                tap_weights(y, py, H, dyw, syw);
                for (int px = max(x - 1, 0); px <= min(x + 1, W - 1); ++px) {
                    float dxw, sxw;
                    tap_weights(x, px, W, dxw, sxw);
                    const float ka = syw * dxw * 0.125f, kb = dyw * sxw * 0.125f;
                    const int q = (py - h0 + 1) * SB_HC + (px - w0 + 1);
                    s.x += ka * abl[0][q] + kb * abl[3][q]; s.y += ka * abl[1][q] + kb * abl[4][q];
                    s.z += ka * abl[2][q] + kb * abl[5][q];
                }
            }
            gd += (x - c.cx) / c.fx * s.x + (y - c.cy) / c.fy * s.y + s.z;
            gout[i] = to_disp ? gd * (-disp_range * d * d) : gd;
        }
    }
}
__global__ __launch_bounds__(LT) void sup_bwd_fused_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                           const float* __restrict__ K, const float4* __restrict__ gtn,
                                                           const float* __restrict__ wts, const double* __restrict__ sums,
                                                           float* __restrict__ gout, int N, int H, int W, float min_d,
                                                           float max_d, float disp_range, int to_disp, int tiles_h,
                                                           int tiles_w, int ntiles) {
    __shared__ float abl[6][SB_HP];
    sup_bwd_fused_body(abl, pred, gt, K, gtn, wts, sums, gout, N, H, W, min_d, max_d, disp_range, to_disp, tiles_h, tiles_w,
                       ntiles, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ edge-aware smoothness
__device__ __forceinline__ void image_mean_body(double (&sm)[16], const float* __restrict__ disp, float* __restrict__ mean,
                                                int P, unsigned img) {
    const float* d = disp + (long)img * P;
    // four independent fp64 chains over 16-byte loads (one chain of scalar loads was latency-bound: 32 us per launch)
    const int P4 = ((P & 3) == 0 && (((size_t)d) & 15) == 0) ? P >> 2 : 0;
    const float4* d4 = reinterpret_cast<const float4*>(d);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
    for (int i = threadIdx.x; i < P4; i += 1024) {
        const float4 v = d4[i];
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    for (int i = 4 * P4 + threadIdx.x; i < P; i += 1024) s0 += d[i];
    double s = (s0 + s1) + (s2 + s3);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += sm[i];
        mean[img] = (float)(t / P);
    }
}
__global__ __launch_bounds__(1024) void image_mean_kernel(const float* __restrict__ disp, float* __restrict__ mean, int P) {
    __shared__ double sm[16];
    image_mean_body(sm, disp, mean, P, blockIdx.x);
}

__device__ __forceinline__ float color_grad(const float* __restrict__ img, long base, long plane, long a, long b) {
    // mean over 3 channels of |I[a] - I[b]|, img NCHW planar
    return (fabsf(img[base + a] - img[base + b]) + fabsf(img[base + plane + a] - img[base + plane + b]) +
            fabsf(img[base + 2 * plane + a] - img[base + 2 * plane + b])) / 3.f;
}

// partial[block][2] = (sum_x |dx norm| e^{-|dx I|}, sum_y ...)
// edge_w (optional) [N,h,w,2] receives e^{-|dx I|}, e^{-|dy I|} of every pixel (0 on the last column / row): they depend
// on the image only and the backward pass needs each of them twice
__device__ __forceinline__ void smooth_fwd_body(float* sm, const float* __restrict__ disp, const float* __restrict__ img,
                                                const float* __restrict__ mean, float* __restrict__ partial,
                                                float2* __restrict__ edge_w, int N, int h, int w, unsigned bid, unsigned nb) {
    const long P = (long)h * w, total = N * P;
    float acc[2] = {0.f, 0.f};
    for (long i = bid * (long)LT + threadIdx.x; i < total; i += (long)nb * LT) {
        int x, y;
        const long t = divmod(i, w, x);
        const long n = divmod(t, h, y);
        const float inv = mean ? 1.f / (mean[n] + 1e-7f) : 1.f;     // mean == nullptr: the raw term (layers.get_smooth_loss)
        const float v = disp[i] * inv;
        const long ib = n * 3 * P, p = (long)y * w + x;
        float ex = 0.f, ey = 0.f;
        if (x + 1 < w) { ex = expf(-color_grad(img, ib, P, p, p + 1)); acc[0] += fabsf(v - disp[i + 1] * inv) * ex; }
        if (y + 1 < h) { ey = expf(-color_grad(img, ib, P, p, p + w)); acc[1] += fabsf(v - disp[i + w] * inv) * ey; }
        if (edge_w) edge_w[i] = make_float2(ex, ey);
    }
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) { partial[bid * 2] = acc[0]; partial[bid * 2 + 1] = acc[1]; }
}
__global__ __launch_bounds__(LT) void smooth_fwd_kernel(const float* __restrict__ disp, const float* __restrict__ img,
                                                        const float* __restrict__ mean, float* __restrict__ partial,
                                                        float2* __restrict__ edge_w, int N, int h, int w) {
    __shared__ float sm[4 * 2];
    smooth_fwd_body(sm, disp, img, mean, partial, edge_w, N, h, w, blockIdx.x, gridDim.x);
}

// G = d smooth / d norm (already times the scale weight); gd_acc[n] += sum_i G_i * disp_i
__device__ __forceinline__ void smooth_bwd_g_body(double (&smd)[4], const float* __restrict__ disp,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ mean, const float* __restrict__ wts,
                                                  const float2* __restrict__ edge_w,
                                                  float* __restrict__ G, double* __restrict__ gd_acc, int N,
                                                  int h, int w, unsigned bid, unsigned nb) {
    const long P = (long)h * w, total = N * P;
    const float wx = wts[2] / (float)((double)N * h * (w - 1)), wy = wts[2] / (float)((double)N * (h - 1) * w);
    // A workgroup walks a CONTIGUOUS range of pixels and keeps the running sum of the image it is in: one fp64 atomic per
    // workgroup and image.  (One atomic per 256 pixels -- 20 000 on 16 addresses at scale 0 -- serialised in the L2:
    // 0.2 ms of a kernel whose loads take 0.03.)
    const long chunk = ((total + nb - 1) / nb + LT - 1) / LT * LT;
    const long beg = bid * chunk, end = min(beg + chunk, total);
    double run = 0.0;
    long run_n = -1;
    auto flush = [&]() {                       // block-uniform
        if (run_n >= 0) {
            double v = run;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((threadIdx.x & 63) == 0) smd[threadIdx.x >> 6] = v;
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(&gd_acc[run_n], (smd[0] + smd[1]) + (smd[2] + smd[3]));
            __syncthreads();
        }
        run = 0.0;
    };
    for (long i0 = beg; i0 < end; i0 += LT) {
        const long i = i0 + threadIdx.x;
        float gsum = 0.f; long n = 0; float gi = 0.f;
        if (i < total) {
            int x, y;
            const long t = divmod(i, w, x);
            n = divmod(t, h, y);
            const float inv = mean ? 1.f / (mean[n] + 1e-7f) : 1.f;
            const float v = disp[i] * inv;
            const long ib = n * 3 * P, p = (long)y * w + x;
            auto sgn = [](float a) { return a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f); };
            if (edge_w) {          // the four edge weights from the forward pass instead of 24 image loads and 4 exp
                if (x + 1 < w) gi += wx * sgn(v - disp[i + 1] * inv) * edge_w[i].x;
                if (x > 0) gi -= wx * sgn(disp[i - 1] * inv - v) * edge_w[i - 1].x;
                if (y + 1 < h) gi += wy * sgn(v - disp[i + w] * inv) * edge_w[i].y;
                if (y > 0) gi -= wy * sgn(disp[i - w] * inv - v) * edge_w[i - w].y;
            } else {
                if (x + 1 < w) gi += wx * sgn(v - disp[i + 1] * inv) * expf(-color_grad(img, ib, P, p, p + 1));
                if (x > 0) gi -= wx * sgn(disp[i - 1] * inv - v) * expf(-color_grad(img, ib, P, p - 1, p));
                if (y + 1 < h) gi += wy * sgn(v - disp[i + w] * inv) * expf(-color_grad(img, ib, P, p, p + w));
                if (y > 0) gi -= wy * sgn(disp[i - w] * inv - v) * expf(-color_grad(img, ib, P, p - w, p));
            }
            G[i] = gi;
            gsum = gi * disp[i];
        }
        const long nfirst = i0 / P, nlast = (min(i0 + LT, total) - 1) / P;
        if (nfirst == nlast) {
            if (nfirst != run_n) { flush(); run_n = nfirst; }
            run += (double)gsum;
        } else {                               // the 256 pixels straddle two images: per-thread atomics
            flush();
            run_n = -1;
            if (i < total) atomicAdd(&gd_acc[n], (double)gsum);
        }
    }
    flush();
}
__global__ __launch_bounds__(LT) void smooth_bwd_g_kernel(const float* __restrict__ disp, const float* __restrict__ img,
                                                          const float* __restrict__ mean, const float* __restrict__ wts,
                                                          const float2* __restrict__ edge_w,
                                                          float* __restrict__ G, double* __restrict__ gd_acc, int N,
                                                          int h, int w) {
    __shared__ double smd[4];
    smooth_bwd_g_body(smd, disp, img, mean, wts, edge_w, G, gd_acc, N, h, w, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void smooth_bwd_final_body(const float* __restrict__ G, const float* __restrict__ mean,
                                                      const double* __restrict__ gd_acc, float* __restrict__ gdisp,
                                                      int N, long P, int accumulate, unsigned bid, unsigned nb) {
    const long total = N * P;
    for (long i = bid * (long)LT + threadIdx.x; i < total; i += (long)nb * LT) {
        const long n = i / P;
        if (!mean) { gdisp[i] = accumulate ? gdisp[i] + G[i] : G[i]; continue; }       // no normalisation: d/d disp = G
        const float me = mean[n] + 1e-7f;
        const float v = G[i] / me - (float)(gd_acc[n] / ((double)me * me * (double)P));
        gdisp[i] = accumulate ? gdisp[i] + v : v;
    }
}
__global__ __launch_bounds__(LT) void smooth_bwd_final_kernel(const float* __restrict__ G, const float* __restrict__ mean,
                                                              const double* __restrict__ gd_acc, float* __restrict__ gdisp,
                                                              int N, long P, int accumulate) {
    smooth_bwd_final_body(G, mean, gd_acc, gdisp, N, P, accumulate, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ all scales in one launch
// The per-scale kernels above, dispatched by blockIdx.y = scale with the SAME per-scale grid (blocks beyond a scale's own
// grid leave at once), so every partial sum, atomic and rounding is the one the single-scale launches produce: the
// multi-scale loss is 6 launches forward (gt normals, depths, supervised terms, image means, smoothness, finalise) and
// 6 backward instead of 18 + 21.  The supervised forward kernel additionally reads gt / gt normals once for all scales.
struct MsArgs {
    int S, N, H, W;
    const float* disp[8]; const float* color[8];
    float* depth[8]; float* mean[8]; float2* edge[8];
    float* gdisp[8]; float* gws[8]; float* gup[8];
    int hs[8], ws[8];
    unsigned nb_full, nb_s[8];          // blocks of a full-resolution / a scale-resolution launch
};

__global__ __launch_bounds__(LT) void ms_disp_to_depth_kernel(const MsArgs a, float min_disp, float max_disp) {
    const int s = blockIdx.y;
    disp_to_depth_body(a.disp[s], a.depth[s], nullptr, a.N, a.hs[s], a.ws[s], a.H, a.W, min_disp, max_disp, blockIdx.x, gridDim.x);
}

// partial[s][block][3]: the sums of sup_fwd_kernel for every scale, one pass over the pixels
__global__ __launch_bounds__(LT) void ms_sup_fwd_kernel(const MsArgs a, const float* __restrict__ gt, const float* __restrict__ K,
                                                        const float4* __restrict__ gtn, float* __restrict__ partial,
                                                        int part_stride, float min_d, float max_d, int with_normals) {
    __shared__ float sm[4 * 3];
    const int N = a.N, H = a.H, W = a.W;
    const long total = (long)N * H * W;
    float acc[8][3];
#pragma unroll
    for (int s = 0; s < 8; ++s) { acc[s][0] = 0.f; acc[s][1] = 0.f; acc[s][2] = 0.f; }
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int x, y;
        const long t = divmod(i, W, x);
        const long n = divmod(t, H, y);
        const float g = gt[i];
        if (!(g >= min_d && g <= max_d)) continue;
        V3 ng{0, 0, 0};
        float n1 = 1.f;
        Cam c{};
        if (with_normals) {
            c = load_cam(K, n);
            if (gtn) { const float4 q = gtn[i]; ng = V3{q.x, q.y, q.z}; }
            else { V3 A, B; float nv; sobel_xyz(gt + n * H * W, H, W, x, y, c, A, B); ng = normalize12(cross(A, B), nv); }
            n1 = fmaxf(sqrtf(dot(ng, ng)), 1e-8f);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s >= a.S) break;
            const float* pred = a.depth[s];
            acc[s][0] += fabsf(g - pred[i]);
            acc[s][2] += 1.f;
            if (with_normals) {
                V3 A, B; float nv;
                sobel_xyz(pred + n * H * W, H, W, x, y, c, A, B);
                const V3 np_ = normalize12(cross(A, B), nv);
                const float n2 = fmaxf(sqrtf(dot(np_, np_)), 1e-8f);
                const float cs = (ng.x / n1) * (np_.x / n2) + (ng.y / n1) * (np_.y / n2) + (ng.z / n1) * (np_.z / n2);
                acc[s][1] += 2.f - cs;
            }
        }
    }
    for (int s = 0; s < a.S; ++s) {
        float v[3] = {acc[s][0], acc[s][1], acc[s][2]};
        __syncthreads();
        block_sum<3>(v, sm);
        if (threadIdx.x == 0) {
            float* p = partial + ((long)s * part_stride + blockIdx.x) * 3;
            p[0] = v[0]; p[1] = v[1]; p[2] = v[2];
        }
    }
}

__global__ __launch_bounds__(1024) void ms_image_mean_kernel(const MsArgs a) {
    __shared__ double sm[16];
    const int s = blockIdx.y;
    image_mean_body(sm, a.disp[s], a.mean[s], a.hs[s] * a.ws[s], blockIdx.x);
}

__global__ __launch_bounds__(LT) void ms_smooth_fwd_kernel(const MsArgs a, float* __restrict__ partial, int part_stride) {
    __shared__ float sm[4 * 2];
    const int s = blockIdx.y;
    if (blockIdx.x >= a.nb_s[s]) return;
    smooth_fwd_body(sm, a.disp[s], a.color[s], a.mean[s], partial + (long)s * part_stride * 2, a.edge[s], a.N, a.hs[s], a.ws[s],
                    blockIdx.x, a.nb_s[s]);
}

__global__ __launch_bounds__(LT) void ms_sup_bwd_kernel(const MsArgs a, const float* __restrict__ gt, const float* __restrict__ K,
                                                        const float4* __restrict__ gtn, const float* __restrict__ wts,
                                                        const double* __restrict__ sums, float min_d, float max_d,
                                                        float disp_range, int tiles_h, int tiles_w, int ntiles) {
    __shared__ float abl[6][SB_HP];
    const int s = blockIdx.y;
    sup_bwd_fused_body(abl, a.depth[s], gt, K, gtn, wts + 3 * s, sums + 5 * s, a.gup[s], a.N, a.H, a.W, min_d, max_d, disp_range, 1,
                       tiles_h, tiles_w, ntiles, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(LT) void ms_up_gather_bwd_kernel(const MsArgs a) {
    const int s = blockIdx.y;
    if (blockIdx.x >= a.nb_s[s]) return;
    const int f = a.H / a.hs[s];
    if (f == 1) up_gather_bwd_f_body<1>(a.gup[s], a.gdisp[s], a.N, a.hs[s], a.ws[s], a.H, a.W, 0, blockIdx.x, a.nb_s[s]);
    else if (f == 2) up_gather_bwd_f_body<2>(a.gup[s], a.gdisp[s], a.N, a.hs[s], a.ws[s], a.H, a.W, 0, blockIdx.x, a.nb_s[s]);
    else if (f == 4) up_gather_bwd_f_body<4>(a.gup[s], a.gdisp[s], a.N, a.hs[s], a.ws[s], a.H, a.W, 0, blockIdx.x, a.nb_s[s]);
    else up_gather_bwd_f_body<8>(a.gup[s], a.gdisp[s], a.N, a.hs[s], a.ws[s], a.H, a.W, 0, blockIdx.x, a.nb_s[s]);
}

__global__ __launch_bounds__(LT) void ms_smooth_bwd_g_kernel(const MsArgs a, const float* __restrict__ wts,
                                                             double* __restrict__ gd_acc) {
    __shared__ double smd[4];
    const int s = blockIdx.y;
    if (blockIdx.x >= a.nb_s[s]) return;
    smooth_bwd_g_body(smd, a.disp[s], a.color[s], a.mean[s], wts + 3 * s, a.edge[s], a.gws[s], gd_acc + (long)s * a.N, a.N,
                      a.hs[s], a.ws[s], blockIdx.x, a.nb_s[s]);
}

__global__ __launch_bounds__(LT) void ms_smooth_bwd_final_kernel(const MsArgs a, const double* __restrict__ gd_acc) {
    const int s = blockIdx.y;
    if (blockIdx.x >= a.nb_s[s]) return;
    smooth_bwd_final_body(a.gws[s], a.mean[s], gd_acc + (long)s * a.N, a.gdisp[s], a.N, (long)a.hs[s] * a.ws[s], 1, blockIdx.x,
                          a.nb_s[s]);
}

// ------------------------------------------------------------------ scalar bookkeeping
// vals layout: [0] loss, then per scale s: [1+3s] loss/s, [2+3s] supervised_depth_loss/s, [3+3s] normals loss
// sums layout (double): per scale 5: sum|.|m, sum(2-cos)m, sum m, smooth_x_sum, smooth_y_sum
struct LossMeta {
    int S;
    int sup_rows[8], sm_rows[8], N[8], h[8], w[8], scale_id[8];
};

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ sup_part,
                                                            const float* __restrict__ sm_part, const LossMeta m,
                                                            int part_stride, float w_normals, float w_smooth,
                                                            double* __restrict__ sums, float* __restrict__ vals) {
    // one workgroup; thread t sums rows t, t+256, ... of every partial array in fp64, then a tree over LDS
    __shared__ double red[256];
    __shared__ double tot[8 * 5];
    for (int s = 0; s < m.S; ++s) {
        for (int j = 0; j < 5; ++j) {
            double v = 0.0;
            if (j < 3) {
                const float* sp = sup_part + (long)s * part_stride * 3;
                for (int r = threadIdx.x; r < m.sup_rows[s]; r += 256) v += sp[r * 3 + j];
            } else {
                const float* mp = sm_part + (long)s * part_stride * 2;
                for (int r = threadIdx.x; r < m.sm_rows[s]; r += 256) v += mp[r * 2 + (j - 3)];
            }
            red[threadIdx.x] = v;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
                __syncthreads();
            }
            if (threadIdx.x == 0) tot[s * 5 + j] = red[0];
            __syncthreads();
        }
    }
    if (threadIdx.x != 0) return;
    double total = 0.0;
    for (int s = 0; s < m.S; ++s) {
        const double* a = tot + s * 5;
        for (int j = 0; j < 5; ++j) sums[s * 5 + j] = a[j];
        const int N = m.N[s], h = m.h[s], w = m.w[s];
        const double l1 = a[0] / a[2], ln = a[1] / a[2];
        const double smooth = a[3] / ((double)N * h * (w - 1)) + a[4] / ((double)N * (h - 1) * w);
        const double ls = l1 + (double)w_normals * ln + (double)w_smooth * smooth / (double)(1 << m.scale_id[s]);
        vals[1 + 3 * s] = (float)ls;
        vals[2 + 3 * s] = (float)l1;
        vals[3 + 3 * s] = (float)ln;
        total += ls;
    }
    vals[0] = (float)(total / m.S);
}

// vals from given sums (S x 5 doubles): the second half of loss_finalize_kernel, for sums that were exchanged
// between data-parallel ranks in between (global mask normalisation, trainer.py:1247,1308 over the global batch)
__global__ void loss_from_sums_kernel(const double* __restrict__ sums, const LossMeta m, float w_normals,
                                      float w_smooth, float* __restrict__ vals) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double total = 0.0;
    for (int s = 0; s < m.S; ++s) {
        const double* a = sums + s * 5;
        const int N = m.N[s], h = m.h[s], w = m.w[s];
        const double l1 = a[0] / a[2], ln = a[1] / a[2];
        const double smooth = a[3] / ((double)N * h * (w - 1)) + a[4] / ((double)N * (h - 1) * w);
        const double ls = l1 + (double)w_normals * ln + (double)w_smooth * smooth / (double)(1 << m.scale_id[s]);
        vals[1 + 3 * s] = (float)ls;
        vals[2 + 3 * s] = (float)l1;
        vals[3 + 3 * s] = (float)ln;
        total += ls;
    }
    vals[0] = (float)(total / m.S);
}

// per-scale weights (w_L1, w_LN, w_sm) from the upstream gradient of vals
__global__ void loss_weights_kernel(const float* __restrict__ gvals, const LossMeta m, float w_normals, float w_smooth,
                                    float* __restrict__ wts) {
    const int s = threadIdx.x;
    if (s >= m.S) return;
    const float gl = gvals[0] / m.S + gvals[1 + 3 * s];
    wts[3 * s] = gl + gvals[2 + 3 * s];
    wts[3 * s + 1] = gl * w_normals + gvals[3 + 3 * s];
    wts[3 * s + 2] = gl * w_smooth / (float)(1 << m.scale_id[s]);
}

}  // namespace

// ============================================================================ C ABI
extern "C" int pd_loss_rows(long n) { return (int)lgrid(n); }

extern "C" int pd_disp_to_depth(const void* disp, void* depth, void* updisp, int N, int hs, int ws, int H, int W,
                                float min_depth, float max_depth, void* stream) {
    PD_REQUIRE(disp && depth && N >= 0 && hs > 0 && ws > 0 && H >= hs && W >= ws, "pd_disp_to_depth: bad arguments");
    PD_REQUIRE(min_depth > 0 && max_depth > min_depth, "pd_disp_to_depth: bad depth range");
    if (N == 0) return PD_OK;
    hipLaunchKernelGGL(disp_to_depth_kernel, dim3(lgrid((long)N * H * W)), dim3(LT), 0, (hipStream_t)stream,
                       (const float*)disp, (float*)depth, (float*)updisp, N, hs, ws, H, W, 1.f / max_depth,
                       1.f / min_depth);
    return pd::check_launch("pd_disp_to_depth");
}

extern "C" int pd_up_gather_bwd(const void* gup, void* gdisp, int N, int hs, int ws, int H, int W, int accumulate,
                                int generic, void* stream) {
    PD_REQUIRE(gup && gdisp && N >= 0 && hs > 0 && ws > 0, "pd_up_gather_bwd: bad arguments");
    PD_REQUIRE(H % hs == 0 && W % ws == 0, "pd_up_gather_bwd: full size must be an integer multiple of the scale size");
    if (N == 0) return PD_OK;
    const int f = generic ? 0 : H / hs;                                // generic != 0: the any-ratio kernel (tests compare the two)
    const unsigned grid = lgrid((long)N * hs * ws);
#define PD_UPG(F) hipLaunchKernelGGL(up_gather_bwd_f_kernel<F>, dim3(grid), dim3(LT), 0, (hipStream_t)stream, \
                                     (const float*)gup, (float*)gdisp, N, hs, ws, H, W, accumulate)
    if (W / ws == f && f == 1) PD_UPG(1);
    else if (W / ws == f && f == 2) PD_UPG(2);
    else if (W / ws == f && f == 4) PD_UPG(4);
    else if (W / ws == f && f == 8) PD_UPG(8);
    else
        hipLaunchKernelGGL(up_gather_bwd_kernel, dim3(grid), dim3(LT), 0, (hipStream_t)stream, (const float*)gup,
                           (float*)gdisp, N, hs, ws, H, W, accumulate);
#undef PD_UPG
    return pd::check_launch("pd_up_gather_bwd");
}

extern "C" int pd_gt_normals(const void* gt, const void* K, void* gtn, int N, int H, int W, float min_depth,
                             float max_depth, void* stream) {
    PD_REQUIRE(gt && K && gtn && N > 0 && H > 0 && W > 0 && pd::aligned16(gtn), "pd_gt_normals: bad arguments");
    hipLaunchKernelGGL(gt_normals_kernel, dim3(lgrid((long)N * H * W)), dim3(LT), 0, (hipStream_t)stream, (const float*)gt,
                       (const float*)K, (float4*)gtn, N, H, W, min_depth, max_depth);
    return pd::check_launch("pd_gt_normals");
}

extern "C" int pd_sup_loss_fwd(const void* pred, const void* gt, const void* K, const void* gt_normals, void* partial, int N,
                               int H, int W, float min_depth, float max_depth, int with_normals, void* stream) {
    PD_REQUIRE(pred && gt && partial && (K || !with_normals) && N > 0 && H > 0 && W > 0 && pd::aligned16(gt_normals),
               "pd_sup_loss_fwd: bad arguments");
    hipLaunchKernelGGL(sup_fwd_kernel, dim3(lgrid((long)N * H * W)), dim3(LT), 0, (hipStream_t)stream,
                       (const float*)pred, (const float*)gt, (const float*)K, (const float4*)gt_normals, (float*)partial, N,
                       H, W, min_depth, max_depth, with_normals);
    return pd::check_launch("pd_sup_loss_fwd");
}

extern "C" int pd_normals_loss_masked(const void* pred, const void* gt, const void* K, const void* mask, void* partial_ws,
                                      void* out, int N, int H, int W, void* stream) {
    PD_REQUIRE(pred && gt && K && mask && partial_ws && out && N > 0 && H > 0 && W > 0, "pd_normals_loss_masked: bad arguments");
    const long total = (long)N * H * W;
    const int rows = (int)lgrid(total);
    hipLaunchKernelGGL(normals_loss_masked_kernel, dim3(rows), dim3(LT), 0, (hipStream_t)stream, (const float*)pred,
                       (const float*)gt, (const float*)K, (const float*)mask, (float*)partial_ws, N, H, W);
    hipLaunchKernelGGL(ratio_of_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial_ws, rows,
                       (float*)out);
    return pd::check_launch("pd_normals_loss_masked");
}

extern "C" int pd_normals_pred_loss_fwd(const void* pred, long ld, const void* gt_normals, const void* gt, void* partial_ws,
                                        void* out, int N, int H, int W, float min_depth, float max_depth, void* stream) {
    PD_REQUIRE(pred && gt_normals && gt && partial_ws && out && N > 0 && H > 0 && W > 0 && ld >= 3 && pd::aligned16(gt_normals),
               "pd_normals_pred_loss_fwd: bad arguments");
    const long total = (long)N * H * W;
    const int rows = (int)lgrid(total);
    hipLaunchKernelGGL(normals_pred_fwd_kernel, dim3(rows), dim3(LT), 0, (hipStream_t)stream, (const float*)pred, ld,
                       (const float4*)gt_normals, (const float*)gt, (float*)partial_ws, total, min_depth, max_depth);
    hipLaunchKernelGGL(ratio_and_count_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial_ws, rows,
                       (float*)out);
    return pd::check_launch("pd_normals_pred_loss_fwd");
}

extern "C" int pd_normals_pred_loss_bwd(const void* pred, long ld, const void* gt_normals, const void* gt, const void* gout,
                                        const void* loss_count, void* dpred, long ld_dpred, int N, int H, int W, float min_depth,
                                        float max_depth, void* stream) {
    PD_REQUIRE(pred && gt_normals && gt && gout && loss_count && dpred && N > 0 && H > 0 && W > 0 && ld >= 3 && ld_dpred >= 3 &&
                   pd::aligned16(gt_normals), "pd_normals_pred_loss_bwd: bad arguments");
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(normals_pred_bwd_kernel, dim3(lgrid(total)), dim3(LT), 0, (hipStream_t)stream, (const float*)pred, ld,
                       (const float4*)gt_normals, (const float*)gt, (const float*)gout, (const float*)loss_count, (float*)dpred,
                       ld_dpred, total, min_depth, max_depth);
    return pd::check_launch("pd_normals_pred_loss_bwd");
}

extern "C" int pd_sup_loss_bwd(const void* pred, const void* gt, const void* K, const void* gt_normals, const void* wts,
                               const void* sums, void* ab_ws, void* gout, int N, int H, int W, float min_depth, float max_depth,
                               int with_normals, int to_disp, int two_pass_form, void* stream) {
    PD_REQUIRE(pred && gt && wts && sums && gout && N > 0 && H > 0 && W > 0 && pd::aligned16(gt_normals),
               "pd_sup_loss_bwd: bad arguments");
    PD_REQUIRE(!with_normals || K, "pd_sup_loss_bwd: the normals term needs K");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = lgrid((long)N * H * W);
    const bool two_pass = two_pass_form != 0;                          // the form with the [N,H,W,6] intermediate (tests compare the two)
    if (with_normals && !two_pass) {
        const int tiles_h = (H + SB_TR - 1) / SB_TR, tiles_w = (W + SB_TW - 1) / SB_TW;
        const long ntiles = (long)N * tiles_h * tiles_w;
        PD_REQUIRE(ntiles < (1L << 31), "pd_sup_loss_bwd: too many tiles");
        hipLaunchKernelGGL(sup_bwd_fused_kernel, dim3((unsigned)(ntiles > 4096 ? 4096 : ntiles)), dim3(LT), 0, st,
                           (const float*)pred, (const float*)gt, (const float*)K, (const float4*)gt_normals, (const float*)wts,
                           (const double*)sums, (float*)gout, N, H, W, min_depth, max_depth, 1.f / min_depth - 1.f / max_depth,
                           to_disp, tiles_h, tiles_w, (int)ntiles);
        return pd::check_launch("pd_sup_loss_bwd");
    }
    PD_REQUIRE(!with_normals || ab_ws, "pd_sup_loss_bwd: the two-pass form needs the [N,H,W,6] workspace");
    if (with_normals)
        hipLaunchKernelGGL(sup_bwd_a_kernel, dim3(grid), dim3(LT), 0, st, (const float*)pred, (const float*)gt,
                           (const float*)K, (const float4*)gt_normals, (const float*)wts, (const double*)sums, (float*)ab_ws,
                           N, H, W, min_depth, max_depth);
    hipLaunchKernelGGL(sup_bwd_b_kernel, dim3(grid), dim3(LT), 0, st, (const float*)pred, (const float*)gt,
                       (const float*)K, (const float*)wts, (const double*)sums, (const float*)ab_ws, (float*)gout, N, H,
                       W, min_depth, max_depth, 1.f / min_depth - 1.f / max_depth, with_normals, to_disp);
    return pd::check_launch("pd_sup_loss_bwd");
}

extern "C" int pd_smooth_fwd(const void* disp, const void* img, void* mean, void* partial, void* edge_w, int N, int h, int w,
                             void* stream) {
    PD_REQUIRE(disp && img && partial && N > 0 && h > 1 && w > 1, "pd_smooth_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (mean) hipLaunchKernelGGL(image_mean_kernel, dim3(N), dim3(1024), 0, st, (const float*)disp, (float*)mean, h * w);
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3(lgrid((long)N * h * w)), dim3(LT), 0, st, (const float*)disp,
                       (const float*)img, (const float*)mean, (float*)partial, (float2*)edge_w, N, h, w);
    return pd::check_launch("pd_smooth_fwd");
}

extern "C" int pd_smooth_bwd(const void* disp, const void* img, const void* mean, const void* wts, const void* edge_w,
                             void* g_ws, void* gd_acc, void* gdisp, int N, int h, int w, int accumulate, void* stream) {
    PD_REQUIRE(disp && img && wts && g_ws && gd_acc && gdisp && N > 0 && h > 1 && w > 1, "pd_smooth_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(gd_acc, 0, sizeof(double) * N, st) != hipSuccess) return pd::fail(PD_ELAUNCH, "pd_smooth_bwd: memset");
    const unsigned grid = lgrid((long)N * h * w);
    hipLaunchKernelGGL(smooth_bwd_g_kernel, dim3(grid), dim3(LT), 0, st, (const float*)disp, (const float*)img,
                       (const float*)mean, (const float*)wts, (const float2*)edge_w, (float*)g_ws, (double*)gd_acc, N, h, w);
    hipLaunchKernelGGL(smooth_bwd_final_kernel, dim3(grid), dim3(LT), 0, st, (const float*)g_ws, (const float*)mean,
                       (const double*)gd_acc, (float*)gdisp, N, (long)h * w, accumulate);
    return pd::check_launch("pd_smooth_bwd");
}

// ---- all scales per launch (trainer.py:1134-1265 loop over scales)
static int ms_fill(MsArgs& a, const void* const* disps, const void* const* colors, void* const* depths, void* const* means,
                   void* const* edge_ws, const int* hs, const int* ws, int S, int N, int H, int W) {
    PD_REQUIRE(S > 0 && S <= 8 && N > 0 && H > 0 && W > 0 && disps && depths && hs && ws, "pd_multiscale_loss: bad arguments");
    a = MsArgs{};
    a.S = S; a.N = N; a.H = H; a.W = W;
    a.nb_full = lgrid((long)N * H * W);
    for (int s = 0; s < S; ++s) {
        PD_REQUIRE(disps[s] && depths[s] && hs[s] > 1 && ws[s] > 1 && H % hs[s] == 0 && W % ws[s] == 0 && H / hs[s] == W / ws[s] &&
                       (H / hs[s] == 1 || H / hs[s] == 2 || H / hs[s] == 4 || H / hs[s] == 8),
                   "pd_multiscale_loss: scale %d: the full size must be 1, 2, 4 or 8 times the scale size", s);
        a.disp[s] = (const float*)disps[s]; a.color[s] = colors ? (const float*)colors[s] : nullptr;
        a.depth[s] = (float*)depths[s]; a.mean[s] = means ? (float*)means[s] : nullptr;
        a.edge[s] = edge_ws ? (float2*)edge_ws[s] : nullptr;
        a.hs[s] = hs[s]; a.ws[s] = ws[s];
        a.nb_s[s] = lgrid((long)N * hs[s] * ws[s]);
    }
    return PD_OK;
}

extern "C" int pd_multiscale_loss_fwd(const void* const* disps, const void* const* colors, const int* hs, const int* ws, int S,
                                      const void* gt, const void* K, const void* gt_normals, void* const* depths,
                                      void* const* means, void* const* edge_ws, void* sup_part, void* sm_part, int part_stride,
                                      int N, int H, int W, float min_depth, float max_depth, int with_normals, void* stream) {
    MsArgs a;
    int rc = ms_fill(a, disps, colors, depths, means, edge_ws, hs, ws, S, N, H, W);
    if (rc) return rc;
    PD_REQUIRE(colors && means && gt && (K || !with_normals) && sup_part && sm_part, "pd_multiscale_loss_fwd: null pointer");
    PD_REQUIRE(min_depth > 0 && max_depth > min_depth, "pd_multiscale_loss_fwd: bad depth range");
    unsigned nb_max = 0;
    for (int s = 0; s < S; ++s) { PD_REQUIRE(colors[s] && means[s], "pd_multiscale_loss_fwd: null scale tensor"); nb_max = std::max(nb_max, a.nb_s[s]); }
    PD_REQUIRE((int)a.nb_full <= part_stride && (int)nb_max <= part_stride, "pd_multiscale_loss_fwd: part_stride too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ms_disp_to_depth_kernel, dim3(a.nb_full, S), dim3(LT), 0, st, a, 1.f / max_depth, 1.f / min_depth);
    hipLaunchKernelGGL(ms_sup_fwd_kernel, dim3(a.nb_full), dim3(LT), 0, st, a, (const float*)gt, (const float*)K,
                       (const float4*)gt_normals, (float*)sup_part, part_stride, min_depth, max_depth, with_normals);
    hipLaunchKernelGGL(ms_image_mean_kernel, dim3(N, S), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(ms_smooth_fwd_kernel, dim3(nb_max, S), dim3(LT), 0, st, a, (float*)sm_part, part_stride);
    return pd::check_launch("pd_multiscale_loss_fwd");
}

extern "C" int pd_multiscale_loss_bwd(const void* const* disps, const void* const* colors, const void* const* depths,
                                      const void* const* means, const void* const* edge_ws, const int* hs, const int* ws, int S,
                                      const void* gt, const void* K, const void* gt_normals, const void* wts, const void* sums,
                                      void* gup_ws, void* g_ws, void* gd_acc, void* const* gdisps, int N, int H, int W,
                                      float min_depth, float max_depth, void* stream) {
    MsArgs a;
    int rc = ms_fill(a, disps, colors, const_cast<void* const*>(depths), const_cast<void* const*>(means),
                     const_cast<void* const*>(edge_ws), hs, ws, S, N, H, W);
    if (rc) return rc;
    PD_REQUIRE(colors && means && gt && K && wts && sums && gup_ws && g_ws && gd_acc && gdisps, "pd_multiscale_loss_bwd: null pointer");
    unsigned nb_max = 0;
    long goff = 0;
    for (int s = 0; s < S; ++s) {
        PD_REQUIRE(colors[s] && means[s] && gdisps[s], "pd_multiscale_loss_bwd: null scale tensor");
        a.gdisp[s] = (float*)gdisps[s];
        a.gup[s] = (float*)gup_ws + (long)s * N * H * W;
        a.gws[s] = (float*)g_ws + goff;
        goff += (long)N * hs[s] * ws[s];
        nb_max = std::max(nb_max, a.nb_s[s]);
    }
    hipStream_t st = (hipStream_t)stream;
    const int tiles_h = (H + SB_TR - 1) / SB_TR, tiles_w = (W + SB_TW - 1) / SB_TW;
    const long ntiles = (long)N * tiles_h * tiles_w;
    PD_REQUIRE(ntiles < (1L << 31), "pd_multiscale_loss_bwd: too many tiles");
    hipLaunchKernelGGL(ms_sup_bwd_kernel, dim3((unsigned)(ntiles > 4096 ? 4096 : ntiles), S), dim3(LT), 0, st, a, (const float*)gt,
                       (const float*)K, (const float4*)gt_normals, (const float*)wts, (const double*)sums, min_depth, max_depth,
                       1.f / min_depth - 1.f / max_depth, tiles_h, tiles_w, (int)ntiles);
    hipLaunchKernelGGL(ms_up_gather_bwd_kernel, dim3(nb_max, S), dim3(LT), 0, st, a);
    if (hipMemsetAsync(gd_acc, 0, sizeof(double) * N * S, st) != hipSuccess) return pd::fail(PD_ELAUNCH, "pd_multiscale_loss_bwd: memset");
    hipLaunchKernelGGL(ms_smooth_bwd_g_kernel, dim3(nb_max, S), dim3(LT), 0, st, a, (const float*)wts, (double*)gd_acc);
    hipLaunchKernelGGL(ms_smooth_bwd_final_kernel, dim3(nb_max, S), dim3(LT), 0, st, a, (const double*)gd_acc);
    return pd::check_launch("pd_multiscale_loss_bwd");
}

extern "C" int pd_loss_finalize(const void* sup_part, const int* sup_rows, const void* sm_part, const int* sm_rows,
                                const int* dims /* per scale N,h,w */, const int* scale_ids, int S, int part_stride,
                                float w_normals, float w_smooth, void* sums, void* vals, void* stream) {
    PD_REQUIRE(sup_part && sup_rows && sm_part && sm_rows && dims && scale_ids && sums && vals && S > 0 && S <= 8,
               "pd_loss_finalize: bad arguments");
    LossMeta m{};
    m.S = S;
    for (int s = 0; s < S; ++s) {
        m.sup_rows[s] = sup_rows[s]; m.sm_rows[s] = sm_rows[s];
        m.N[s] = dims[3 * s]; m.h[s] = dims[3 * s + 1]; m.w[s] = dims[3 * s + 2]; m.scale_id[s] = scale_ids[s];
        PD_REQUIRE(sup_rows[s] <= part_stride && sm_rows[s] <= part_stride, "pd_loss_finalize: rows exceed part_stride");
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)sup_part,
                       (const float*)sm_part, m, part_stride, w_normals, w_smooth, (double*)sums, (float*)vals);
    return pd::check_launch("pd_loss_finalize");
}

extern "C" int pd_loss_from_sums(const void* sums, const int* dims, const int* scale_ids, int S, float w_normals,
                                 float w_smooth, void* vals, void* stream) {
    PD_REQUIRE(sums && dims && scale_ids && vals && S > 0 && S <= 8, "pd_loss_from_sums: bad arguments");
    LossMeta m{};
    m.S = S;
    for (int s = 0; s < S; ++s) {
        m.N[s] = dims[3 * s]; m.h[s] = dims[3 * s + 1]; m.w[s] = dims[3 * s + 2]; m.scale_id[s] = scale_ids[s];
    }
    hipLaunchKernelGGL(loss_from_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)sums, m,
                       w_normals, w_smooth, (float*)vals);
    return pd::check_launch("pd_loss_from_sums");
}

extern "C" int pd_loss_weights(const void* gvals, const int* scale_ids, int S, float w_normals, float w_smooth,
                               void* wts, void* stream) {
    PD_REQUIRE(gvals && scale_ids && wts && S > 0 && S <= 8, "pd_loss_weights: bad arguments");
    LossMeta m{};
    m.S = S;
    for (int s = 0; s < S; ++s) m.scale_id[s] = scale_ids[s];
    hipLaunchKernelGGL(loss_weights_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)gvals, m,
                       w_normals, w_smooth, (float*)wts);
    return pd::check_launch("pd_loss_weights");
}

// ============================================================================ SSIM + depth metrics
// SSIM (layers.py:468-499: ReflectionPad2d(1), 3x3 average pools, C1 = 1e-4, C2 = 9e-4, clamp((1-n/d)/2, 0, 1))
// and the photometric mix of trainer.py:1069-1081 (0.85 * mean_c SSIM + 0.15 * mean_c |target - pred|).
// Constructed by the reference trainer but inactive under --depth_supervision_only; exposed because
// north_star names it.  Planar NCHW fp32, one thread per pixel, all channels.
namespace {

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// mode 0: out[N,C,H,W] = SSIM map;  mode 1: out[N,1,H,W] = reprojection loss (no_ssim -> plain L1 mean)
__global__ __launch_bounds__(LT) void ssim_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                  float* __restrict__ out, int N, int C, int H, int W, int mode,
                                                  int no_ssim) {
    const long P = (long)H * W, total = N * P;
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int px, py;
        const long t = divmod(i, W, px);
        const long n = divmod(t, H, py);
        float s_ssim = 0.f, s_l1 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* xp = x + (n * C + c) * P;
            const float* yp = y + (n * C + c) * P;
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = reflect1(py + dy, H);
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = reflect1(px + dx, W);
                    const float a = xp[(long)yy * W + xx], b = yp[(long)yy * W + xx];
                    sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
                }
            }
            const float k = 1.f / 9.f;
            const float mu_x = sx * k, mu_y = sy * k;
            const float sig_x = sxx * k - mu_x * mu_x, sig_y = syy * k - mu_y * mu_y, sig_xy = sxy * k - mu_x * mu_y;
            const float nn = (2.f * mu_x * mu_y + 1e-4f) * (2.f * sig_xy + 9e-4f);
            const float dd = (mu_x * mu_x + mu_y * mu_y + 1e-4f) * (sig_x + sig_y + 9e-4f);
            const float v = fminf(fmaxf((1.f - nn / dd) * 0.5f, 0.f), 1.f);
            if (mode == 0) out[(n * C + c) * P + (long)py * W + px] = v;
            s_ssim += v;
            s_l1 += fabsf(yp[(long)py * W + px] - xp[(long)py * W + px]);
        }
        if (mode == 1) {
            const float l1 = s_l1 / C;
            out[i] = no_ssim ? l1 : 0.85f * (s_ssim / C) + 0.15f * l1;
        }
    }
}

// ---- SSIM backward (layers.py:468-499 differentiated; trainer.py:1069-1081 for the photometric mix)
// Per pixel q and channel: f = clamp((1 - n/d) / 2, 0, 1), n = (2 mx my + C1)(2 sxy + C2), d = (mx^2 + my^2 + C1)(sx + sy + C2),
// with the window means Sx, Sy, Sxx, Syy, Sxy (mx = Sx, sx = Sxx - Sx^2, sxy = Sxy - Sx Sy).  Pass A writes the five
// coefficients G_S(q) = g_q [0 < f < 1] df/dS of every pixel; pass B gathers, for a source pixel p,
//   dL/dx_p = 1/9 sum over windows q and taps t of q that land on p (reflection!) of G_Sx(q) + 2 x_p G_Sxx(q) + y_p G_Sxy(q)
// and the same with x <-> y.  mode 1: g_q = 0.85 / C * gout[n, 0, q] for every channel, plus the L1 term 0.15 / C sign(x - y).
__global__ __launch_bounds__(LT) void ssim_bwd_a_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ gout, float* __restrict__ coef, int N, int C,
                                                        int H, int W, int mode) {
    const long P = (long)H * W, total = (long)N * C * P;
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int px, py;
        long t = divmod(i, W, px);
        t = divmod(t, H, py);
        const long n = t / C;
        const float* xp = x + t * P;
        const float* yp = y + t * P;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = reflect1(py + dy, H);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = reflect1(px + dx, W);
                const float a = xp[(long)yy * W + xx], b = yp[(long)yy * W + xx];
                sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
            }
        }
        const float k = 1.f / 9.f;
        const float mx = sx * k, my = sy * k;
        const float vx = sxx * k - mx * mx, vy = syy * k - my * my, vxy = sxy * k - mx * my;
        const float a = 2.f * mx * my + 1e-4f, b = 2.f * vxy + 9e-4f, c = mx * mx + my * my + 1e-4f, e = vx + vy + 9e-4f;
        const float nn = a * b, dd = c * e;
        const float f = (1.f - nn / dd) * 0.5f;
        float g = mode == 0 ? gout[i] : gout[n * P + (long)py * W + px] * (0.85f / C);
        if (!(f > 0.f && f < 1.f)) g = 0.f;                       // clamp
        // df/dS = -(dn d - n dd') / (2 d^2)
        const float s = -0.5f * g / (dd * dd);
        const float dn_sx = 2.f * my * (b - a), dd_sx = 2.f * mx * (e - c);
        const float dn_sy = 2.f * mx * (b - a), dd_sy = 2.f * my * (e - c);
        float* o = coef + i;
        o[0] = s * (dn_sx * dd - nn * dd_sx);                     // G_Sx
        o[total] = s * (dn_sy * dd - nn * dd_sy);                 // G_Sy
        o[2 * total] = s * (-nn * c);                             // G_Sxx  (dd/dSxx = c, dn/dSxx = 0)
        o[3 * total] = s * (-nn * c);                             // G_Syy
        o[4 * total] = s * (2.f * a * dd);                        // G_Sxy  (dn/dSxy = 2a, dd/dSxy = 0)
    }
}

__global__ __launch_bounds__(LT) void ssim_bwd_b_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ gout, const float* __restrict__ coef,
                                                        float* __restrict__ gx, float* __restrict__ gy, int N, int C, int H,
                                                        int W, int mode, int no_ssim) {
    const long P = (long)H * W, total = (long)N * C * P;
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < total; i += (long)gridDim.x * LT) {
        int px, py;
        long t = divmod(i, W, px);
        t = divmod(t, H, py);
        const long n = t / C;
        const float xv = x[i], yv = y[i];
        float ax = 0.f, ay = 0.f;
        if (!no_ssim) {
            const float* cf = coef + t * P;
            for (int qy = max(py - 2, 0); qy <= min(py + 2, H - 1); ++qy)
                for (int qx = max(px - 2, 0); qx <= min(px + 2, W - 1); ++qx) {
                    int mult = 0;                             // taps of window q that land on p after reflection
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx)
                            mult += (reflect1(qy + dy, H) == py && reflect1(qx + dx, W) == px) ? 1 : 0;
                    if (!mult) continue;
                    const long q = (long)qy * W + qx;
                    const float m = (float)mult;
                    ax += m * (cf[q] + 2.f * xv * cf[2 * total + q] + yv * cf[4 * total + q]);
                    ay += m * (cf[total + q] + 2.f * yv * cf[3 * total + q] + xv * cf[4 * total + q]);
                }
            ax *= 1.f / 9.f; ay *= 1.f / 9.f;
        }
        if (mode == 1) {
            const float g = gout[n * P + (long)py * W + px] * ((no_ssim ? 1.f : 0.15f) / C);
            const float sg = xv > yv ? 1.f : (xv < yv ? -1.f : 0.f);   // d|y - x| / dx = sign(x - y)
            ax += g * sg; ay -= g * sg;
        }
        gx[i] = ax;
        if (gy) gy[i] = ay;
    }
}

// compute_depth_errors (layers.py:539-557) over the pixels selected by lo < gt < hi [and mask == mask_value],
// per image: partial[block][9] = (count, sum|d|/gt, sum d^2/gt, sum d^2, sum dlog^2, a1, a2, a3, 0); the
// prediction is clamped to [lo, hi] first like trainer.py:1422-1423.  One image per blockIdx.y.
__global__ __launch_bounds__(LT) void depth_metrics_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                           const int* __restrict__ mask, int mask_value,
                                                           float* __restrict__ partial, long P, float lo, float hi) {
    __shared__ float sm[4 * 8];
    const long n = blockIdx.y;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long i = blockIdx.x * (long)LT + threadIdx.x; i < P; i += (long)gridDim.x * LT) {
        const float g = gt[n * P + i];
        if (!(g > lo && g < hi)) continue;
        if (mask && mask[n * P + i] != mask_value) continue;
        const float p = fminf(fmaxf(pred[n * P + i], lo), hi);
        const float d = g - p, r = fmaxf(g / p, p / g), dl = logf(g) - logf(p);
        acc[0] += 1.f; acc[1] += fabsf(d) / g; acc[2] += d * d / g; acc[3] += d * d; acc[4] += dl * dl;
        acc[5] += r < 1.25f ? 1.f : 0.f; acc[6] += r < 1.5625f ? 1.f : 0.f; acc[7] += r < 1.953125f ? 1.f : 0.f;
    }
    block_sum<8>(acc, sm);
    if (threadIdx.x == 0) {
        float* o = partial + (n * gridDim.x + blockIdx.x) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = acc[k];
    }
}

// metrics[n][8] = (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, count) from the per-block partials
__global__ void depth_metrics_finalize_kernel(const float* __restrict__ partial, float* __restrict__ metrics,
                                              int blocks) {
    const int n = blockIdx.x;
    if (threadIdx.x != 0) return;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b)
        for (int k = 0; k < 8; ++k) a[k] += partial[((long)n * blocks + b) * 8 + k];
    float* o = metrics + n * 8;
    const double c = a[0];
    o[0] = (float)(a[1] / c); o[1] = (float)(a[2] / c); o[2] = (float)sqrt(a[3] / c); o[3] = (float)sqrt(a[4] / c);
    o[4] = (float)(a[5] / c); o[5] = (float)(a[6] / c); o[6] = (float)(a[7] / c); o[7] = (float)c;
}

}  // namespace

extern "C" int pd_ssim_fwd(const void* x, const void* y, void* out, int N, int C, int H, int W, int mode, int no_ssim,
                           void* stream) {
    PD_REQUIRE(x && y && out && N >= 0 && C > 0 && H >= 2 && W >= 2 && (mode == 0 || mode == 1), "pd_ssim_fwd: bad arguments");
    if (N == 0) return PD_OK;
    hipLaunchKernelGGL(ssim_kernel, dim3(lgrid((long)N * H * W)), dim3(LT), 0, (hipStream_t)stream, (const float*)x,
                       (const float*)y, (float*)out, N, C, H, W, mode, no_ssim);
    return pd::check_launch("pd_ssim_fwd");
}

extern "C" int pd_ssim_bwd(const void* x, const void* y, const void* gout, void* coef_ws, void* gx, void* gy, int N, int C, int H,
                           int W, int mode, int no_ssim, void* stream) {
    PD_REQUIRE(x && y && gout && coef_ws && gx && N > 0 && C > 0 && H > 1 && W > 1, "pd_ssim_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = lgrid((long)N * C * H * W);
    if (!no_ssim)
        hipLaunchKernelGGL(ssim_bwd_a_kernel, dim3(grid), dim3(LT), 0, st, (const float*)x, (const float*)y, (const float*)gout,
                           (float*)coef_ws, N, C, H, W, mode);
    hipLaunchKernelGGL(ssim_bwd_b_kernel, dim3(grid), dim3(LT), 0, st, (const float*)x, (const float*)y, (const float*)gout,
                       (const float*)coef_ws, (float*)gx, (float*)gy, N, C, H, W, mode, no_ssim);
    return pd::check_launch("pd_ssim_bwd");
}

extern "C" int pd_depth_metrics(const void* gt, const void* pred, const void* mask, int mask_value, void* partial_ws,
                                void* metrics, int N, long P, float min_depth, float max_depth, void* stream) {
    PD_REQUIRE(gt && pred && partial_ws && metrics && N > 0 && P > 0, "pd_depth_metrics: bad arguments");
    const int blocks = 64;   // partial_ws: N * 64 * 8 floats
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(depth_metrics_kernel, dim3(blocks, N), dim3(LT), 0, st, (const float*)gt, (const float*)pred,
                       (const int*)mask, mask_value, (float*)partial_ws, P, min_depth, max_depth);
    hipLaunchKernelGGL(depth_metrics_finalize_kernel, dim3(N), dim3(64), 0, st, (const float*)partial_ws, (float*)metrics,
                       blocks);
    return pd::check_launch("pd_depth_metrics");
}
