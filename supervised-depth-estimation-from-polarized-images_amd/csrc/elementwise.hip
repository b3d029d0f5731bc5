// K3 -- fused memory-bound kernels around the convolutions (all NHWC fp32, 16 bytes per lane):
//   BatchNorm statistics finalisation (fp64), BN-apply + ReLU + 2x2 max-pool + dropout + residual
//   (+ReLU) forward, and the two-pass backward of the same chain;
//   3x3/s2 max-pool (ResNet stem), bilinear x2 upsample + skip concat (decoder), activation
//   derivative, reflection-padding fold, fused Adam.
// Reference ops replaced: pre_encoders.py:27-34,43-46 (ConvBlock / ResidualBlock tail),
// torchvision BasicBlock bn/relu/add, resnet maxpool (resnet_encoder.py:813-818),
// layers.py:446-449 upsample + depth_decoder.py:64-67 cat, trainer.py:238-240,442 Adam.
#include "pd_common.h"
#include <cstdlib>
#include <cstdint>

namespace {

constexpr int EW_T = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// write-once streams (activations / gradients larger than the caches, optimizer state): nontemporal stores
typedef float ew_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
    const ew_f4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<ew_f4*>(p));
}
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }

// ---------------------------------------------------------------- Philox4x32-10 (dropout masks)
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0; k.y += W1;
    }
    return c;
}

// keep-mask scaled by 1/(1-p) for the 4 channels of output element quad `q`
// Philox offset of a dropout site: the call's own index (kernel argument) + 4096 x the training-step counter kept in
// device memory, so that a captured hipGraph of the step draws fresh masks on every replay with frozen arguments
__device__ __forceinline__ uint64_t site_offset(uint64_t offset, const long long* step_state) {
    return step_state ? offset + ((uint64_t)step_state[0] << 12) : offset;
}

__device__ __forceinline__ float4 dropout_scale(long q, uint64_t seed, uint64_t offset, float p) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)offset,
                                             (uint32_t)(offset >> 32)),
                                  make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967040.f);
    const float s = 1.f / (1.f - p);
    return make_float4(r.x >= thr ? s : 0.f, r.y >= thr ? s : 0.f, r.z >= thr ? s : 0.f, r.w >= thr ? s : 0.f);
}

// ---------------------------------------------------------------- BatchNorm statistics: column sums + finalisation
// partial [R][C][2] fp32 -> acc [C][2] fp64 (a handful of fp64 atomics per workgroup), then the workgroup that
// arrives LAST (agent-scope ticket, release before / acquire after: cdna_hip_programming.md Guideline 16) turns the
// sums into the layer's coefficients and leaves acc and the ticket zero for the next call.  One launch per
// BatchNorm and direction instead of two (176 -> 88 per step).
__device__ __forceinline__ void colsum_block(const float* __restrict__ part, double* __restrict__ acc, long R, int C,
                                             int rows_per_block) {
    // thread -> (channel pair slot, row lane): 32 channels x 8 row lanes
    const int cgroups = (C + 31) / 32;
    const int cg = blockIdx.x % cgroups;
    const long rb = (long)(blockIdx.x / cgroups) * rows_per_block;
    const int c = cg * 32 + (threadIdx.x & 31);
    const int rl = threadIdx.x >> 5;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        const long rend = rb + rows_per_block < R ? rb + rows_per_block : R;
        for (long r = rb + rl; r < rend; r += 8) {
            const float2 v = *reinterpret_cast<const float2*>(part + (r * C + c) * 2);
            s0 += v.x; s1 += v.y;
        }
    }
    __shared__ double red[8][32][2];
    red[rl][threadIdx.x & 31][0] = s0; red[rl][threadIdx.x & 31][1] = s1;
    __syncthreads();
    if (threadIdx.x < 32 && c < C) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0 += red[i][threadIdx.x][0]; t1 += red[i][threadIdx.x][1]; }
        atomicAdd(&acc[2 * c], t0);
        atomicAdd(&acc[2 * c + 1], t1);
    }
}

// true in every thread of exactly one workgroup: the one whose ticket is the last.  The ticket word lives behind
// the sums (acc[2C], as an unsigned) and is left zero.
__device__ __forceinline__ bool last_block_arrives(double* acc, int C) {
    __shared__ unsigned ticket_s;
    unsigned* ticket = reinterpret_cast<unsigned*>(acc + 2 * C);
    __syncthreads();                                   // this workgroup's atomics have been issued
    if (threadIdx.x == 0) {
        __threadfence();                               // release: they are performed before the ticket is taken
        ticket_s = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool last = ticket_s == gridDim.x - 1;
    if (last) {
        if (threadIdx.x == 0) { __threadfence(); *ticket = 0u; }     // acquire; reset for the next call
        __syncthreads();
    }
    return last;
}
__device__ __forceinline__ double acc_take(double* p) {             // read a finished sum past the L1 and clear it
    const double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// training-mode statistics -> affine coefficients (+ running stats update, torch semantics)
__device__ __forceinline__ void bn_fwd_finalize_channel(int c, double s, double ss, const float* gamma, const float* beta,
                                                        float* rmean, float* rvar, float* scale, float* shift,
                                                        float* smean, float* sinvstd, double count, float momentum,
                                                        float eps, int training) {
    float mean, invstd;
    if (training) {
        const double m = s / count;
        double var = ss / count - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    } else {
        mean = rmean[c];
        invstd = 1.f / sqrtf(rvar[c] + eps);
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - mean * sc;
    if (smean) { smean[c] = mean; sinvstd[c] = invstd; }
}

__global__ __launch_bounds__(EW_T) void bn_fwd_stats_kernel(const float* __restrict__ part, double* __restrict__ acc, long R,
                                                            int C, int rows_per_block, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, float* __restrict__ scale,
                                                            float* __restrict__ shift, float* __restrict__ smean,
                                                            float* __restrict__ sinvstd, double count, float momentum,
                                                            float eps) {
    colsum_block(part, acc, R, C, rows_per_block);
    if (!last_block_arrives(acc, C)) return;
    for (int c = threadIdx.x; c < C; c += EW_T) {
        const double s = acc_take(acc + 2 * c), ss = acc_take(acc + 2 * c + 1);
        bn_fwd_finalize_channel(c, s, ss, gamma, beta, rmean, rvar, scale, shift, smean, sinvstd, count, momentum, eps, 1);
    }
}

// eval mode (running statistics): no sums
__global__ void bn_fwd_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                                   float* __restrict__ rvar, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ smean, float* __restrict__ sinvstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) bn_fwd_finalize_channel(c, 0.0, 0.0, gamma, beta, rmean, rvar, scale, shift, smean, sinvstd, 1.0, 0.f, eps, 0);
}

// backward: acc = (sum g, sum g*xhat) -> dgamma, dbeta, coef = (mean g, mean g*xhat)
__global__ __launch_bounds__(EW_T) void bn_bwd_stats_kernel(const float* __restrict__ part, double* __restrict__ acc, long R,
                                                            int C, int rows_per_block, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ coef,
                                                            double count, int accumulate) {
    colsum_block(part, acc, R, C, rows_per_block);
    if (!last_block_arrives(acc, C)) return;
    for (int c = threadIdx.x; c < C; c += EW_T) {
        const double sg = acc_take(acc + 2 * c), sgx = acc_take(acc + 2 * c + 1);
        if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sgx;
        if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sg;
        coef[c] = (float)(sg / count);
        coef[C + c] = (float)(sgx / count);
    }
}

// Few partial rows (every layer below 256x320: <= 2560 tiles): ONE workgroup per 32 channels sums all rows -- 32 channels x
// 32 row lanes, fp64, fixed order -- and finalises its channels itself: no atomics, no ticket, no second trip through L2
// (the multi-workgroup version above is a chain of load -> atomic -> fence -> ticket -> atomic load latencies, ~14 us
// whatever the size).
template <bool FWD>
__global__ __launch_bounds__(1024) void bn_stats_small_kernel(const float* __restrict__ part, long R, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ rmean, float* __restrict__ rvar,
                                                              float* __restrict__ scale, float* __restrict__ shift,
                                                              float* __restrict__ smean, float* __restrict__ sinvstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef, double count, float momentum,
                                                              float eps, int accumulate) {
    __shared__ double red[32][32][2];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0, u0 = 0.0, u1 = 0.0, w0 = 0.0, w1 = 0.0;
    if (c < C) {
        long r = rl;
        for (; r + 96 < R; r += 128) {          // four independent loads in flight per thread
            const float2 a = *reinterpret_cast<const float2*>(part + (r * C + c) * 2);
            const float2 b = *reinterpret_cast<const float2*>(part + ((r + 32) * C + c) * 2);
            const float2 d = *reinterpret_cast<const float2*>(part + ((r + 64) * C + c) * 2);
            const float2 e = *reinterpret_cast<const float2*>(part + ((r + 96) * C + c) * 2);
            s0 += a.x; s1 += a.y; t0 += b.x; t1 += b.y; u0 += d.x; u1 += d.y; w0 += e.x; w1 += e.y;
        }
        for (; r < R; r += 32) { const float2 a = *reinterpret_cast<const float2*>(part + (r * C + c) * 2); s0 += a.x; s1 += a.y; }
    }
    red[rl][cl][0] = (s0 + t0) + (u0 + w0); red[rl][cl][1] = (s1 + t1) + (u1 + w1);
    __syncthreads();
    if (rl == 0 && c < C) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) { a0 += red[i][cl][0]; a1 += red[i][cl][1]; }
        if (FWD) {
            bn_fwd_finalize_channel(c, a0, a1, gamma, beta, rmean, rvar, scale, shift, smean, sinvstd, count, momentum, eps, 1);
        } else {
            if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)a1;
            if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a0;
            coef[c] = (float)(a0 / count);
            coef[C + c] = (float)(a1 / count);
        }
    }
}

// ---------------------------------------------------------------- BN-apply chain
struct ChainArgs {
    const float* x;        // raw conv output [N,H,W,C] contiguous
    const float* scale;    // [C] or null (identity)
    const float* shift;
    const float* res;      // residual on the output grid, row stride ld_res, or null
    float* out;            // output grid [N,Ho,Wo,C], row stride ld_out
    long ld_res, ld_out;
    int N, H, W, C;        // input grid
    int relu_pre, pool, relu_post;
    float drop_p;
    uint64_t seed, offset;
    const long long* step_state;   // device step counters (pd_step_tick) or null: offset += step_state[0] << 12
    // backward only
    const float* dy;       // grad of out, row stride ld_dy
    long ld_dy;
    const float* mean;     // saved batch mean / invstd (null in identity mode)
    const float* invstd;
    const float* coef;     // [2][C] (apply pass)
    float* dx;             // [N,H,W,C] contiguous
    float* dres;           // grad of the residual input, contiguous [N,Ho,Wo,C], or null
    float* partial;        // [blocks][C][2]
};

__device__ __forceinline__ float4 affine4(float4 v, float4 s, float4 b) {
    return make_float4(v.x * s.x + b.x, v.y * s.y + b.y, v.z * s.z + b.z, v.w * s.w + b.w);
}
__device__ __forceinline__ float4 relu4(float4 v) {
    return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// 32-bit element indices (the host checks N*H*W*C/4 < 2^31): 64-bit divisions per element cost more than the math
__global__ __launch_bounds__(EW_T) void chain_fwd_kernel(const ChainArgs a) {
    const unsigned cq = a.C >> 2;
    const unsigned Ho = a.pool ? a.H >> 1 : a.H, Wo = a.pool ? a.W >> 1 : a.W;
    const unsigned total = (unsigned)a.N * Ho * Wo * cq;
    // 256 % cq == 0 (checked on the host): a thread keeps its channel quad across the loop, so the per-channel
    // coefficients are loaded once instead of twice per element
    const unsigned i0 = blockIdx.x * (unsigned)EW_T + threadIdx.x;
    const int c4 = (int)(i0 % cq) * 4;
    const float4 s = a.scale ? ld4(a.scale + c4) : f4(1.f);
    const float4 b = a.scale ? ld4(a.shift + c4) : f4(0.f);
    for (unsigned i = i0; i < total; i += gridDim.x * (unsigned)EW_T) {
        const unsigned pix = i / cq;                // output pixel index
        float4 v;
        if (a.pool) {
            const unsigned t = pix / Wo;
            const unsigned wo = pix - t * Wo;
            const unsigned n = t / Ho;
            const unsigned ho = t - n * Ho;
            const float* p = a.x + (((size_t)n * a.H + 2 * ho) * a.W + 2 * wo) * a.C + c4;
            float4 v00 = affine4(ld4(p), s, b), v01 = affine4(ld4(p + a.C), s, b);
            float4 v10 = affine4(ld4(p + (long)a.W * a.C), s, b), v11 = affine4(ld4(p + (long)a.W * a.C + a.C), s, b);
            if (a.relu_pre) { v00 = relu4(v00); v01 = relu4(v01); v10 = relu4(v10); v11 = relu4(v11); }
            v = max4(max4(v00, v01), max4(v10, v11));
        } else {
            v = affine4(ld4(a.x + (size_t)pix * a.C + c4), s, b);
            if (a.relu_pre) v = relu4(v);
        }
        if (a.drop_p > 0.f) v = mul4(v, dropout_scale(i, a.seed, site_offset(a.offset, a.step_state), a.drop_p));
        if (a.res) v = add4(v, ld4(a.res + (size_t)pix * a.ld_res + c4));
        if (a.relu_post) v = relu4(v);
        st4_nt(a.out + (size_t)pix * a.ld_out + c4, v);
    }
}

// gradient w.r.t. the BN output z at input pixel (n,h,w), channels c4..c4+3; also returns xhat.
// Recomputes the forward masks from x (and `out` for the post-add ReLU).
__device__ __forceinline__ float4 chain_grad(const ChainArgs& a, unsigned n, unsigned h, unsigned w, int c4, float4 xv,
                                             float4 s, float4 b, float4* g_post = nullptr) {
    const unsigned Ho = a.pool ? a.H >> 1 : a.H, Wo = a.pool ? a.W >> 1 : a.W;
    const unsigned ho = a.pool ? h >> 1 : h, wo = a.pool ? w >> 1 : w;
    if (a.pool && (ho >= Ho || wo >= Wo)) return f4(0.f);   // odd trailing row/col is dropped by the pool
    const unsigned opix = (n * Ho + ho) * Wo + wo;
    float4 g = ld4(a.dy + (size_t)opix * a.ld_dy + c4);
    if (a.relu_post) {
        const float4 o = ld4(a.out + (size_t)opix * a.ld_out + c4);
        g = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
    }
    if (g_post) *g_post = g;            // gradient behind the post-add ReLU = gradient of the residual input
    if (a.drop_p > 0.f) g = mul4(g, dropout_scale(opix * (a.C >> 2) + (c4 >> 2), a.seed, site_offset(a.offset, a.step_state), a.drop_p));
    float4 z = affine4(xv, s, b);
    if (a.pool) {
        // route to the first maximum of the 2x2 window (scan order), like torch max_pool2d
        const float* p = a.x + (((size_t)n * a.H + 2 * ho) * a.W + 2 * wo) * a.C + c4;
        float4 v[4];
        v[0] = affine4(ld4(p), s, b); v[1] = affine4(ld4(p + a.C), s, b);
        v[2] = affine4(ld4(p + (long)a.W * a.C), s, b); v[3] = affine4(ld4(p + (long)a.W * a.C + a.C), s, b);
        if (a.relu_pre) { v[0] = relu4(v[0]); v[1] = relu4(v[1]); v[2] = relu4(v[2]); v[3] = relu4(v[3]); }
        const int me = ((h & 1) << 1) | (w & 1);
        int ax = 0, ay = 0, az = 0, aw = 0;
        float mx = v[0].x, my = v[0].y, mz = v[0].z, mw = v[0].w;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            if (v[j].x > mx) { mx = v[j].x; ax = j; }
            if (v[j].y > my) { my = v[j].y; ay = j; }
            if (v[j].z > mz) { mz = v[j].z; az = j; }
            if (v[j].w > mw) { mw = v[j].w; aw = j; }
        }
        g = make_float4(ax == me ? g.x : 0.f, ay == me ? g.y : 0.f, az == me ? g.z : 0.f, aw == me ? g.w : 0.f);
    }
    if (a.relu_pre)
        g = make_float4(z.x > 0.f ? g.x : 0.f, z.y > 0.f ? g.y : 0.f, z.z > 0.f ? g.z : 0.f, z.w > 0.f ? g.w : 0.f);
    return g;
}

template <bool APPLY>
__global__ __launch_bounds__(EW_T) void chain_bwd_kernel(const ChainArgs a) {
    const unsigned cq = a.C >> 2;
    const unsigned total = (unsigned)a.N * a.H * a.W * cq;
    float4 sg = f4(0.f), sgx = f4(0.f);
    // 256 % cq == 0 (checked on the host): every thread keeps its channel quad across the loop -- the six per-channel
    // coefficient vectors are loaded once, not once per element
    const unsigned i0 = blockIdx.x * (unsigned)EW_T + threadIdx.x;
    const int c4 = (int)(i0 % cq) * 4;
    const float4 s = a.scale ? ld4(a.scale + c4) : f4(1.f);
    const float4 b = a.scale ? ld4(a.shift + c4) : f4(0.f);
    const float4 m = a.mean ? ld4(a.mean + c4) : f4(0.f), is = a.mean ? ld4(a.invstd + c4) : f4(0.f);
    const float4 c1 = (APPLY && a.mean) ? ld4(a.coef + c4) : f4(0.f), c2 = (APPLY && a.mean) ? ld4(a.coef + a.C + c4) : f4(0.f);
    if (a.pool && !(a.H & 1) && !(a.W & 1)) {
        // pooled layers, even grid: one thread per 2x2 WINDOW -- four loads of z and one of dy serve four input
        // pixels (per input pixel the window would be re-read four times: 2.1 / 3.5 TB/s instead of ~5)
        const unsigned Ho = a.H >> 1, Wo = a.W >> 1;
        const unsigned wtotal = (unsigned)a.N * Ho * Wo * cq;
        for (unsigned i = i0; i < wtotal; i += gridDim.x * (unsigned)EW_T) {
            const unsigned opix = i / cq;
            const unsigned t = opix / Wo;
            const unsigned wo = opix - t * Wo;
            const unsigned n = t / Ho;
            const unsigned ho = t - n * Ho;
            const size_t p00 = (((size_t)n * a.H + 2 * ho) * a.W + 2 * wo) * a.C + c4;
            const size_t offs[4] = {p00, p00 + a.C, p00 + (size_t)a.W * a.C, p00 + (size_t)a.W * a.C + a.C};
            float4 xv[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xv[j] = ld4(a.x + offs[j]);
                v[j] = affine4(xv[j], s, b);
                if (a.relu_pre) v[j] = relu4(v[j]);
            }
            float4 g = ld4(a.dy + (size_t)opix * a.ld_dy + c4);
            if (a.relu_post) {
                const float4 o = ld4(a.out + (size_t)opix * a.ld_out + c4);
                g = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
            }
            if (a.drop_p > 0.f) g = mul4(g, dropout_scale(i, a.seed, site_offset(a.offset, a.step_state), a.drop_p));
            // first maximum of the window in scan order, like torch max_pool2d
            int ax = 0, ay = 0, az = 0, aw = 0;
            float mx = v[0].x, my = v[0].y, mz = v[0].z, mw = v[0].w;
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                if (v[j].x > mx) { mx = v[j].x; ax = j; }
                if (v[j].y > my) { my = v[j].y; ay = j; }
                if (v[j].z > mz) { mz = v[j].z; az = j; }
                if (v[j].w > mw) { mw = v[j].w; aw = j; }
            }
            if (a.relu_pre)      // the pooled value passes the gradient only where the pre-pool ReLU was open
                g = make_float4(mx > 0.f ? g.x : 0.f, my > 0.f ? g.y : 0.f, mz > 0.f ? g.z : 0.f, mw > 0.f ? g.w : 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 gj = make_float4(ax == j ? g.x : 0.f, ay == j ? g.y : 0.f, az == j ? g.z : 0.f, aw == j ? g.w : 0.f);
                float4 xh = f4(0.f);
                if (a.mean) xh = make_float4((xv[j].x - m.x) * is.x, (xv[j].y - m.y) * is.y, (xv[j].z - m.z) * is.z, (xv[j].w - m.w) * is.w);
                if (!APPLY) {
                    sg = add4(sg, gj);
                    sgx = add4(sgx, mul4(gj, xh));
                } else {
                    float4 d;
                    if (a.mean) d = make_float4(s.x * (gj.x - c1.x - xh.x * c2.x), s.y * (gj.y - c1.y - xh.y * c2.y),
                                                s.z * (gj.z - c1.z - xh.z * c2.z), s.w * (gj.w - c1.w - xh.w * c2.w));
                    else d = mul4(gj, s);
                    st4_nt(a.dx + offs[j], d);
                }
            }
        }
    } else
    for (unsigned i = i0; i < total; i += gridDim.x * (unsigned)EW_T) {
        const unsigned pix = i / cq;
        const unsigned t = pix / a.W;
        const unsigned w = pix - t * a.W;
        const unsigned n = t / a.H;
        const unsigned h = t - n * a.H;
        const float4 xv = ld4(a.x + (size_t)pix * a.C + c4);
        float4 gp;
        const float4 g = chain_grad(a, n, h, w, c4, xv, s, b, &gp);
        if (APPLY && a.dres) st4(a.dres + (size_t)pix * a.C + c4, gp);     // (unpooled chains only, see the launcher)
        float4 xh = f4(0.f);
        if (a.mean) xh = make_float4((xv.x - m.x) * is.x, (xv.y - m.y) * is.y, (xv.z - m.z) * is.z, (xv.w - m.w) * is.w);
        if (!APPLY) {
            sg = add4(sg, g);
            sgx = add4(sgx, mul4(g, xh));
        } else {
            float4 d;
            if (a.mean) {
                d = make_float4(s.x * (g.x - c1.x - xh.x * c2.x), s.y * (g.y - c1.y - xh.y * c2.y),
                                s.z * (g.z - c1.z - xh.z * c2.z), s.w * (g.w - c1.w - xh.w * c2.w));
            } else {
                d = mul4(g, s);
            }
            st4_nt(a.dx + (size_t)pix * a.C + c4, d);
        }
    }
    if (!APPLY) {
        __shared__ float red[EW_T][8];
        float* r = red[threadIdx.x];
        r[0] = sg.x; r[1] = sg.y; r[2] = sg.z; r[3] = sg.w; r[4] = sgx.x; r[5] = sgx.y; r[6] = sgx.z; r[7] = sgx.w;
        __syncthreads();
        // threads t, t+cq, t+2cq ... share a channel quad
        if ((int)threadIdx.x < cq) {
            float acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = threadIdx.x; t < EW_T; t += cq)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] += red[t][j];
            float* o = a.partial + ((long)blockIdx.x * a.C + 4 * threadIdx.x) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[2 * j] = acc8[j]; o[2 * j + 1] = acc8[4 + j]; }
        }
    }
}

// i -> (i / d, i % d) for a positive divisor: 32-bit unsigned arithmetic whenever the index fits (every grid-stride index
// of the training step does); a 64-bit division is ~5x the instructions, and the pixel decode below needs three
__device__ __forceinline__ long divmod(long i, int d, int& rem) {
    if (i >> 32) { const long q = i / d; rem = (int)(i - q * d); return q; }
    const unsigned u = (unsigned)i, q = u / (unsigned)d;
    rem = (int)(u - q * (unsigned)d);
    return (long)q;
}

// grad of the residual input of a post-add ReLU block: dres = dy * (out > 0)
__global__ __launch_bounds__(EW_T) void relu_mask_kernel(const float* __restrict__ dy, long ld_dy,
                                                         const float* __restrict__ out, long ld_out,
                                                         float* __restrict__ d, long npix, int C) {
    const int cq = C >> 2;
    const long total = npix * cq;
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < total; i += (long)gridDim.x * EW_T) {
        int c4;
        const long pix = divmod(i, cq, c4); c4 *= 4;
        const float4 g = ld4(dy + pix * ld_dy + c4), o = ld4(out + pix * ld_out + c4);
        st4(d + pix * C + c4, make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f,
                                          o.w > 0.f ? g.w : 0.f));
    }
}

// ---------------------------------------------------------------- 3x3 stride-2 pad-1 max pool
// idx (optional): per output element the window position dh*3+dw of its first maximum in scan order (torch
// max_pool2d_with_indices), one byte per channel -- the backward pass then needs no second look at x.
__global__ __launch_bounds__(EW_T) void maxpool3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            uint32_t* __restrict__ idx, int N, int H, int W, int C, int Ho,
                                                            int Wo) {
    const int cq = C >> 2;
    const long total = (long)N * Ho * Wo * cq;
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < total; i += (long)gridDim.x * EW_T) {
        int c4, wo, ho;
        long pix = divmod(i, cq, c4); c4 *= 4;
        pix = divmod(pix, Wo, wo);
        const long n = divmod(pix, Ho, ho);
        // all nine loads are issued before the first comparison (clamped addresses, taps outside the image are skipped
        // by predicate): behind a branch each load waited for the previous compare -- nine HBM latencies in a row
        float4 v[9];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int h = min(max(2 * ho - 1 + dh, 0), H - 1);
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int w = min(max(2 * wo - 1 + dw, 0), W - 1);
                v[dh * 3 + dw] = ld4(x + ((n * H + h) * W + w) * C + c4);
            }
        }
        float4 m = f4(-INFINITY);
        uint32_t ix = 0, iy = 0, iz = 0, iw = 0;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int h = 2 * ho - 1 + dh;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int w = 2 * wo - 1 + dw;
                const bool in = h >= 0 && h < H && w >= 0 && w < W;
                const float4 t = v[dh * 3 + dw];
                const uint32_t id = (uint32_t)(dh * 3 + dw);
                if (in && t.x > m.x) { m.x = t.x; ix = id; }
                if (in && t.y > m.y) { m.y = t.y; iy = id; }
                if (in && t.z > m.z) { m.z = t.z; iz = id; }
                if (in && t.w > m.w) { m.w = t.w; iw = id; }
            }
        }
        st4(y + i * 4, m);
        if (idx) idx[i] = ix | (iy << 8) | (iz << 16) | (iw << 24);
    }
}

// dx from the saved window positions: an input pixel lies in 1, 2 or 4 windows (ho in {h/2, (h+1)/2}); it receives
// dy of those whose first maximum it is.  One read of idx (4 B) and dy (16 B) per covering window instead of the nine
// taps of x: 0.66 ms -> ~0.1 ms on the ResNet stem.
// add (optional, row stride ld_add): a second gradient of the pooled tensor (the decoder's skip connection reads the same
// feature map), summed here instead of by a separate pass over three 335 MB tensors
__global__ __launch_bounds__(EW_T) void maxpool3_bwd_kernel(const uint32_t* __restrict__ idx, const float* __restrict__ dy,
                                                            const float* __restrict__ add, long ld_add,
                                                            float* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                                            int Wo) {
    const int cq = C >> 2;
    const long total = (long)N * H * W * cq;
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < total; i += (long)gridDim.x * EW_T) {
        int cqi, w, h;
        long pix = divmod(i, cq, cqi);
        pix = divmod(pix, W, w);
        const long n = divmod(pix, H, h);
        // the (up to) four covering windows: ho in {h/2, (h+1)/2}, wo likewise; all loads first (clamped, duplicates and
        // windows beyond the grid are masked out), same summation order as the nested loops
        const int hoa = h >> 1, hob = (h + 1) >> 1, woa = w >> 1, wob = (w + 1) >> 1;
        uint32_t ids[4]; float4 gs[4]; bool use[4]; uint32_t mes[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ho = (k >> 1) ? hob : hoa, wo = (k & 1) ? wob : woa;
            use[k] = ho < Ho && wo < Wo && !((k >> 1) && hob == hoa) && !((k & 1) && wob == woa);
            const int hc = min(ho, Ho - 1), wc = min(wo, Wo - 1);
            const long o = ((n * Ho + hc) * Wo + wc) * cq + cqi;
            ids[k] = idx[o];
            gs[k] = ld4(dy + o * 4);
            mes[k] = (uint32_t)(h - (2 * ho - 1)) * 3u + (uint32_t)(w - (2 * wo - 1));
        }
        float4 acc = f4(0.f);
        float4 extra = f4(0.f);
        if (add) extra = ld4(add + ((n * H + h) * (long)W + w) * ld_add + cqi * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!use[k]) continue;
            const uint32_t id = ids[k], me = mes[k];
            acc.x += (id & 0xffu) == me ? gs[k].x : 0.f;
            acc.y += ((id >> 8) & 0xffu) == me ? gs[k].y : 0.f;
            acc.z += ((id >> 16) & 0xffu) == me ? gs[k].z : 0.f;
            acc.w += (id >> 24) == me ? gs[k].w : 0.f;
        }
        if (add) { acc.x += extra.x; acc.y += extra.y; acc.z += extra.z; acc.w += extra.w; }
        st4(dx + i * 4, acc);
    }
}

// ---------------------------------------------------------------- bilinear x2 upsample (+ concat)
// out[n, y, x, 0:Ca] = bilinear(a)[n, y, x]  (align_corners=False, scale 2), out[..., Ca:Ca+Cs] = skip
__global__ __launch_bounds__(EW_T) void upcat_fwd_kernel(const float* __restrict__ a, const float* __restrict__ skip,
                                                         long ld_skip, float* __restrict__ out, int N, int H, int W,
                                                         int Ca, int Cs) {
    const int Ct = Ca + Cs, cq = Ct >> 2;
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = (long)N * Ho * Wo * cq;
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < total; i += (long)gridDim.x * EW_T) {
        int c4, x, y;
        long pix = divmod(i, cq, c4); c4 *= 4;
        const long opix = pix;
        pix = divmod(pix, Wo, x);
        const long n = divmod(pix, Ho, y);
        float4 v;
        if (c4 < Ca) {
            // torch: src = max(0, 0.5*(dst+0.5)-0.5); i0 = floor(src); i1 = min(i0+1, size-1); l1 = src-i0
            const float sy = fmaxf(0.5f * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (x + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
            const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const float* b = a + n * H * W * Ca + c4;
            const float4 v00 = ld4(b + ((long)y0 * W + x0) * Ca), v01 = ld4(b + ((long)y0 * W + x1) * Ca);
            const float4 v10 = ld4(b + ((long)y1 * W + x0) * Ca), v11 = ld4(b + ((long)y1 * W + x1) * Ca);
            v = make_float4(ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x),
                            ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y),
                            ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z),
                            ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w));
        } else {
            v = ld4(skip + opix * ld_skip + (c4 - Ca));
        }
        st4(out + opix * Ct + c4, v);
    }
}

// ---------------------------------------------------------------- bilinear x2, align_corners = True (DPT)
// nn.functional.interpolate(scale_factor=2, mode="bilinear", align_corners=True) of the DPT fusion blocks and output head
// (reference manydepth/dpt/blocks.py:138-172, 375-377): src = dst * (in - 1) / (out - 1), evaluated in fp32 like torch's
// area_pixel_compute_source_index; i0 = int(src), i1 = i0 + (i0 < in - 1), l1 = src - i0.
__device__ __forceinline__ void ac_tap(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    const float s = scale * (float)o;
    i0 = (int)s; i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0; l0 = 1.f - l1;
}
__global__ __launch_bounds__(EW_T) void up2x_ac_fwd_kernel(const float* __restrict__ a, float* __restrict__ out, int N, int H,
                                                           int W, int C) {
    const int cq = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long total = (long)N * Ho * Wo * cq;
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < total; i += (long)gridDim.x * EW_T) {
        int c4, x, y;
        long pix = divmod(i, cq, c4); c4 *= 4;
        const long opix = pix;
        pix = divmod(pix, Wo, x);
        const long n = divmod(pix, Ho, y);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        ac_tap(y, sh, H, y0, y1, ly0, ly1);
        ac_tap(x, sw, W, x0, x1, lx0, lx1);
        const float* b = a + n * H * W * C + c4;
        const float4 v00 = ld4(b + ((long)y0 * W + x0) * C), v01 = ld4(b + ((long)y0 * W + x1) * C);
        const float4 v10 = ld4(b + ((long)y1 * W + x0) * C), v11 = ld4(b + ((long)y1 * W + x1) * C);
        st4(out + opix * C + c4, make_float4(ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x),
                                             ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y),
                                             ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z),
                                             ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w)));
    }
}
// da[n,h,w,:] = sum over the output pixels whose taps include (h, w), in a fixed order (gather: deterministic).  The rows
// that read input row h lie in [ceil((h-1)/s), floor((h+1)/s)] (s = (H-1)/(2H-1) ~ 1/2): at most five, tested one by one
__global__ __launch_bounds__(EW_T) void up2x_ac_bwd_kernel(const float* __restrict__ dout, float* __restrict__ da, int N, int H,
                                                           int W, int C) {
    const int cq = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long total = (long)N * H * W * cq;
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < total; i += (long)gridDim.x * EW_T) {
        int c4, w, h;
        long pix = divmod(i, cq, c4); c4 *= 4;
        pix = divmod(pix, W, w);
        const long n = divmod(pix, H, h);
        const int oy_lo = max(2 * h - 3, 0), oy_hi = min(2 * h + 3, Ho - 1), ox_lo = max(2 * w - 3, 0), ox_hi = min(2 * w + 3, Wo - 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* d = dout + n * Ho * Wo * C + c4;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float ly0, ly1;
            ac_tap(oy, sh, H, y0, y1, ly0, ly1);
            if (y0 != h && y1 != h) continue;
            const float wy = (y0 == h ? ly0 : 0.f) + (y1 == h ? ly1 : 0.f);      // (last row: y0 == y1 == h, both weights)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float lx0, lx1;
                ac_tap(ox, sw, W, x0, x1, lx0, lx1);
                if (x0 != w && x1 != w) continue;
                const float wx = (x0 == w ? lx0 : 0.f) + (x1 == w ? lx1 : 0.f);
                const float4 g = ld4(d + ((long)oy * Wo + ox) * C);
                const float k = wy * wx;
                acc.x += k * g.x; acc.y += k * g.y; acc.z += k * g.z; acc.w += k * g.w;
            }
        }
        st4(da + (n * H * W + (long)h * W + w) * C + c4, acc);
    }
}

// out = max(x, 0) [+ res]  /  out = x + res: the element-wise glue of the DPT residual units (blocks.py:289-307)
__global__ __launch_bounds__(EW_T) void relu_add_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        float* __restrict__ out, long n4, int relu) {
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < n4; i += (long)gridDim.x * EW_T) {
        float4 v = ld4(x + 4 * i);
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (res) { const float4 r = ld4(res + 4 * i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        st4(out + 4 * i, v);
    }
}

// da[n,h,w,:] = sum over the (up to 4x4) output pixels that read a[h,w], times their bilinear weights
// elu_y != NULL: a = ELU(z) came from a ConvBlock; da is multiplied by ELU'(z) (through a) and leaves as dz
__global__ __launch_bounds__(EW_T) void up_bwd_kernel(const float* __restrict__ dout, long ld_d, const float* __restrict__ elu_y,
                                                      float* __restrict__ da, int N, int H, int W, int Ca) {
    const int cq = Ca >> 2;
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = (long)N * H * W * cq;
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < total; i += (long)gridDim.x * EW_T) {
        int c4, w, h;
        long pix = divmod(i, cq, c4); c4 *= 4;
        pix = divmod(pix, W, w);
        const long n = divmod(pix, H, h);
        // a[h] feeds output rows 2h-1 .. 2h+2 with weights 1/4, 3/4, 3/4, 1/4; at the image border the clamped source row
        // takes the whole weight (rows 0 and 2H-1) and the outer neighbour does not exist.  Same weights, products and
        // summation order as the generic source-index form.
        float wy[4], wx[4];
        wy[0] = h > 0 ? 0.25f : 0.f; wy[1] = h > 0 ? 0.75f : 1.f; wy[2] = h < H - 1 ? 0.75f : 1.f; wy[3] = h < H - 1 ? 0.25f : 0.f;
        wx[0] = w > 0 ? 0.25f : 0.f; wx[1] = w > 0 ? 0.75f : 1.f; wx[2] = w < W - 1 ? 0.75f : 1.f; wx[3] = w < W - 1 ? 0.25f : 0.f;
        float4 gv[16];                                                 // all sixteen loads before the first use
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int y = min(max(2 * h - 1 + a, 0), Ho - 1);          // weight 0 where clamped
            const float* row = dout + ((n * Ho + y) * (long)Wo) * ld_d + c4;
#pragma unroll
            for (int b = 0; b < 4; ++b) gv[a * 4 + b] = ld4(row + (long)min(max(2 * w - 1 + b, 0), Wo - 1) * ld_d);
        }
        float4 acc = f4(0.f);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float ww = wy[a] * wx[b];
                const float4 g = gv[a * 4 + b];
                if (ww != 0.f) { acc.x += ww * g.x; acc.y += ww * g.y; acc.z += ww * g.z; acc.w += ww * g.w; }
            }
        if (elu_y) {
            const float4 o = ld4(elu_y + i * 4);
            acc.x = o.x > 0.f ? acc.x : acc.x * (o.x + 1.f); acc.y = o.y > 0.f ? acc.y : acc.y * (o.y + 1.f);
            acc.z = o.z > 0.f ? acc.z : acc.z * (o.z + 1.f); acc.w = o.w > 0.f ? acc.w : acc.w * (o.w + 1.f);
        }
        st4(da + i * 4, acc);
    }
}

// ---------------------------------------------------------------- activation derivative
// dz = dy * f'(.) expressed through the activation OUTPUT y.  act: 1 ReLU, 2 ELU, 3 sigmoid
__global__ __launch_bounds__(EW_T) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dz, long n, int act) {
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < n; i += (long)gridDim.x * EW_T) {
        const float g = dy[i], o = y[i];
        float d;
        if (act == 1) d = o > 0.f ? g : 0.f;
        else if (act == 2) d = o > 0.f ? g : g * (o + 1.f);
        else d = g * o * (1.f - o);
        dz[i] = d;
    }
}

// ---------------------------------------------------------------- reflection-pad fold
// dxp [N,H+2p,W+2p,C] (gradient on the padded grid, pad p) -> dx [N,H,W,C]; V = channels per thread (4 or 1)
template <int V>
__global__ __launch_bounds__(EW_T) void reflect_fold_kernel(const float* __restrict__ dxp, float* __restrict__ dx,
                                                            int N, int H, int W, int C, int pad) {
    const int cq = C / V;
    const long total = (long)N * H * W * cq;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    for (long i = blockIdx.x * (long)EW_T + threadIdx.x; i < total; i += (long)gridDim.x * EW_T) {
        int c, w, h;
        long pix = divmod(i, cq, c); c *= V;
        pix = divmod(pix, W, w);
        const long n = divmod(pix, H, h);
        // padded rows that map to h: h+p always; p-h for 1 <= h <= p (top border); 2(H-1)-h+p for H-1-p <= h <= H-2
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = h + pad; if (h >= 1 && h <= pad) hs[nh++] = pad - h; if (h >= H - 1 - pad && h <= H - 2) hs[nh++] = 2 * (H - 1) - h + pad;
        ws[nw++] = w + pad; if (w >= 1 && w <= pad) ws[nw++] = pad - w; if (w >= W - 1 - pad && w <= W - 2) ws[nw++] = 2 * (W - 1) - w + pad;
        float4 acc = f4(0.f);
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) {
                const float* p = dxp + ((n * Hp + hs[a]) * Wp + ws[b]) * C + c;
                if (V == 4) { const float4 v = ld4(p); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
                else acc.x += p[0];
            }
        if (V == 4) st4(dx + i * 4, acc); else dx[i] = acc.x;
    }
}

// ---------------------------------------------------------------- reflection-pad border terms of a data gradient
// dX of y = conv3x3(reflection_pad1(x)): the gradient on the padded grid, dxp[p] = sum_k dz[p + 1 - k] w[k], folded back
// (padded row -1 onto row 1, row H onto row H-2, same for columns).  Its INTERIOR is exactly the zero-padding (pad 1)
// data gradient, which the conv kernels write straight into dx; what remains are the four border strips of dxp, each
// a 1x3 / 3x1 slice of the filter applied to the first / last row or column of dz -- added here to rows 1, H-2 and
// columns 1, W-2 of dx: replaces the full-tensor pass of pd_reflect_fold (read the padded gradient, write dx) by a
// kernel that touches 2(H+W) pixels per image.
// dz [N,H,W,Co] (row stride ldd), w [Co][3][3][Ci], dx [N,H,W,Ci] in/out.
// sum_co d[co] * w[co * ws]: eight independent partial sums keep eight (lane-coalesced) weight loads in flight
__device__ __forceinline__ float dot_co(const float* __restrict__ d, const float* __restrict__ wp, int Co, long ws) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int co = 0;
    for (; co + 8 <= Co; co += 8) {
        const float4 d0 = ld4(d + co), d1 = ld4(d + co + 4);
        s[0] += d0.x * wp[(co + 0) * ws]; s[1] += d0.y * wp[(co + 1) * ws]; s[2] += d0.z * wp[(co + 2) * ws]; s[3] += d0.w * wp[(co + 3) * ws];
        s[4] += d1.x * wp[(co + 4) * ws]; s[5] += d1.y * wp[(co + 5) * ws]; s[6] += d1.z * wp[(co + 6) * ws]; s[7] += d1.w * wp[(co + 7) * ws];
    }
    for (; co < Co; ++co) s[0] += d[co] * wp[co * ws];
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// One thread per (target pixel, ci).  Meant for the shallow decoder levels (few channels, thousands of border pixels);
// in the deep ones every target re-streams its slice of a multi-megabyte filter and the fold pass over the small
// tensor is cheaper (functional.ReflectConvActFn picks per layer).
__global__ __launch_bounds__(EW_T) void reflect_dgrad_border_flat_kernel(const float* __restrict__ dz, long ldd,
                                                                         const float* __restrict__ w, float* __restrict__ dx,
                                                                         int N, int H, int W, int Co, int Ci) {
    const int per_img = 2 * W + 2 * H;
    const long total = (long)N * per_img * Ci;
    for (long idx = blockIdx.x * (long)EW_T + threadIdx.x; idx < total; idx += (long)gridDim.x * EW_T) {
        const int ci = (int)(idx % Ci);
        long t = idx / Ci;
        const int e = (int)(t % per_img);
        const long n = t / per_img;
        int i, j;
        if (e < 2 * W) {
            i = e < W ? 1 : H - 2; j = e < W ? e : e - W;
            if (e >= W && H - 2 == 1) continue;
        } else {
            const int f = e - 2 * W;
            j = f < H ? 1 : W - 2; i = f < H ? f : f - H;
            if ((f >= H && W - 2 == 1) || i == 1 || i == H - 2) continue;
        }
        const float* dzn = dz + n * (long)H * W * ldd;
        const float* wb = w + ci;
        auto strip_h = [&](int zr, int kh, int q) {
            float s = 0.f;
            for (int kw = 0; kw < 3; ++kw) {
                const int c = q + 1 - kw;
                if (c < 0 || c >= W) continue;
                s += dot_co(dzn + ((long)zr * W + c) * ldd, wb + ((long)kh * 3 + kw) * Ci, Co, 9L * Ci);
            }
            return s;
        };
        auto strip_v = [&](int zc, int kw, int p) {
            float s = 0.f;
            for (int kh = 0; kh < 3; ++kh) {
                const int r = p + 1 - kh;
                if (r < 0 || r >= H) continue;
                s += dot_co(dzn + ((long)r * W + zc) * ldd, wb + ((long)kh * 3 + kw) * Ci, Co, 9L * Ci);
            }
            return s;
        };
        auto full_h = [&](int zr, int kh) {
            float s = strip_h(zr, kh, j);
            if (j == 1) s += strip_h(zr, kh, -1);
            if (j == W - 2) s += strip_h(zr, kh, W);
            return s;
        };
        float corr = 0.f;
        if (i == 1) corr += full_h(0, 0);
        if (i == H - 2) corr += full_h(H - 1, 2);
        if (j == 1) corr += strip_v(0, 0, i);
        if (j == W - 2) corr += strip_v(W - 1, 2, i);
        dx[((n * H + i) * (long)W + j) * Ci + ci] += corr;
    }
}

// ---------------------------------------------------------------- fused Adam over a flat buffer
// Parameters, moments and (ZERO_G) the cleared gradient leave with nontemporal stores: 340 MB that nothing reads
// before the next step's kernels have streamed gigabytes -- left dirty in the caches, the kernel that follows
// (K1 of the next step) pays for their write-back (measured: 50 -> 87 us, tools/k1_instep_probe.py).

// beta^t by repeated squaring in double: the same IEEE operations on the host (step passed by value) and on the device
// (step read from the step-state words), so that a captured step and an eager step update bit-identically
__host__ __device__ inline double ipow(double b, long t) {
    double r = 1.0;
    while (t > 0) { if (t & 1) r *= b; b *= b; t >>= 1; }
    return r;
}

// step_state = int64[4] in device memory: [0] training steps begun (dropout epoch), [1] optimizer steps (Adam's t),
// [2] the fp32 bit patterns of lr (low word) and grad_scale (high word) for a captured optimizer step
__global__ void step_tick_kernel(long long* __restrict__ st, int bump_dropout, int bump_adam) {
    if (bump_dropout) st[0] += 1;
    if (bump_adam) st[1] += 1;
}
__global__ void step_set_hyper_kernel(long long* __restrict__ st, float lr, float grad_scale) {
    st[2] = (long long)(((unsigned long long)__float_as_uint(grad_scale) << 32) | (unsigned long long)__float_as_uint(lr));
}

template <bool ZERO_G>
__global__ __launch_bounds__(EW_T) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, float grad_scale,
                                                    const long long* __restrict__ step_state) {
    if (step_state) {       // step count, lr and grad_scale in device memory (captured step): nothing of a replayed
        // optimizer step is frozen at capture time -- the bias corrections are those of pd_adam_step's host branch
        const unsigned long long hy = (unsigned long long)step_state[2];
        lr = __uint_as_float((unsigned)hy);
        grad_scale = __uint_as_float((unsigned)(hy >> 32));
        const long t = (long)step_state[1];
        bc1 = (float)(1.0 - ipow((double)beta1, t));
        bc2_sqrt = sqrtf((float)(1.0 - ipow((double)beta2, t)));
    }
    for (long i = (blockIdx.x * (long)EW_T + threadIdx.x) * 4; i < n; i += (long)gridDim.x * EW_T * 4) {
        if (i + 3 < n) {
            float4 pp = ld4(p + i), gg = ld4(g + i), mm = ld4(m + i), vv = ld4(v + i);
            float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float gr = G[j] * grad_scale + wd * P[j];
                M[j] = beta1 * M[j] + (1.f - beta1) * gr;
                V[j] = beta2 * V[j] + (1.f - beta2) * gr * gr;
                // torch.optim.Adam: p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
                P[j] -= (lr / bc1) * (M[j] / (sqrtf(V[j]) / bc2_sqrt + eps));
            }
            st4_nt(p + i, pp); st4_nt(m + i, mm); st4_nt(v + i, vv);
            if (ZERO_G) st4_nt(g + i, make_float4(0.f, 0.f, 0.f, 0.f));
        } else {
            for (long j = i; j < n; ++j) {
                float gr = g[j] * grad_scale + wd * p[j];
                m[j] = beta1 * m[j] + (1.f - beta1) * gr;
                v[j] = beta2 * v[j] + (1.f - beta2) * gr * gr;
                p[j] -= (lr / bc1) * (m[j] / (sqrtf(v[j]) / bc2_sqrt + eps));
                if (ZERO_G) g[j] = 0.f;
            }
        }
    }
}

inline unsigned ew_grid(long work_items) {
    long b = (work_items + EW_T - 1) / EW_T;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

// ============================================================================ C ABI
extern "C" int pd_bn_fwd_finalize(const void* partial, long R, int C, double count, const void* gamma,
                                  const void* beta, void* running_mean, void* running_var, float momentum, float eps,
                                  void* acc_ws, long acc_len, void* scale, void* shift, void* save_mean,
                                  void* save_invstd, int training, void* stream) {
    PD_REQUIRE(C > 0 && scale && shift, "pd_bn_fwd_finalize: bad arguments");
    PD_REQUIRE(!training || acc_len >= 2L * C + 1, "pd_bn_fwd_finalize: accumulator needs 2 C + 1 doubles (%ld given)", acc_len);
    PD_REQUIRE(!training || (partial && acc_ws && R > 0 && count > 0), "pd_bn_fwd_finalize: training needs partials");
    PD_REQUIRE(training || (running_mean && running_var), "pd_bn_fwd_finalize: eval needs running stats");
    hipStream_t st = (hipStream_t)stream;
    if (training && R <= 4096) {
        hipLaunchKernelGGL(bn_stats_small_kernel<true>, dim3((C + 31) / 32), dim3(1024), 0, st, (const float*)partial, R, C,
                           (const float*)gamma, (const float*)beta, (float*)running_mean, (float*)running_var, (float*)scale,
                           (float*)shift, (float*)save_mean, (float*)save_invstd, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr, count, momentum, eps, 0);
    } else if (training) {
        const int cgroups = (C + 31) / 32;
        const int rpb = 256;
        const long rblocks = (R + rpb - 1) / rpb;
        hipLaunchKernelGGL(bn_fwd_stats_kernel, dim3((unsigned)(cgroups * rblocks)), dim3(EW_T), 0, st,
                           (const float*)partial, (double*)acc_ws, R, C, rpb, (const float*)gamma, (const float*)beta,
                           (float*)running_mean, (float*)running_var, (float*)scale, (float*)shift, (float*)save_mean,
                           (float*)save_invstd, count, momentum, eps);
    } else {
        hipLaunchKernelGGL(bn_fwd_eval_kernel, dim3((C + 127) / 128), dim3(128), 0, st, (const float*)gamma,
                           (const float*)beta, (float*)running_mean, (float*)running_var, (float*)scale, (float*)shift,
                           (float*)save_mean, (float*)save_invstd, C, eps);
    }
    return pd::check_launch("pd_bn_fwd_finalize");
}

extern "C" int pd_bn_bwd_finalize(const void* partial, long R, int C, double count, void* acc_ws, long acc_len,
                                  void* dgamma, void* dbeta, void* coef, int accumulate, void* stream) {
    PD_REQUIRE(partial && acc_ws && coef && C > 0 && R > 0 && count > 0, "pd_bn_bwd_finalize: bad arguments");
    PD_REQUIRE(acc_len >= 2L * C + 1, "pd_bn_bwd_finalize: accumulator needs 2 C + 1 doubles (%ld given)", acc_len);
    hipStream_t st = (hipStream_t)stream;
    if (R <= 4096) {
        hipLaunchKernelGGL(bn_stats_small_kernel<false>, dim3((C + 31) / 32), dim3(1024), 0, st, (const float*)partial, R, C,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)dgamma, (float*)dbeta, (float*)coef,
                           count, 0.f, 0.f, accumulate);
        return pd::check_launch("pd_bn_bwd_finalize");
    }
    const int cgroups = (C + 31) / 32;
    const int rpb = 256;
    const long rblocks = (R + rpb - 1) / rpb;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3((unsigned)(cgroups * rblocks)), dim3(EW_T), 0, st, (const float*)partial,
                       (double*)acc_ws, R, C, rpb, (float*)dgamma, (float*)dbeta, (float*)coef, count, accumulate);
    return pd::check_launch("pd_bn_bwd_finalize");
}

static int chain_check(int N, int H, int W, int C, int pool) {
    PD_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "pd_chain: bad dims (C must be a multiple of 4)");
    PD_REQUIRE(!pool || (H >= 2 && W >= 2), "pd_chain: pooling needs H,W >= 2");
    PD_REQUIRE(EW_T % (C / 4) == 0, "pd_chain: C/4 must divide %d (a thread keeps its channel quad across the loop)", EW_T);
    PD_REQUIRE((long)N * H * W * (C / 4) < (1L << 31) - (1L << 22), "pd_chain: tensor too large for 32-bit element indices");
    return PD_OK;
}

extern "C" long pd_chain_bwd_rows(int N, int H, int W, int C) {
    const long items = (long)N * H * W * (C / 4);
    long b = (items + EW_T - 1) / EW_T;
    constexpr long cap = 1024;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return b;
}

extern "C" int pd_chain_fwd(const void* x, const void* scale, const void* shift, const void* res, void* out,
                            int N, int H, int W, int C, long ld_res, long ld_out, int relu_pre, int pool,
                            float drop_p, uint64_t seed, uint64_t offset, const void* step_state, int relu_post,
                            void* stream) {
    int rc = chain_check(N, H, W, C, pool);
    if (rc) return rc;
    PD_REQUIRE(x && out && (!scale || shift), "pd_chain_fwd: null tensor");
    PD_REQUIRE(ld_out >= C && ld_out % 4 == 0 && (!res || (ld_res >= C && ld_res % 4 == 0)), "pd_chain_fwd: bad row stride");
    PD_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "pd_chain_fwd: dropout p must be in [0,1)");
    if (N == 0) return PD_OK;
    ChainArgs a{};
    a.x = (const float*)x; a.scale = (const float*)scale; a.shift = (const float*)shift; a.res = (const float*)res;
    a.out = (float*)out; a.ld_res = ld_res; a.ld_out = ld_out; a.N = N; a.H = H; a.W = W; a.C = C;
    a.relu_pre = relu_pre; a.pool = pool; a.relu_post = relu_post; a.drop_p = drop_p; a.seed = seed; a.offset = offset;
    a.step_state = (const long long*)step_state;
    const long items = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C / 4);
    hipLaunchKernelGGL(chain_fwd_kernel, dim3(ew_grid(items)), dim3(EW_T), 0, (hipStream_t)stream, a);
    return pd::check_launch("pd_chain_fwd");
}

static int chain_bwd_common(ChainArgs& a, const void* dy, long ld_dy, const void* x, const void* out, long ld_out,
                            const void* scale, const void* shift, const void* mean, const void* invstd, int N, int H,
                            int W, int C, int relu_pre, int pool, float drop_p, uint64_t seed, uint64_t offset,
                            const void* step_state, int relu_post) {
    int rc = chain_check(N, H, W, C, pool);
    if (rc) return rc;
    PD_REQUIRE(dy && x && (!relu_post || out) && (!scale || shift) && (!mean || invstd), "pd_chain_bwd: null tensor");
    PD_REQUIRE(ld_dy >= C && ld_dy % 4 == 0 && (!relu_post || (ld_out >= C && ld_out % 4 == 0)), "pd_chain_bwd: bad row stride");
    a.dy = (const float*)dy; a.ld_dy = ld_dy; a.x = (const float*)x; a.out = (float*)out; a.ld_out = ld_out;
    a.scale = (const float*)scale; a.shift = (const float*)shift; a.mean = (const float*)mean; a.invstd = (const float*)invstd;
    a.N = N; a.H = H; a.W = W; a.C = C; a.relu_pre = relu_pre; a.pool = pool; a.relu_post = relu_post;
    a.drop_p = drop_p; a.seed = seed; a.offset = offset; a.step_state = (const long long*)step_state;
    return PD_OK;
}

extern "C" int pd_chain_bwd_reduce(const void* dy, long ld_dy, const void* x, const void* out, long ld_out,
                                   const void* scale, const void* shift, const void* mean, const void* invstd,
                                   void* partial, int N, int H, int W, int C, int relu_pre, int pool, float drop_p,
                                   uint64_t seed, uint64_t offset, const void* step_state, int relu_post, void* stream) {
    ChainArgs a{};
    int rc = chain_bwd_common(a, dy, ld_dy, x, out, ld_out, scale, shift, mean, invstd, N, H, W, C, relu_pre, pool,
                              drop_p, seed, offset, step_state, relu_post);
    if (rc) return rc;
    PD_REQUIRE(partial && mean, "pd_chain_bwd_reduce: partial and mean/invstd are required");
    PD_REQUIRE(EW_T % (C / 4) == 0, "pd_chain_bwd_reduce: C/4 must divide 256");
    if (N == 0) return PD_OK;
    a.partial = (float*)partial;
    hipLaunchKernelGGL(chain_bwd_kernel<false>, dim3((unsigned)pd_chain_bwd_rows(N, H, W, C)), dim3(EW_T), 0,
                       (hipStream_t)stream, a);
    return pd::check_launch("pd_chain_bwd_reduce");
}

extern "C" int pd_chain_bwd_apply(const void* dy, long ld_dy, const void* x, const void* out, long ld_out,
                                  const void* scale, const void* shift, const void* mean, const void* invstd,
                                  const void* coef, void* dx, void* dres, int N, int H, int W, int C, int relu_pre,
                                  int pool, float drop_p, uint64_t seed, uint64_t offset, const void* step_state,
                                  int relu_post, void* stream) {
    ChainArgs a{};
    int rc = chain_bwd_common(a, dy, ld_dy, x, out, ld_out, scale, shift, mean, invstd, N, H, W, C, relu_pre, pool,
                              drop_p, seed, offset, step_state, relu_post);
    if (rc) return rc;
    PD_REQUIRE(dx && (!mean || coef), "pd_chain_bwd_apply: dx (and coef with batch statistics) required");
    PD_REQUIRE(!dres || relu_post, "pd_chain_bwd_apply: dres is only produced for post-add ReLU blocks");
    if (N == 0) return PD_OK;
    a.coef = (const float*)coef; a.dx = (float*)dx;
    // the residual gradient dy * [out > 0] is a by-product of the apply pass (same loads); pooled chains keep the
    // separate pass over the output grid
    a.dres = pool ? nullptr : (float*)dres;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chain_bwd_kernel<true>, dim3((unsigned)pd_chain_bwd_rows(N, H, W, C)), dim3(EW_T), 0, st, a);
    if (dres && pool) {
        const long npix = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W);
        hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_grid(npix * (C / 4))), dim3(EW_T), 0, st, (const float*)dy, ld_dy,
                           (const float*)out, ld_out, (float*)dres, npix, C);
    }
    return pd::check_launch("pd_chain_bwd_apply");
}

extern "C" int pd_maxpool3s2_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream) {
    PD_REQUIRE(x && y && N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "pd_maxpool3s2_fwd: bad arguments");
    if (N == 0) return PD_OK;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(ew_grid((long)N * Ho * Wo * (C / 4))), dim3(EW_T), 0,
                       (hipStream_t)stream, (const float*)x, (float*)y, (uint32_t*)idx, N, H, W, C, Ho, Wo);
    return pd::check_launch("pd_maxpool3s2_fwd");
}

extern "C" int pd_maxpool3s2_bwd_add(const void* idx, const void* dy, const void* addend, long ld_add, void* dx, int N, int H,
                                    int W, int C, void* stream) {
    PD_REQUIRE(idx && dy && dx && N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "pd_maxpool3s2_bwd: bad arguments");
    PD_REQUIRE(!addend || (ld_add >= C && ld_add % 4 == 0 && pd::aligned16(addend)), "pd_maxpool3s2_bwd: bad addend");
    if (N == 0) return PD_OK;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(ew_grid((long)N * H * W * (C / 4))), dim3(EW_T), 0,
                       (hipStream_t)stream, (const uint32_t*)idx, (const float*)dy, (const float*)addend, ld_add, (float*)dx, N,
                       H, W, C, Ho, Wo);
    return pd::check_launch("pd_maxpool3s2_bwd");
}

extern "C" int pd_maxpool3s2_bwd(const void* idx, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
    return pd_maxpool3s2_bwd_add(idx, dy, nullptr, 0, dx, N, H, W, C, stream);
}

extern "C" int pd_upcat_fwd(const void* a, const void* skip, long ld_skip, void* out, int N, int H, int W, int Ca,
                            int Cs, void* stream) {
    PD_REQUIRE(a && out && N >= 0 && H > 0 && W > 0 && Ca > 0 && Ca % 4 == 0 && Cs >= 0 && Cs % 4 == 0,
               "pd_upcat_fwd: bad arguments");
    PD_REQUIRE(Cs == 0 || (skip && ld_skip >= Cs && ld_skip % 4 == 0), "pd_upcat_fwd: bad skip tensor");
    if (N == 0) return PD_OK;
    hipLaunchKernelGGL(upcat_fwd_kernel, dim3(ew_grid((long)N * 4 * H * W * ((Ca + Cs) / 4))), dim3(EW_T), 0,
                       (hipStream_t)stream, (const float*)a, (const float*)skip, ld_skip, (float*)out, N, H, W, Ca, Cs);
    return pd::check_launch("pd_upcat_fwd");
}

extern "C" int pd_up2x_ac_fwd(const void* a, void* out, int N, int H, int W, int C, void* stream) {
    PD_REQUIRE(a && out && N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "pd_up2x_ac_fwd: bad arguments");
    if (N == 0) return PD_OK;
    hipLaunchKernelGGL(up2x_ac_fwd_kernel, dim3(ew_grid((long)N * 4 * H * W * (C / 4))), dim3(EW_T), 0, (hipStream_t)stream,
                       (const float*)a, (float*)out, N, H, W, C);
    return pd::check_launch("pd_up2x_ac_fwd");
}

extern "C" int pd_up2x_ac_bwd(const void* dout, void* da, int N, int H, int W, int C, void* stream) {
    PD_REQUIRE(dout && da && N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "pd_up2x_ac_bwd: bad arguments");
    if (N == 0) return PD_OK;
    hipLaunchKernelGGL(up2x_ac_bwd_kernel, dim3(ew_grid((long)N * H * W * (C / 4))), dim3(EW_T), 0, (hipStream_t)stream,
                       (const float*)dout, (float*)da, N, H, W, C);
    return pd::check_launch("pd_up2x_ac_bwd");
}

extern "C" int pd_relu_add(const void* x, const void* res, void* out, long n, int relu, void* stream) {
    PD_REQUIRE(x && out && n >= 0 && n % 4 == 0 && pd::aligned16(x) && pd::aligned16(out) && pd::aligned16(res),
               "pd_relu_add: bad arguments (element count must be a multiple of 4, pointers 16-byte aligned)");
    if (n == 0) return PD_OK;
    hipLaunchKernelGGL(relu_add_kernel, dim3(ew_grid(n / 4)), dim3(EW_T), 0, (hipStream_t)stream, (const float*)x,
                       (const float*)res, (float*)out, n / 4, relu);
    return pd::check_launch("pd_relu_add");
}

extern "C" int pd_up_bwd_elu(const void* dout, long ld_d, const void* elu_y, void* da, int N, int H, int W, int Ca,
                             void* stream) {
    PD_REQUIRE(dout && da && N >= 0 && H > 0 && W > 0 && Ca > 0 && Ca % 4 == 0 && ld_d >= Ca && ld_d % 4 == 0,
               "pd_up_bwd: bad arguments");
    if (N == 0) return PD_OK;
    hipLaunchKernelGGL(up_bwd_kernel, dim3(ew_grid((long)N * H * W * (Ca / 4))), dim3(EW_T), 0, (hipStream_t)stream,
                       (const float*)dout, ld_d, (const float*)elu_y, (float*)da, N, H, W, Ca);
    return pd::check_launch("pd_up_bwd");
}

extern "C" int pd_up_bwd(const void* dout, long ld_d, void* da, int N, int H, int W, int Ca, void* stream) {
    return pd_up_bwd_elu(dout, ld_d, nullptr, da, N, H, W, Ca, stream);
}

extern "C" int pd_act_bwd(const void* dy, const void* y, void* dz, long n, int act, void* stream) {
    PD_REQUIRE(dy && y && dz && n >= 0 && act >= 1 && act <= 3, "pd_act_bwd: bad arguments");
    if (n == 0) return PD_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(EW_T), 0, (hipStream_t)stream, (const float*)dy,
                       (const float*)y, (float*)dz, n, act);
    return pd::check_launch("pd_act_bwd");
}

extern "C" int pd_reflect_fold_pad(const void* dxp, void* dx, int N, int H, int W, int C, int pad, void* stream) {
    PD_REQUIRE(dxp && dx && N >= 0 && H >= 2 && W >= 2 && C > 0, "pd_reflect_fold: bad arguments");
    PD_REQUIRE(pad >= 1 && pad < H && pad < W, "pd_reflect_fold: pad must be in [1, min(H, W))");
    if (N == 0) return PD_OK;
    if (C % 4 == 0)
        hipLaunchKernelGGL(reflect_fold_kernel<4>, dim3(ew_grid((long)N * H * W * (C / 4))), dim3(EW_T), 0,
                           (hipStream_t)stream, (const float*)dxp, (float*)dx, N, H, W, C, pad);
    else
        hipLaunchKernelGGL(reflect_fold_kernel<1>, dim3(ew_grid((long)N * H * W * C)), dim3(EW_T), 0,
                           (hipStream_t)stream, (const float*)dxp, (float*)dx, N, H, W, C, pad);
    return pd::check_launch("pd_reflect_fold");
}

extern "C" int pd_reflect_fold(const void* dxp, void* dx, int N, int H, int W, int C, void* stream) {
    return pd_reflect_fold_pad(dxp, dx, N, H, W, C, 1, stream);
}

extern "C" int pd_reflect_dgrad_border(const void* dz, long ldd, const void* w, void* dx, int N, int H, int W, int Co,
                                       int Ci, void* stream) {
    PD_REQUIRE(dz && w && dx && N >= 0 && H >= 2 && W >= 2 && Co > 0 && Ci > 0 && ldd >= Co, "pd_reflect_dgrad_border: bad arguments");
    if (N == 0) return PD_OK;
    hipLaunchKernelGGL(reflect_dgrad_border_flat_kernel, dim3(ew_grid((long)N * (2 * W + 2 * H) * Ci)), dim3(EW_T), 0,
                       (hipStream_t)stream, (const float*)dz, ldd, (const float*)w, (float*)dx, N, H, W, Co, Ci);
    return pd::check_launch("pd_reflect_dgrad_border");
}

extern "C" int pd_step_tick(void* step_state, int bump_dropout, int bump_adam, void* stream) {
    PD_REQUIRE(step_state, "pd_step_tick: null state");
    hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)step_state, bump_dropout, bump_adam);
    return pd::check_launch("pd_step_tick");
}

extern "C" int pd_step_set_hyper(void* step_state, float lr, float grad_scale, void* stream) {
    PD_REQUIRE(step_state, "pd_step_set_hyper: null state");
    hipLaunchKernelGGL(step_set_hyper_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)step_state, lr, grad_scale);
    return pd::check_launch("pd_step_set_hyper");
}

extern "C" int pd_adam_step(void* p, void* g, void* m, void* v, long n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, long step, const void* step_state, float grad_scale,
                            int zero_grad, void* stream) {
    PD_REQUIRE(p && g && m && v && n >= 0 && (step >= 1 || step_state), "pd_adam_step: bad arguments");
    PD_REQUIRE(pd::aligned16(p) && pd::aligned16(g) && pd::aligned16(m) && pd::aligned16(v), "pd_adam_step: unaligned");
    if (n == 0) return PD_OK;
    float bc1 = 0.f, bc2s = 0.f;
    if (!step_state) {
        bc1 = (float)(1.0 - ipow((double)beta1, step));
        bc2s = sqrtf((float)(1.0 - ipow((double)beta2, step)));
    }
    const long long* ss = (const long long*)step_state;
    if (zero_grad)
        hipLaunchKernelGGL(adam_kernel<true>, dim3(ew_grid((n + 3) / 4)), dim3(EW_T), 0, (hipStream_t)stream, (float*)p,
                           (float*)g, (float*)m, (float*)v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, ss);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3(ew_grid((n + 3) / 4)), dim3(EW_T), 0, (hipStream_t)stream, (float*)p,
                           (float*)g, (float*)m, (float*)v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, ss);
    return pd::check_launch("pd_adam_step");
}

// ============================================================================ row softmax (attention variant)
// Single-head self-attention at the JointEncoder merge (BASELINE config 5 / SURVEY A17; the reference branch
// arch1++_attention is absent from the checkout, so the block is defined by this build: see DESIGN.md).
// The score matrix is materialised in fp32 (T = H/8*W/8 = 5120 tokens -> 105 MB per image, trivially resident in
// 288 GB of HBM) and the two GEMMs run on the fp32-MFMA implicit-GEMM kernels; these kernels do the row softmax.
namespace {

__device__ __forceinline__ float block_reduce(float v, float* sm, bool is_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o); v = is_max ? fmaxf(v, t) : v + t; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float r = sm[0];
#pragma unroll
    for (int w = 1; w < EW_T / 64; ++w) r = is_max ? fmaxf(r, sm[w]) : r + sm[w];
    return r;
}

// in place: x[r][:] = softmax(scale * x[r][:]); one workgroup per row
__global__ __launch_bounds__(EW_T) void softmax_fwd_kernel(float* __restrict__ x, long L, float scale) {
    __shared__ float sm[EW_T / 64];
    float* row = x + (long)blockIdx.x * L;
    float m = -INFINITY;
    for (long i = threadIdx.x; i < L; i += EW_T) m = fmaxf(m, row[i] * scale);
    m = block_reduce(m, sm, true);
    float s = 0.f;
    for (long i = threadIdx.x; i < L; i += EW_T) { const float e = expf(row[i] * scale - m); row[i] = e; s += e; }
    s = block_reduce(s, sm, false);
    const float inv = 1.f / s;
    for (long i = threadIdx.x; i < L; i += EW_T) row[i] *= inv;
}

// in place on dp: ds[r][:] = scale * p[r][:] * (dp[r][:] - sum_j dp[r][j] p[r][j])
__global__ __launch_bounds__(EW_T) void softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long L,
                                                           float scale) {
    __shared__ float sm[EW_T / 64];
    const float* pr = p + (long)blockIdx.x * L;
    float* dr = dp + (long)blockIdx.x * L;
    float s = 0.f;
    for (long i = threadIdx.x; i < L; i += EW_T) s += pr[i] * dr[i];
    s = block_reduce(s, sm, false);
    for (long i = threadIdx.x; i < L; i += EW_T) dr[i] = scale * pr[i] * (dr[i] - s);
}

}  // namespace

extern "C" int pd_softmax_rows_fwd(void* x, long R, long L, float scale, void* stream) {
    PD_REQUIRE(x && R >= 0 && L > 0, "pd_softmax_rows_fwd: bad arguments");
    if (R == 0) return PD_OK;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)R), dim3(EW_T), 0, (hipStream_t)stream, (float*)x, L, scale);
    return pd::check_launch("pd_softmax_rows_fwd");
}

extern "C" int pd_softmax_rows_bwd(const void* p, void* dp, long R, long L, float scale, void* stream) {
    PD_REQUIRE(p && dp && R >= 0 && L > 0, "pd_softmax_rows_bwd: bad arguments");
    if (R == 0) return PD_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)R), dim3(EW_T), 0, (hipStream_t)stream, (const float*)p,
                       (float*)dp, L, scale);
    return pd::check_launch("pd_softmax_rows_bwd");
}
