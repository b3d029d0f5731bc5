// Fused single-head softmax attention over the token grid of the joint encoder (SURVEY.md §8 row A17,
// BASELINE configs[4]) on the fp32 matrix cores: O = softmax(Q K^T * scale) V without materialising the
// T x T score matrix (flash-attention recurrence).  q, k, v, o: [N][T][128] fp32 (NHWC tokens), lse: [N][T].
//
// Layout trick: every wave owns 32 queries and keeps them on the *lane* axis of all its MFMA tiles --
//   S^T (keys x queries)  = K_blk (32 x 128)  .  Q^T          A from LDS, B = the wave's Q in registers
//   O^T (chan x queries) += V_blk^T (128 x 32) . P^T (keys x queries)   A from LDS, B = the S^T accumulator itself
// In the 32x32 C/D layout a lane holds column (query) lane&31 and rows (r&3)+8(r>>2)+4(lane>>5): the softmax
// statistics of a query are lane-local (16 values + one cross-half shuffle), the rescale of O^T is a per-lane
// scalar, and accumulator register r of S^T is exactly the B operand of MFMA step r of the second product (the
// contraction index is permuted identically on the V side), so P never leaves the registers.
#include "pd_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 128;            // head dimension (channels after fc2)
constexpr int KB = 32;             // keys per block
constexpr int QW = 32;             // queries per wave
constexpr int ATT_T = 256;         // 4 waves -> 128 queries per workgroup

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

// LDS tiles: K block [32 keys][128] with 16-byte slots XOR-swizzled by the key (conflict-free ds_read_b128 of one
// slot column over 32 rows); V block [32 keys][128] with the 32-float halves swapped on rows with bit 2 set (the two
// half-waves of a ds_read_b32 read rows 4 apart).
__device__ __forceinline__ int kslot(int row, int slot) { return 4 * (slot ^ (row & 31)); }
__device__ __forceinline__ int vcol(int row, int col) { return col ^ (((row >> 2) & 1) << 5); }

__global__ __launch_bounds__(ATT_T, 2) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ o,
                                                            float* __restrict__ lse, int T, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) float Ks[2][KB][HD];
    __shared__ __attribute__((aligned(16))) float Vs[2][KB][HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int q0 = blockIdx.x * (4 * QW) + wave * QW;
    const int ql = lane & 31, h = lane >> 5;
    const float* qn = q + (long)n * T * HD;
    const float* kn = k + (long)n * T * HD;
    const float* vn = v + (long)n * T * HD;

    // the wave's queries, pre-scaled: qv[g] = Q[q0+ql][8g + 4h .. +3] * scale * log2(e)
    float4 qv[HD / 8];
    {
        const int qi = q0 + ql < T ? q0 + ql : T - 1;
#pragma unroll
        for (int g = 0; g < HD / 8; ++g) {
            float4 t = ld4g(qn + (long)qi * HD + 8 * g + 4 * h);
            qv[g] = make_float4(t.x * scale_log2e, t.y * scale_log2e, t.z * scale_log2e, t.w * scale_log2e);
        }
    }
    f32x16 oacc[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[c][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging: 32 rows x 32 slots per tile = 1024 float4 -> 4 per thread and tile
    float4 pk[4], pv[4];
    auto load_block = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + ATT_T * i, row = p >> 5, slot = p & 31;
            const long off = (long)(kb * KB + row) * HD + 4 * slot;
            pk[i] = ld4g(kn + off);
            pv[i] = ld4g(vn + off);
        }
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + ATT_T * i, row = p >> 5, slot = p & 31;
            *reinterpret_cast<float4*>(&Ks[buf][row][kslot(row, slot)]) = pk[i];
            *reinterpret_cast<float4*>(&Vs[buf][row][vcol(row, 4 * slot)]) = pv[i];
        }
    };

    const int nkb = T / KB;
    load_block(0);
    store_block(0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) load_block(kb + 1);
        // ---- S^T = K_blk . Q^T  (keys on rows, this wave's queries on lanes)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < HD / 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(&Ks[buf][ql][kslot(ql, 2 * g + h)]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qv[g].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qv[g].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qv[g].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qv[g].w, s, 0, 0, 0);
        }
        // ---- online softmax of the lane's query (16 keys here, 16 in the other half-wave)
        float mloc = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = exp2f(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - m_new); lsum += s[r]; }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        // ---- O^T = alpha * O^T + V_blk^T . P^T : MFMA step r contracts keys (r&3)+8(r>>2) [+4 for the upper half-wave]
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[c][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float a = Vs[buf][row][vcol(row, 32 * c + ql)];
                oacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], oacc[c], 0, 0, 0);
            }
        }
        if (kb + 1 < nkb) store_block(buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue: O[query][c] = O^T[c][query] / l ; lse = (m + log2 l) * ln 2
    const int qi = q0 + ql;
    if (qi < T) {
        const float inv = 1.f / l_run;
        float* on = o + ((long)n * T + qi) * HD;
#pragma unroll
        for (int c = 0; c < HD / 32; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) on[32 * c + (r & 3) + 8 * (r >> 2) + 4 * h] = oacc[c][r] * inv;
        if (h == 0) lse[(long)n * T + qi] = (m_run + log2f(l_run)) * 0.6931471805599453f;
    }
}

// delta[n][t] = sum_c dO[t][c] * O[t][c]   (one wave per token pair; memory-bound)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
                                                         float* __restrict__ delta, long ntok) {
    const long t = blockIdx.x * 8L + (threadIdx.x >> 5);      // 32 lanes per token, float4 each
    if (t >= ntok) return;
    const int l = threadIdx.x & 31;
    const float4 a = ld4g(o + t * HD + 4 * l), b = ld4g(d_o + t * HD + 4 * l);
    float sum = (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    if (l == 0) delta[t] = sum;
}

// dQ: same geometry as the forward (a wave's 32 queries on the lane axis).  Per key block:
//   S^T  = K_blk . Q^T (recomputed),  P = exp(S - lse),  dP^T = V_blk . dO^T,  dS = P (dP - delta),
//   dQ^T (chan x queries) += K_blk^T (128 x 32 keys) . dS (keys x queries)      -- dS registers are the B operand
__global__ __launch_bounds__(ATT_T, 1) void attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ d_o,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dq, int T, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[2][KB][HD];
    __shared__ __attribute__((aligned(16))) float Vs[2][KB][HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int q0 = blockIdx.x * (4 * QW) + wave * QW;
    const int ql = lane & 31, h = lane >> 5;
    const long base = (long)n * T * HD;
    const float* kn = k + base;
    const float* vn = v + base;
    const float scale_log2e = scale * 1.4426950408889634f;
    const int qi = q0 + ql < T ? q0 + ql : T - 1;
    float4 qv[HD / 8], dov[HD / 8];
#pragma unroll
    for (int g = 0; g < HD / 8; ++g) {
        const float4 t = ld4g(q + base + (long)qi * HD + 8 * g + 4 * h);
        qv[g] = make_float4(t.x * scale_log2e, t.y * scale_log2e, t.z * scale_log2e, t.w * scale_log2e);
        dov[g] = ld4g(d_o + base + (long)qi * HD + 8 * g + 4 * h);
    }
    const float lse2 = lse[(long)n * T + qi] * 1.4426950408889634f;
    const float dl = delta[(long)n * T + qi];
    f32x16 acc[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    float4 pk[4], pv[4];
    auto load_block = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + ATT_T * i, row = p >> 5, slot = p & 31;
            const long off = (long)(kb * KB + row) * HD + 4 * slot;
            pk[i] = ld4g(kn + off);
            pv[i] = ld4g(vn + off);
        }
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + ATT_T * i, row = p >> 5, slot = p & 31;
            *reinterpret_cast<float4*>(&Ks[buf][row][kslot(row, slot)]) = pk[i];
            *reinterpret_cast<float4*>(&Vs[buf][row][kslot(row, slot)]) = pv[i];
        }
    };
    const int nkb = T / KB;
    load_block(0);
    store_block(0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) load_block(kb + 1);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int g = 0; g < HD / 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(&Ks[buf][ql][kslot(ql, 2 * g + h)]);
            const float4 b = *reinterpret_cast<const float4*>(&Vs[buf][ql][kslot(ql, 2 * g + h)]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qv[g].x, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, dov[g].x, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qv[g].y, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, dov[g].y, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qv[g].z, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, dov[g].z, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qv[g].w, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, dov[g].w, dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = exp2f(s[r] - lse2) * (dp[r] - dl);      // dS (keys x queries)
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = 32 * c + ql;
                const float a = Ks[buf][row][kslot(row, col >> 2) + (col & 3)];
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], acc[c], 0, 0, 0);
            }
        }
        if (kb + 1 < nkb) store_block(buf ^ 1);
        __syncthreads();
    }
    if (q0 + ql < T) {
        float* out = dq + base + (long)(q0 + ql) * HD;
#pragma unroll
        for (int c = 0; c < HD / 32; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[32 * c + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[c][r] * scale;
    }
}

// dK, dV: a wave owns 32 *keys* on the lane axis; per query block
//   S (queries x keys) = Q_blk . K^T,  P = exp(S - lse[row]),  dP = dO_blk . V^T,  dS = P (dP - delta[row]),
//   dV^T (chan x keys) += dO_blk^T . P,   dK^T (chan x keys) += Q_blk^T . dS
__global__ __launch_bounds__(ATT_T, 1) void attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ d_o,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                float* __restrict__ dk, float* __restrict__ dv, int T,
                                                                float scale) {
    __shared__ __attribute__((aligned(16))) float Qs[2][KB][HD];
    __shared__ __attribute__((aligned(16))) float Ds[2][KB][HD];
    __shared__ __attribute__((aligned(16))) float Ls[2][KB], Dl[2][KB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int k0 = blockIdx.x * (4 * QW) + wave * QW;
    const int kl = lane & 31, h = lane >> 5;
    const long base = (long)n * T * HD;
    const float* qn = q + base;
    const float* dn = d_o + base;
    const float scale_log2e = scale * 1.4426950408889634f;
    const int ki = k0 + kl < T ? k0 + kl : T - 1;
    float4 kv[HD / 8], vv[HD / 8];
#pragma unroll
    for (int g = 0; g < HD / 8; ++g) {
        const float4 t = ld4g(k + base + (long)ki * HD + 8 * g + 4 * h);
        kv[g] = make_float4(t.x * scale_log2e, t.y * scale_log2e, t.z * scale_log2e, t.w * scale_log2e);
        vv[g] = ld4g(v + base + (long)ki * HD + 8 * g + 4 * h);
    }
    f32x16 akk[HD / 32], avv[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { akk[c][r] = 0.f; avv[c][r] = 0.f; }

    float4 pq[4], pd[4];
    float pl = 0.f, pdl = 0.f;
    auto load_block = [&](int qb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + ATT_T * i, row = p >> 5, slot = p & 31;
            const long off = (long)(qb * KB + row) * HD + 4 * slot;
            pq[i] = ld4g(qn + off);
            pd[i] = ld4g(dn + off);
        }
        if (tid < KB) { pl = lse[(long)n * T + qb * KB + tid] * 1.4426950408889634f; pdl = delta[(long)n * T + qb * KB + tid]; }
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + ATT_T * i, row = p >> 5, slot = p & 31;
            *reinterpret_cast<float4*>(&Qs[buf][row][kslot(row, slot)]) = pq[i];
            *reinterpret_cast<float4*>(&Ds[buf][row][kslot(row, slot)]) = pd[i];
        }
        if (tid < KB) { Ls[buf][tid] = pl; Dl[buf][tid] = pdl; }
    };
    const int nqb = T / KB;
    load_block(0);
    store_block(0);
    __syncthreads();
    for (int qb = 0; qb < nqb; ++qb) {
        const int buf = qb & 1;
        if (qb + 1 < nqb) load_block(qb + 1);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int g = 0; g < HD / 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(&Qs[buf][kl][kslot(kl, 2 * g + h)]);   // row = query (lane&31)
            const float4 b = *reinterpret_cast<const float4*>(&Ds[buf][kl][kslot(kl, 2 * g + h)]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kv[g].x, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, vv[g].x, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kv[g].y, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, vv[g].y, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kv[g].z, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, vv[g].z, dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kv[g].w, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, vv[g].w, dp, 0, 0, 0);
        }
        // rows of the tile are queries: row(r) = (r&3) + 8(r>>2) + 4h; their lse / delta come from LDS (float4 per r>>2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 l4 = *reinterpret_cast<const float4*>(&Ls[buf][8 * j + 4 * h]);
            const float4 d4 = *reinterpret_cast<const float4*>(&Dl[buf][8 * j + 4 * h]);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = exp2f(s[4 * j + i] - lv[i]);
                s[4 * j + i] = p;                                  // P  (queries x keys)
                dp[4 * j + i] = p * (dp[4 * j + i] - dvv[i]);      // dS (queries x keys)
            }
        }
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = 32 * c + kl;
                const int off = kslot(row, col >> 2) + (col & 3);
                avv[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[buf][row][off], s[r], avv[c], 0, 0, 0);
                akk[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[buf][row][off], dp[r], akk[c], 0, 0, 0);
            }
        }
        if (qb + 1 < nqb) store_block(buf ^ 1);
        __syncthreads();
    }
    if (k0 + kl < T) {
        float* ok = dk + base + (long)(k0 + kl) * HD;
        float* ov = dv + base + (long)(k0 + kl) * HD;
#pragma unroll
        for (int c = 0; c < HD / 32; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cc = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * h;
                ok[cc] = akk[c][r] * scale;
                ov[cc] = avv[c][r];
            }
    }
}

}  // namespace

extern "C" int pd_attn_fwd(const void* q, const void* k, const void* v, void* o, void* lse, int N, int T, int C,
                           float scale, void* stream) {
    PD_REQUIRE(N >= 0 && T > 0, "pd_attn_fwd: bad shape N=%d T=%d", N, T);
    PD_REQUIRE(C == HD, "pd_attn_fwd: head dimension must be %d (got %d)", HD, C);
    PD_REQUIRE(T % KB == 0, "pd_attn_fwd: the token count must be a multiple of %d (got %d)", KB, T);
    if (N == 0) return PD_OK;
    PD_REQUIRE(q && k && v && o && lse, "pd_attn_fwd: null tensor");
    PD_REQUIRE(pd::aligned16(q) && pd::aligned16(k) && pd::aligned16(v), "pd_attn_fwd: q, k, v must be 16-byte aligned");
    const dim3 grid((unsigned)((T + 4 * QW - 1) / (4 * QW)), (unsigned)N);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(ATT_T), 0, (hipStream_t)stream, (const float*)q, (const float*)k,
                       (const float*)v, (float*)o, (float*)lse, T, scale * 1.4426950408889634f);
    return pd::check_launch("pd_attn_fwd");
}

extern "C" int pd_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const void* lse,
                           void* delta, void* dq, void* dk, void* dv, int N, int T, int C, float scale, void* stream) {
    PD_REQUIRE(N >= 0 && T > 0, "pd_attn_bwd: bad shape N=%d T=%d", N, T);
    PD_REQUIRE(C == HD, "pd_attn_bwd: head dimension must be %d (got %d)", HD, C);
    PD_REQUIRE(T % KB == 0, "pd_attn_bwd: the token count must be a multiple of %d (got %d)", KB, T);
    if (N == 0) return PD_OK;
    PD_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "pd_attn_bwd: null tensor");
    PD_REQUIRE(pd::aligned16(q) && pd::aligned16(k) && pd::aligned16(v) && pd::aligned16(o) && pd::aligned16(d_o),
               "pd_attn_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long ntok = (long)N * T;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((ntok + 7) / 8)), dim3(256), 0, st, (const float*)o,
                       (const float*)d_o, (float*)delta, ntok);
    const dim3 grid((unsigned)((T + 4 * QW - 1) / (4 * QW)), (unsigned)N);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(ATT_T), 0, st, (const float*)q, (const float*)k, (const float*)v,
                       (const float*)d_o, (const float*)lse, (const float*)delta, (float*)dk, (float*)dv, T, scale);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(ATT_T), 0, st, (const float*)q, (const float*)k, (const float*)v,
                       (const float*)d_o, (const float*)lse, (const float*)delta, (float*)dq, T, scale);
    return pd::check_launch("pd_attn_bwd");
}
