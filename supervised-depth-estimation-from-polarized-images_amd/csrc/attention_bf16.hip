// bf16-MFMA variant of the fused single-head attention (BASELINE configs[4]: "bf16, MFMA attention path"; SURVEY.md
// §8 row A17).  Same algorithm, interface and layouts as csrc/attention.hip (q, k, v, o, dq, dk, dv: [N][T][128]
// fp32 NHWC tokens; lse, delta: [N][T] fp32), but every matrix product runs on v_mfma_f32_32x32x16_bf16: operands
// are rounded to bf16 when they are staged (LDS tiles) or loaded (the wave's own tokens), the softmax statistics,
// exp2, the rescale and all accumulators stay fp32.  16 MFMAs replace 128 per 32-key block.
//
// Layout trick (as in the fp32 kernels): a wave keeps its 32 tokens on the LANE axis of every MFMA tile.
//   S^T (keys x queries) = K_blk . Q^T            A = K rows from LDS (ds_read_b128), B = the wave's Q in registers
//   O^T (chan x queries) += V_blk^T . P^T         A = V^T rows from LDS (2 x ds_read_b64), B = the S^T accumulator
// An accumulator of the 32x32 tile holds column lane&31 and rows (r&3)+8(r>>2)+4(lane>>5): converted pairwise to
// bf16, registers 8s..8s+7 ARE the B fragment of k-step s of the next product, with the contraction index permuted
// to 16s + 8(j>>2) + 4h + (j&3); the A fragments are read in the same order from a TRANSPOSED LDS tile
// ([channel][token], 8-byte units of 4 tokens), which the staging threads write next to the row-major tile.
#include "pd_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 128;            // head dimension (channels after fc2)
constexpr int KB = 32;             // tokens per staged block
constexpr int QW = 32;             // tokens per wave (lane axis)
constexpr int ATT_T = 256;         // 4 waves -> 128 tokens per workgroup
constexpr int TILE = KB * HD * 2;  // bytes of one bf16 tile (either layout): 8 KB
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ bf16x4 cvt4(float a, float b, float c, float d) {
    bf16x4 r; r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d; return r;
}
__device__ __forceinline__ bf16x8 cvt8(float4 a, float4 b, float s) {
    bf16x8 r;
    r[0] = (__bf16)(a.x * s); r[1] = (__bf16)(a.y * s); r[2] = (__bf16)(a.z * s); r[3] = (__bf16)(a.w * s);
    r[4] = (__bf16)(b.x * s); r[5] = (__bf16)(b.y * s); r[6] = (__bf16)(b.z * s); r[7] = (__bf16)(b.w * s);
    return r;
}
// accumulator registers 8s .. 8s+7 -> the bf16 B fragment of k-step s
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s + j];
    return r;
}

// ---- LDS tiles
// row-major [32 tokens][128 ch] bf16: 256-byte rows, 16-byte slots XOR-swizzled by the token (conflict-free
// ds_read_b128 of one slot column over 32 rows)
__device__ __forceinline__ int row_off(int tok, int slot) { return tok * 256 + 16 * (slot ^ (tok & 15)); }
// transposed [128 ch][32 tokens] bf16: 64-byte rows, 8-byte units (4 tokens) XOR-swizzled by channel / 4
__device__ __forceinline__ int tr_off(int ch, int unit) { return ch * 64 + 8 * (unit ^ ((ch >> 2) & 7)); }

// staging: thread t holds tokens 4u .. 4u+3 (u = t >> 5) x channels 4c4 .. 4c4+3 (c4 = t & 31) of a block
struct Stage { float4 v[4]; };
__device__ __forceinline__ Stage load_stage(const float* base, int tok0, int tid) {
    Stage s;
    const int c4 = tid & 31, u = tid >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) s.v[i] = ld4g(base + (long)(tok0 + 4 * u + i) * HD + 4 * c4);
    return s;
}
__device__ __forceinline__ void store_rows(char* tile, const Stage& s, int tid) {
    const int c4 = tid & 31, u = tid >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tok = 4 * u + i;
        *reinterpret_cast<bf16x4*>(tile + row_off(tok, c4 >> 1) + 8 * (c4 & 1)) = cvt4(s.v[i].x, s.v[i].y, s.v[i].z, s.v[i].w);
    }
}
__device__ __forceinline__ void store_transposed(char* tile, const Stage& s, int tid) {
    const int c4 = tid & 31, u = tid >> 5;
    *reinterpret_cast<bf16x4*>(tile + tr_off(4 * c4 + 0, u)) = cvt4(s.v[0].x, s.v[1].x, s.v[2].x, s.v[3].x);
    *reinterpret_cast<bf16x4*>(tile + tr_off(4 * c4 + 1, u)) = cvt4(s.v[0].y, s.v[1].y, s.v[2].y, s.v[3].y);
    *reinterpret_cast<bf16x4*>(tile + tr_off(4 * c4 + 2, u)) = cvt4(s.v[0].z, s.v[1].z, s.v[2].z, s.v[3].z);
    *reinterpret_cast<bf16x4*>(tile + tr_off(4 * c4 + 3, u)) = cvt4(s.v[0].w, s.v[1].w, s.v[2].w, s.v[3].w);
}
// A fragment of k-step g (channels 16g + 8h ..) of row `tok` of a row-major tile
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int tok, int g, int h) {
    return *reinterpret_cast<const bf16x8*>(tile + row_off(tok, 2 * g + h));
}
// A fragment of k-step s (tokens 16s + 8(j>>2) + 4h + (j&3)) of channel row `ch` of a transposed tile
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int ch, int s, int h) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(tile + tr_off(ch, 4 * s + h));
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(tile + tr_off(ch, 4 * s + 2 + h));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// the wave's own token as eight B fragments: X[tok][16g + 8h + j] * s
__device__ __forceinline__ void own_frags(const float* row, int h, float s, bf16x8 (&f)[HD / 16]) {
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) f[g] = cvt8(ld4g(row + 16 * g + 8 * h), ld4g(row + 16 * g + 8 * h + 4), s);
}
__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(ATT_T, 2) void attn_fwd_bf16_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, float* __restrict__ o,
                                                                 float* __restrict__ lse, int T, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char Ks[2][TILE];
    __shared__ __attribute__((aligned(16))) char Vt[2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int q0 = blockIdx.x * (4 * QW) + wave * QW;
    const int ql = lane & 31, h = lane >> 5;
    const float* kn = k + (long)n * T * HD;
    const float* vn = v + (long)n * T * HD;
    bf16x8 qb[HD / 16];
    own_frags(q + ((long)n * T + (q0 + ql < T ? q0 + ql : T - 1)) * HD, h, scale_log2e, qb);
    f32x16 oacc[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) zero(oacc[c]);
    float m_run = -INFINITY, l_run = 0.f;

    const int nkb = T / KB;
    Stage sk = load_stage(kn, 0, tid), sv = load_stage(vn, 0, tid);
    store_rows(Ks[0], sk, tid);
    store_transposed(Vt[0], sv, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) { sk = load_stage(kn, (kb + 1) * KB, tid); sv = load_stage(vn, (kb + 1) * KB, tid); }
        // ---- S^T = K_blk . Q^T  (keys on rows, this wave's queries on lanes)
        f32x16 s;
        zero(s);
#pragma unroll
        for (int g = 0; g < HD / 16; ++g)
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks[buf], ql, g, h), qb[g], s, 0, 0, 0);
        // ---- online softmax of the lane's query (16 keys here, 16 in the other half-wave), fp32
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
        // the rescale of O^T is skipped (exactly: alpha == 1) while no query of the wave has met a new maximum
        if (__any(m_new > m_run)) {
#pragma unroll
            for (int c = 0; c < HD / 32; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[c][r] *= alpha;
        }
        m_run = m_new;
        // ---- O^T += V_blk^T . P^T
        const bf16x8 p0 = acc_frag(s, 0), p1 = acc_frag(s, 1);
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
            oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vt[buf], 32 * c + ql, 0, h), p0, oacc[c], 0, 0, 0);
            oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vt[buf], 32 * c + ql, 1, h), p1, oacc[c], 0, 0, 0);
        }
        if (kb + 1 < nkb) { store_rows(Ks[buf ^ 1], sk, tid); store_transposed(Vt[buf ^ 1], sv, tid); }
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
            for (int j = 0; j < 4; ++j)      // registers 4j .. 4j+3 are four consecutive channels
                *reinterpret_cast<float4*>(on + 32 * c + 8 * j + 4 * h) =
                    make_float4(oacc[c][4 * j] * inv, oacc[c][4 * j + 1] * inv, oacc[c][4 * j + 2] * inv, oacc[c][4 * j + 3] * inv);
        if (h == 0) lse[(long)n * T + qi] = (m_run + log2f(l_run)) * 0.6931471805599453f;
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// queries on lanes.  Per key block:  S^T = K.Q^T,  dP^T = V.dO^T,  dS = exp2(S - lse) (dP - delta),
//   dQ^T (chan x queries) += K_blk^T (chan x keys) . dS (keys x queries)
__global__ __launch_bounds__(ATT_T, 2) void attn_bwd_dq_bf16_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                    const float* __restrict__ v, const float* __restrict__ d_o,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    float* __restrict__ dq, int T, float scale) {
    __shared__ __attribute__((aligned(16))) char Ks[2][TILE];
    __shared__ __attribute__((aligned(16))) char Vs[2][TILE];
    __shared__ __attribute__((aligned(16))) char Kt[2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int q0 = blockIdx.x * (4 * QW) + wave * QW;
    const int ql = lane & 31, h = lane >> 5;
    const long base = (long)n * T * HD;
    const float* kn = k + base;
    const float* vn = v + base;
    const int qi = q0 + ql < T ? q0 + ql : T - 1;
    bf16x8 qb[HD / 16], dob[HD / 16];
    own_frags(q + base + (long)qi * HD, h, scale * kLog2e, qb);
    own_frags(d_o + base + (long)qi * HD, h, 1.f, dob);
    const float lse2 = lse[(long)n * T + qi] * kLog2e;
    const float dl = delta[(long)n * T + qi];
    f32x16 acc[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) zero(acc[c]);

    const int nkb = T / KB;
    Stage sk = load_stage(kn, 0, tid), sv = load_stage(vn, 0, tid);
    store_rows(Ks[0], sk, tid); store_transposed(Kt[0], sk, tid); store_rows(Vs[0], sv, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) { sk = load_stage(kn, (kb + 1) * KB, tid); sv = load_stage(vn, (kb + 1) * KB, tid); }
        f32x16 s, dp;
        zero(s); zero(dp);
#pragma unroll
        for (int g = 0; g < HD / 16; ++g) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks[buf], ql, g, h), qb[g], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Vs[buf], ql, g, h), dob[g], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = exp2f(s[r] - lse2) * (dp[r] - dl);      // dS (keys x queries), fp32
        const bf16x8 d0 = acc_frag(s, 0), d1 = acc_frag(s, 1);
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kt[buf], 32 * c + ql, 0, h), d0, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kt[buf], 32 * c + ql, 1, h), d1, acc[c], 0, 0, 0);
        }
        if (kb + 1 < nkb) {
            store_rows(Ks[buf ^ 1], sk, tid); store_transposed(Kt[buf ^ 1], sk, tid); store_rows(Vs[buf ^ 1], sv, tid);
        }
        __syncthreads();
    }
    if (q0 + ql < T) {
        float* out = dq + base + (long)(q0 + ql) * HD;
#pragma unroll
        for (int c = 0; c < HD / 32; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(out + 32 * c + 8 * j + 4 * h) =
                    make_float4(acc[c][4 * j] * scale, acc[c][4 * j + 1] * scale, acc[c][4 * j + 2] * scale, acc[c][4 * j + 3] * scale);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// keys on lanes.  Per query block:  S = Q_blk.K^T,  dP = dO_blk.V^T  (rows = queries),
//   P = exp2(S - lse[row]),  dS = P (dP - delta[row]),
//   dV^T (chan x keys) += dO_blk^T . P,   dK^T (chan x keys) += Q_blk^T . dS
__global__ __launch_bounds__(ATT_T, 1) void attn_bwd_dkv_bf16_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                     const float* __restrict__ v, const float* __restrict__ d_o,
                                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                                     float* __restrict__ dk, float* __restrict__ dv, int T,
                                                                     float scale) {
    // 4 tiles x 2 buffers + the row statistics = 64.5 KB: dynamic LDS (above the 64 KB static limit)
    extern __shared__ __attribute__((aligned(16))) char dkv_smem[];
    char (*Qs)[TILE] = reinterpret_cast<char (*)[TILE]>(dkv_smem);
    char (*Ds)[TILE] = reinterpret_cast<char (*)[TILE]>(dkv_smem + 2 * TILE);
    char (*Qt)[TILE] = reinterpret_cast<char (*)[TILE]>(dkv_smem + 4 * TILE);
    char (*Dt)[TILE] = reinterpret_cast<char (*)[TILE]>(dkv_smem + 6 * TILE);
    float (*Ls)[KB] = reinterpret_cast<float (*)[KB]>(dkv_smem + 8 * TILE);
    float (*Dl)[KB] = reinterpret_cast<float (*)[KB]>(dkv_smem + 8 * TILE + 2 * KB * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int k0 = blockIdx.x * (4 * QW) + wave * QW;
    const int kl = lane & 31, h = lane >> 5;
    const long base = (long)n * T * HD;
    const float* qn = q + base;
    const float* dn = d_o + base;
    const int ki = k0 + kl < T ? k0 + kl : T - 1;
    bf16x8 kfr[HD / 16], vfr[HD / 16];
    own_frags(k + base + (long)ki * HD, h, scale * kLog2e, kfr);
    own_frags(v + base + (long)ki * HD, h, 1.f, vfr);
    f32x16 akk[HD / 32], avv[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) { zero(akk[c]); zero(avv[c]); }

    float pl = 0.f, pdl = 0.f;
    auto load_stats = [&](int qb) {
        if (tid < KB) { pl = lse[(long)n * T + qb * KB + tid] * kLog2e; pdl = delta[(long)n * T + qb * KB + tid]; }
    };
    auto store_all = [&](int buf, const Stage& sq, const Stage& sd) {
        store_rows(Qs[buf], sq, tid); store_transposed(Qt[buf], sq, tid);
        store_rows(Ds[buf], sd, tid); store_transposed(Dt[buf], sd, tid);
        if (tid < KB) { Ls[buf][tid] = pl; Dl[buf][tid] = pdl; }
    };
    const int nqb = T / KB;
    Stage sq = load_stage(qn, 0, tid), sd = load_stage(dn, 0, tid);
    load_stats(0);
    store_all(0, sq, sd);
    __syncthreads();
    for (int qb = 0; qb < nqb; ++qb) {
        const int buf = qb & 1;
        if (qb + 1 < nqb) { sq = load_stage(qn, (qb + 1) * KB, tid); sd = load_stage(dn, (qb + 1) * KB, tid); load_stats(qb + 1); }
        f32x16 s, dp;
        zero(s); zero(dp);
#pragma unroll
        for (int g = 0; g < HD / 16; ++g) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qs[buf], kl, g, h), kfr[g], s, 0, 0, 0);    // row = query lane&31
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ds[buf], kl, g, h), vfr[g], dp, 0, 0, 0);
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
        const bf16x8 p0 = acc_frag(s, 0), p1 = acc_frag(s, 1), d0 = acc_frag(dp, 0), d1 = acc_frag(dp, 1);
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
            avv[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Dt[buf], 32 * c + kl, 0, h), p0, avv[c], 0, 0, 0);
            avv[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Dt[buf], 32 * c + kl, 1, h), p1, avv[c], 0, 0, 0);
            akk[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt[buf], 32 * c + kl, 0, h), d0, akk[c], 0, 0, 0);
            akk[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qt[buf], 32 * c + kl, 1, h), d1, akk[c], 0, 0, 0);
        }
        if (qb + 1 < nqb) store_all(buf ^ 1, sq, sd);
        __syncthreads();
    }
    if (k0 + kl < T) {
        float* ok = dk + base + (long)(k0 + kl) * HD;
        float* ov = dv + base + (long)(k0 + kl) * HD;
#pragma unroll
        for (int c = 0; c < HD / 32; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = 32 * c + 8 * j + 4 * h;
                *reinterpret_cast<float4*>(ok + cc) = make_float4(akk[c][4 * j] * scale, akk[c][4 * j + 1] * scale,
                                                                  akk[c][4 * j + 2] * scale, akk[c][4 * j + 3] * scale);
                *reinterpret_cast<float4*>(ov + cc) = make_float4(avv[c][4 * j], avv[c][4 * j + 1], avv[c][4 * j + 2], avv[c][4 * j + 3]);
            }
    }
}

// delta[n][t] = sum_c dO[t][c] * O[t][c]   (fp32; memory-bound)
__global__ __launch_bounds__(256) void attn_delta_bf16_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
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

}  // namespace

extern "C" int pd_attn_bf16_fwd(const void* q, const void* k, const void* v, void* o, void* lse, int N, int T, int C,
                                float scale, void* stream) {
    PD_REQUIRE(N >= 0 && T > 0, "pd_attn_bf16_fwd: bad shape N=%d T=%d", N, T);
    PD_REQUIRE(C == HD, "pd_attn_bf16_fwd: head dimension must be %d (got %d)", HD, C);
    PD_REQUIRE(T % KB == 0, "pd_attn_bf16_fwd: the token count must be a multiple of %d (got %d)", KB, T);
    if (N == 0) return PD_OK;
    PD_REQUIRE(q && k && v && o && lse, "pd_attn_bf16_fwd: null tensor");
    PD_REQUIRE(pd::aligned16(q) && pd::aligned16(k) && pd::aligned16(v) && pd::aligned16(o), "pd_attn_bf16_fwd: tensors must be 16-byte aligned");
    const dim3 grid((unsigned)((T + 4 * QW - 1) / (4 * QW)), (unsigned)N);
    hipLaunchKernelGGL(attn_fwd_bf16_kernel, grid, dim3(ATT_T), 0, (hipStream_t)stream, (const float*)q, (const float*)k,
                       (const float*)v, (float*)o, (float*)lse, T, scale * kLog2e);
    return pd::check_launch("pd_attn_bf16_fwd");
}

extern "C" int pd_attn_bf16_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const void* lse,
                                void* delta, void* dq, void* dk, void* dv, int N, int T, int C, float scale, void* stream) {
    PD_REQUIRE(N >= 0 && T > 0, "pd_attn_bf16_bwd: bad shape N=%d T=%d", N, T);
    PD_REQUIRE(C == HD, "pd_attn_bf16_bwd: head dimension must be %d (got %d)", HD, C);
    PD_REQUIRE(T % KB == 0, "pd_attn_bf16_bwd: the token count must be a multiple of %d (got %d)", KB, T);
    if (N == 0) return PD_OK;
    PD_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "pd_attn_bf16_bwd: null tensor");
    PD_REQUIRE(pd::aligned16(q) && pd::aligned16(k) && pd::aligned16(v) && pd::aligned16(o) && pd::aligned16(d_o) &&
                   pd::aligned16(dq) && pd::aligned16(dk) && pd::aligned16(dv),
               "pd_attn_bf16_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long ntok = (long)N * T;
    hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3((unsigned)((ntok + 7) / 8)), dim3(256), 0, st, (const float*)o,
                       (const float*)d_o, (float*)delta, ntok);
    const dim3 grid((unsigned)((T + 4 * QW - 1) / (4 * QW)), (unsigned)N);
    constexpr int kDkvLds = 8 * TILE + 4 * KB * 4;
    static const hipError_t lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_bf16_kernel),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kDkvLds);
    PD_REQUIRE(lds_ok == hipSuccess, "pd_attn_bf16_bwd: cannot reserve %d bytes of LDS", kDkvLds);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel, grid, dim3(ATT_T), kDkvLds, st, (const float*)q, (const float*)k, (const float*)v,
                       (const float*)d_o, (const float*)lse, (const float*)delta, (float*)dk, (float*)dv, T, scale);
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, grid, dim3(ATT_T), 0, st, (const float*)q, (const float*)k, (const float*)v,
                       (const float*)d_o, (const float*)lse, (const float*)delta, (float*)dq, T, scale);
    return pd::check_launch("pd_attn_bf16_bwd");
}
