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
// ([channel][token], 16-byte units holding a fragment's eight tokens in fragment order), which attn_pack_kernel writes next
// to the row-major tile.
#include "pd_common.h"
#include <cstdlib>
#include <type_traits>

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

// v_exp_f32 as it is (1 ulp).  exp2f() wraps it in a denormal-range fix-up (compare, select, add, exp, ldexp: five
// instructions per score, ~70 of the ~250 vector instructions a wave issues per 32-key block beside its 16 MFMAs); the
// arguments here are score - max <= 0 or score - lse, and a result that underflows to zero instead of a denormal is beyond
// what the bf16 probabilities resolve.  Forward 0.46 -> 0.43 ms; with the operands pre-packed into bf16 tile images (below:
// no conversion / address arithmetic in the loop) 0.37 ms.  Measured without effect or worse on top of that (round 3): a
// prefetch distance of two blocks, two query groups per wave (half the LDS reads per MFMA, but one wave per SIMD:
// 0.79 ms), packed fp32 softmax arithmetic with compile-time LDS buffers (0.39 ms); 64 keys per step with two independent
// score chains, one softmax pass and one barrier per 32 MFMAs, tiles staged by direct-to-LDS loads (64 KB of LDS: two
// workgroups per CU): 0.39 ms, and with five or six waves per workgroup (one round of the chip instead of 1.25) 0.56-0.60.
// What bounds the loop is LDS bandwidth: with 32 queries per wave every A fragment (1 KiB of K rows or V^T rows) read
// from LDS feeds exactly ONE 32x32x16 MFMA (32 cycles of one SIMD); four SIMDs at full rate would need 128 B/clk -- all the
// LDS of a CU delivers -- and the 8-byte reads of the transposed tile run at half that.  A quarter of the bf16 peak is the
// ceiling of this layout; beyond it a wave has to keep 64 queries (two B operands per A fragment), i.e. one wave per SIMD
// with the softmax of one query group scheduled by hand into the MFMAs of the other.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
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
// transposed [128 ch][32 tokens] bf16: 64-byte rows of four 16-byte units; unit 2s + h holds the eight tokens of the A
// fragment of k-step s, half-wave h, IN FRAGMENT ORDER (16s + 4h + 0..3, then 16s + 8 + 4h + 0..3): one ds_read_b128 per
// fragment (two ds_read_b64 of a token-ordered row ran the LDS at half its rate -- and the LDS feeds every MFMA of these
// kernels, see fast_exp2).  Units XOR-swizzled by channel / 2: eight consecutive channels cover all 32 banks.
__device__ __forceinline__ int tr_off(int ch, int unit16) { return ch * 64 + 16 * (unit16 ^ ((ch >> 1) & 3)); }

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
    // tokens 4u .. 4u+3 = k-step u >> 2, upper / lower four of the fragment (u >> 1) & 1, half-wave u & 1
    const int unit = 2 * (u >> 2) + (u & 1), half = 8 * ((u >> 1) & 1);
    *reinterpret_cast<bf16x4*>(tile + tr_off(4 * c4 + 0, unit) + half) = cvt4(s.v[0].x, s.v[1].x, s.v[2].x, s.v[3].x);
    *reinterpret_cast<bf16x4*>(tile + tr_off(4 * c4 + 1, unit) + half) = cvt4(s.v[0].y, s.v[1].y, s.v[2].y, s.v[3].y);
    *reinterpret_cast<bf16x4*>(tile + tr_off(4 * c4 + 2, unit) + half) = cvt4(s.v[0].z, s.v[1].z, s.v[2].z, s.v[3].z);
    *reinterpret_cast<bf16x4*>(tile + tr_off(4 * c4 + 3, unit) + half) = cvt4(s.v[0].w, s.v[1].w, s.v[2].w, s.v[3].w);
}
// A fragment of k-step g (channels 16g + 8h ..) of row `tok` of a row-major tile
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int tok, int g, int h) {
    return *reinterpret_cast<const bf16x8*>(tile + row_off(tok, 2 * g + h));
}
// A fragment of k-step s (tokens 16s + 8(j>>2) + 4h + (j&3)) of channel row `ch` of a transposed tile
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int ch, int s, int h) {
    return *reinterpret_cast<const bf16x8*>(tile + tr_off(ch, 2 * s + h));
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

// ---- operands packed once per call: bf16 tile IMAGES in global memory
// Every workgroup of the attention kernels stages every block of the other operand; converting fp32 -> bf16 and scattering
// into the swizzled layouts there repeated that work 40 times per image, on the VALU, beside the MFMAs.  attn_pack_kernel
// writes, per 32-token block, the row-major and / or the transposed LDS tile verbatim (8 KB each); the attention kernels
// then copy a tile with two 16-byte loads + two ds_write_b128 per thread: no conversion, no address arithmetic.
// several tensors in ONE launch (blockIdx.y selects the job): the forward pass packs K and V, the backward pass K, V, Q and dO
struct PackJobs { const float* x[4]; char* rows[4]; char* trans[4]; };
__global__ __launch_bounds__(ATT_T) void attn_pack_multi_kernel(const PackJobs jobs, long nblocks) {
    const float* x = jobs.x[blockIdx.y];
    char* rows = jobs.rows[blockIdx.y];
    char* trans = jobs.trans[blockIdx.y];
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const Stage s = load_stage(x + b * (long)KB * HD, 0, threadIdx.x);
        if (rows) store_rows(rows + b * TILE, s, threadIdx.x);
        if (trans) store_transposed(trans + b * TILE, s, threadIdx.x);
    }
}

struct TileRegs { uint4 a, b; };
__device__ __forceinline__ TileRegs load_tile(const char* __restrict__ g, int tid) {
    TileRegs t;
    t.a = *reinterpret_cast<const uint4*>(g + 16 * tid);
    t.b = *reinterpret_cast<const uint4*>(g + TILE / 2 + 16 * tid);
    return t;
}
__device__ __forceinline__ void store_tile(char* lds, const TileRegs& t, int tid) {
    *reinterpret_cast<uint4*>(lds + 16 * tid) = t.a;
    *reinterpret_cast<uint4*>(lds + TILE / 2 + 16 * tid) = t.b;
}

// ------------------------------------------------------------------------------------------------ forward
// Three workgroups per CU: 16 x 5120 queries are 640 workgroups of 128 -- 1.25 rounds of the chip at two per CU, one at three.
__global__ __launch_bounds__(ATT_T, 3) void attn_fwd_bf16_kernel(const float* __restrict__ q, const char* __restrict__ kr,
                                                                 const char* __restrict__ vt, float* __restrict__ o,
                                                                 float* __restrict__ lse, int T, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char Ks[2][TILE];
    __shared__ __attribute__((aligned(16))) char Vt[2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int q0 = blockIdx.x * (4 * QW) + wave * QW;
    const int ql = lane & 31, h = lane >> 5;
    const int nkb = T / KB;
    const char* kn = kr + (long)n * nkb * TILE;
    const char* vn = vt + (long)n * nkb * TILE;
    bf16x8 qb[HD / 16];
    own_frags(q + ((long)n * T + (q0 + ql < T ? q0 + ql : T - 1)) * HD, h, scale_log2e, qb);
    f32x16 oacc[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) zero(oacc[c]);
    float m_run = -INFINITY, l_run = 0.f;

    TileRegs k1 = load_tile(kn, tid), v1 = load_tile(vn, tid);
    store_tile(Ks[0], k1, tid);
    store_tile(Vt[0], v1, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) { k1 = load_tile(kn + (long)(kb + 1) * TILE, tid); v1 = load_tile(vn + (long)(kb + 1) * TILE, tid); }
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
        const float alpha = fast_exp2(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(s[r] - m_new); lsum += s[r]; }
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
        if (kb + 1 < nkb) { store_tile(Ks[buf ^ 1], k1, tid); store_tile(Vt[buf ^ 1], v1, tid); }
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

// max / sum of a value over the two half-waves (lanes l and l ^ 32) on the VALU: v_permlane32_swap (gfx950) instead of the
// ds_bpermute of __shfl_xor, whose LDS round trip sat in the middle of the softmax's dependency chain
__device__ __forceinline__ float halves_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float halves_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// direct-to-LDS copy of 1 KiB (one wave instruction: LDS address = M0 + 16 * lane), issued from inline asm: no staging
// registers, and the compiler -- which drains vmcnt in front of every LDS read once it knows of a pending LDS-DMA -- does not
// see it; the loop retires it with an explicit s_waitcnt vmcnt(0) in front of its barrier.  (M0 is a reserved register of
// the backend, which rewrites it in front of each of its own uses: csrc/conv.hip:dma16.)
typedef __attribute__((address_space(3))) void att_lds_t;
__device__ __forceinline__ void att_dma16(__amdgpu_buffer_rsrc_t r, const void* lds_wave_base, unsigned voff) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(att_lds_t*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(m0v), "v"(voff), "s"(r) : "memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t att_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__device__ __forceinline__ const char* uniform_cptr(const char* p) {      // a wave-uniform pointer the compiler keeps in an SGPR pair
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u >> 32));
    return reinterpret_cast<const char*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// ------------------------------------------------------------------------------------------------ forward, software-pipelined
// SQ counters of the loop above (tools/sq_prof_attn.sh, profiles/r03_attention_sq_counters.txt): the vector pipe is the
// bottleneck, not the matrix pipe -- 8 to 10 vector instructions per MFMA (17 v_exp_f32 at quarter rate, 17 subtractions,
// 18 sums, the O rescale in ~45 % of the key blocks, ~30 LDS / global address instructions, 8 accumulator clears per 16
// MFMAs) against a matrix pipe that is busy a third of the time; and the wave runs S MFMAs, softmax and P.V MFMAs one after
// the other.  This kernel removes vector work and overlaps the rest:
//   * lazy maximum: the scores leave the matrix pipe already shifted -- the accumulator of S starts at -m instead of 0 -- and m
//     only moves when a block exceeds it by more than 2^8 (then O, l and the scores in flight are rescaled in a cold path):
//     no subtraction per score, no rescale of O in the common case.  Exact: any offset cancels in O / l, and exp2 of a
//     shifted score stays below 2^8 (bf16 keeps its relative precision);
//   * the scores of block kb + 1 are computed WHILE the softmax of block kb runs: the eight chained S MFMAs are issued one
//     by one between the exponentials (source order is issue order: sched_barrier), so each finds its predecessor finished;
//   * four K buffers / two V buffers addressed statically (the loop is unrolled four times): fragment addresses are eight
//     loop-invariant registers + immediate offsets; tiles go global -> LDS by direct-to-LDS loads (no staging registers);
//   * the half-wave exchanges of the softmax are v_permlane32_swap (VALU) instead of ds_bpermute (an LDS round trip).
// 48 KB of LDS, three workgroups per CU.  Needs T % 128 == 0.
__global__ __launch_bounds__(ATT_T, 3) void attn_fwd_bf16_pipe_kernel(const float* __restrict__ q, const char* __restrict__ kr,
                                                                      const char* __restrict__ vt, float* __restrict__ o,
                                                                      float* __restrict__ lse, int T, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char Ks[4][TILE];
    __shared__ __attribute__((aligned(16))) char Vt[2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int q0 = blockIdx.x * (4 * QW) + wave * QW;
    const int ql = lane & 31, h = lane >> 5;
    const int nkb = T / KB;
    const unsigned img_bytes = (unsigned)((long)nkb * TILE);
    const __amdgpu_buffer_rsrc_t rk = att_rsrc(kr + (long)n * img_bytes, img_bytes);
    const __amdgpu_buffer_rsrc_t rv = att_rsrc(vt + (long)n * img_bytes, img_bytes);
    // a tile image = eight 1-KiB pieces: wave w copies pieces 2w and 2w + 1
    const unsigned dma_lane = (2 * wave) * 1024 + lane * 16;
    const unsigned wave_lds = __builtin_amdgcn_readfirstlane((unsigned)((2 * wave) * 1024));      // uniform: SALU arithmetic below
    auto dma_tile = [&](const __amdgpu_buffer_rsrc_t r, const char* lds, int blk) {
        const unsigned m0a = (unsigned)(size_t)(att_lds_t*)lds + wave_lds;
        const unsigned off = (unsigned)blk * TILE + dma_lane;
        // (the second piece: + 1 KiB through M0 for the LDS side and through the scalar offset for the global side -- an
        //  instruction offset would be added to both)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(m0a), "v"(off), "s"(r), "s"(1024u), "s"(m0a + 1024u) : "memory");
    };
    bf16x8 qb[HD / 16];
    own_frags(q + ((long)n * T + (q0 + ql < T ? q0 + ql : T - 1)) * HD, h, scale_log2e, qb);
    // loop-invariant fragment offsets inside a tile (the buffer and, for V^T, the channel tile are immediates)
    int koff[HD / 16];
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) koff[g] = row_off(ql, 2 * g + h);
    const int voff0 = tr_off(ql, h), voff1 = tr_off(ql, 2 + h);          // k-steps 0 / 1; channel tile c adds c * 32 * 64 bytes
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) asm volatile("" : "+v"(koff[g]));   // keep them as registers (no re-derivation per block)
    f32x16 oacc[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) zero(oacc[c]);

    // prologue: K0, V0, K1 in LDS; S of block 0, unshifted; m = its exact row maximum
    dma_tile(rk, Ks[0], 0); dma_tile(rv, Vt[0], 0); dma_tile(rk, Ks[1], 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 sA, sB;                       // the score tiles of blocks kb (even: sA) and kb + 1: they alternate roles, no copies
    zero(sA);
#pragma unroll
    for (int g = 0; g < HD / 16; ++g)
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Ks[0] + koff[g]), qb[g], sA, 0, 0, 0);
    float m_run = sA[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m_run = fmaxf(m_run, sA[r]);
    m_run = halves_max(m_run);
#pragma unroll
    for (int r = 0; r < 16; ++r) sA[r] -= m_run;
    float l_run = 0.f;

#define PD_PIN() __builtin_amdgcn_sched_barrier(0)
    // one key block; J = kb % 4 (compile time): K of block kb in Ks[J], of kb + 1 in Ks[J + 1], V of kb in Vt[J & 1]
    auto step = [&](auto jtag, int kb) {
        constexpr int J = decltype(jtag)::value;
        f32x16& s = (J & 1) ? sB : sA;
        f32x16& sn = (J & 1) ? sA : sB;
        const char* const Kn = Ks[(J + 1) & 3];
        const char* const Vc = Vt[J & 1];
        dma_tile(rk, Ks[(J + 2) & 3], kb + 2 < nkb ? kb + 2 : nkb - 1);      // (clamped at the end: those copies are unused)
        dma_tile(rv, Vt[(J + 1) & 1], kb + 1 < nkb ? kb + 1 : nkb - 1);
        // ---- S^T of block kb + 1, shifted by -m, in the shadow of the softmax of block kb
        {
            const float nm = -m_run;
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[r] = nm;
        }
        bf16x8 ka0 = *reinterpret_cast<const bf16x8*>(Kn + koff[0]), ka1 = *reinterpret_cast<const bf16x8*>(Kn + koff[1]), ka2;
        PD_PIN();
        float t = fmaxf(fmaxf(fmaxf(s[0], s[1]), s[2]), fmaxf(fmaxf(s[3], s[4]), s[5]));
        t = fmaxf(fmaxf(t, s[6]), s[7]);
        PD_PIN();
        sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qb[0], sn, 0, 0, 0);
        ka2 = *reinterpret_cast<const bf16x8*>(Kn + koff[2]);
        PD_PIN();
        float u = fmaxf(fmaxf(fmaxf(s[8], s[9]), s[10]), fmaxf(fmaxf(s[11], s[12]), s[13]));
        u = fmaxf(fmaxf(u, s[14]), s[15]);
        t = halves_max(fmaxf(t, u));                 // the block's largest shifted score of this query
        PD_PIN();
        sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, qb[1], sn, 0, 0, 0);
        ka0 = *reinterpret_cast<const bf16x8*>(Kn + koff[3]);
        PD_PIN();
        // the maximum moves only when a score exceeds it by more than 2^8 (cold: O, l, and both score tiles are rescaled)
        float shift = 0.f;
        if (__builtin_expect(__any(t > 8.f), 0)) {
            shift = fmaxf(t, 0.f);
            const float a = fast_exp2(-shift);
            m_run += shift;
            l_run *= a;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] -= shift;
#pragma unroll
            for (int c = 0; c < HD / 32; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[c][r] *= a;
        }
        float lsum = 0.f;
        // six MFMAs, behind each the exponentials of (up to) three scores: 3 x (v_exp at quarter rate, v_add)
#define PD_STEP(G, KA, KNEXT, R0, R1, R2, NR)                                                              \
        sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KA, qb[G], sn, 0, 0, 0);                              \
        if ((G) + 2 < HD / 16) KNEXT = *reinterpret_cast<const bf16x8*>(Kn + koff[((G) + 2) & 7]);        \
        PD_PIN();                                                                                          \
        s[R0] = fast_exp2(s[R0]); lsum += s[R0];                                                           \
        if ((NR) > 1) { s[R1] = fast_exp2(s[R1]); lsum += s[R1]; }                                         \
        if ((NR) > 2) { s[R2] = fast_exp2(s[R2]); lsum += s[R2]; }                                         \
        PD_PIN();
        PD_STEP(2, ka2, ka1, 0, 1, 2, 3)
        PD_STEP(3, ka0, ka2, 3, 4, 5, 3)
        PD_STEP(4, ka1, ka0, 6, 7, 8, 3)
        PD_STEP(5, ka2, ka1, 9, 10, 11, 3)
        PD_STEP(6, ka0, ka2, 12, 13, 13, 2)
        PD_STEP(7, ka1, ka0, 14, 15, 15, 2)
#undef PD_STEP
        l_run += halves_sum(lsum);
        const bf16x8 p0 = acc_frag(s, 0), p1 = acc_frag(s, 1);
        if (__builtin_expect(shift != 0.f, 0)) {      // (per lane: the scores in flight were shifted by the old maximum)
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[r] -= shift;
        }
        // ---- O^T += V_blk^T . P^T
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
            oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Vc + c * 2048 + voff0), p0, oacc[c], 0, 0, 0);
            oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Vc + c * 2048 + voff1), p1, oacc[c], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the tiles to come
        __syncthreads();                                       // everyone's pieces; everyone done with this block's tiles
    };
    for (int kb = 0; kb < nkb; kb += 4) {
        step(std::integral_constant<int, 0>{}, kb);
        step(std::integral_constant<int, 1>{}, kb + 1);
        step(std::integral_constant<int, 2>{}, kb + 2);
        step(std::integral_constant<int, 3>{}, kb + 3);
    }
#undef PD_PIN
    const int qi = q0 + ql;
    if (qi < T) {
        const float inv = 1.f / l_run;
        float* on = o + ((long)n * T + qi) * HD;
#pragma unroll
        for (int c = 0; c < HD / 32; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(on + 32 * c + 8 * j + 4 * h) =
                    make_float4(oacc[c][4 * j] * inv, oacc[c][4 * j + 1] * inv, oacc[c][4 * j + 2] * inv, oacc[c][4 * j + 3] * inv);
        if (h == 0) lse[(long)n * T + qi] = (m_run + log2f(l_run)) * 0.6931471805599453f;
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// queries on lanes.  Per key block:  S^T = K.Q^T,  dP^T = V.dO^T,  dS = exp2(S - lse) (dP - delta),
//   dQ^T (chan x queries) += K_blk^T (chan x keys) . dS (keys x queries)
__global__ __launch_bounds__(ATT_T, 2) void attn_bwd_dq_bf16_kernel(const float* __restrict__ q, const char* __restrict__ kr,
                                                                    const char* __restrict__ kt, const char* __restrict__ vr,
                                                                    const float* __restrict__ d_o,
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
    const int nkb = T / KB;
    const char* krn = kr + (long)n * nkb * TILE;
    const char* ktn = kt + (long)n * nkb * TILE;
    const char* vrn = vr + (long)n * nkb * TILE;
    const int qi = q0 + ql < T ? q0 + ql : T - 1;
    bf16x8 qb[HD / 16], dob[HD / 16];
    own_frags(q + base + (long)qi * HD, h, scale * kLog2e, qb);
    own_frags(d_o + base + (long)qi * HD, h, 1.f, dob);
    const float lse2 = lse[(long)n * T + qi] * kLog2e;
    const float dl = delta[(long)n * T + qi];
    f32x16 acc[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) zero(acc[c]);

    TileRegs a1 = load_tile(krn, tid), b1 = load_tile(ktn, tid), c1 = load_tile(vrn, tid);
    store_tile(Ks[0], a1, tid); store_tile(Kt[0], b1, tid); store_tile(Vs[0], c1, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) {
            const long off = (long)(kb + 1) * TILE;
            a1 = load_tile(krn + off, tid); b1 = load_tile(ktn + off, tid); c1 = load_tile(vrn + off, tid);
        }
        f32x16 s, dp;
        zero(s); zero(dp);
#pragma unroll
        for (int g = 0; g < HD / 16; ++g) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks[buf], ql, g, h), qb[g], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Vs[buf], ql, g, h), dob[g], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r] - lse2) * (dp[r] - dl);      // dS (keys x queries), fp32
        const bf16x8 d0 = acc_frag(s, 0), d1 = acc_frag(s, 1);
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kt[buf], 32 * c + ql, 0, h), d0, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Kt[buf], 32 * c + ql, 1, h), d1, acc[c], 0, 0, 0);
        }
        if (kb + 1 < nkb) { store_tile(Ks[buf ^ 1], a1, tid); store_tile(Kt[buf ^ 1], b1, tid); store_tile(Vs[buf ^ 1], c1, tid); }
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

// ------------------------------------------------------------------------------------------------ backward: dQ, pipelined
// The same treatment as the forward pass.  lse and delta of the lane's query are constants of the whole loop, so the score
// accumulators start at -lse (log2 units) and the dP accumulators at -delta: P = exp2(S') and dS = P * dP' need one exponential
// and one multiplication per score, nothing else.  The 16 chained MFMAs of block kb + 1 (S and dP, two chains taking turns)
// are issued one by one between the exponentials / products of block kb; tiles arrive by direct-to-LDS loads into statically
// addressed double buffers (loop unrolled twice).  Needs T % 64 == 0.
__global__ __launch_bounds__(ATT_T, 2) void attn_bwd_dq_bf16_pipe_kernel(const float* __restrict__ q, const char* __restrict__ kr,
                                                                         const char* __restrict__ kt, const char* __restrict__ vr,
                                                                         const float* __restrict__ d_o,
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
    const int nkb = T / KB;
    const unsigned img_bytes = (unsigned)((long)nkb * TILE);
    const __amdgpu_buffer_rsrc_t rkr = att_rsrc(kr + (long)n * img_bytes, img_bytes);
    const __amdgpu_buffer_rsrc_t rkt = att_rsrc(kt + (long)n * img_bytes, img_bytes);
    const __amdgpu_buffer_rsrc_t rvr = att_rsrc(vr + (long)n * img_bytes, img_bytes);
    const unsigned dma_lane = (2 * wave) * 1024 + lane * 16;
    const unsigned wave_lds = __builtin_amdgcn_readfirstlane((unsigned)((2 * wave) * 1024));
    auto dma_tile = [&](const __amdgpu_buffer_rsrc_t r, const char* lds, int blk) {      // eight 1-KiB pieces, two per wave
        const unsigned m0a = (unsigned)(size_t)(att_lds_t*)lds + wave_lds;
        const unsigned off = (unsigned)blk * TILE + dma_lane;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(m0a), "v"(off), "s"(r), "s"(1024u), "s"(m0a + 1024u) : "memory");
    };
    const int qi = q0 + ql < T ? q0 + ql : T - 1;
    bf16x8 qb[HD / 16], dob[HD / 16];
    own_frags(q + base + (long)qi * HD, h, scale * kLog2e, qb);
    own_frags(d_o + base + (long)qi * HD, h, 1.f, dob);
    const float nlse = -lse[(long)n * T + qi] * kLog2e;
    const float ndl = -delta[(long)n * T + qi];
    int koff[HD / 16];
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) koff[g] = row_off(ql, 2 * g + h);
    const int toff0 = tr_off(ql, h), toff1 = tr_off(ql, 2 + h);
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) asm volatile("" : "+v"(koff[g]));
    f32x16 acc[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) zero(acc[c]);

    // prologue: rows of blocks 0 and 1, K^T of block 0; S' and dP' of block 0
    dma_tile(rkr, Ks[0], 0); dma_tile(rvr, Vs[0], 0); dma_tile(rkt, Kt[0], 0);
    dma_tile(rkr, Ks[1], 1); dma_tile(rvr, Vs[1], 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 sA, pA, sB, pB;               // (S', dP') of blocks kb (even: A) and kb + 1: they alternate roles
#pragma unroll
    for (int r = 0; r < 16; ++r) { sA[r] = nlse; pA[r] = ndl; }
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) {
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Ks[0] + koff[g]), qb[g], sA, 0, 0, 0);
        pA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Vs[0] + koff[g]), dob[g], pA, 0, 0, 0);
    }
#define PD_PIN() __builtin_amdgcn_sched_barrier(0)
    // one key block; J = kb % 2 (compile time): rows of block kb + 1 in Ks / Vs[J ^ 1], K^T of block kb in Kt[J]
    auto step = [&](auto jtag, int kb) {
        constexpr int J = decltype(jtag)::value;
        f32x16& s = J ? sB : sA;
        f32x16& dp = J ? pB : pA;
        f32x16& sn = J ? sA : sB;
        f32x16& dn = J ? pA : pB;
        const char* const Kn = Ks[J ^ 1];
        const char* const Vn = Vs[J ^ 1];
        const char* const Ktc = Kt[J];
        // rows of block kb + 2 replace those of block kb (read during the previous step), K^T of block kb + 1 that of kb - 1
        dma_tile(rkr, Ks[J], kb + 2 < nkb ? kb + 2 : nkb - 1);
        dma_tile(rvr, Vs[J], kb + 2 < nkb ? kb + 2 : nkb - 1);
        dma_tile(rkt, Kt[J ^ 1], kb + 1 < nkb ? kb + 1 : nkb - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { sn[r] = nlse; dn[r] = ndl; }
        bf16x8 ka = *reinterpret_cast<const bf16x8*>(Kn + koff[0]), va = *reinterpret_cast<const bf16x8*>(Vn + koff[0]);
        bf16x8 kb2, vb2;
        PD_PIN();
        // sixteen MFMAs (the S and dP chains of block kb + 1 taking turns), behind each one score of block kb:
        // P = exp2(S'), dS = P * dP'  (v_exp at quarter rate + v_mul)
#define PD_PAIR(G, KA, VA, KB2, VB2)                                                                        \
        sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(KA, qb[G], sn, 0, 0, 0);                              \
        if ((G) + 1 < HD / 16) KB2 = *reinterpret_cast<const bf16x8*>(Kn + koff[((G) + 1) & 7]);          \
        PD_PIN();                                                                                          \
        s[2 * (G)] = fast_exp2(s[2 * (G)]) * dp[2 * (G)];                                                  \
        PD_PIN();                                                                                          \
        dn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(VA, dob[G], dn, 0, 0, 0);                             \
        if ((G) + 1 < HD / 16) VB2 = *reinterpret_cast<const bf16x8*>(Vn + koff[((G) + 1) & 7]);          \
        PD_PIN();                                                                                          \
        s[2 * (G) + 1] = fast_exp2(s[2 * (G) + 1]) * dp[2 * (G) + 1];                                      \
        PD_PIN();
        PD_PAIR(0, ka, va, kb2, vb2)
        PD_PAIR(1, kb2, vb2, ka, va)
        PD_PAIR(2, ka, va, kb2, vb2)
        PD_PAIR(3, kb2, vb2, ka, va)
        PD_PAIR(4, ka, va, kb2, vb2)
        PD_PAIR(5, kb2, vb2, ka, va)
        PD_PAIR(6, ka, va, kb2, vb2)
        PD_PAIR(7, kb2, vb2, ka, va)
#undef PD_PAIR
        const bf16x8 d0 = acc_frag(s, 0), d1 = acc_frag(s, 1);       // dS (keys x queries)
        // ---- dQ^T (chan x queries) += K_blk^T (chan x keys) . dS
#pragma unroll
        for (int c = 0; c < HD / 32; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Ktc + c * 2048 + toff0), d0, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Ktc + c * 2048 + toff1), d1, acc[c], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        step(std::integral_constant<int, 0>{}, kb);
        step(std::integral_constant<int, 1>{}, kb + 1);
    }
#undef PD_PIN
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
__global__ __launch_bounds__(ATT_T, 1) void attn_bwd_dkv_bf16_kernel(const char* __restrict__ qr, const char* __restrict__ qt,
                                                                     const float* __restrict__ k, const float* __restrict__ v,
                                                                     const char* __restrict__ dr, const char* __restrict__ dt,
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
    const int nqb = T / KB;
    const char* qrn = qr + (long)n * nqb * TILE;
    const char* qtn = qt + (long)n * nqb * TILE;
    const char* drn = dr + (long)n * nqb * TILE;
    const char* dtn = dt + (long)n * nqb * TILE;
    const int ki = k0 + kl < T ? k0 + kl : T - 1;
    bf16x8 kfr[HD / 16], vfr[HD / 16];
    own_frags(k + base + (long)ki * HD, h, scale * kLog2e, kfr);
    own_frags(v + base + (long)ki * HD, h, 1.f, vfr);
    f32x16 akk[HD / 32], avv[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) { zero(akk[c]); zero(avv[c]); }

    struct St { TileRegs qr, qt, dr, dt; float pl, pdl; };
    auto load_all = [&](int qb) {
        St t;
        const long off = (long)qb * TILE;
        t.qr = load_tile(qrn + off, tid); t.qt = load_tile(qtn + off, tid);
        t.dr = load_tile(drn + off, tid); t.dt = load_tile(dtn + off, tid);
        t.pl = 0.f; t.pdl = 0.f;
        if (tid < KB) { t.pl = lse[(long)n * T + qb * KB + tid] * kLog2e; t.pdl = delta[(long)n * T + qb * KB + tid]; }
        return t;
    };
    auto store_all = [&](int buf, const St& t) {
        store_tile(Qs[buf], t.qr, tid); store_tile(Qt[buf], t.qt, tid);
        store_tile(Ds[buf], t.dr, tid); store_tile(Dt[buf], t.dt, tid);
        if (tid < KB) { Ls[buf][tid] = t.pl; Dl[buf][tid] = t.pdl; }
    };
    St s1 = load_all(0);
    store_all(0, s1);
    __syncthreads();
    for (int qb = 0; qb < nqb; ++qb) {
        const int buf = qb & 1;
        if (qb + 1 < nqb) s1 = load_all(qb + 1);
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
                const float p = fast_exp2(s[4 * j + i] - lv[i]);
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
        if (qb + 1 < nqb) store_all(buf ^ 1, s1);
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

// ------------------------------------------------------------------------------------------------ backward: dK, dV, pipelined
// One wave per SIMD (the two 32x128 accumulators + the wave's own K and V leave no room for a second), so the overlap of the
// matrix and the vector pipe has to come from inside the wave: the 16 chained MFMAs of query block qb + 1 (S' = Q.K^T - lse and
// dP' = dO.V^T - delta, the accumulators starting at the rows' -lse / -delta read from LDS) are issued one by one between the
// exponentials / products of block qb (P = exp2(S'), dS = P * dP'), then the 16 MFMAs of dV^T += dO^T.P and dK^T += Q^T.dS.
// Tiles arrive by direct-to-LDS loads into statically addressed double buffers (loop unrolled twice).  Needs T % 64 == 0.
__global__ __launch_bounds__(ATT_T, 1) void attn_bwd_dkv_bf16_pipe_kernel(const char* __restrict__ qr, const char* __restrict__ qt,
                                                                          const float* __restrict__ k, const float* __restrict__ v,
                                                                          const char* __restrict__ dr, const char* __restrict__ dt,
                                                                          const float* __restrict__ nstat,
                                                                          float* __restrict__ dk, float* __restrict__ dv, int T, int N,
                                                                          float scale) {
    extern __shared__ __attribute__((aligned(16))) char dkvp_smem[];
    // Four slots per tile kind (128 KB): tiles are requested two blocks before their first use.  The LDS base of a
    // direct-to-LDS load is the low 16 bits of M0: only the first 64 KB can be its target (measured: a target above lands
    // 64 KB lower).  The row tiles (the ones S' / dP' need first) live there and arrive by direct-to-LDS loads; the transposed
    // tiles and the row statistics live above and pass through registers -- loaded by inline-asm loads two steps ahead
    // (invisible to the compiler, whose own loads would be waited for with a full drain), written to LDS a step later.
    char (*Qs)[TILE] = reinterpret_cast<char (*)[TILE]>(dkvp_smem);
    char (*Ds)[TILE] = reinterpret_cast<char (*)[TILE]>(dkvp_smem + 4 * TILE);
    char (*Qt)[TILE] = reinterpret_cast<char (*)[TILE]>(dkvp_smem + 8 * TILE);
    char (*Dt)[TILE] = reinterpret_cast<char (*)[TILE]>(dkvp_smem + 12 * TILE);
    float (*St)[2 * KB] = reinterpret_cast<float (*)[2 * KB]>(dkvp_smem + 16 * TILE);       // per slot: -lse (log2 units) | -delta of the rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int k0 = blockIdx.x * (4 * QW) + wave * QW;
    const int kl = lane & 31, h = lane >> 5;
    const long base = (long)n * T * HD;
    const int nqb = T / KB;
    const unsigned img_bytes = (unsigned)((long)nqb * TILE);
    const __amdgpu_buffer_rsrc_t rqr = att_rsrc(qr + (long)n * img_bytes, img_bytes);
    const __amdgpu_buffer_rsrc_t rdr = att_rsrc(dr + (long)n * img_bytes, img_bytes);
    const char* const qt_n = uniform_cptr(qt + (long)n * img_bytes);
    const char* const dt_n = uniform_cptr(dt + (long)n * img_bytes);
    const unsigned dma_lane = (2 * wave) * 1024 + lane * 16;
    const unsigned wave_lds = __builtin_amdgcn_readfirstlane((unsigned)((2 * wave) * 1024));
    auto dma_tile = [&](const __amdgpu_buffer_rsrc_t r, const char* lds, int blk) {      // eight 1-KiB pieces, two per wave
        const unsigned m0a = (unsigned)(size_t)(att_lds_t*)lds + wave_lds;
        const unsigned off = (unsigned)blk * TILE + dma_lane;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(m0a), "v"(off), "s"(r), "s"(1024u), "s"(m0a + 1024u) : "memory");
    };
    const int ki = k0 + kl < T ? k0 + kl : T - 1;
    bf16x8 kfr[HD / 16], vfr[HD / 16];
    own_frags(k + base + (long)ki * HD, h, scale * kLog2e, kfr);
    own_frags(v + base + (long)ki * HD, h, 1.f, vfr);
    int koff[HD / 16];
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) koff[g] = row_off(kl, 2 * g + h);
    const int toff0 = tr_off(kl, h), toff1 = tr_off(kl, 2 + h);
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) asm volatile("" : "+v"(koff[g]));
    f32x16 akk[HD / 32], avv[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) { zero(akk[c]); zero(avv[c]); }
    // register-staged operands: a transposed tile pair (thread t copies bytes [16 t, +16) and [TILE / 2 + 16 t, +16) of each image)
    // and the block's 64 statistics (attn_delta_bf16_kernel: nstat = [-lse * log2 e | -delta], N * T floats each; lanes 0..31 /
    // 32..63 of every wave load the same values, wave 0 writes them)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));       // (a register tuple: the asm operands below need one)
    struct Staged { u32x4 q0, q1, d0, d1; float st; };
    const char* const st_n = uniform_cptr(reinterpret_cast<const char*>(nstat));
    const unsigned st_lane = (unsigned)(((lane >> 5) * (long)N * T + (long)n * T + (lane & 31)) * 4);
    auto issue_staged = [&](Staged& r, int blk_t, int blk_s) {          // five loads
        const unsigned off = (unsigned)blk_t * TILE + 16u * tid;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.q0) : "v"(off), "s"(qt_n) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.q1) : "v"(off + TILE / 2), "s"(qt_n) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.d0) : "v"(off), "s"(dt_n) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.d1) : "v"(off + TILE / 2), "s"(dt_n) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "=v"(r.st) : "v"(st_lane + (unsigned)blk_s * (KB * 4)), "s"(st_n) : "memory");
    };
    auto store_staged = [&](const Staged& r, int slot_t, int slot_s) {
        *reinterpret_cast<u32x4*>(Qt[slot_t] + 16 * tid) = r.q0;
        *reinterpret_cast<u32x4*>(Qt[slot_t] + TILE / 2 + 16 * tid) = r.q1;
        *reinterpret_cast<u32x4*>(Dt[slot_t] + 16 * tid) = r.d0;
        *reinterpret_cast<u32x4*>(Dt[slot_t] + TILE / 2 + 16 * tid) = r.d1;
        if (wave == 0) St[slot_s][lane] = r.st;
    };
    // the accumulators of a block start at its rows' statistics: element r of the tile is row (r & 3) + 8 (r >> 2) + 4 h
    auto init_rows = [&](f32x16& a, const float* stat) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(stat + 8 * j + 4 * h);
            a[4 * j] = t.x; a[4 * j + 1] = t.y; a[4 * j + 2] = t.z; a[4 * j + 3] = t.w;
        }
    };
    // prologue: rows of blocks 0..2 (direct to LDS), transposed tiles of blocks 0 and 1 + statistics of blocks 0..2 (registers)
    dma_tile(rqr, Qs[0], 0); dma_tile(rdr, Ds[0], 0);
    dma_tile(rqr, Qs[1], 1); dma_tile(rdr, Ds[1], 1);
    dma_tile(rqr, Qs[2], 2); dma_tile(rdr, Ds[2], 2);
    Staged sgA, sgB;                     // staged operands requested at even / odd steps
    issue_staged(sgA, 0, 0);
    issue_staged(sgB, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sgA.q0), "+v"(sgA.q1), "+v"(sgA.d0), "+v"(sgA.d1), "+v"(sgA.st), "+v"(sgB.q0), "+v"(sgB.q1),
                 "+v"(sgB.d0), "+v"(sgB.d1), "+v"(sgB.st) :: "memory");
    store_staged(sgA, 0, 0);
    store_staged(sgB, 1, 1);
    issue_staged(sgB, 1, 2);             // (its transposed pair is rewritten unchanged at the end of step 0, its statistics are block 2's)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sgB.q0), "+v"(sgB.q1), "+v"(sgB.d0), "+v"(sgB.d1), "+v"(sgB.st) :: "memory");
    store_staged(sgB, 1, 2);
    issue_staged(sgB, 1, 2);             // in flight into step 0, which retires it like any other step's
    __syncthreads();
    f32x16 sA, pA, sB, pB;               // (S', dP') of blocks qb (even: A) and qb + 1: they alternate roles
    init_rows(sA, St[0]); init_rows(pA, St[0] + KB);
#pragma unroll
    for (int g = 0; g < HD / 16; ++g) {
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Qs[0] + koff[g]), kfr[g], sA, 0, 0, 0);
        pA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Ds[0] + koff[g]), vfr[g], pA, 0, 0, 0);
    }
#define PD_PIN() __builtin_amdgcn_sched_barrier(0)
    // one query block; J = qb % 4: rows + statistics of block qb + 1 in slot J + 1, transposed tiles of block qb in slot J
    auto step = [&](auto jtag, int qb) {
        constexpr int J = decltype(jtag)::value;
        f32x16& s = (J & 1) ? sB : sA;
        f32x16& dp = (J & 1) ? pB : pA;
        f32x16& sn = (J & 1) ? sA : sB;
        f32x16& dn = (J & 1) ? pA : pB;
        const char* const Qn = Qs[(J + 1) & 3];
        const char* const Dn = Ds[(J + 1) & 3];
        const char* const Qtc = Qt[J];
        const char* const Dtc = Dt[J];
        // rows of block qb + 3 into the slot of block qb - 1 (direct to LDS: four operations); transposed tiles of block qb + 2
        // and statistics of block qb + 3 into this step's register set (five): nine operations per wave and step
        Staged& mine = (J & 1) ? sgB : sgA;          // requested now, written to LDS at the end of the NEXT step
        Staged& older = (J & 1) ? sgA : sgB;         // requested a step ago, written at the end of this one
        const int b3 = qb + 3 < nqb ? qb + 3 : nqb - 1, b2 = qb + 2 < nqb ? qb + 2 : nqb - 1;
        dma_tile(rqr, Qs[(J + 3) & 3], b3); dma_tile(rdr, Ds[(J + 3) & 3], b3);
        issue_staged(mine, b2, b3);
        init_rows(sn, St[(J + 1) & 3]); init_rows(dn, St[(J + 1) & 3] + KB);
        // Row fragments of block qb + 1 three MFMA slots ahead (a ring of four per chain), the sixteen transposed fragments of
        // block qb one per slot.  (All sixteen row fragments up front cost 48 more registers and pushed the score tiles -- which
        // the vector pipe works on -- into AGPRs: a v_accvgpr move around every exponential.)
        bf16x8 qf[4], df[4], tf[4];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            qf[g] = *reinterpret_cast<const bf16x8*>(Qn + koff[g]);
            df[g] = *reinterpret_cast<const bf16x8*>(Dn + koff[g]);
        }
        PD_PIN();
        // sixteen MFMAs (the S and dP chains of block qb + 1 taking turns), behind each one score of block qb:
        // P = exp2(S') (kept in s), dS = P * dP' (kept in dp)
        // transposed fragment i of the second phase (i = 0..15): channel tile i >> 2, k-step (i >> 1) & 1, dO^T (even i) or Q^T
#define PD_TADDR(I) ((((I) & 1) ? Qtc : Dtc) + ((I) >> 2) * 2048 + ((((I) >> 1) & 1) ? toff1 : toff0))
#define PD_PAIR(G)                                                                                          \
        sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[(G) & 3], kfr[G], sn, 0, 0, 0);                    \
        if ((G) + 3 < HD / 16) qf[((G) + 3) & 3] = *reinterpret_cast<const bf16x8*>(Qn + koff[((G) + 3) & 7]); \
        PD_PIN();                                                                                          \
        s[2 * (G)] = fast_exp2(s[2 * (G)]); dp[2 * (G)] *= s[2 * (G)];                                     \
        PD_PIN();                                                                                          \
        dn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[(G) & 3], vfr[G], dn, 0, 0, 0);                    \
        if ((G) + 3 < HD / 16) df[((G) + 3) & 3] = *reinterpret_cast<const bf16x8*>(Dn + koff[((G) + 3) & 7]); \
        if ((G) >= 5) tf[(G) - 5] = *reinterpret_cast<const bf16x8*>(PD_TADDR((G) - 5));                 \
        PD_PIN();                                                                                          \
        s[2 * (G) + 1] = fast_exp2(s[2 * (G) + 1]); dp[2 * (G) + 1] *= s[2 * (G) + 1];                     \
        PD_PIN();
        PD_PAIR(0) PD_PAIR(1) PD_PAIR(2) PD_PAIR(3) PD_PAIR(4) PD_PAIR(5) PD_PAIR(6) PD_PAIR(7)
#undef PD_PAIR
        const bf16x8 p0 = acc_frag(s, 0), p1 = acc_frag(s, 1), d0 = acc_frag(dp, 0), d1 = acc_frag(dp, 1);
        // ---- dV^T (chan x keys) += dO_blk^T . P,   dK^T (chan x keys) += Q_blk^T . dS: sixteen MFMAs, their transposed
        // fragments three slots ahead (a ring of four; the first three were requested behind the last S' / dP' MFMAs)
#define PD_MF(I)                                                                                               \
        if ((I) & 1) akk[(I) >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[(I) & 3], ((I) & 2) ? d1 : d0, akk[(I) >> 2], 0, 0, 0); \
        else avv[(I) >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[(I) & 3], ((I) & 2) ? p1 : p0, avv[(I) >> 2], 0, 0, 0);       \
        if ((I) + 3 < 16) tf[((I) + 3) & 3] = *reinterpret_cast<const bf16x8*>(PD_TADDR((I) + 3));             \
        PD_PIN();
        PD_MF(0) PD_MF(1) PD_MF(2) PD_MF(3) PD_MF(4) PD_MF(5) PD_MF(6) PD_MF(7)
        PD_MF(8) PD_MF(9) PD_MF(10) PD_MF(11) PD_MF(12) PD_MF(13) PD_MF(14) PD_MF(15)
#undef PD_MF
#undef PD_TADDR
        // what the NEXT step reads was requested a step ago: the nine operations of this step may stay in flight.  The older
        // register set has landed: transposed tiles of block qb + 1 -> slot J + 1, statistics of block qb + 2 -> slot J + 2
        asm volatile("s_waitcnt vmcnt(9)" : "+v"(older.q0), "+v"(older.q1), "+v"(older.d0), "+v"(older.d1), "+v"(older.st) :: "memory");
        store_staged(older, (J + 1) & 3, (J + 2) & 3);
        __syncthreads();
    };
    for (int qb = 0; qb < nqb; qb += 4) {
        step(std::integral_constant<int, 0>{}, qb);
        step(std::integral_constant<int, 1>{}, qb + 1);
        step(std::integral_constant<int, 2>{}, qb + 2);
        step(std::integral_constant<int, 3>{}, qb + 3);
    }
#undef PD_PIN
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
// nstat (optional): [-lse * log2 e | -delta], ntok floats each -- the form the pipelined dK / dV kernel starts its accumulators from
__global__ __launch_bounds__(256) void attn_delta_bf16_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
                                                              float* __restrict__ delta, long ntok,
                                                              const float* __restrict__ lse, float* __restrict__ nstat) {
    const long t = blockIdx.x * 8L + (threadIdx.x >> 5);      // 32 lanes per token, float4 each
    if (t >= ntok) return;
    const int l = threadIdx.x & 31;
    const float4 a = ld4g(o + t * HD + 4 * l), b = ld4g(d_o + t * HD + 4 * l);
    float sum = (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    if (l == 0) {
        delta[t] = sum;
        if (nstat) { nstat[t] = -lse[t] * kLog2e; nstat[ntok + t] = -sum; }
    }
}

}  // namespace

extern "C" size_t pd_attn_bf16_workspace(int N, int T, int C, int backward) {
    if (N <= 0 || T <= 0 || C != HD || T % KB) return 0;
    // bf16 tile images: forward K rows + V^T; backward Kr Kt Vr Qr Qt dOr dOt + the negated row statistics (2 N T floats)
    return (size_t)N * T * HD * 2 * (backward ? 7 : 2) + (backward ? (size_t)N * T * 8 : 0);
}

static void pack_multi(const PackJobs& jobs, int njobs, long nblocks, hipStream_t st) {
    hipLaunchKernelGGL(attn_pack_multi_kernel, dim3((unsigned)(nblocks > 4096 ? 4096 : nblocks), (unsigned)njobs), dim3(ATT_T), 0, st,
                       jobs, nblocks);
}

extern "C" int pd_attn_bf16_fwd(const void* q, const void* k, const void* v, void* o, void* lse, void* workspace,
                                size_t ws_bytes, int N, int T, int C, float scale, void* stream) {
    PD_REQUIRE(N >= 0 && T > 0, "pd_attn_bf16_fwd: bad shape N=%d T=%d", N, T);
    PD_REQUIRE(C == HD, "pd_attn_bf16_fwd: head dimension must be %d (got %d)", HD, C);
    PD_REQUIRE(T % KB == 0, "pd_attn_bf16_fwd: the token count must be a multiple of %d (got %d)", KB, T);
    if (N == 0) return PD_OK;
    PD_REQUIRE(q && k && v && o && lse && workspace, "pd_attn_bf16_fwd: null tensor");
    PD_REQUIRE(ws_bytes >= pd_attn_bf16_workspace(N, T, C, 0), "pd_attn_bf16_fwd: workspace too small");
    PD_REQUIRE(pd::aligned16(q) && pd::aligned16(k) && pd::aligned16(v) && pd::aligned16(o) && pd::aligned16(workspace),
               "pd_attn_bf16_fwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const size_t one = (size_t)N * T * HD * 2;
    char* kr = (char*)workspace; char* vt = kr + one;
    const long nblocks = (long)N * T / KB;
    {
        PackJobs jobs{};
        jobs.x[0] = (const float*)k; jobs.rows[0] = kr;
        jobs.x[1] = (const float*)v; jobs.trans[1] = vt;
        pack_multi(jobs, 2, nblocks, st);
    }
    const dim3 grid((unsigned)((T + 4 * QW - 1) / (4 * QW)), (unsigned)N);
    if (T % (4 * KB) == 0 && nblocks / N >= 4)
        hipLaunchKernelGGL(attn_fwd_bf16_pipe_kernel, grid, dim3(ATT_T), 0, st, (const float*)q, (const char*)kr,
                           (const char*)vt, (float*)o, (float*)lse, T, scale * kLog2e);
    else
        hipLaunchKernelGGL(attn_fwd_bf16_kernel, grid, dim3(ATT_T), 0, st, (const float*)q, (const char*)kr,
                           (const char*)vt, (float*)o, (float*)lse, T, scale * kLog2e);
    return pd::check_launch("pd_attn_bf16_fwd");
}

extern "C" int pd_attn_bf16_bwd_parts(const void* q, const void* k, const void* v, const void* o, const void* d_o, const void* lse,
                                      void* delta, void* dq, void* dk, void* dv, void* workspace, size_t ws_bytes, int N, int T,
                                      int C, float scale, int parts, void* stream);

extern "C" int pd_attn_bf16_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const void* lse,
                                void* delta, void* dq, void* dk, void* dv, void* workspace, size_t ws_bytes, int N, int T, int C,
                                float scale, void* stream) {
    return pd_attn_bf16_bwd_parts(q, k, v, o, d_o, lse, delta, dq, dk, dv, workspace, ws_bytes, N, T, C, scale, 7, stream);
}

// parts: 1 = delta + operand packing, 2 = dK / dV, 4 = dQ.  The two gradient kernels are independent of each other: a caller
// that enqueues parts 2 and 4 on two streams behind part 1 lets the workgroups of one fill the last, partly empty round of
// the other (their grids are 2.5 and 1.25 rounds of the chip).
extern "C" int pd_attn_bf16_bwd_parts(const void* q, const void* k, const void* v, const void* o, const void* d_o, const void* lse,
                                      void* delta, void* dq, void* dk, void* dv, void* workspace, size_t ws_bytes, int N, int T,
                                      int C, float scale, int parts, void* stream) {
    PD_REQUIRE(parts > 0 && parts <= 7, "pd_attn_bf16_bwd_parts: parts must be a combination of 1 | 2 | 4");
    PD_REQUIRE(N >= 0 && T > 0, "pd_attn_bf16_bwd: bad shape N=%d T=%d", N, T);
    PD_REQUIRE(C == HD, "pd_attn_bf16_bwd: head dimension must be %d (got %d)", HD, C);
    PD_REQUIRE(T % KB == 0, "pd_attn_bf16_bwd: the token count must be a multiple of %d (got %d)", KB, T);
    if (N == 0) return PD_OK;
    PD_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv && workspace, "pd_attn_bf16_bwd: null tensor");
    PD_REQUIRE(ws_bytes >= pd_attn_bf16_workspace(N, T, C, 1), "pd_attn_bf16_bwd: workspace too small");
    PD_REQUIRE(pd::aligned16(q) && pd::aligned16(k) && pd::aligned16(v) && pd::aligned16(o) && pd::aligned16(d_o) &&
                   pd::aligned16(dq) && pd::aligned16(dk) && pd::aligned16(dv) && pd::aligned16(workspace),
               "pd_attn_bf16_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long ntok = (long)N * T;
    const size_t one = (size_t)N * T * HD * 2;
    char* w = (char*)workspace;
    float* nstat = reinterpret_cast<float*>(w + 7 * one);
    if (parts & 1)
        hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3((unsigned)((ntok + 7) / 8)), dim3(256), 0, st, (const float*)o,
                           (const float*)d_o, (float*)delta, ntok, (const float*)lse, nstat);
    char *kr = w, *kt = w + one, *vr = w + 2 * one, *qr = w + 3 * one, *qt = w + 4 * one, *dr = w + 5 * one, *dt = w + 6 * one;
    const long nblocks = ntok / KB;
    if (parts & 1) {
        PackJobs jobs{};
        jobs.x[0] = (const float*)k; jobs.rows[0] = kr; jobs.trans[0] = kt;
        jobs.x[1] = (const float*)v; jobs.rows[1] = vr;
        jobs.x[2] = (const float*)q; jobs.rows[2] = qr; jobs.trans[2] = qt;
        jobs.x[3] = (const float*)d_o; jobs.rows[3] = dr; jobs.trans[3] = dt;
        pack_multi(jobs, 4, nblocks, st);
    }
    const dim3 grid((unsigned)((T + 4 * QW - 1) / (4 * QW)), (unsigned)N);
    constexpr int kDkvLds = 8 * TILE + 4 * KB * 4, kDkvPipeLds = 16 * TILE + 8 * KB * 4;
    if (!(parts & 2)) {
    } else if (T % (4 * KB) == 0 && 2L * N * T * 4 < (1L << 31)) {
        static const hipError_t lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_bf16_pipe_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kDkvPipeLds);
        PD_REQUIRE(lds_ok == hipSuccess, "pd_attn_bf16_bwd: cannot reserve %d bytes of LDS", kDkvPipeLds);
        hipLaunchKernelGGL(attn_bwd_dkv_bf16_pipe_kernel, grid, dim3(ATT_T), kDkvPipeLds, st, (const char*)qr, (const char*)qt,
                           (const float*)k, (const float*)v, (const char*)dr, (const char*)dt, (const float*)nstat,
                           (float*)dk, (float*)dv, T, N, scale);
    } else {
        static const hipError_t lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_bf16_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kDkvLds);
        PD_REQUIRE(lds_ok == hipSuccess, "pd_attn_bf16_bwd: cannot reserve %d bytes of LDS", kDkvLds);
        hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel, grid, dim3(ATT_T), kDkvLds, st, (const char*)qr, (const char*)qt, (const float*)k,
                           (const float*)v, (const char*)dr, (const char*)dt, (const float*)lse, (const float*)delta, (float*)dk,
                           (float*)dv, T, scale);
    }
    if (!(parts & 4)) {
    } else if (T % (2 * KB) == 0)
        hipLaunchKernelGGL(attn_bwd_dq_bf16_pipe_kernel, grid, dim3(ATT_T), 0, st, (const float*)q, (const char*)kr, (const char*)kt,
                           (const char*)vr, (const float*)d_o, (const float*)lse, (const float*)delta, (float*)dq, T, scale);
    else
        hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, grid, dim3(ATT_T), 0, st, (const float*)q, (const char*)kr, (const char*)kt,
                           (const char*)vr, (const float*)d_o, (const float*)lse, (const float*)delta, (float*)dq, T, scale);
    return pd::check_launch("pd_attn_bf16_bwd");
}
