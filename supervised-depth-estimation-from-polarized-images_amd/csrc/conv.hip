// K2 -- implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// One kernel template serves
//   * forward convolution, zero padding      (nn.Conv2d in pre_encoders.py:15-25, torchvision resnet18)
//   * forward convolution, reflection padding (layers.py:364-380 Conv3x3 = ReflectionPad2d(1) + Conv2d(3))
//   * data gradient (transposed convolution, any stride) of either
// as C[m][n] = sum_k A[m][k] * Bt[n][k]  ("NT" GEMM, both operands K-contiguous):
//   m = output pixel (n, oh, ow), n = output channel, k = (kh, kw, cin)
//   A is gathered on the fly from the NHWC (or arbitrarily strided) activation tensor,
//   Bt is the weight tensor in [Cout][kh][kw][Cin] order (= torch channels_last storage).
// Tiles: BM x BN x 32 per 256-thread workgroup, staged through LDS (rows padded to 36 floats so
// that ds_read_b128 fragment reads and ds_write_b128 stores are bank-conflict free), software
// double-buffered (global->register prefetch of chunk q+1 while chunk q feeds the MFMAs).
// Each lane reads four consecutive k per ds_read_b128; lanes 0-31 take k = 8g..8g+3 and lanes
// 32-63 take k = 8g+4..8g+7, which is a legal permutation of the contraction index because A and
// B use the same one.
// Epilogue: bias, activation (none / ReLU / ELU / sigmoid), optional per-workgroup column
// sums and sums of squares (training-mode BatchNorm statistics, reduced later in fp64),
// NHWC store with an arbitrary row stride (lets a layer write into a slice of a concat buffer).
//
// The weight-gradient kernel (contraction over pixels, "TN" GEMM) lives below in the same file.
#include "pd_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;        // contraction chunk
constexpr int LDT = BK;       // LDS row = 128 B; 16-byte slots are XOR-swizzled: conflict-free b128 reads and writes
constexpr int NT = 256;       // threads per workgroup (4 waves)
constexpr unsigned OOB = 0x80000000u;   // byte offset beyond every buffer extent (< 2 GiB, checked on the host)

enum { MODE_ZERO = 0, MODE_REFLECT = 1, MODE_TRANSPOSED = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_SIGMOID = 3 };

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* oscale;     // per-output-channel multiplier of the accumulator (folded BatchNorm) or null
    float* y;
    float* stats;            // [gridM][Co][2] or null
    const float* add;        // NHWC tensor added to the result before the store (row stride ld_add) or null; may alias y
    long ld_add;
    int N, H, W, C;          // logical dims of x
    long sN, sH, sW, sC;     // element strides of x
    int Ho, Wo, Co;          // output grid / channels
    int KH, KW, stride, pad;
    int pad_w;               // column padding when it differs from `pad` (rows); uniform-tap kernel only
    int mode, act;
    int affine;
    float sub, div;
    int K;
    long M;
    long ldy;                // row stride of y in elements
    int mtiles, ntiles;
    int sshift;              // log2(stride) (transposed mode: stride is a power of two)
    unsigned w_bytes;
    unsigned mg_hw, sh_hw, mg_wo, sh_wo;   // n / (Ho*Wo) and n / Wo as mulhi + shift (n < 2^31), see magic_div()
    unsigned flags;          // PD_CONV_* kernel-family selection of the caller (include/polardepth.h)
    int stats_rows;          // output rows per row of `stats` = pd_conv2d_tile_m(M, Co): the one rule host and kernels share
    int nmajor;              // bf16-split kernels: consecutive workgroups (one XCD's L2) walk the row tiles of ONE column tile
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// float offset of 16-byte slot `slot` (0..7) in LDS row `row`: slot ^ ((row >> 1) & 7).  With 128-byte rows two
// rows cover the 64 banks; the XOR makes the eight even (odd) rows of every ds_read_b128 lane group hit
// eight different slots, and the eight slots of one row (a ds_write_b128 group) are always distinct.
__device__ __forceinline__ int swz(int row, int slot) { return 4 * (slot ^ ((row >> 1) & 7)); }

// Hardware-bounds-checked loads: an offset >= the descriptor's extent returns 0 without a branch.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// Input coordinate of tap (kh,kw) for a row whose (rh,rw) were prepared for MODE:
//   zero / reflect: rh = oh*stride - pad        -> ih = rh + kh
//   transposed    : rh = oh + pad               -> ih = (rh - kh) / stride when divisible
template <int MODE>
__device__ __forceinline__ bool tap_in(int rh, int rw, int kh, int kw, int H, int W, int sshift, int& ih, int& iw) {
    if (MODE == MODE_TRANSPOSED) {
        const int th = rh - kh, tw = rw - kw;
        const int smask = (1 << sshift) - 1;
        ih = th >> sshift; iw = tw >> sshift;
        return ((th | tw) >= 0) & (((th | tw) & smask) == 0) & (ih < H) & (iw < W);
    }
    ih = rh + kh; iw = rw + kw;
    if (MODE == MODE_REFLECT) {
        ih = ih < 0 ? -ih : ih; ih = ih >= H ? 2 * H - 2 - ih : ih;
        iw = iw < 0 ? -iw : iw; iw = iw >= W ? 2 * W - 2 - iw : iw;
        return true;
    }
    return ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
}

typedef __attribute__((address_space(3))) void lds_ptr_t;

// One direct-to-LDS piece (1 KiB per wave: LDS address = M0 + 16 * lane), issued from inline asm: told about a pending
// LDS-DMA, hipcc drains vmcnt before EVERY fragment read (ds_read) of the loop, which serialises the prefetch;
// unseen, the piece stays in flight across the MFMAs and is retired by the explicit dma_wait() before the barrier.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, const void* lds_wave_base, unsigned voff) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(m0v), "v"(voff), "s"(r) : "memory");
    // (M0 cannot be declared clobbered: the AMDGPU backend treats it as a RESERVED register -- "inline asm clobber list
    //  contains reserved registers: m0" -- i.e. it never keeps a value in M0 across statements and rewrites M0 right in
    //  front of each of its own uses (LDS-param / readlane / GWS), so an asm statement that overwrites it is safe.)
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Epilogue shared by the implicit-GEMM kernels: bias / folded-BN scale, activation, per-tile BatchNorm partial sums,
// NHWC store.  C/D layout of the MFMA: col = lane % MT; 32x32: row = (r&3) + 8*(r>>2) + 4*(lane>>5), r < 16;
// 16x16: row = r + 4*(lane>>4), r < 4.
template <int BM, int BN, int WM, int WN, int MT, typename AccT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, AccT (&acc)[WM / MT][WN / MT], float* scratch, int mt,
                                              long m0, int n0, int wm, int wn, int lane, int wave, int tid) {
    constexpr int TM = WM / MT, TN = WN / MT, ACCN = MT == 32 ? 16 : 4, WAVES_N = BN / WN;
    float (*red)[WN][2] = reinterpret_cast<float (*)[WN][2]>(scratch);
    if constexpr (MT == 32 && WN == 32) {
        // Full tile, no bias / scale / activation (every BatchNorm-followed convolution and every data gradient): the
        // wave transposes its WM x 32 block through LDS (the A/B tiles are dead by now) and leaves with WM/8
        // buffer_store_dwordx4 of eight full 128-byte lines each instead of WM/2 dword stores, and the per-element
        // work shrinks to the two statistics updates -- on this chip every VALU instruction is taken from the
        // matrix pipe of the other workgroups on the SIMD (tools/mfma_peak.hip).
        const bool lean = a.M - m0 >= BM && n0 + BN <= a.Co && !a.bias && !a.oscale && a.act == ACT_NONE &&
                          (a.ldy & 3) == 0 && ((size_t)a.y & 15) == 0 &&
                          (!a.add || ((a.ld_add & 3) == 0 && ((size_t)a.add & 15) == 0));
        if (lean) {
            float* T = scratch + wave * (WM * WN);                 // [WM][32] floats, private to the wave
            red = reinterpret_cast<float (*)[WN][2]>(scratch + 4 * WM * WN);
            const int col_l = lane & 31, rbase = 4 * (lane >> 5);
            // addend (the residual gradient a data gradient is summed with): its loads fly during the transposition
            float4 addv[WM / 8];
            if (a.add) {
                const __amdgpu_buffer_rsrc_t ra = make_rsrc(a.add + m0 * a.ld_add, (unsigned)((long)BM * a.ld_add * 4));
                const unsigned off_a = (unsigned)((wm * WM + (lane >> 3)) * (int)a.ld_add + n0 + wn * WN + 4 * (lane & 7)) * 4u;
#pragma unroll
                for (int t = 0; t < WM / 8; ++t) addv[t] = buf_ld4(ra, off_a + (unsigned)(t * 8 * (int)a.ld_add * 4));
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][0][r];
                    T[(MT * i + (r & 3) + 8 * (r >> 2) + rbase) * WN + col_l] = v;
                    s1 += v;
                    s2 = __builtin_fmaf(v, v, s2);
                }
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y + m0 * a.ldy, (unsigned)((long)BM * a.ldy * 4));
            const unsigned off_l = (unsigned)((wm * WM + (lane >> 3)) * (int)a.ldy + n0 + wn * WN + 4 * (lane & 7)) * 4u;
            const float4* Tq = reinterpret_cast<const float4*>(T) + lane;      // row lane/8, column quad lane%8
#pragma unroll
            for (int t = 0; t < WM / 8; ++t) {
                float4 v = Tq[t * 64];
                if (a.add) { v.x += addv[t].x; v.y += addv[t].y; v.z += addv[t].z; v.w += addv[t].w; }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, off_l, t * 8 * (int)a.ldy * 4, 0);
            }
            if (a.stats) {
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (lane < 32) { red[wave][col_l][0] = s1; red[wave][col_l][1] = s2; }
                __syncthreads();
                if (tid < BN) {
                    const int w_n = tid / WN, cl = tid - w_n * WN;
                    float t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int w_m = 0; w_m < BM / WM; ++w_m) {
                        t1 += red[w_m * WAVES_N + w_n][cl][0];
                        t2 += red[w_m * WAVES_N + w_n][cl][1];
                    }
                    float* o = a.stats + ((long)mt * a.Co + n0 + tid) * 2;
                    o[0] = t1; o[1] = t2;
                }
            }
            return;
        }
    }
    // ---- epilogue: C/D layout of the MFMA: col = lane % MT; 32x32: row = (r&3) + 8*(r>>2) + 4*(lane>>5), r < 16;
    // 16x16: row = r + 4*(lane>>4), r < 4.
    // Straight-line code: the activation is a template argument, the row step of every store is a scalar offset
    // and out-of-range elements are dropped by the bounds check of the buffer store (offset OOB), so the stores
    // of a wave issue back to back (with per-element branches the compiler drained vmcnt to zero after each one).
    const int col_l = lane % MT;
    const int rbase = 4 * (lane / MT);
    {
        const long rows_left = a.M - m0;
        const int rows = rows_left < BM ? (int)rows_left : BM;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y + m0 * a.ldy, (unsigned)((long)rows * a.ldy * 4));
        const __amdgpu_buffer_rsrc_t rb_ = make_rsrc(a.bias ? a.bias : a.w, a.bias ? (unsigned)a.Co * 4u : 0u);
        const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(a.oscale ? a.oscale : a.w, a.oscale ? (unsigned)a.Co * 4u : 0u);
        const __amdgpu_buffer_rsrc_t ra_ = make_rsrc(a.add ? a.add + m0 * a.ld_add : a.w, a.add ? (unsigned)((long)rows * a.ld_add * 4) : 0u);
        auto body = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WN + MT * j + col_l;
                const bool cv = col < a.Co;
                const float bv = a.bias ? buf_ld1(rb_, cv ? (unsigned)col * 4u : OOB) : 0.f;
                const float sv = a.oscale ? buf_ld1(rs_, cv ? (unsigned)col * 4u : OOB) : 1.f;
                float s1 = 0.f, s2 = 0.f;
                const int row_l = wm * WM + rbase;                      // this lane's first row in the tile
                const unsigned off_l = (unsigned)(row_l * (int)a.ldy + col) * 4u;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < ACCN; ++r) {
                        const int rr = MT * i + (MT == 32 ? (r & 3) + 8 * (r >> 2) : r);   // compile-time row step: scalar offset
                        const bool ok = cv & (row_l < rows - rr);
                        float v = fmaf(acc[i][j][r], sv, bv);    // sv == 1: exactly acc + bv
                        const float vs = ok ? v : 0.f;
                        s1 += vs;
                        s2 += vs * vs;
                        if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
                        // ELU as torch evaluates it, exp(x) - 1, on the hardware exp2 (|err| < 2e-7 absolute; expm1f costs
                        // ~40 VALU instructions per element, which the fp32 matrix pipe pays for)
                        else if (ACT == ACT_ELU) v = v > 0.f ? v : __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.f;
                        else if (ACT == ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
                        if (a.add) v += buf_ld1(ra_, ok ? (unsigned)((row_l + rr) * (int)a.ld_add + col) * 4u : OOB);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, ok ? off_l : OOB,
                                                              rr * (int)a.ldy * 4, 0);
                    }
                }
                if (a.stats) {
#pragma unroll
                    for (int m = MT; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
                    if (lane < MT) { red[wave][MT * j + col_l][0] = s1; red[wave][MT * j + col_l][1] = s2; }
                }
            }
        };
        if (a.act == ACT_NONE) body(std::integral_constant<int, ACT_NONE>{});
        else if (a.act == ACT_RELU) body(std::integral_constant<int, ACT_RELU>{});
        else if (a.act == ACT_ELU) body(std::integral_constant<int, ACT_ELU>{});
        else body(std::integral_constant<int, ACT_SIGMOID>{});
    }
    if (a.stats) {
        __syncthreads();
        if (tid < BN) {   // column tid of the block tile: sum over the waves that own it
            const int w_n = tid / WN, cl = tid - w_n * WN;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w_m = 0; w_m < BM / WM; ++w_m) {
                t1 += red[w_m * WAVES_N + w_n][cl][0];
                t2 += red[w_m * WAVES_N + w_n][cl][1];
            }
            const int col = n0 + tid;
            if (col < a.Co) {
                float* o = a.stats + ((long)mt * a.Co + col) * 2;
                o[0] = t1; o[1] = t2;
            }
        }
    }
}

// DMA (vector path only): the A and B chunks go global -> LDS directly (buffer_load_dwordx4 ... lds: 1 KiB = eight
// 128-byte tile rows per wave instruction, lane l -> row l/8, physical slot l%8), no staging registers and no
// ds_write pass.  The LDS image stays the XOR-swizzled one the fragment reads expect: a thread loads the LOGICAL
// slot (l%8) ^ ((row/2)%8) of its row -- the swizzle moves to the source address (same for all of its rows).
template <int BM, int BN, int WM, int WN, bool VEC, int MODE, bool DMA = false>
__global__ __launch_bounds__(NT) void conv_igemm_kernel(const ConvArgs a) {
    static_assert(!DMA || VEC, "direct-to-LDS staging needs the 16-byte gather path");
    constexpr int WAVES_N = BN / WN;
    // MFMA shape: 32x32x2 tiles, or 16x16x4 tiles for the 16-wide block tile (layers with <= 16 output channels
    // would leave half of a 32-wide tile empty; both shapes have the same flops per cycle)
    constexpr int MT = BN == 16 ? 16 : 32;
    constexpr int TM = WM / MT, TN = WN / MT;     // MFMA tiles per wave
    constexpr int ACCN = MT == 32 ? 16 : 4;       // accumulator registers per tile
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
    constexpr int A_VEC_ITERS = BM / 32;          // 16-byte pieces per thread per chunk (A)
    constexpr int B_VEC_ITERS = (BN + 31) / 32;   // (rows >= BN are skipped)
    constexpr int A_SC_ITERS = BM / 8;            // scalar elements per thread per chunk (A)
    constexpr int B_SC_ITERS = BN / 8;

    // one LDS object (a second __shared__ array next to a direct-to-LDS destination makes hipcc drain vmcnt before
    // every fragment read): A tiles, B tiles, then the scalar path's row table
    __shared__ __attribute__((aligned(16))) float smem_all[2 * BM * LDT + 2 * BN * LDT + (VEC ? 0 : BM * 3)];
    float (*As)[BM][LDT] = reinterpret_cast<float (*)[BM][LDT]>(smem_all);
    float (*Bs)[BN][LDT] = reinterpret_cast<float (*)[BN][LDT]>(smem_all + 2 * BM * LDT);
    int (*rowinfo)[3] = reinterpret_cast<int (*)[3]>(smem_all + 2 * BM * LDT + 2 * BN * LDT);   // scalar path: element offset of the image (or -1), rh, rw

    // XCD-aware mapping: consecutive logical tiles (which share input halos / the A tile) land
    // on the same XCD and hence the same L2.  Grid is padded to a multiple of 8.
    const int nblk = a.mtiles * a.ntiles;
    const int per_xcd = (int)gridDim.x >> 3;
    const int logical = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (logical >= nblk) return;
    const int mt = logical / a.ntiles, nt = logical - mt * a.ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int hw = a.Ho * a.Wo;

    // Buffer descriptors.  The activation descriptor starts at the first image this tile touches, so
    // that 32-bit byte offsets stay small whatever the batch size.
    const int img0 = (int)(m0 / hw);
    const long rest = ((long)a.N - img0) * a.sN * 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)img0 * a.sN, rest > 0x7fffffffL ? 0x7fffffffu : (unsigned)rest);
    const __amdgpu_buffer_rsrc_t rw_ = make_rsrc(a.w, a.w_bytes);

    auto row_coords = [&](long m, int& rb, int& rh, int& rw) {
        if (m < a.M) {
            const int mr = (int)(m - (long)img0 * hw);      // pixel index relative to the tile's first image
            const int dn = mr / hw;
            const int rem = mr - dn * hw;
            const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            rb = dn * (int)a.sN;
            if (MODE == MODE_TRANSPOSED) { rh = oh + a.pad; rw = ow + a.pad; }
            else { rh = oh * a.stride - a.pad; rw = ow * a.stride - a.pad; }
        } else {
            rb = -1; rh = 0; rw = 0;
        }
    };

    // ---- per-thread gather state
    // Linear addressing (zero padding, or transposed with stride 1): the element offset of tap (kh,kw,c) for a row
    // is rowbase + tapoff with tapoff = +-(kh*sH + kw*sW) + c kept incrementally -- no multiplications in the loop.
    // Then rb[] holds rowbase and rows beyond M carry a poisoned rh that fails every bounds test.
    const bool lin = VEC && (MODE == MODE_ZERO || (MODE == MODE_TRANSPOSED && a.sshift == 0));
    const int dW = MODE == MODE_TRANSPOSED ? -(int)a.sW : (int)a.sW;
    const int dH = MODE == MODE_TRANSPOSED ? -(int)a.sH : (int)a.sH;
    int rb[VEC ? A_VEC_ITERS : 1], rh[VEC ? A_VEC_ITERS : 1], rwc[VEC ? A_VEC_ITERS : 1];
    if (VEC) {
#pragma unroll
        for (int i = 0; i < A_VEC_ITERS; ++i) {
            row_coords(m0 + (tid >> 3) + 32 * i, rb[i], rh[i], rwc[i]);
            if (lin) {
                if (rb[i] < 0) { rb[i] = 0; rh[i] = -(1 << 28); }
                else rb[i] += rh[i] * (int)a.sH + rwc[i] * (int)a.sW;
            }
        }
    } else {
        for (int r = tid; r < BM; r += NT) row_coords(m0 + r, rowinfo[r][0], rowinfo[r][1], rowinfo[r][2]);
        __syncthreads();
    }
    // tap of this thread's k column, advanced by BK per chunk (no divisions in the loop)
    int tk = VEC ? (DMA ? 4 * ((tid & 7) ^ ((tid >> 4) & 7)) : 4 * (tid & 7)) : (tid & 31);
    int tkh, tkw, tc, tapoff;
    {
        const int tap = tk / a.C;
        tc = tk - tap * a.C; tkh = tap / a.KW; tkw = tap - tkh * a.KW;
        tapoff = tkh * dH + tkw * dW + tc;
    }
    auto advance_tap = [&]() {
        tk += BK; tc += BK; tapoff += BK;
        while (tc >= a.C) {
            tc -= a.C; tapoff += dW - a.C;
            if (++tkw == a.KW) { tkw = 0; ++tkh; tapoff += dH - a.KW * dW; }
        }
    };

    const int nchunks = (a.K + BK - 1) / BK;
    float4 pa[VEC ? A_VEC_ITERS : 1], pb[VEC ? B_VEC_ITERS : 1];
    float sa[VEC ? 1 : A_SC_ITERS], sb[VEC ? 1 : B_SC_ITERS];

    auto load_chunk = [&](int dst) {   // loads the chunk the tap state points at (DMA: into LDS buffer dst), then advances it
        const bool kv = tk < a.K;
        if (VEC) {
#pragma unroll
            for (int i = 0; i < A_VEC_ITERS; ++i) {
                unsigned off;
                if (lin) {
                    const int ih = MODE == MODE_TRANSPOSED ? rh[i] - tkh : rh[i] + tkh;
                    const int iw = MODE == MODE_TRANSPOSED ? rwc[i] - tkw : rwc[i] + tkw;
                    const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & kv;
                    off = ok ? (unsigned)(rb[i] + tapoff) * 4u : OOB;
                } else {
                    int ih, iw;
                    const bool ok = tap_in<MODE>(rh[i], rwc[i], tkh, tkw, a.H, a.W, a.sshift, ih, iw) & kv & (rb[i] >= 0);
                    off = ok ? (unsigned)(rb[i] + ih * (int)a.sH + iw * (int)a.sW + tc) * 4u : OOB;
                }
                if (DMA) dma16(rx, &As[dst][8 * wave + 32 * i][0], off);
                else pa[i] = buf_ld4(rx, off);
            }
#pragma unroll
            for (int i = 0; i < B_VEC_ITERS; ++i) {
                const int nl = (tid >> 3) + 32 * i, nr = n0 + nl;
                const unsigned off = (kv & (nr < a.Co)) ? (unsigned)(nr * a.K + tk) * 4u : OOB;
                if (BN % 32 == 0 || 8 * wave + 32 * i < BN) {        // (wave-uniform: a wave covers eight rows)
                    if (DMA) dma16(rw_, &Bs[dst][8 * wave + 32 * i][0], off);
                    else if (BN % 32 == 0 || nl < BN) pb[i] = buf_ld4(rw_, off);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_SC_ITERS; ++i) {
                const int r = (tid >> 5) + 8 * i;
                const int rbv = rowinfo[r][0];
                int ih, iw;
                const bool ok = tap_in<MODE>(rowinfo[r][1], rowinfo[r][2], tkh, tkw, a.H, a.W, a.sshift, ih, iw) & kv & (rbv >= 0);
                const unsigned off = (unsigned)(rbv + ih * (int)a.sH + iw * (int)a.sW + tc * (int)a.sC) * 4u;
                float v = buf_ld1(rx, ok ? off : OOB);
                if (a.affine) v = ok ? (v - a.sub) / a.div : 0.f;
                sa[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B_SC_ITERS; ++i) {
                const int nr = n0 + (tid >> 5) + 8 * i;
                sb[i] = buf_ld1(rw_, (kv & (nr < a.Co)) ? (unsigned)(nr * a.K + tk) * 4u : OOB);
            }
        }
        advance_tap();
    };
    auto store_chunk = [&](int buf) {
        if (DMA) return;                 // the chunk is already on its way into LDS
        if (VEC) {
            const int pc = tid & 7;
#pragma unroll
            for (int i = 0; i < A_VEC_ITERS; ++i)
                *reinterpret_cast<float4*>(&As[buf][(tid >> 3) + 32 * i][swz((tid >> 3) + 32 * i, pc)]) = pa[i];
#pragma unroll
            for (int i = 0; i < B_VEC_ITERS; ++i)
                if (BN % 32 == 0 || (tid >> 3) + 32 * i < BN)
                    *reinterpret_cast<float4*>(&Bs[buf][(tid >> 3) + 32 * i][swz((tid >> 3) + 32 * i, pc)]) = pb[i];
        } else {
            const int kc = tid & 31;
#pragma unroll
            for (int i = 0; i < A_SC_ITERS; ++i) As[buf][(tid >> 5) + 8 * i][swz((tid >> 5) + 8 * i, kc >> 2) + (kc & 3)] = sa[i];
#pragma unroll
            for (int i = 0; i < B_SC_ITERS; ++i) Bs[buf][(tid >> 5) + 8 * i][swz((tid >> 5) + 8 * i, kc >> 2) + (kc & 3)] = sb[i];
        }
    };

    typedef float accv_t __attribute__((ext_vector_type(ACCN)));
    accv_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < ACCN; ++r) acc[i][j][r] = 0.f;

    load_chunk(0);
    store_chunk(0);
    if (DMA) dma_wait();
    __syncthreads();
    // Fragment reads: lane -> (row = lane % MT, slot group = lane / MT).  One ds_read_b128 hands a lane four
    // consecutive k; MFMA number c of a group contracts component c of every lane, i.e. k = 4*slot + c over the
    // 2 (32x32x2) or 4 (16x16x4) slot groups -- a permutation of k shared by A and B.
    constexpr int SG = 64 / MT;                    // slot groups per read: 2 or 4
    const int frow = lane % MT, fh = lane / MT;
    for (int q = 0; q < nchunks; ++q) {
        const int buf = q & 1;
        if (q + 1 < nchunks) load_chunk(buf ^ 1);
#pragma unroll
        for (int g = 0; g < BK / (4 * SG); ++g) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const float4*>(&As[buf][wm * WM + MT * i + frow][swz(wm * WM + MT * i + frow, SG * g + fh)]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const float4*>(&Bs[buf][wn * WN + MT * j + frow][swz(wn * WN + MT * j + frow, SG * g + fh)]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (MT == 32) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (q + 1 < nchunks) store_chunk(buf ^ 1);
        if (DMA) dma_wait();
        __syncthreads();
    }

    conv_epilogue<BM, BN, WM, WN, MT>(a, acc, &As[0][0][0], mt, m0, n0, wm, wn, lane, wave, tid);
}

// ===================================================================== uniform-tap variant
// On gfx950 the fp32 MFMA runs on the SIMD's fp32 FMA lanes: tools/mfma_peak.hip shows that LDS reads, LDS-DMA issues,
// barriers and scalar instructions beside a stream of v_mfma_f32_32x32x2_f32 cost nothing (98-99 % of 157.3 TF at
// 2-3 waves per SIMD), while every VALU instruction takes ~4.5 cycles away from the matrix pipe whichever wave
// issues it (30 VALU per 8 MFMAs: 77 %).  The general kernel above spends ~100 VALU instructions per K-chunk on
// gather addresses (bounds tests, tap bookkeeping, LDS destination, fragment addresses) against 32 MFMAs -- its 81 %.
// This variant serves the layers that carry the step (16-byte path, C % 32 == 0, zero padding or the stride-1 data
// gradient): a K-chunk of 32 then lies inside ONE filter tap for the whole workgroup, so
//   * tap bookkeeping and the tap's byte offset live in SGPRs and enter the load as its scalar offset,
//   * a row's validity for every tap is one bit of a mask built once per tile (bit kh*KW+kw set = outside the image);
//     an invalid piece gets bit 31 of its offset set (beyond the descriptor -> the hardware returns zeros),
//   * the descriptor's base is moved back by the padding, so row offsets and tap offsets are both non-negative,
//   * LDS destinations (M0) are scalar, fragment addresses are eight loop-invariant registers + immediate offsets
//     (the loop is unrolled over the two LDS buffers).
// Per K-chunk: 32 MFMAs, 12 ds_read_b128, 6 LDS-DMA issues and 2 VALU per A piece (v_bfe_u32 + v_lshl_add_u32).
__device__ __forceinline__ unsigned magic_div(unsigned n, unsigned mg, unsigned sh) {
    return mg ? __umulhi(n, mg) >> sh : n;     // mg == 0: divisor 1
}

__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t r, unsigned m0v, unsigned voff, unsigned soff) {
    // s_nop 4: the scalar offset may have been written by the SALU instruction right in front (5 wait states to a
    // VMEM read of it); s_nop 0: M0 write -> LDS-DMA
    // M0: reserved register of the backend, see dma16.  No "memory" clobber on purpose: the LDS rows this piece fills belong to a ring slot that
    // no ds_read of the current chunk touches, and the s_waitcnt + barrier that publish the slot carry the clobber --
    // with it here hipcc pins every fragment read behind the issue and the loop loses its overlap
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(m0v), "v"(voff), "s"(r), "s"(soff));
}

// compile-time loop over LDS-DMA piece numbers [P0, P1)
template <int P0, int P1, typename F, typename D>
__device__ __forceinline__ void pieces(F& f, D dst) {
    if constexpr (P0 < P1) {
        f(dst, std::integral_constant<int, P0>{});
        pieces<P0 + 1, P1>(f, dst);
    }
}

template <int BM, int BN, int WM, int WN, int MODE, int NS>
__global__ __launch_bounds__(NT) void conv_igemm_uni_kernel(const ConvArgs a) {
    // MODE_REFLECT (the decoder's ReflectionPad2d(1) + Conv3x3): no tap is ever invalid, but a border row's offset is
    // not linear in the tap.  The per-row offsets of the CURRENT tap are recomputed on the VALU only when the tap
    // changes (every C/32 chunks, a uniform branch); the pieces then take them as they are -- no VALU per piece.
    constexpr int MT = 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / MT, TN = WN / MT;
    static_assert((BM / WM) * WAVES_N == 4 && BN % 32 == 0, "4 waves per workgroup");
    constexpr int A_IT = BM / 32, B_IT = BN / 32, NP = A_IT + B_IT;   // LDS-DMA pieces per thread and chunk
    constexpr unsigned A_BYTES = BM * LDT * 4, B_BYTES = BN * LDT * 4;
    constexpr unsigned B_BASE = NS * A_BYTES;

    // ring of NS chunk buffers: A tiles, then B tiles
    __shared__ __attribute__((aligned(16))) float smem_all[NS * (BM + BN) * LDT];

    const int nblk = a.mtiles * a.ntiles;
    const int per_xcd = (int)gridDim.x >> 3;
    const int logical = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (logical >= nblk) return;
    const int mt = logical / a.ntiles, nt = logical - mt * a.ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int hw = a.Ho * a.Wo;

    // Descriptor of the activation: first image of the tile, moved back by the padding (zero mode) or by the part of
    // the filter extent the padding does not cover (data gradient, taps walk backwards) -- see the offsets below.
    const int img0 = (int)(m0 / hw);
    const long shift = MODE == MODE_REFLECT ? 0L : MODE == MODE_ZERO ? (long)a.pad * a.sH + (long)a.pad_w * a.sW
                                         : (long)(a.KH - 1 - a.pad) * a.sH + (long)(a.KW - 1 - a.pad_w) * a.sW;
    const long rest = (((long)a.N - img0) * a.sN + shift) * 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)img0 * a.sN - shift, rest > 0x7fffffffL ? 0x7fffffffu : (unsigned)rest);
    const __amdgpu_buffer_rsrc_t rw_ = make_rsrc(a.w, a.w_bytes);

    // ---- per-thread gather state (loop invariant)
    const int prow = tid >> 3;                                   // row of this thread inside every 32-row piece
    const unsigned col4 = 16u * ((tid & 7) ^ ((prow >> 1) & 7)); // byte offset of its LOGICAL 16-byte slot (swizzle at the source)
    const unsigned ones_kw = (1u << a.KW) - 1u;
    unsigned va[A_IT], inv[A_IT], vb[B_IT];
    int rfh[MODE == MODE_REFLECT ? A_IT : 1], rfw[MODE == MODE_REFLECT ? A_IT : 1];   // reflect: oh*stride - pad, ow*stride - pad
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const long m = m0 + prow + 32 * i;
        const unsigned mr = (unsigned)(m - (long)img0 * hw);
        const unsigned dn = magic_div(mr, a.mg_hw, a.sh_hw);
        const unsigned rem = mr - dn * (unsigned)hw;
        const int oh = (int)magic_div(rem, a.mg_wo, a.sh_wo);
        const int ow = (int)rem - oh * a.Wo;
        if constexpr (MODE == MODE_REFLECT) {
            va[i] = (unsigned)((int)dn * (int)a.sN) * 4u + col4;      // image base; the pixel part follows the tap
            rfh[i] = oh * a.stride - a.pad; rfw[i] = ow * a.stride - a.pad_w;
            inv[i] = m < a.M ? 0u : 0xffffffffu;
            continue;
        }
        va[i] = (unsigned)((int)dn * (int)a.sN + oh * a.stride * (int)a.sH + ow * a.stride * (int)a.sW) * 4u + col4;
        // taps outside the image, per dimension: a prefix [0, lo) and a suffix [hi, K) of the tap range
        int lo_h, hi_h, lo_w, hi_w;
        if (MODE == MODE_ZERO) {            // ih = oh*stride - pad + kh
            lo_h = a.pad - oh * a.stride; hi_h = a.H + lo_h;
            lo_w = a.pad_w - ow * a.stride; hi_w = a.W + lo_w;
        } else {                            // ih = oh + pad - kh
            hi_h = oh + a.pad + 1; lo_h = hi_h - a.H;
            hi_w = ow + a.pad_w + 1; lo_w = hi_w - a.W;
        }
        lo_h = min(max(lo_h, 0), a.KH); hi_h = min(max(hi_h, 0), a.KH);
        lo_w = min(max(lo_w, 0), a.KW); hi_w = min(max(hi_w, 0), a.KW);
        const unsigned bad_h = ((1u << lo_h) - 1u) | ~((1u << hi_h) - 1u);     // bit kh (bits >= KH are never read)
        const unsigned bad_w = (((1u << lo_w) - 1u) | ~((1u << hi_w) - 1u)) & ones_kw;
        unsigned mask = 0;
        for (int kh = a.KH - 1; kh >= 0; --kh)
            mask = (mask << a.KW) | (((bad_h >> kh) & 1u) ? ones_kw : bad_w);
        inv[i] = m < a.M ? (mask | 0x80000000u) : 0xffffffffu;   // bit 31: the tap index of chunks beyond K (ring run-out)
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int nr = n0 + prow + 32 * i;
        vb[i] = nr < a.Co ? (unsigned)(nr * a.K) * 4u + col4 : OOB;
    }

    // ---- scalar tap state of the chunk the next load_piece() calls fetch: tap index, channel offset inside the tap,
    // byte offset of (tap, channel chunk).  Chunks past K (the ring keeps loading NS-1 ahead) fetch nothing: tap
    // index 31 is invalid in every mask, the weight pieces re-read chunk 0.
    const int dW4 = (MODE == MODE_ZERO ? (int)a.sW : -(int)a.sW) * 4, dH4 = (MODE == MODE_ZERO ? (int)a.sH : -(int)a.sH) * 4;
    const int nchunks = a.K / BK;          // K % 32 == 0 on this path
    int s_q = 0, s_tap = 0, s_kw = 0, s_c = 0;
    unsigned s_aoff = MODE != MODE_TRANSPOSED ? 0u : (unsigned)(((a.KH - 1) * (int)a.sH + (a.KW - 1) * (int)a.sW) * 4);
    unsigned s_boff = 0;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t*)smem_all;
    const unsigned m0_a = lds0 + 1024u * (unsigned)wave, m0_b = lds0 + B_BASE + 1024u * (unsigned)wave;

    // reflect: byte offsets of the rows for the tap (s_kh, s_kw), redone when the tap changes
    unsigned vr[MODE == MODE_REFLECT ? A_IT : 1];
    int s_kh = 0;
    auto reflect_rows = [&]() {
        if constexpr (MODE == MODE_REFLECT) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int ih = rfh[i] + s_kh, iw = rfw[i] + s_kw;
                ih = ih < 0 ? -ih : ih; ih = ih >= a.H ? 2 * a.H - 2 - ih : ih;
                iw = iw < 0 ? -iw : iw; iw = iw >= a.W ? 2 * a.W - 2 - iw : iw;
                vr[i] = inv[i] ? OOB : va[i] + (unsigned)(ih * (int)a.sH + iw * (int)a.sW) * 4u;
            }
        }
    };
    reflect_rows();
    auto load_piece = [&](auto dst_tag, auto piece_tag) {
        constexpr unsigned DST = decltype(dst_tag)::value;
        constexpr int P = decltype(piece_tag)::value;
        if constexpr (P < A_IT && MODE == MODE_REFLECT) {
            dma16s(rx, m0_a + DST * A_BYTES + 4096u * P, vr[P], s_aoff);      // s_aoff = channel chunk inside the tap
        } else if constexpr (P < A_IT) {
            const unsigned bad = __builtin_amdgcn_ubfe(inv[P], (unsigned)s_tap, 1u);
            dma16s(rx, m0_a + DST * A_BYTES + 4096u * P, (bad << 31) + va[P], s_aoff);
        } else {
            dma16s(rw_, m0_b + DST * B_BYTES + 4096u * (P - A_IT), vb[P - A_IT], s_boff);
        }
    };
    auto advance_chunk = [&]() {
        ++s_q;
        if constexpr (MODE == MODE_REFLECT) {
            s_boff += BK * 4; s_aoff += BK * 4; s_c += BK;
            if (s_q >= nchunks) { s_boff = 0; s_aoff = 0; s_c = 0; }     // ring run-out: re-read chunk 0 of the last tap (harmless)
            else if (s_c == a.C) {
                s_c = 0; s_aoff = 0;
                if (++s_kw == a.KW) { s_kw = 0; ++s_kh; }
                reflect_rows();
            }
            return;
        }
        s_boff += BK * 4; s_aoff += BK * 4; s_c += BK;
        if (s_c == a.C) {
            s_c = 0; ++s_tap; s_aoff += (unsigned)(dW4 - a.C * 4);
            if (++s_kw == a.KW) { s_kw = 0; s_aoff += (unsigned)(dH4 - a.KW * dW4); }
        }
        if (s_q >= nchunks) { s_tap = 31; s_boff = 0; s_aoff = 0; }
    };

    // ---- fragment addresses: lane -> (row = lane % 32, slot group = lane / 32); the swizzle (row>>1)&7 is the same
    // for rows 32 apart, so the tiles of a wave differ by immediate offsets only
    const int frow = lane & 31, fh = lane >> 5;
    unsigned fa_off[4], fb_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const unsigned sw = 16u * ((2 * g + fh) ^ ((frow >> 1) & 7));
        fa_off[g] = (unsigned)(wm * WM + frow) * (LDT * 4) + sw;
        fb_off[g] = B_BASE + (unsigned)(wn * WN + frow) * (LDT * 4) + sw;
        asm volatile("" : "+v"(fa_off[g]), "+v"(fb_off[g]));     // keep them as registers (no re-derivation per chunk)
    }
    const char* lds_c = reinterpret_cast<const char*>(smem_all);

    typedef float accv_t __attribute__((ext_vector_type(16)));
    accv_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One K-chunk: the MFMAs of buffer BUF with the LDS-DMA pieces of the chunk NS-1 ahead spread between the four
    // fragment groups (a piece issued in the shadow of the wave's own MFMAs costs it nothing; six in a row at the
    // head of the chunk are ~400 cycles in which the wave feeds no MFMA).
    auto chunk = [&](auto buf_tag) {
        constexpr unsigned BUF = decltype(buf_tag)::value;
        constexpr unsigned DST = (BUF + NS - 1) % NS;
        const std::integral_constant<unsigned, DST> dst{};
        float4 fa[2][TM], fb[2][TN];          // fragments of group g+1 are read while the MFMAs of group g run
        auto read_frags = [&](int g, int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[set][i] = *reinterpret_cast<const float4*>(lds_c + fa_off[g] + (BUF * A_BYTES + (unsigned)i * MT * LDT * 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[set][j] = *reinterpret_cast<const float4*>(lds_c + fb_off[g] + (BUF * B_BYTES + (unsigned)j * MT * LDT * 4));
        };
        read_frags(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) read_frags(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);     // (left alone, the scheduler sinks the reads below the MFMAs again)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i].x, fb[g & 1][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i].y, fb[g & 1][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i].z, fb[g & 1][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i].w, fb[g & 1][j].w, acc[i][j], 0, 0, 0);
                }
            // pieces g*NP/4 .. (g+1)*NP/4 - 1 of the prefetched chunk
            if (g == 0) { pieces<0 * NP / 4, 1 * NP / 4>(load_piece, dst); }
            else if (g == 1) { pieces<1 * NP / 4, 2 * NP / 4>(load_piece, dst); }
            else if (g == 2) { pieces<2 * NP / 4, 3 * NP / 4>(load_piece, dst); }
            else { pieces<3 * NP / 4, NP>(load_piece, dst); }
        }
        advance_chunk();
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * NP) : "memory");
        __syncthreads();
    };

    // ---- prologue: chunks 0 .. NS-2 in flight, chunk 0 landed
    {
        const std::integral_constant<unsigned, 0> d0{};
        pieces<0, NP>(load_piece, d0);
        advance_chunk();
        if constexpr (NS == 3) {
            const std::integral_constant<unsigned, 1> d1{};
            pieces<0, NP>(load_piece, d1);
            advance_chunk();
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * NP) : "memory");
        __syncthreads();
    }
    for (int q = 0; q < nchunks; q += NS) {
        chunk(std::integral_constant<unsigned, 0>{});
        if (q + 1 < nchunks) chunk(std::integral_constant<unsigned, 1>{});
        if constexpr (NS == 3) {
            if (q + 2 < nchunks) chunk(std::integral_constant<unsigned, 2>{});
        }
    }
    if constexpr (NS > 2) {                // run-out pieces (zeros / chunk 0 of the weights) still target the ring
        dma_wait();
        __syncthreads();
    }
    conv_epilogue<BM, BN, WM, WN, MT>(a, acc, smem_all, mt, m0, n0, wm, wn, lane, wave, tid);
}

// ===================================================================== fp32 products on the bf16 matrix cores ("x3")
// gfx950 multiplies bf16 sixteen times faster than fp32 (v_mfma_f32_32x32x16_bf16: 32768 flop in 32 cycles of a SIMD;
// v_mfma_f32_32x32x2_f32: 4096 flop in 64).  An fp32 number is the exact sum of three bf16 numbers of 8 significant
// bits each -- hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest, both subtractions exact,
// |x - hi - mid - lo| <= 2^-27 |x| -- and a product a*b is
//     hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi)   + terms <= 2^-24 |a*b| (mid*lo, lo*mid, lo*lo: dropped),
// every partial product exact in the fp32 accumulator's input (8 x 8 bits).  Six bf16 MFMAs per 32x32x16 block (issued
// largest term first, in the order the split yields the terms) instead of eight fp32 MFMAs of twice the duration each: the contraction keeps fp32 accuracy (the
// dropped terms are of the size of ONE fp32 rounding of the product; tests/test_conv_gpu.py measures the kernel's error
// against an fp64 convolution next to the fp32 kernel's) at 2.7x the matrix rate.  What it costs is the split: ~4.5
// vector instructions per operand element (tools/bf16x3_peak.hip: the probe that sized this kernel).  So
//   * the wave tile is 64 rows x 64 columns: an activation element is split by exactly one wave (4 x 1 waves, each
//     stages, splits and multiplies its own 64 rows), 3 vector instructions per MFMA;
//   * the weights of a chunk (64 x 16) are split ONCE per workgroup, by all 256 threads, one chunk ahead, from an fp32
//     staging tile into three bf16 planes in LDS: no pre-split copies in memory that could go stale, no extra launch;
//   * chunks are 16 deep (one MFMA k-step: a 16-channel group of one tap; the channel group is the OUTER loop), so that
//     three workgroups fit a CU (52 KB each) and the split of one wave runs under the MFMAs of the two others;
//   * the vector work is threaded between the MFMAs in program order (see chunk()).
// Gather, masks and scalar tap state are those of conv_igemm_uni_kernel.  Output tile 256 x 64 (128 x 64 with one row
// block per wave); the BatchNorm partial sums keep their 128- / 64-row granularity, so pd_conv2d_stats_rows() does not
// change.  (The constants below are those of the 256-row tile; the kernel derives its own from RB.)
namespace x3 {
constexpr int BM = 256, BN = 64, CK = 16;
constexpr unsigned A_BYTES = BM * CK * 4;          // 16 KB per ring slot
constexpr unsigned BS_BYTES = BN * CK * 4;         // 4 KB fp32 staging tile of the weights
constexpr unsigned BP_BYTES = BN * CK * 2;         // 2 KB per bf16 plane
constexpr unsigned BS_BASE = 2 * A_BYTES, BP_BASE = BS_BASE + 2 * BS_BYTES;
constexpr unsigned LDS_BYTES = BP_BASE + 2 * 3 * BP_BYTES;     // 52 KB
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Split { u32x4 hi, mid, lo; };
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32 (round to nearest even) -- through the compiler, not inline asm: the hazard recognizer does not look
// into asm statements, and a conversion feeding an MFMA right behind it needs its wait states
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// two fp32 -> (hi, mid, lo) packs of two bf16 each
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const f32x2 v = {x0, x1};
    hi = cvt_pk_bf16(x0, x1);
    const f32x2 h = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    const f32x2 r1 = v - h;
    mid = cvt_pk_bf16(r1.x, r1.y);
    const f32x2 m = {__uint_as_float(mid << 16), __uint_as_float(mid & 0xffff0000u)};
    const f32x2 r2 = r1 - m;
    lo = cvt_pk_bf16(r2.x, r2.y);
}
__device__ __forceinline__ Split split8(const float4 a, const float4 b) {
    unsigned h[4], m[4], l[4];
    split2(a.x, a.y, h[0], m[0], l[0]);
    split2(a.z, a.w, h[1], m[1], l[1]);
    split2(b.x, b.y, h[2], m[2], l[2]);
    split2(b.z, b.w, h[3], m[3], l[3]);
    Split s;
    s.hi = u32x4{h[0], h[1], h[2], h[3]}; s.mid = u32x4{m[0], m[1], m[2], m[3]}; s.lo = u32x4{l[0], l[1], l[2], l[3]};
    return s;
}
// split2() on pair P of a value array in three steps -- hi (1 instruction), mid (5), lo (3) -- so that a kernel can
// thread them between its MFMAs (program order is issue order)
struct Terms { unsigned h[4], m[4], l[4]; f32x2 tf[4], tr[4]; };
// (W marks the split of a weight chunk.  PD_PROBE_NOSPLIT / PD_PROBE_NOSPLIT_W exist for tools/build_probe.sh only: timing
//  probes -- never defined for libpolardepth.so -- in which the terms are the operand's raw bits, i.e. the instruction mix of
//  a kernel whose operands arrive pre-split; their results are meaningless.)
#if defined(PD_PROBE_NOSPLIT)
#define PD_PROBE_RAW(W) true
#elif defined(PD_PROBE_NOSPLIT_W)
#define PD_PROBE_RAW(W) (W)
#else
#define PD_PROBE_RAW(W) false
#endif
template <int P, bool W = false> __device__ __forceinline__ void sp_h(const float* x, Terms& t) {
    if constexpr (PD_PROBE_RAW(W)) { t.h[P] = __float_as_uint(x[2 * P]); return; }
    t.h[P] = cvt_pk_bf16(x[2 * P], x[2 * P + 1]);
}
template <int P, bool W = false> __device__ __forceinline__ void sp_m(const float* x, Terms& t) {
    if constexpr (PD_PROBE_RAW(W)) { t.m[P] = __float_as_uint(x[2 * P + 1]); return; }
    t.tf[P] = f32x2{__uint_as_float(t.h[P] << 16), __uint_as_float(t.h[P] & 0xffff0000u)};
    t.tr[P] = f32x2{x[2 * P], x[2 * P + 1]} - t.tf[P];
    t.m[P] = cvt_pk_bf16(t.tr[P].x, t.tr[P].y);
    t.tf[P].x = __uint_as_float(t.m[P] << 16);
}
template <int P, bool W = false> __device__ __forceinline__ void sp_l(Terms& t) {
    if constexpr (PD_PROBE_RAW(W)) { t.l[P] = t.h[P]; return; }
    t.tf[P].y = __uint_as_float(t.m[P] & 0xffff0000u);
    const f32x2 r2 = t.tr[P] - t.tf[P];
    t.l[P] = cvt_pk_bf16(r2.x, r2.y);
}
}  // namespace x3

// RB = row blocks (of 32) per wave: 2 -> 256-row tiles (52 KB of LDS, three workgroups per CU); 1 -> 128-row tiles (36 KB,
// four per CU) for the layers whose 256-row tiles would not fill the chip twice.
// (Round 4, measured and dropped: EIGHT waves per workgroup -- two sets of four, each a complete copy of the pipeline on its own
//  half of LDS, taking every other 16-channel group and meeting in LDS at the end -- for the 320-tile 3x3x512 @16x20 layers:
//  no change, 113 TF either way; the two sets wait at the same barriers for the same loads.  What those layers lacked was L2
//  locality: x3_nmajor below.)
template <int MODE, int RB = 2>
__global__ __launch_bounds__(NT, RB == 2 ? 3 : 4) void conv_igemm_x3_kernel(const ConvArgs a) {
    using namespace x3;
    constexpr int BM = 128 * RB;
    constexpr unsigned A_BYTES = BM * CK * 4, BS_BASE = 2 * A_BYTES, BP_BASE = BS_BASE + 2 * BS_BYTES;
    constexpr unsigned LDS_BYTES = BP_BASE + 2 * 3 * BP_BYTES;
    __shared__ __attribute__((aligned(16))) float smem_all[LDS_BYTES / 4];

    const int nblk = a.mtiles * a.ntiles;
    const int per_xcd = (int)gridDim.x >> 3;
    const int logical = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (logical >= nblk) return;
    // (a.nmajor: column tile outside -- an XCD then streams one 64-column slice of a multi-megabyte filter instead of all of them)
    const int mt = a.nmajor ? logical % a.mtiles : logical / a.ntiles, nt = a.nmajor ? logical / a.mtiles : logical - mt * a.ntiles;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = a.Ho * a.Wo;

    // every other tile computes the negated convolution and negates back in the epilogue: the bf16 MFMA's truncation bias
    // (toward -infinity, whatever the signs) then cancels in sums over outputs -- see conv_halo_x3_kernel
    const unsigned wsign = (mt & 1) ? 0x80008000u : 0u;
    const int img0 = (int)(m0 / hw);
    // MODE_REFLECT (the decoder's ReflectionPad2d(1) + Conv3x3, same-size output): offsets as with zero padding; a tap that
    // leaves the image is moved two rows / columns inwards -- a per-row correction (hcor, wcor: +2 rows at the top row, -2
    // at the bottom one, 0 elsewhere; same for columns) that applies when the tap is the first / last of its dimension.
    const long shift = MODE != MODE_TRANSPOSED ? (long)a.pad * a.sH + (long)a.pad_w * a.sW
                                               : (long)(a.KH - 1 - a.pad) * a.sH + (long)(a.KW - 1 - a.pad_w) * a.sW;
    const long rest = (((long)a.N - img0) * a.sN + shift) * 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)img0 * a.sN - shift, rest > 0x7fffffffL ? 0x7fffffffu : (unsigned)rest);
    const __amdgpu_buffer_rsrc_t rw_ = make_rsrc(a.w, a.w_bytes);

    // ---- gather state: a wave stages its own 64 rows, 16 per LDS-DMA instruction (lane -> row lane/4, physical 16-byte
    // slot lane%4 of the 64-byte row; it fetches the LOGICAL slot (lane%4) ^ ((row/4)%4): conflict-free fragment reads)
    const int prow = lane >> 2;
    const unsigned col4 = 16u * ((lane & 3) ^ ((lane >> 4) & 3));
    const unsigned ones_kw = (1u << a.KW) - 1u;
    unsigned va[2 * RB], inv[2 * RB];
    int hcor[MODE == MODE_REFLECT ? 2 * RB : 1], wcor[MODE == MODE_REFLECT ? 2 * RB : 1];
#pragma unroll
    for (int i = 0; i < 2 * RB; ++i) {
        const long m = m0 + 32 * RB * wave + 16 * i + prow;
        const unsigned mr = (unsigned)(m - (long)img0 * hw);
        const unsigned dn = magic_div(mr, a.mg_hw, a.sh_hw);
        const unsigned rem = mr - dn * (unsigned)hw;
        const int oh = (int)magic_div(rem, a.mg_wo, a.sh_wo);
        const int ow = (int)rem - oh * a.Wo;
        va[i] = (unsigned)((int)dn * (int)a.sN + oh * a.stride * (int)a.sH + ow * a.stride * (int)a.sW) * 4u + col4;
        if constexpr (MODE == MODE_REFLECT) {
            hcor[i] = oh == 0 ? 2 * (int)a.sH * 4 : oh == a.H - 1 ? -2 * (int)a.sH * 4 : 0;
            wcor[i] = ow == 0 ? 2 * (int)a.sW * 4 : ow == a.W - 1 ? -2 * (int)a.sW * 4 : 0;
            inv[i] = 0;
            continue;
        }
        int lo_h, hi_h, lo_w, hi_w;       // taps outside the image: a prefix [0, lo) and a suffix [hi, K) per dimension
        if (MODE == MODE_ZERO) {
            lo_h = a.pad - oh * a.stride; hi_h = a.H + lo_h;
            lo_w = a.pad_w - ow * a.stride; hi_w = a.W + lo_w;
        } else {
            hi_h = oh + a.pad + 1; lo_h = hi_h - a.H;
            hi_w = ow + a.pad_w + 1; lo_w = hi_w - a.W;
        }
        lo_h = min(max(lo_h, 0), a.KH); hi_h = min(max(hi_h, 0), a.KH);
        lo_w = min(max(lo_w, 0), a.KW); hi_w = min(max(hi_w, 0), a.KW);
        const unsigned bad_h = ((1u << lo_h) - 1u) | ~((1u << hi_h) - 1u);
        const unsigned bad_w = (((1u << lo_w) - 1u) | ~((1u << hi_w) - 1u)) & ones_kw;
        unsigned mask = 0;
        for (int kh = a.KH - 1; kh >= 0; --kh)
            mask = (mask << a.KW) | (((bad_h >> kh) & 1u) ? ones_kw : bad_w);
        inv[i] = m < a.M ? (mask | 0x80000000u) : 0xffffffffu;
    }
    // weights: wave w stages rows 16w .. 16w+15 of the 64 x 16 chunk (same row / slot geometry)
    const unsigned vb = (unsigned)((n0 + 16 * wave + prow) * a.K) * 4u + col4;

    // ---- scalar state: the activation pieces run one chunk ahead, the weight pieces two
    const int dW4 = (MODE != MODE_TRANSPOSED ? (int)a.sW : -(int)a.sW) * 4, dH4 = (MODE != MODE_TRANSPOSED ? (int)a.sH : -(int)a.sH) * 4;
    // (a channel count that is no multiple of 16 -- the space-to-depth stems: 8, 12, 36 -- gets a last, partly empty channel
    //  group: the lanes whose 16-byte slot lies beyond C fetch nothing)
    const int ngroups = (a.C + CK - 1) / CK;
    const int nchunks = ngroups * a.KH * a.KW;
    const int c_lim = a.C - 4 * ((lane & 3) ^ ((lane >> 4) & 3));      // this lane's slot is valid while s_c < c_lim
    // Chunk order: the 16-channel group is the OUTER loop, the taps the inner one.  With the taps outside (the order of
    // the fp32 kernels) every input row is fetched once per filter row: the workgroups resident on an XCD hold ~25 000
    // pixels x 256 B = 6.3 MB of input in flight, more than its 4 MB L2, and the row a tile reads at tap row kh is read by
    // its neighbour one tap row later -- rocprofv3 FETCH_SIZE was 4.3x the input on the 5x5 layers.  With the channel
    // group outside, all KH*KW taps of a 64-byte segment follow each other and the footprint between reuses is a quarter.
    const int ntaps = a.KH * a.KW;
    const unsigned aoff0 = MODE != MODE_TRANSPOSED ? 0u : (unsigned)(((a.KH - 1) * (int)a.sH + (a.KW - 1) * (int)a.sW) * 4);
    int s_q = 0, s_tap = 0, s_kw = 0, s_kh = 0, s_c = 0;
    unsigned s_aoff = aoff0;
    int s_qb = 0, sb_tap = 0;
    unsigned s_boff = 0, sb_c4 = 0;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t*)smem_all;
    const unsigned m0_a = lds0 + 2048u * RB * (unsigned)wave, m0_b = lds0 + BS_BASE + 1024u * (unsigned)wave;

    auto load_a = [&](auto dst_tag, auto piece_tag) {
        constexpr unsigned DST = decltype(dst_tag)::value;
        constexpr int P = decltype(piece_tag)::value;
        if constexpr (MODE == MODE_REFLECT) {
            const int hs = s_kh == 0 ? max(hcor[P], 0) : s_kh == 2 ? min(hcor[P], 0) : 0;       // (uniform selects)
            const int ws = s_kw == 0 ? max(wcor[P], 0) : s_kw == 2 ? min(wcor[P], 0) : 0;
            const unsigned bad = (s_tap == 31 ? 1u : 0u) | ((unsigned)(c_lim - 1 - s_c) >> 31);
            dma16s(rx, m0_a + DST * A_BYTES + 1024u * P, (bad << 31) + (va[P] + (unsigned)(hs + ws)), s_aoff);
        } else {
        const unsigned bad = __builtin_amdgcn_ubfe(inv[P], (unsigned)s_tap, 1u) | ((unsigned)(c_lim - 1 - s_c) >> 31);
        dma16s(rx, m0_a + DST * A_BYTES + 1024u * P, (bad << 31) + va[P], s_aoff);
        }
    };
    auto advance_a = [&]() {
        ++s_q;
        ++s_tap; s_aoff += (unsigned)dW4;
        if (++s_kw == a.KW) { s_kw = 0; ++s_kh; s_aoff += (unsigned)(dH4 - a.KW * dW4); }
        if (s_tap == ntaps) { s_tap = 0; s_kh = 0; s_c += CK; s_aoff = aoff0 + (unsigned)s_c * 4u; }
        if (s_q >= nchunks) { s_tap = 31; s_aoff = 0; }          // run-out: every mask drops tap 31
    };
    // (in a partly empty channel group the staged weight columns beyond C are the next tap's -- finite values that meet
    //  zero activations)
    auto load_b = [&](auto dst_tag) {                             // chunk s_qb (past the end: chunk 0 again, never used)
        constexpr unsigned DST = decltype(dst_tag)::value;
        dma16s(rw_, m0_b + DST * BS_BYTES, vb, s_qb < nchunks ? s_boff : 0u);
        ++s_qb;
        s_boff += (unsigned)a.C * 4u;                             // next tap of the same channel group
        if (++sb_tap == ntaps) { sb_tap = 0; sb_c4 += CK * 4; s_boff = sb_c4; }
    };

    // ---- fragment addresses (bytes from the start of LDS)
    // activations: lane -> (row = lane%32 of the block, k half = lane/32): logical slots 2*half, 2*half+1
    const int frow = lane & 31, fh = lane >> 5;
    unsigned fa_off[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        fa_off[e] = (unsigned)(32 * RB * wave + frow) * (CK * 4) + 16u * ((2 * fh + e) ^ ((frow >> 2) & 3));
        asm volatile("" : "+v"(fa_off[e]));
    }
    // weight planes: 32-byte rows, two 16-byte k halves, half ^ ((row/8)%2)
    unsigned fb_off = BP_BASE + (unsigned)frow * (CK * 2) + 16u * (fh ^ ((frow >> 3) & 1));
    asm volatile("" : "+v"(fb_off));
    // split pass: thread -> (row tid/4, physical slot tid%4) of the staging tile = its 16 bytes number tid
    const int srow = tid >> 2, sls = (tid & 3) ^ ((srow >> 2) & 3);            // logical slot: k = 4*sls .. 4*sls+3
    const unsigned ss_off = BS_BASE + 16u * (unsigned)tid;
    const unsigned sp_off = BP_BASE + (unsigned)srow * (CK * 2) + 16u * ((sls >> 1) ^ ((srow >> 3) & 1)) + 8u * (sls & 1);
    char* lds_c = reinterpret_cast<char*>(smem_all);

    typedef float accv_t __attribute__((ext_vector_type(16)));
    accv_t acc[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto split_b = [&](auto src_tag) {          // staging tile SRC -> planes SRC
        constexpr unsigned SRC = decltype(src_tag)::value;
        const float4 w4 = *reinterpret_cast<const float4*>(lds_c + ss_off + SRC * BS_BYTES);
        uint2 h, m, l;
        split2(w4.x, w4.y, h.x, m.x, l.x);
        split2(w4.z, w4.w, h.y, m.y, l.y);
        *reinterpret_cast<uint2*>(lds_c + sp_off + (SRC * 3 + 0) * BP_BYTES) = uint2{h.x ^ wsign, h.y ^ wsign};
        *reinterpret_cast<uint2*>(lds_c + sp_off + (SRC * 3 + 1) * BP_BYTES) = uint2{m.x ^ wsign, m.y ^ wsign};
        *reinterpret_cast<uint2*>(lds_c + sp_off + (SRC * 3 + 2) * BP_BYTES) = uint2{l.x ^ wsign, l.y ^ wsign};
    };
    auto bf = [](u32x4 v) { return __builtin_bit_cast(bf16x8, v); };

    // One chunk: pieces of the next chunks first (they land under everything below), fragments, then 24 MFMAs with the
    // vector work threaded between them IN PROGRAM ORDER: a wave issues in order, so a split placed behind a run of MFMAs
    // waits until the matrix pipe has accepted the last of them (32 cycles each) -- SQ counters of the straightforward
    // order: matrix pipe busy 61 %, waves waiting for issue 59 % of their cycles.  Here only the split of row block 0 is
    // exposed... and of that only its four hi conversions: the products are issued largest terms first (hi*hi, hi*mid, hi*lo,
    // mid*hi, mid*mid, lo*hi), in the order the split yields the terms, and mid / lo of a block are computed behind the
    // MFMAs that need only hi; the split of row block 1 and of the next chunk's weights ride behind the rest.  The two column
    // blocks alternate, so consecutive MFMAs never share an accumulator.
    auto chunk = [&](auto buf_tag) {
        constexpr unsigned BUF = decltype(buf_tag)::value;
        const std::integral_constant<unsigned, BUF ^ 1> nxt{};
        constexpr unsigned NXT = BUF ^ 1;
        pieces<0, 2 * RB>(load_a, nxt);
        load_b(buf_tag);
        advance_a();
        float4 fa[RB][2];
        u32x4 fb[2][3];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                fa[i][e] = *reinterpret_cast<const float4*>(lds_c + fa_off[e] + (BUF * A_BYTES + (unsigned)i * 32 * CK * 4));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                fb[j][t] = *reinterpret_cast<const u32x4*>(lds_c + fb_off + ((BUF * 3 + t) * BP_BYTES + (unsigned)j * 32 * CK * 2));
        const float4 w4 = *reinterpret_cast<const float4*>(lds_c + ss_off + NXT * BS_BYTES);      // next chunk's weights (fp32)
        const float x0[8] = {fa[0][0].x, fa[0][0].y, fa[0][0].z, fa[0][0].w, fa[0][1].x, fa[0][1].y, fa[0][1].z, fa[0][1].w};
        const float x1[8] = {fa[RB - 1][0].x, fa[RB - 1][0].y, fa[RB - 1][0].z, fa[RB - 1][0].w,
                             fa[RB - 1][1].x, fa[RB - 1][1].y, fa[RB - 1][1].z, fa[RB - 1][1].w};     // (RB == 1: unused)
        const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
        x3::Terms t0, t1, tw;                            // staged split (x3::sp_h / sp_m / sp_l) of row blocks 0, 1 and the weights
        // MFMA number N (0..11) of row block I, largest terms first (what a block needs first is what its split yields
        // first): hi*hi, hi*mid, hi*lo, mid*hi, mid*mid, lo*hi; column block N % 2
        auto mm = [&](const x3::Terms& t, auto i_tag, auto n_tag) {
            constexpr int I = decltype(i_tag)::value, N = decltype(n_tag)::value, T = N / 2, J = N % 2;
            const unsigned* av = T < 3 ? t.h : T < 5 ? t.m : t.l;
            const u32x4 bv = T == 0 ? fb[J][0] : T == 1 ? fb[J][1] : T == 2 ? fb[J][2] : T == 3 ? fb[J][0] : T == 4 ? fb[J][1] : fb[J][0];
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(u32x4{av[0], av[1], av[2], av[3]}), bf(bv), acc[I][J], 0, 0, 0);
        };
#define PD_I(n) std::integral_constant<int, n>{}
#define PD_SB __builtin_amdgcn_sched_barrier(0);
        x3::sp_h<0>(x0, t0); x3::sp_h<1>(x0, t0); x3::sp_h<2>(x0, t0); x3::sp_h<3>(x0, t0);
        PD_SB
        if constexpr (RB == 1) {       // one row block: 12 MFMAs, its mid / lo terms and the weights' split behind them
            mm(t0, PD_I(0), PD_I(0)); x3::sp_m<0>(x0, t0); PD_SB
            mm(t0, PD_I(0), PD_I(1)); x3::sp_m<1>(x0, t0); PD_SB
            mm(t0, PD_I(0), PD_I(2)); x3::sp_m<2>(x0, t0); PD_SB
            mm(t0, PD_I(0), PD_I(3)); x3::sp_m<3>(x0, t0); PD_SB
            mm(t0, PD_I(0), PD_I(4)); x3::sp_h<0, true>(ws, tw); x3::sp_h<1, true>(ws, tw); PD_SB
            mm(t0, PD_I(0), PD_I(5)); x3::sp_m<0, true>(ws, tw); PD_SB
            mm(t0, PD_I(0), PD_I(6)); x3::sp_l<0>(t0); x3::sp_l<0, true>(tw); PD_SB
            mm(t0, PD_I(0), PD_I(7)); x3::sp_l<1>(t0); x3::sp_m<1, true>(ws, tw); PD_SB
            mm(t0, PD_I(0), PD_I(8)); x3::sp_l<2>(t0); x3::sp_l<1, true>(tw); PD_SB
            mm(t0, PD_I(0), PD_I(9)); x3::sp_l<3>(t0); PD_SB
            *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 0) * BP_BYTES) = uint2{tw.h[0] ^ wsign, tw.h[1] ^ wsign};
            *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 1) * BP_BYTES) = uint2{tw.m[0] ^ wsign, tw.m[1] ^ wsign};
            *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 2) * BP_BYTES) = uint2{tw.l[0] ^ wsign, tw.l[1] ^ wsign};
            mm(t0, PD_I(0), PD_I(10)); mm(t0, PD_I(0), PD_I(11));
        } else {
        // row block 0: its mid terms behind MFMAs 0..3, the weights' first pair behind 4..5, its lo terms (and block 1's hi)
        // behind 6..9
        mm(t0, PD_I(0), PD_I(0)); x3::sp_m<0>(x0, t0); PD_SB
        mm(t0, PD_I(0), PD_I(1)); x3::sp_m<1>(x0, t0); PD_SB
        mm(t0, PD_I(0), PD_I(2)); x3::sp_m<2>(x0, t0); PD_SB
        mm(t0, PD_I(0), PD_I(3)); x3::sp_m<3>(x0, t0); PD_SB
        mm(t0, PD_I(0), PD_I(4)); x3::sp_h<0, true>(ws, tw); x3::sp_h<1, true>(ws, tw); PD_SB
        mm(t0, PD_I(0), PD_I(5)); x3::sp_m<0, true>(ws, tw); PD_SB
        mm(t0, PD_I(0), PD_I(6)); x3::sp_l<0>(t0); x3::sp_h<0>(x1, t1); PD_SB
        mm(t0, PD_I(0), PD_I(7)); x3::sp_l<1>(t0); x3::sp_h<1>(x1, t1); PD_SB
        mm(t0, PD_I(0), PD_I(8)); x3::sp_l<2>(t0); x3::sp_h<2>(x1, t1); PD_SB
        mm(t0, PD_I(0), PD_I(9)); x3::sp_l<3>(t0); x3::sp_h<3>(x1, t1); PD_SB
        mm(t0, PD_I(0), PD_I(10)); x3::sp_m<1, true>(ws, tw); PD_SB
        mm(t0, PD_I(0), PD_I(11)); x3::sp_l<0, true>(tw); PD_SB
        // row block 1
        mm(t1, PD_I(RB - 1), PD_I(0)); x3::sp_m<0>(x1, t1); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(1)); x3::sp_m<1>(x1, t1); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(2)); x3::sp_m<2>(x1, t1); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(3)); x3::sp_m<3>(x1, t1); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(4)); x3::sp_l<1, true>(tw); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(5));
        *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 0) * BP_BYTES) = uint2{tw.h[0] ^ wsign, tw.h[1] ^ wsign};
        *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 1) * BP_BYTES) = uint2{tw.m[0] ^ wsign, tw.m[1] ^ wsign};
        *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 2) * BP_BYTES) = uint2{tw.l[0] ^ wsign, tw.l[1] ^ wsign};
        PD_SB
        mm(t1, PD_I(RB - 1), PD_I(6)); x3::sp_l<0>(t1); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(7)); x3::sp_l<1>(t1); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(8)); x3::sp_l<2>(t1); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(9)); x3::sp_l<3>(t1); PD_SB
        mm(t1, PD_I(RB - 1), PD_I(10)); mm(t1, PD_I(RB - 1), PD_I(11));
        }
#undef PD_SB
#undef PD_I
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // ---- prologue: activations of chunk 0, weights of chunks 0 and 1; planes of chunk 0
    {
        const std::integral_constant<unsigned, 0> d0{};
        const std::integral_constant<unsigned, 1> d1{};
        pieces<0, 2 * RB>(load_a, d0);
        advance_a();
        load_b(d0);
        load_b(d1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        split_b(d0);
        __syncthreads();
    }
    for (int q = 0; q < nchunks; q += 2) {
        chunk(std::integral_constant<unsigned, 0>{});
        if (q + 1 < nchunks) chunk(std::integral_constant<unsigned, 1>{});
    }

    // ---- epilogue (full tiles, no scale, activation none | ELU: the host routes nothing else here): per column block the
    // wave transposes its 64 x 32 block through LDS and leaves with full 128-byte lines; BatchNorm partial sums per
    // 128-row half of the tile
    {
        constexpr int RW = 32 * RB;                                    // rows of the wave
        float* T = smem_all + wave * (RW * 32);                        // 4 | 8 KB per wave; the ring is dead
        float (*red)[BN][2] = reinterpret_cast<float (*)[BN][2]>(smem_all + 4 * RW * 32);
        const int col_l = lane & 31, rbase = 4 * (lane >> 5);
        const bool elu = a.act == ACT_ELU;
        // (a last, partial tile -- data gradients on grids that are no multiple of 128 pixels: the descriptors end at row M, the
        //  hardware drops the stores beyond)
        const long rows_here = a.M - m0 < BM ? a.M - m0 : BM;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y + m0 * a.ldy, (unsigned)(rows_here * a.ldy * 4));
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(a.add ? a.add + m0 * a.ld_add : a.y, a.add ? (unsigned)(rows_here * a.ld_add * 4) : 0u);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 addv[4 * RB];
            if (a.add) {
                const unsigned off_a = (unsigned)((RW * wave + (lane >> 3)) * (int)a.ld_add + n0 + 32 * j + 4 * (lane & 7)) * 4u;
#pragma unroll
                for (int t = 0; t < 4 * RB; ++t) addv[t] = buf_ld4(ra, off_a + (unsigned)(t * 8 * (int)a.ld_add * 4));
            }
            float s1 = 0.f, s2 = 0.f;
            const float bv = a.bias ? a.bias[n0 + 32 * j + col_l] : 0.f;     // (the statistics are those of conv + bias)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (wsign ? -acc[i][j][r] : acc[i][j][r]) + bv;
                    // (ELU as torch evaluates it, exp(x) - 1, on the hardware exp2: see conv_epilogue)
                    T[(32 * i + (r & 3) + 8 * (r >> 2) + rbase) * 32 + col_l] =
                        elu ? (v > 0.f ? v : __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.f) : v;
                    s1 += v;
                    s2 = __builtin_fmaf(v, v, s2);
                }
            const unsigned off_l = (unsigned)((RW * wave + (lane >> 3)) * (int)a.ldy + n0 + 32 * j + 4 * (lane & 7)) * 4u;
            const float4* Tq = reinterpret_cast<const float4*>(T) + lane;
            // All eight lines are formed first and the stores leave back to back: a 128-bit buffer store reads its data
            // registers for several cycles after issue, and with a scalar offset operand hipcc does not guard them against
            // the next vector instruction (seen here as the last lanes of every 16 storing the NEXT line's values when the
            // sum for line t+1 was written into the registers of store t).
            float4 v[4 * RB];
#pragma unroll
            for (int t = 0; t < 4 * RB; ++t) {
                v[t] = Tq[t * 64];
                if (a.add) { v[t].x += addv[t].x; v[t].y += addv[t].y; v[t].z += addv[t].z; v[t].w += addv[t].w; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (rows_here == BM) {
#pragma unroll
                for (int t = 0; t < 4 * RB; ++t)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[t]), ry, off_l, t * 8 * (int)a.ldy * 4, 0);
            } else {                            // partial tile: a row test per lane (the scalar offset is outside the range check)
#pragma unroll
                for (int t = 0; t < 4 * RB; ++t)
                    if (RW * wave + (lane >> 3) + 8 * t < rows_here)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[t]), ry, off_l + (unsigned)(t * 8 * (int)a.ldy * 4), 0, 0);
            }
            asm volatile("s_nop 1");
            __builtin_amdgcn_sched_barrier(0);
            if (a.stats) {
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (lane < 32) { red[wave][32 * j + col_l][0] = s1; red[wave][32 * j + col_l][1] = s2; }
            }
        }
        if (a.stats) {
            // one row of `stats` per a.stats_rows output pixels (128, or 64 when M < 65536: pd_conv2d_tile_m): WPS waves each
            const int sgr = a.stats_rows, wps = sgr / RW, nrow = BM / sgr;
            __syncthreads();
            if (tid < nrow * BN) {
                const int row_l = tid >> 6, cl = tid & 63;
                float t1 = 0.f, t2 = 0.f;
                for (int w = 0; w < wps; ++w) { t1 += red[row_l * wps + w][cl][0]; t2 += red[row_l * wps + w][cl][1]; }
                float* o = a.stats + ((long)(mt * nrow + row_l) * a.Co + n0 + cl) * 2;
                o[0] = t1; o[1] = t2;
            }
        }
    }
}

// Fewest workgroups that take the bf16-split kernel by default: two per CU.  Below that a SIMD holds one wave most of the
// time and the split is not hidden by another wave's MFMAs: 320 workgroups of 256 rows run at the fp32 kernel's pace
// (3x3x256 @32x40: 116 vs 114 TF), 640 at 1.3x (5x5 256 -> 512 @32x40: 167 vs 127).  128-row tiles (four workgroups per CU)
// are worth it from 320 on (3x3x512 @16x20: 117 -> 134 TF).  PD_CONV_BF16X3 in the caller's flags waives both counts.
constexpr long X3_MIN_WG = 512, X3_MIN_WG1 = 320;

// 0: not for the bf16-split kernel; 2 | 1: its row blocks per wave (256- | 128-row tiles)
// what every bf16-split forward / data-gradient kernel needs of a launch (tile shapes and channel counts are the kernels' own)
static bool x3_common_ok(const ConvArgs& a, bool vec) {
    const bool on = !(a.flags & PD_CONV_FP32_MFMA);
    return on && vec && a.KH * a.KW <= 31 && a.pad < a.KH && a.pad_w < a.KW &&
           (a.mode == MODE_ZERO || (a.mode == MODE_TRANSPOSED && a.sshift == 0) ||
            (a.mode == MODE_REFLECT && a.KH == 3 && a.KW == 3 && a.pad == 1 && a.pad_w == 1 && a.stride == 1 && a.Ho == a.H &&
             a.Wo == a.W && a.H >= 3 && a.W >= 3)) &&
           !a.oscale && (a.act == ACT_NONE || a.act == ACT_ELU) && (a.ldy & 3) == 0 && ((size_t)a.y & 15) == 0 &&
           (!a.add || ((a.ld_add & 3) == 0 && ((size_t)a.add & 15) == 0)) && (long)a.Co * a.K * 4 < 0x7fffffffL;
}
static int x3_eligible(const ConvArgs& a, bool vec) {
    const bool force = (a.flags & PD_CONV_BF16X3) != 0;
    // (C % 4 == 0 is part of `vec`; a partly empty last channel group may at most double the contraction: C >= 8)
    // (whole 128-row tiles, except for a data gradient without BatchNorm statistics: its last tile may be partial)
    if (!(x3_common_ok(a, vec) && (a.C + x3::CK - 1) / x3::CK * x3::CK <= 2 * a.C && a.Co % x3::BN == 0 &&
          (a.M % 128 == 0 || (a.mode == MODE_TRANSPOSED && !a.stats))))
        return 0;
    const long ct = a.Co / x3::BN;
    if (a.M % 256 == 0 && (a.M / 256) * ct >= X3_MIN_WG) return 2;
    // a layer with 320 tiles of 256 rows is better off with its 640 tiles of 128 (64 -> 64 @64x80: 113 vs 102 TF, 3x3x256
    // @32x40: 161 vs 145)
    return (force || (a.M / 128) * ct >= X3_MIN_WG1) ? 1 : 0;
}

// Tile order of the bf16-split kernels.  Default: row tiles outside -- the workgroups that share an XCD's 4 MB L2 work on
// neighbouring pixels and all column tiles (a 64-channel filter is 150 KB).  A filter of several megabytes (3x3x512x512: 9.4 MB,
// 5x5x256x512: 13 MB) would then be streamed through every L2 by every row tile -- 1.1 GB per launch from the Infinity Cache
// for 3x3x512 @16x20, which bounds it (twice the waves per CU changed nothing) -- so with >= 8 column tiles the COLUMN tile goes
// outside: an XCD keeps its slice of the filter and streams the (smaller) activations once.
static int x3_nmajor(const ConvArgs& a) { return (long)a.Co * a.K * 4 > (3L << 20) && a.ntiles >= 8 && a.ntiles % 8 == 0; }
#include "conv_x3_halo.hpp"

static int launch_conv_x3(ConvArgs& a, int rb, hipStream_t st) {
    a.mtiles = (int)((a.M + 128 * rb - 1) / (128 * rb));
    a.ntiles = a.Co / x3::BN;
    a.nmajor = x3_nmajor(a);
    const long nblk = (long)a.mtiles * a.ntiles;
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8)), block(NT);
    if (rb == 2) {
        if (a.mode == MODE_ZERO) hipLaunchKernelGGL((conv_igemm_x3_kernel<MODE_ZERO, 2>), grid, block, 0, st, a);
        else if (a.mode == MODE_REFLECT) hipLaunchKernelGGL((conv_igemm_x3_kernel<MODE_REFLECT, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_x3_kernel<MODE_TRANSPOSED, 2>), grid, block, 0, st, a);
    } else {
        if (a.mode == MODE_ZERO) hipLaunchKernelGGL((conv_igemm_x3_kernel<MODE_ZERO, 1>), grid, block, 0, st, a);
        else if (a.mode == MODE_REFLECT) hipLaunchKernelGGL((conv_igemm_x3_kernel<MODE_REFLECT, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_x3_kernel<MODE_TRANSPOSED, 1>), grid, block, 0, st, a);
    }
    return pd::check_launch("pd_conv2d");
}

template <int BM, int BN, int WM, int WN>
int launch_conv(ConvArgs& a, bool vec, hipStream_t st) {
    a.mtiles = (int)((a.M + BM - 1) / BM);
    a.ntiles = (a.Co + BN - 1) / BN;
    const long nblk = (long)a.mtiles * a.ntiles;
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8)), block(NT);
#define PD_LAUNCH(V, MD) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, V, MD>), grid, block, 0, st, a)
#define PD_LAUNCH_DMA(MD) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true, MD, true>), grid, block, 0, st, a)
    constexpr bool dma = true;                                    // direct-to-LDS staging (the register-staged path serves the scalar gather)
    const bool uni_on = !(a.flags & PD_CONV_GENERAL_KERNELS);       // uniform-tap kernel unless the caller asks for the general one
    if constexpr (BN % 32 == 0 && WN == 32) {
        const bool uni = vec && dma && uni_on && a.C % BK == 0 && a.KH * a.KW <= 31 && a.pad < a.KH && a.pad_w < a.KW &&
                         (a.mode == MODE_ZERO || a.mode == MODE_REFLECT || (a.mode == MODE_TRANSPOSED && a.sshift == 0));
        if (uni && a.mode == MODE_REFLECT) {       // (pad < H, W is checked by pd_conv2d)
            hipLaunchKernelGGL((conv_igemm_uni_kernel<BM, BN, WM, WN, MODE_REFLECT, 2>), grid, block, 0, st, a);
            return pd::check_launch("pd_conv2d");
        }
        if (uni) {
            // ring depth: 2 for the 128-row tiles (3 workgroups per CU); 3 for the 64x64 tile of the small-M layers, whose
            // short chunks (16 MFMAs per wave) leave the loads half the time to land: 118 -> 135 TF on 3x3x256 @32x40
            const int ns = BM == 64 ? 3 : 2;
            if (ns == 2) {
                if (a.mode == MODE_ZERO) hipLaunchKernelGGL((conv_igemm_uni_kernel<BM, BN, WM, WN, MODE_ZERO, 2>), grid, block, 0, st, a);
                else hipLaunchKernelGGL((conv_igemm_uni_kernel<BM, BN, WM, WN, MODE_TRANSPOSED, 2>), grid, block, 0, st, a);
            } else {
                if (a.mode == MODE_ZERO) hipLaunchKernelGGL((conv_igemm_uni_kernel<BM, BN, WM, WN, MODE_ZERO, 3>), grid, block, 0, st, a);
                else hipLaunchKernelGGL((conv_igemm_uni_kernel<BM, BN, WM, WN, MODE_TRANSPOSED, 3>), grid, block, 0, st, a);
            }
            return pd::check_launch("pd_conv2d");
        }
    }
    if (a.pad_w != a.pad) return pd::fail(PD_EINVAL, "pd_conv2d_rect: shape outside the uniform-tap kernel (C %% 32, 16-byte aligned NHWC)");
    if (vec && dma) {
        if (a.mode == MODE_ZERO) PD_LAUNCH_DMA(MODE_ZERO);
        else if (a.mode == MODE_REFLECT) PD_LAUNCH_DMA(MODE_REFLECT);
        else PD_LAUNCH_DMA(MODE_TRANSPOSED);
    } else if (vec) {
        if (a.mode == MODE_ZERO) PD_LAUNCH(true, MODE_ZERO);
        else if (a.mode == MODE_REFLECT) PD_LAUNCH(true, MODE_REFLECT);
        else PD_LAUNCH(true, MODE_TRANSPOSED);
    } else {
        if (a.mode == MODE_ZERO) PD_LAUNCH(false, MODE_ZERO);
        else if (a.mode == MODE_REFLECT) PD_LAUNCH(false, MODE_REFLECT);
        else PD_LAUNCH(false, MODE_TRANSPOSED);
    }
#undef PD_LAUNCH
#undef PD_LAUNCH_DMA
    return pd::check_launch("pd_conv2d");
}

}  // namespace

extern "C" int pd_conv2d_uses_x3(long M, int Co, int C, int KH, int KW, int stride, int pad, int mode, int act,
                                 int has_out_scale, int Ho, int Wo, unsigned flags) {
    ConvArgs a{};
    a.flags = flags;
    a.M = M; a.Co = Co; a.C = C; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = a.pad_w = pad; a.mode = mode; a.act = act;
    a.K = KH * KW * C; a.ldy = Co;
    a.H = a.Ho = Ho > 0 ? Ho : 16;            // (reflection padding: a same-size 3x3 layer is assumed)
    a.W = a.Wo = Wo > 0 ? Wo : 16;
    a.N = Ho > 0 && Wo > 0 ? (int)(M / ((long)Ho * Wo)) : 1;
    a.sC = 1; a.sW = C; a.sH = (long)a.W * C; a.sN = a.sH * a.H;
    a.oscale = has_out_scale ? reinterpret_cast<const float*>(16) : nullptr;
    while ((1 << a.sshift) < stride) ++a.sshift;
    a.stats_rows = pd_conv2d_tile_m(M, Co);
    if (x3_common_ok(a, true) && !(flags & PD_CONV_X3_IM2COL) && Ho > 0 && Wo > 0 && x3_halo_eligible(a)) return 3;
    return x3_eligible(a, true);
}

extern "C" int pd_conv2d_tile_m(long M, int Co) {
    if (Co <= 32) return 128;   // 128x32 tile: four waves of one 32x32 MFMA tile each
    return M >= 64 * 1024 ? 128 : 64;
}

extern "C" long pd_conv2d_stats_rows(long M, int Co) {
    const int bm = pd_conv2d_tile_m(M, Co);
    return (M + bm - 1) / bm;
}

static int conv2d_impl(const void* x, const void* w, const void* bias, const void* out_scale, void* y, void* stats,
                       const void* addend, long ld_add,
                       int N, int H, int W, int C, long sN, long sH, long sW, long sC,
                       int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int mode, int act,
                       int affine, float sub, float div, long ldy, unsigned flags, void* stream, int pad_w = -1);

extern "C" int pd_conv2d(const void* x, const void* w, const void* bias, const void* out_scale, void* y, void* stats,
                         int N, int H, int W, int C, long sN, long sH, long sW, long sC,
                         int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int mode, int act,
                         int affine, float sub, float div, long ldy, unsigned flags, void* stream) {
    return conv2d_impl(x, w, bias, out_scale, y, stats, nullptr, 0, N, H, W, C, sN, sH, sW, sC, Ho, Wo, Co, KH, KW, stride,
                       pad, mode, act, affine, sub, div, ldy, flags, stream);
}

extern "C" int pd_conv2d_add(const void* x, const void* w, const void* addend, long ld_add, void* y,
                             int N, int H, int W, int C, long sN, long sH, long sW, long sC,
                             int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int mode, long ldy, unsigned flags,
                             void* stream) {
    PD_REQUIRE(addend && ld_add >= Co, "pd_conv2d_add: bad addend");
    return conv2d_impl(x, w, nullptr, nullptr, y, nullptr, addend, ld_add, N, H, W, C, sN, sH, sW, sC, Ho, Wo, Co, KH, KW,
                       stride, pad, mode, ACT_NONE, 0, 0.f, 1.f, ldy, flags, stream);
}

// rows x columns filter with its own column padding: the exact sub-filters (1x1, 1x2, 2x1, 2x2) of a stride-2 data
// gradient split by output parity (pd_dgrad_s2_filters); stride 1, no bias / activation, uniform-tap kernel only
extern "C" int pd_conv2d_rect(const void* x, const void* w, void* y, int N, int H, int W, int C, long sN, long sH, long sW,
                              long sC, int Ho, int Wo, int Co, int KH, int KW, int pad_h, int pad_w, int mode, long ldy,
                              unsigned flags, void* stream) {
    PD_REQUIRE(pad_w >= 0 && mode != MODE_REFLECT, "pd_conv2d_rect: bad padding / mode");
    PD_REQUIRE(!(flags & PD_CONV_GENERAL_KERNELS), "pd_conv2d_rect: the general kernel has no separate column padding");
    return conv2d_impl(x, w, nullptr, nullptr, y, nullptr, nullptr, 0, N, H, W, C, sN, sH, sW, sC, Ho, Wo, Co, KH, KW, 1,
                       pad_h, mode, ACT_NONE, 0, 0.f, 1.f, ldy, flags, stream, pad_w);
}

static int conv2d_impl(const void* x, const void* w, const void* bias, const void* out_scale, void* y, void* stats,
                       const void* addend, long ld_add,
                       int N, int H, int W, int C, long sN, long sH, long sW, long sC,
                       int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int mode, int act,
                       int affine, float sub, float div, long ldy, unsigned flags, void* stream, int pad_w) {
    if (pad_w < 0) pad_w = pad;
    PD_REQUIRE(pd::conv_flags_ok(flags), "pd_conv2d: bad flags 0x%x", flags);
    PD_REQUIRE(x && w && y, "pd_conv2d: null tensor");
    PD_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && Co > 0, "pd_conv2d: bad dims");
    PD_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0, "pd_conv2d: bad filter geometry");
    PD_REQUIRE(mode >= 0 && mode <= 2 && act >= 0 && act <= 3, "pd_conv2d: bad mode/act");
    PD_REQUIRE(mode != MODE_REFLECT || (pad < H && pad < W), "pd_conv2d: reflect pad must be < input size");
    PD_REQUIRE(ldy >= Co && ldy < (1L << 22), "pd_conv2d: bad output row stride %ld (Cout=%d)", ldy, Co);
    int sshift = 0;
    while ((1 << sshift) < stride) ++sshift;
    if (mode == MODE_TRANSPOSED) {
        PD_REQUIRE((1 << sshift) == stride, "pd_conv2d: transposed mode needs a power-of-two stride");
        // (a SMALLER x grid is the leading part of that output grid: the missing rows / columns read as zero -- the
        //  sub-filters of a phase-decomposed stride-2 data gradient need exactly that, see pd_dgrad_s2_filters)
        PD_REQUIRE(H <= (Ho + 2 * pad - KH) / stride + 1 && W <= (Wo + 2 * pad_w - KW) / stride + 1,
                   "pd_conv2d: transposed: x grid exceeds the forward output grid of a %dx%d input", Ho, Wo);
    } else {
        // a smaller output grid computes the leading Ho x Wo outputs only (asymmetric bottom/right padding)
        PD_REQUIRE(Ho <= (H + 2 * pad - KH) / stride + 1 && Wo <= (W + 2 * pad_w - KW) / stride + 1,
                   "pd_conv2d: output grid does not match input/filter geometry");
    }
    if (N == 0) return PD_OK;
    ConvArgs a;
    a.x = (const float*)x; a.w = (const float*)w; a.bias = (const float*)bias; a.oscale = (const float*)out_scale; a.y = (float*)y; a.stats = (float*)stats;
    a.add = (const float*)addend; a.ld_add = ld_add;
    a.N = N; a.H = H; a.W = W; a.C = C; a.sN = sN; a.sH = sH; a.sW = sW; a.sC = sC;
    a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.pad_w = pad_w;
    a.mode = mode; a.act = act; a.affine = affine; a.sub = sub; a.div = div;
    a.K = KH * KW * C; a.M = (long)N * Ho * Wo; a.ldy = ldy; a.sshift = sshift;
    a.flags = flags; a.stats_rows = pd_conv2d_tile_m(a.M, Co);
    auto magic = [](long d, unsigned& mg, unsigned& sh) {       // n / d == mulhi(n, mg) >> sh for 0 <= n < 2^31, d >= 2
        if (d <= 1 || d >= (1L << 31)) { mg = 0; sh = 0; return; }
        int l = 0;
        while ((1L << l) < d) ++l;                                // ceil(log2 d) >= 1
        mg = (unsigned)((((unsigned long long)1 << (31 + l)) + (unsigned long long)d - 1) / (unsigned long long)d);
        sh = (unsigned)(l - 1);
    };
    magic((long)Ho * Wo, a.mg_hw, a.sh_hw);
    magic(Wo, a.mg_wo, a.sh_wo);
    const long wbytes = (long)Co * a.K * 4;
    PD_REQUIRE(wbytes < 0x7fffffffL, "pd_conv2d: weight tensor too large for 32-bit offsets");
    a.w_bytes = (unsigned)wbytes;
    // 32-bit byte offsets are relative to the first image of a tile: a tile (up to 256 rows) may span ceil(256/(Ho*Wo))+1 images
    const long span = 256 / ((long)Ho * Wo) + 2;
    PD_REQUIRE(span * sN * 4 < 0x7fffffffL, "pd_conv2d: image too large for 32-bit offsets (%ld bytes per image)", sN * 4);
    const bool vec = (C % 4 == 0) && sC == 1 && (sW % 4 == 0) && (sH % 4 == 0) && (sN % 4 == 0) && !affine &&
                     pd::aligned16(x) && pd::aligned16(w);
    hipStream_t st = (hipStream_t)stream;
    const int bm = pd_conv2d_tile_m(a.M, Co);
    // 96 output columns (the data gradient of the decoder's 96 -> 32 layer): three 32-wide column tiles instead of a full and a
    // half-empty 64-wide one (a quarter of the matrix work of that launch was padding)
    // the halo-tile kernel (every activation element staged and split once per tile) where its tiling fits
    if (x3_common_ok(a, vec) && !(flags & PD_CONV_X3_IM2COL) && x3_halo_eligible(a)) return launch_conv_x3_halo(a, st);
    if (const int rb = x3_eligible(a, vec)) return launch_conv_x3(a, rb, st);
    if (Co == 96 && bm == 128) return launch_conv<128, 32, 32, 32>(a, vec, st);
    if (Co > 32) return bm == 128 ? launch_conv<128, 64, 64, 32>(a, vec, st) : launch_conv<64, 64, 32, 32>(a, vec, st);
    if (Co > 16) return launch_conv<128, 32, 32, 32>(a, vec, st);
    return launch_conv<128, 16, 32, 16>(a, vec, st);   // 16x16x4 MFMA tiles
}

// ===================================================================== weight gradient
// dW[co][kh][kw][ci] = sum_m dY[m][co] * X[pix(m) + tap][ci]   ("TN" GEMM: contraction over pixels).
// Workgroup = one TCO(co) x 128(k) tile of dW for one slice of the pixel range; slices write
// partial tiles to a workspace that pd_reduce_rows() sums deterministically (no float atomics).
// Both operands keep their natural [pixel][channel] layout in LDS; the MFMA fragment of a
// 32-wide channel group at a fixed pixel is 32 consecutive floats (conflict-free ds_read_b32).
// A thread's k columns never change, so its tap (kh,kw,ci) is decoded once; pixel rows advance
// incrementally; all global reads are hardware-bounds-checked buffer loads (no branches).
namespace {

constexpr int WG_K = 128;         // k tile
constexpr int WG_MC = 32;         // pixels per chunk
constexpr int WG_LDX = WG_K + 8;  // padded LDS rows (floats)

struct WgradArgs {
    const float* x;
    const float* dy;
    float* part;      // [S][Co][K]
    float* bpart;     // [S][Co] or null
    int N, H, W, C;
    long sN, sH, sW, sC;
    int Ho, Wo, Co;
    int KH, KW, stride, pad, mode;
    int affine;
    float sub, div;
    int K;
    long M, ldd;      // pixels, row stride of dy
    long mper;        // pixels per slice (multiple of WG_MC)
    int S, ktiles, ctiles;
};

template <int TCO, bool VEC, int MODE>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int LDD = TCO + 8;
    constexpr int MT = TCO == 16 ? 16 : 32;       // MFMA tile: 32x32x2, or 16x16x4 for <= 16 output channels
    constexpr int ACCN = MT == 32 ? 16 : 4;
    constexpr int WAVES_CO = TCO / MT;            // 2 (TCO=64) or 1
    constexpr int WAVES_K = 4 / WAVES_CO;         // 2 or 4
    constexpr int TK = WG_K / WAVES_K / MT;       // MFMA tiles along k per wave: 2, 1 or (16-wide) 2
    constexpr int DY_ITERS = (WG_MC * TCO / 4 + NT - 1) / NT;   // 16-byte dY pieces per thread per chunk: 2 or 1
    constexpr int DY_PPR = TCO / 4;                  // pieces per row
    constexpr bool DY_ALL = WG_MC * TCO / 4 >= NT;   // false: only the first WG_MC*TCO/4 threads load dY
    __shared__ __attribute__((aligned(16))) float Ds[2][WG_MC][LDD];
    __shared__ __attribute__((aligned(16))) float Xs[2][WG_MC][WG_LDX];
    __shared__ int ktab[VEC ? 1 : WG_K][3];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_K, wn = wave - wm * WAVES_K;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so consecutive *logical* tiles -- the
    // k-tiles of one pixel slice, which read the same dY rows and overlapping X rows -- are mapped to one XCD
    // and share its L2 (measured HBM traffic of this kernel was 4x its algorithmic bytes without it).
    const int nwg = a.ktiles * a.ctiles * a.S;
    int b = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (b >= nwg) return;
    const int kt = b % a.ktiles; b /= a.ktiles;
    const int ct = b % a.ctiles; b /= a.ctiles;
    const int s = b;
    const int co0 = ct * TCO, k0 = kt * WG_K;
    const long mbeg = (long)s * a.mper;
    const long mend = (mbeg + a.mper < a.M) ? mbeg + a.mper : a.M;
    const int nrows = (int)(mend - mbeg);
    const int nchunks = nrows > 0 ? (nrows + WG_MC - 1) / WG_MC : 0;
    const int hw = a.Ho * a.Wo;

    const int img0 = (int)(mbeg / hw);
    const long rest = ((long)a.N - img0) * a.sN * 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)img0 * a.sN, rest > 0x7fffffffL ? 0x7fffffffu : (unsigned)rest);
    const long dbytes = (long)nrows * a.ldd * 4;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(a.dy + mbeg * a.ldd, dbytes > 0 ? (unsigned)dbytes : 0u);

    // ---- fixed k columns of this thread
    // VEC: one 16-byte piece k = k0 + 4*(tid&31), rows (tid>>5) + 8*i, i < 4
    // scalar: row tid>>3, sixteen columns (tid&7) + 8*j through the LDS tap table
    int vkh = 0, vkw = 0, vc = 0; bool vkv = false;
    if (VEC) {
        const int k = k0 + 4 * (tid & 31);
        vkv = k < a.K;
        const int tap = k / a.C;
        vc = k - tap * a.C; vkh = tap / a.KW; vkw = tap - vkh * a.KW;
    } else {
        if (tid < WG_K) {
            const int k = k0 + tid;
            const int tap = k / a.C;
            const int kh = tap / a.KW;
            ktab[tid][0] = k < a.K ? kh : -1; ktab[tid][1] = tap - kh * a.KW; ktab[tid][2] = (k - tap * a.C) * (int)a.sC;
        }
        __syncthreads();
    }
    // ---- pixel rows of this thread, advanced by WG_MC per chunk
    constexpr int XR = VEC ? 4 : 1;
    int rb[XR], rloc[XR];                         // image offset (elements), row index inside the slice
    int roh[XR], row_[XR];
    for (int i = 0; i < XR; ++i) {
        rloc[i] = VEC ? (tid >> 5) + 8 * i : (tid >> 3);
        const int m = (int)(mbeg - (long)img0 * hw) + rloc[i];   // pixel index relative to image img0
        const int n = m / hw;
        const int rem = m - n * hw;
        roh[i] = rem / a.Wo; row_[i] = rem - roh[i] * a.Wo;
        rb[i] = n * (int)a.sN;
    }
    // Zero-padding mode, 16-byte path: input coordinates and the element offset of this thread's tap are carried
    // incrementally from chunk to chunk (additions only; the general form below multiplies per row and chunk).
    constexpr bool INC = VEC && MODE == MODE_ZERO;
    int xih[INC ? XR : 1], xiw[INC ? XR : 1], xoff[INC ? XR : 1];
    const int c_w = WG_MC * a.stride, c_wo = c_w * (int)a.sW;
    const int c_row = a.Wo * a.stride, c_rowo = a.stride * (int)a.sH - c_row * (int)a.sW;
    const int c_img = a.Ho * a.stride, c_imgo = (int)a.sN - c_img * (int)a.sH;
    if (INC) {
        for (int i = 0; i < XR; ++i) {
            xih[i] = roh[i] * a.stride - a.pad + vkh;
            xiw[i] = row_[i] * a.stride - a.pad + vkw;
            xoff[i] = rb[i] + xih[i] * (int)a.sH + xiw[i] * (int)a.sW + vc;
        }
    }
    auto advance = [&](int i) {
        rloc[i] += WG_MC;
        row_[i] += WG_MC;
        if (INC) { xiw[i] += c_w; xoff[i] += c_wo; }
        while (row_[i] >= a.Wo) {
            row_[i] -= a.Wo;
            if (INC) { xiw[i] -= c_row; xih[i] += a.stride; xoff[i] += c_rowo; }
            if (++roh[i] == a.Ho) {
                roh[i] = 0; rb[i] += (int)a.sN;
                if (INC) { xih[i] -= c_img; xoff[i] += c_imgo; }
            }
        }
    };

    float4 pd[DY_ITERS], px[VEC ? 4 : 1];
    float sx[VEC ? 1 : 16];

    auto load_chunk = [&](int q) {
        const int rbase = q * WG_MC;
#pragma unroll
        for (int i = 0; i < DY_ITERS; ++i) {
            const int r = rbase + tid / DY_PPR + (NT / DY_PPR) * i;
            const int co = co0 + 4 * (tid % DY_PPR);
            const unsigned off = (unsigned)(r * (int)a.ldd + co) * 4u;
            if (!DY_ALL && tid >= WG_MC * DY_PPR) {
                pd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if ((a.Co & 3) == 0) {
                pd[i] = buf_ld4(rd, (r < nrows && co < a.Co) ? off : OOB);
            } else {   // ragged channel count (e.g. the 1-channel disparity heads): element-wise
                float4 v;
                v.x = buf_ld1(rd, (r < nrows && co < a.Co) ? off : OOB);
                v.y = buf_ld1(rd, (r < nrows && co + 1 < a.Co) ? off + 4 : OOB);
                v.z = buf_ld1(rd, (r < nrows && co + 2 < a.Co) ? off + 8 : OOB);
                v.w = buf_ld1(rd, (r < nrows && co + 3 < a.Co) ? off + 12 : OOB);
                pd[i] = v;
            }
        }
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (INC) {
                    const bool ok = ((unsigned)xih[i] < (unsigned)a.H) & ((unsigned)xiw[i] < (unsigned)a.W) & vkv & (rloc[i] < nrows);
                    px[i] = buf_ld4(rx, ok ? (unsigned)xoff[i] * 4u : OOB);
                } else {
                    int ih, iw;
                    const bool ok = tap_in<MODE>(roh[i] * a.stride - a.pad, row_[i] * a.stride - a.pad, vkh, vkw, a.H, a.W, 0, ih, iw) &
                                    vkv & (rloc[i] < nrows);
                    const unsigned off = (unsigned)(rb[i] + ih * (int)a.sH + iw * (int)a.sW + vc) * 4u;
                    px[i] = buf_ld4(rx, ok ? off : OOB);
                }
                advance(i);
            }
        } else {
            const int rh0 = roh[0] * a.stride - a.pad, rw0 = row_[0] * a.stride - a.pad;
            const bool rv = rloc[0] < nrows;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int kc = (tid & 7) + 8 * j;
                const int kh = ktab[kc][0];
                int ih, iw;
                const bool ok = tap_in<MODE>(rh0, rw0, kh, ktab[kc][1], a.H, a.W, 0, ih, iw) & (kh >= 0) & rv;
                const unsigned off = (unsigned)(rb[0] + ih * (int)a.sH + iw * (int)a.sW + ktab[kc][2]) * 4u;
                float v = buf_ld1(rx, ok ? off : OOB);
                if (a.affine) v = ok ? (v - a.sub) / a.div : 0.f;
                sx[j] = v;
            }
            advance(0);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DY_ITERS; ++i)
            if (DY_ALL || tid < WG_MC * DY_PPR)
                *reinterpret_cast<float4*>(&Ds[buf][tid / DY_PPR + (NT / DY_PPR) * i][4 * (tid % DY_PPR)]) = pd[i];
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Xs[buf][(tid >> 5) + 8 * i][4 * (tid & 31)]) = px[i];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) Xs[buf][tid >> 3][(tid & 7) + 8 * j] = sx[j];
        }
    };

    typedef float accv_t __attribute__((ext_vector_type(ACCN)));
    accv_t acc[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int r = 0; r < ACCN; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = a.bpart != nullptr && kt == 0;

    if (nchunks > 0) { load_chunk(0); store_chunk(0); }
    __syncthreads();
    // fragment lane -> (index inside the MFMA tile, pixel of the step): 2 pixels per 32x32x2, 4 per 16x16x4
    constexpr int PS = 64 / MT;
    const int fi = lane % MT, fk = lane / MT;
    for (int q = 0; q < nchunks; ++q) {
        const int buf = q & 1;
        if (q + 1 < nchunks) load_chunk(q + 1);
        // Fragments are read one group (four pixel steps) ahead of their MFMAs; scheduling barriers keep the order
        // (left alone, the compiler reads each step right before its two MFMAs and waits for LDS every time).
        constexpr int NST = WG_MC / PS, UG = 4, NG = NST / UG;
        float av[2][UG], bv[2][UG][TK];
        auto read_group = [&](int sg, int slot) {
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                av[slot][u] = Ds[buf][PS * (UG * sg + u) + fk][wm * MT + fi];
#pragma unroll
                for (int t = 0; t < TK; ++t) bv[slot][u][t] = Xs[buf][PS * (UG * sg + u) + fk][(wn * TK + t) * MT + fi];
            }
        };
        read_group(0, 0);
#pragma unroll
        for (int sg = 0; sg < NG; ++sg) {
            if (sg + 1 < NG) read_group(sg + 1, (sg + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UG; ++u)
#pragma unroll
                for (int t = 0; t < TK; ++t) {
                    if constexpr (MT == 32) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sg & 1][u], bv[sg & 1][u][t], acc[t], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sg & 1][u], bv[sg & 1][u][t], acc[t], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_bias && tid < TCO) {
#pragma unroll 8
            for (int r = 0; r < WG_MC; ++r) bsum += Ds[buf][r][tid];
        }
        if (q + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // C/D layout: col = lane % MT -> k; row -> co: (r&3) + 8*(r>>2) + 4*(lane>>5) (32x32), r + 4*(lane>>4) (16x16)
#pragma unroll
    for (int t = 0; t < TK; ++t) {
        const int k = k0 + (wn * TK + t) * MT + fi;
#pragma unroll
        for (int r = 0; r < ACCN; ++r) {
            const int co = co0 + wm * MT + (MT == 32 ? (r & 3) + 8 * (r >> 2) : r) + 4 * fk;
            if (co < a.Co && k < a.K) a.part[((long)s * a.Co + co) * a.K + k] = acc[t][r];
        }
    }
    if (do_bias && tid < TCO && co0 + tid < a.Co) a.bpart[(long)s * a.Co + co0 + tid] = bsum;
}

// ===================================================================== weight gradient, scalar-pixel variant
// Same reasoning as conv_igemm_uni_kernel: the fp32 MFMA shares the SIMD's FMA lanes with every VALU instruction, and
// the kernel above pays ~100 of them per 32-pixel chunk (tap bounds tests, incremental pixel coordinates, LDS stores)
// against 32 MFMAs.  Here both operands go global -> LDS directly and the per-lane part of every address is fixed:
//   * a lane's k column never changes, so its tap offset (and the +1 pixel of the upper half-wave) is loop invariant;
//   * one LDS-DMA instruction of X covers ONE pixel pair (ow, ow+1) of one output row (Wo is even), so the pixel's
//     offset is a scalar (soffset) and the pair's border class -- (top rows | interior | bottom rows) x (left pair |
//     interior | right pair) -- is a scalar index into a per-lane bit mask built once ("this lane's tap leaves the
//     image for pixels of that class"); bit 31 of the offset drops an invalid piece in the range check;
//   * dY rows are contiguous: four pixels per instruction, scalar row offset, the slice end is the descriptor's end;
//   * unpadded LDS rows; the two pixels of an MFMA step sit 64/128 floats apart, so odd pixels store their row
//     XOR 32 floats (applied to the source address of the DMA) -- conflict-free ds_read_b32 fragments.
// Per chunk and wave: 32 MFMAs, 48 ds_read_b32, 6 LDS-DMA issues, 8 VALU.
struct WgradUniArgs {
    WgradArgs g;
    int nb, nbw;          // border rows / border pixel pairs that can hold an invalid tap
};

// REFLECT (ReflectionPad2d(1) + Conv3x3, the decoder): no tap is invalid; a lane whose tap leaves the image for the
// pair's row / column gets +-2 rows / columns added, selected by four scalar flags of the pair (top, bottom, first,
// last column): 6 VALU per X piece instead of 2.  BIAS: the kt == 0 tiles also sum dY over their pixels.
// TCO = 64: waves 2 (co) x 2 (k), 32x64 wave tiles; TCO = 32 (17..32 output channels): waves 1 x 4, 32x32 wave tiles.
// X3: the products run on the bf16 matrix cores as in conv_igemm_x3_kernel (three-way split of BOTH operands in
// registers -- dY and X are activations, there is nothing to pre-split): a lane's eight values of a 16-pixel MFMA step
// are the eight ds_read_b32 the fp32 path issues for those pixels (pixel 2e + lane/32 of the step, e = 0..7: the
// contraction index may be permuted as long as both operands agree), so staging, LDS image and addresses do not change.
// Per chunk and wave: 24 bf16 MFMAs (768 cycles instead of 2048) and ~220 vector instructions.
template <bool REFLECT, bool BIAS, int TCO = 64, bool X3 = false>
__global__ __launch_bounds__(NT) void conv_wgrad_uni_kernel(const WgradUniArgs ua) {
    const WgradArgs& a = ua.g;
    constexpr int MT = 32, NW_K = TCO == 64 ? 2 : 4, TK = WG_K / MT / NW_K;   // k waves, 32-wide k sub-tiles per wave
    constexpr int PPD = 1024 / (TCO * 4), NDP = 8 / PPD;                         // pixels per dY piece (4 | 8), dY pieces per wave and chunk (2 | 1)
    constexpr int NP = NDP + 4;                                                   // LDS-DMA pieces per wave and chunk
    constexpr unsigned D_ROW = TCO * 4, X_ROW = WG_K * 4;                 // bytes per pixel row in LDS
    constexpr unsigned D_BYTES = WG_MC * D_ROW, X_BYTES = WG_MC * X_ROW;   // per buffer: 8 KB, 16 KB
    __shared__ __attribute__((aligned(16))) float smem_all[(2 * D_BYTES + 2 * X_BYTES) / 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NW_K, wn = wave % NW_K;
    const int nwg = a.ktiles * a.ctiles * a.S;
    int b = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (b >= nwg) return;
    const int kt = b % a.ktiles; b /= a.ktiles;
    const int ct = b % a.ctiles; b /= a.ctiles;
    const int s = b;
    const int co0 = ct * TCO, k0 = kt * WG_K;
    const long mbeg = (long)s * a.mper;
    const long mend = (mbeg + a.mper < a.M) ? mbeg + a.mper : a.M;
    const int nrows = (int)(mend - mbeg);
    const int nchunks = nrows > 0 ? (nrows + WG_MC - 1) / WG_MC : 0;
    const int hw = a.Ho * a.Wo;

    const int img0 = (int)(mbeg / hw);
    const long shift = (long)a.pad * (a.sH + a.sW);                 // descriptor base moved back by the padding
    const long rest = (((long)a.N - img0) * a.sN + shift) * 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)img0 * a.sN - shift, rest > 0x7fffffffL ? 0x7fffffffu : (unsigned)rest);
    const float* dy0 = a.dy + mbeg * a.ldd;

    // ---- per-lane invariants
    // X: lane -> (pixel of the pair = lane/32, physical 16-byte slot = lane%32, logical slot = slot ^ 8*(pixel&1))
    const int xh = lane >> 5;
    const int kx0 = k0 + 4 * ((lane & 31) ^ (8 * xh));
    const int kx = REFLECT ? min(kx0, a.K - 4) : kx0;   // reflect: lanes past K re-read the last columns (never stored)
    const int tap = kx / a.C;
    const int xc = kx - tap * a.C, xkh = tap / a.KW, xkw = tap - xkh * a.KW;
    const unsigned vx = (unsigned)(xkh * (int)a.sH + (xkw + xh * a.stride) * (int)a.sW + xc) * 4u;
    const int ncw = 2 * ua.nbw + 1;
    // reflect corrections of this lane's tap (pad 1, 3x3, stride 1): row -1 -> 1, row H -> H-2, same for columns; the
    // column flags refer to the pair's first pixel, so they apply to one half-wave each
    const unsigned c_top = REFLECT && xkh == 0 ? (unsigned)(2 * (int)a.sH * 4) : 0u;
    const unsigned c_bot = REFLECT && xkh == 2 ? (unsigned)(-2 * (int)a.sH * 4) : 0u;
    const unsigned c_left = REFLECT && xkw == 0 && xh == 0 ? (unsigned)(2 * (int)a.sW * 4) : 0u;
    const unsigned c_right = REFLECT && xkw == 2 && xh == 1 ? (unsigned)(-2 * (int)a.sW * 4) : 0u;
    unsigned xmask = 0x80000000u;                                    // bit 31: the "beyond the tensor" class
    if (REFLECT) {
        xmask = 0;
    } else if (kx < a.K) {
        for (int ch = 0; ch <= 2 * ua.nb; ++ch) {
            const int oh = ch <= ua.nb ? ch : a.Ho - ua.nb + (ch - ua.nb - 1);
            const bool bad_h = (unsigned)(oh * a.stride - a.pad + xkh) >= (unsigned)a.H;
            for (int cw = 0; cw < ncw; ++cw) {
                const int ow = (cw <= ua.nbw ? 2 * cw : a.Wo - 2 * ua.nbw + 2 * (cw - ua.nbw - 1)) + xh;
                const bool bad_w = (unsigned)(ow * a.stride - a.pad + xkw) >= (unsigned)a.W;
                xmask |= (unsigned)(bad_h | bad_w) << (ch * ncw + cw);
            }
        }
    } else {
        xmask = 0xffffffffu;
    }
    // dY: lane -> (pixel of the quad = lane/16, physical slot = lane%16, logical slot = slot ^ 8*(pixel&1))
    const int dpi = lane / (TCO / 4);
    // (64-float rows: the two pixels of an MFMA step share banks -> odd pixels are stored XOR 32 floats; 32-float rows
    // of adjacent pixels already sit on the two bank halves)
    const int dco = co0 + 4 * (TCO == 64 ? ((lane & 15) ^ (8 * (dpi & 1))) : (lane & 7));
    const unsigned vd = dco < a.Co ? (unsigned)(dpi * (int)a.ldd + dco) * 4u : OOB;

    // ---- scalar pixel state of this wave: it stages pixels 8*wave .. 8*wave+7 of every chunk
    const int st_h4 = a.stride * (int)a.sH * 4, st_w4 = a.stride * (int)a.sW * 4, sn4 = (int)a.sN * 4;
    const int d_row4 = st_h4 - a.Wo * st_w4, d_img4 = sn4 - a.Ho * st_h4;   // offset steps at a row / image wrap
    int s_p = 8 * wave;                                              // pixel index inside the slice
    int s_oh, s_ow, s_soff;                                          // its row, column and byte offset (without the tap)
    {
        const int m = (int)(mbeg - (long)img0 * hw) + s_p;
        const int n = m / hw;
        const int rem = m - n * hw;
        s_oh = rem / a.Wo; s_ow = rem - s_oh * a.Wo;
        s_soff = n * sn4 + s_oh * st_h4 + s_ow * st_w4;
        // the divisions run on the VALU: pin the results to SGPRs, or the whole bookkeeping chain follows them there
        s_oh = __builtin_amdgcn_readfirstlane(s_oh); s_ow = __builtin_amdgcn_readfirstlane(s_ow);
        s_soff = __builtin_amdgcn_readfirstlane(s_soff);
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t*)smem_all;
    const unsigned m0_d = lds0 + (unsigned)wave * 8 * D_ROW, m0_x = lds0 + 2 * D_BYTES + (unsigned)wave * 8 * X_ROW;

    // LDS-DMA piece P of the chunk the scalar state points at: 0, 1 = dY (four pixels each), 2..5 = X (one pixel pair
    // each; the pair state moves on).  Chunks past the slice fetch nothing: the dY descriptor has no records left and
    // the pair's class is the "beyond the tensor" bit.
    auto load_piece = [&](auto dst_tag, auto piece_tag) {
        constexpr unsigned DST = decltype(dst_tag)::value;
        constexpr int P = decltype(piece_tag)::value;
        if constexpr (P < NDP) {
            // the scalar offset of a buffer access is outside the range check, so the slice end is enforced by a
            // descriptor per instruction (base and record count are SALU arithmetic): rows >= nrows read as zero
            const int r0 = s_p + PPD * P, left = nrows - r0;
            const __amdgpu_buffer_rsrc_t rd = make_rsrc(dy0 + (long)r0 * a.ldd, left > 0 ? (unsigned)(left * (int)a.ldd) * 4u : 0u);
            dma16s(rd, m0_d + DST * D_BYTES + P * PPD * D_ROW, vd, 0u);
        } else {
            constexpr int J = P - NDP;
            // border class of the pair: rows 0..nb-1 | interior | last nb rows  x  the same over column pairs (min/max, no branches)
            const int ch = min(s_oh, ua.nb) + max(s_oh - (a.Ho - ua.nb) + 1, 0);
            const int pw = s_ow >> 1;
            const int cw = min(pw, ua.nbw) + max(pw - ((a.Wo >> 1) - ua.nbw) + 1, 0);
            if constexpr (REFLECT) {
                // pairs past the slice read an interior pair instead (their dY rows are zero): nothing leaves the tensor
                const bool live = s_p < nrows;
                const unsigned f_t = live && s_oh == 0 ? ~0u : 0u, f_b = live && s_oh == a.Ho - 1 ? ~0u : 0u;
                const unsigned f_l = live && s_ow == 0 ? ~0u : 0u, f_r = live && s_ow == a.Wo - 2 ? ~0u : 0u;
                const unsigned voff = vx + (f_t & c_top) + (f_b & c_bot) + ((f_l & c_left) + (f_r & c_right));
                dma16s(rx, m0_x + DST * X_BYTES + J * 2 * X_ROW, voff,
                       live ? (unsigned)s_soff : (unsigned)((int)a.sH + 2 * (int)a.sW) * 4u);
            } else {
            const unsigned cls = (unsigned)(ch * ncw + cw) | (s_p < nrows ? 0u : 31u);   // (classes are < 31)
            const unsigned bad = __builtin_amdgcn_ubfe(xmask, cls, 1u);
            dma16s(rx, m0_x + DST * X_BYTES + J * 2 * X_ROW, (bad << 31) + vx, (unsigned)s_soff);
            }
            // Pixel bookkeeping as selects, not branches: straight-line scalar code that the scheduler can slide under
            // the 64-cycle MFMAs around it (a branchy version costs the wave ~150 cycles per piece with nothing issued).
            constexpr int STEP = J < 3 ? 2 : WG_MC - 6;        // next pair of the row / this wave's first pair of the next chunk
            s_p += STEP; s_ow += STEP; s_soff += STEP * st_w4;
#pragma unroll
            for (int w = 0; w < (J < 3 ? 1 : 2); ++w) {        // Wo >= 14 (host check): 26 columns wrap at most twice
                const int ow0 = s_ow;
                s_ow = ow0 >= a.Wo ? ow0 - a.Wo : ow0;
                s_soff += ow0 >= a.Wo ? d_row4 : 0;
                const int oh1 = ow0 >= a.Wo ? s_oh + 1 : s_oh;
                s_soff += oh1 == a.Ho ? d_img4 : 0;            // (without a wrap oh1 = oh < Ho)
                s_oh = oh1 == a.Ho ? 0 : oh1;
            }
        }
    };

    // A k tile with at most 64 valid columns (the last tile of K = 576: 4.5 tiles) would leave the wn = 1 waves idle:
    // there both wave columns take the first 64 columns and split the pixel steps (fragment groups alternate); the
    // two partial tiles are added through LDS at the end.
    const bool half = k0 + 2 * MT >= a.K;
    const int wn_k = half ? wn % (NW_K / 2) : wn, hs = wn / (NW_K / 2);     // (half mode: k wave, helper index 0 | 1)
    // ---- fragments: lane -> (index inside the 32-wide tile = lane%32, pixel of the step = lane/32)
    const int fi = lane & 31, fk = lane >> 5;
    unsigned fd_off = fk * D_ROW + 4u * (TCO == 64 ? ((wm * MT + fi) ^ (32 * fk)) : fi);
    unsigned fx_off[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) fx_off[t] = 2 * D_BYTES + fk * X_ROW + 4u * (((wn_k * TK + t) * MT + fi) ^ (32 * fk));
    if (half) {     // the wave's first fragment group is group wn: folded into the lane base, the chunk reads "groups 0 and 2"
        fd_off += hs * (4 * 2 * D_ROW);
#pragma unroll
        for (int t = 0; t < TK; ++t) fx_off[t] += hs * (4 * 2 * X_ROW);
    }
    asm volatile("" : "+v"(fd_off), "+v"(fx_off[0]), "+v"(fx_off[TK - 1]));
    const char* lds_c = reinterpret_cast<const char*>(smem_all);

    typedef float accv_t __attribute__((ext_vector_type(16)));
    accv_t acc[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = BIAS && a.bpart != nullptr && kt == 0;

    // One chunk: the MFMAs of buffer BUF; the six pieces of the next chunk go out between the four fragment groups,
    // in the shadow of the wave's own MFMAs.  HALF (compile time: the two loops below never meet inside a chunk, or
    // the accumulators would be copied between their register sets every chunk): groups wn and wn + 2 only.
    auto chunk = [&](auto buf_tag, auto half_tag) {
        constexpr unsigned BUF = decltype(buf_tag)::value;
        constexpr bool HALF = decltype(half_tag)::value;
        const std::integral_constant<unsigned, BUF ^ 1> dst{};
        constexpr int NST = WG_MC / 2, UG = 4, NG = NST / UG;
        float av[2][UG], bv[2][UG][TK];
        auto read_group = [&](int sg, int slot) {
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const unsigned step = UG * sg + u;
                av[slot][u] = *reinterpret_cast<const float*>(lds_c + fd_off + (BUF * D_BYTES + step * 2 * D_ROW));
#pragma unroll
                for (int t = 0; t < TK; ++t)
                    bv[slot][u][t] = *reinterpret_cast<const float*>(lds_c + fx_off[t] + (BUF * X_BYTES + step * 2 * X_ROW));
            }
        };
        if constexpr (X3) {
            // eight steps (16 pixels) per MFMA k-step; HALF: the wave's two fragment groups are ONE k-step
            constexpr int NKS = HALF ? 1 : 2;
            float a8[NKS][8], b8[NKS][TK][8];
#pragma unroll
            for (int g = 0; g < NKS; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned step = HALF ? (e < 4 ? e : e + 4) : 8 * g + e;
                    a8[g][e] = *reinterpret_cast<const float*>(lds_c + fd_off + (BUF * D_BYTES + step * 2 * D_ROW));
#pragma unroll
                    for (int t = 0; t < TK; ++t)
                        b8[g][t][e] = *reinterpret_cast<const float*>(lds_c + fx_off[t] + (BUF * X_BYTES + step * 2 * X_ROW));
                }
            auto bf = [](u32x4 v) { return __builtin_bit_cast(x3::bf16x8, v); };
#pragma unroll
            for (int g = 0; g < NKS; ++g) {
                const x3::Split sa = x3::split8(make_float4(a8[g][0], a8[g][1], a8[g][2], a8[g][3]),
                                                make_float4(a8[g][4], a8[g][5], a8[g][6], a8[g][7]));
#pragma unroll
                for (int t = 0; t < TK; ++t) {
                    const x3::Split sb = x3::split8(make_float4(b8[g][t][0], b8[g][t][1], b8[g][t][2], b8[g][t][3]),
                                                    make_float4(b8[g][t][4], b8[g][t][5], b8[g][t][6], b8[g][t][7]));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa.lo), bf(sb.hi), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa.hi), bf(sb.lo), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa.mid), bf(sb.mid), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa.mid), bf(sb.hi), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa.hi), bf(sb.mid), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(sa.hi), bf(sb.hi), acc[t], 0, 0, 0);
                }
                if (g == 0) pieces<0, NP / NKS>(load_piece, dst);
                else pieces<NP / 2, NP>(load_piece, dst);
            }
        } else if constexpr (!HALF) {
            read_group(0, 0);
#pragma unroll
            for (int sg = 0; sg < NG; ++sg) {
                if (sg + 1 < NG) read_group(sg + 1, (sg + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UG; ++u)
#pragma unroll
                    for (int t = 0; t < TK; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sg & 1][u], bv[sg & 1][u][t], acc[t], 0, 0, 0);
                if (sg == 0) pieces<0, NP / 3>(load_piece, dst);
                else if (sg == 1) pieces<NP / 3, 2 * NP / 3>(load_piece, dst);
                else if (sg == 2) pieces<2 * NP / 3, NP>(load_piece, dst);
            }
        } else {
            read_group(0, 0);
            read_group(2, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int u = 0; u < UG; ++u)
#pragma unroll
                    for (int t = 0; t < TK; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[h][u], bv[h][u][t], acc[t], 0, 0, 0);
                if (h == 0) pieces<0, NP / 2>(load_piece, dst);
                else pieces<NP / 2, NP>(load_piece, dst);
            }
        }
        if constexpr (BIAS) {
            if (do_bias && tid < TCO) {
#pragma unroll 8
                for (int r = 0; r < WG_MC; ++r)
                    bsum += *reinterpret_cast<const float*>(lds_c + BUF * D_BYTES + r * D_ROW + 4u * (TCO == 64 ? (tid ^ (32 * (r & 1))) : tid));
            }
        }
        dma_wait();
        __syncthreads();
    };

    pieces<0, NP>(load_piece, std::integral_constant<unsigned, 0>{});
    dma_wait();
    __syncthreads();
    if (!half) {
        for (int q = 0; q < nchunks; q += 2) {
            chunk(std::integral_constant<unsigned, 0>{}, std::false_type{});
            if (q + 1 < nchunks) chunk(std::integral_constant<unsigned, 1>{}, std::false_type{});
        }
    } else {
        for (int q = 0; q < nchunks; q += 2) {
            chunk(std::integral_constant<unsigned, 0>{}, std::true_type{});
            if (q + 1 < nchunks) chunk(std::integral_constant<unsigned, 1>{}, std::true_type{});
        }
    }

    if (half) {                             // (uniform per workgroup; the ring is drained: every chunk ends with a barrier)
        float* xch = smem_all + ((wm * (NW_K / 2) + wn_k) * 64 + lane) * 33;   // 33-float rows: conflict-free
        if (hs == 1) {
#pragma unroll
            for (int t = 0; t < TK; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[16 * t + r] = acc[t][r];
        }
        __syncthreads();
        if (hs == 1) return;
#pragma unroll
        for (int t = 0; t < TK; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += xch[16 * t + r];
    }
    // C/D layout: col = lane % 32 -> k; row -> co: (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int t = 0; t < TK; ++t) {
        const int k = k0 + (wn_k * TK + t) * MT + fi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * MT + (r & 3) + 8 * (r >> 2) + 4 * fk;
            if (co < a.Co && k < a.K) a.part[((long)s * a.Co + co) * a.K + k] = acc[t][r];
        }
    }
    if (do_bias && tid < TCO && co0 + tid < a.Co) a.bpart[(long)s * a.Co + co0 + tid] = bsum;
}

// ===================================================================== weight gradient, bf16-split products ("x3c")
// The weight gradient contracts over pixels and BOTH operands are activations: split in registers by every wave that
// needs them (the X3 branch above: 2 x 2 waves, each element split twice) the vector pipe has 216 instructions per 24
// MFMAs and the kernel loses.  Here every element is split once:
//   * the four waves take 32 of the 128 k columns each and all 64 output channels (wave tile 64 x 32): a wave's X values
//     are its own -- eight ds_read_b32 per 16-pixel MFMA step, split in registers (36 instructions per step);
//   * dY (32 pixels x 64 channels per chunk) is needed by all four waves: it is split ONCE per workgroup, one chunk
//     ahead, by all 256 threads (thread = channel x pixel octet: eight ds_read_b32 of the fp32 staging tile, 36
//     instructions, three ds_write_b128) into bf16 planes laid out in fragment order -- an A fragment of the MFMA is one
//     ds_read_b128;
//   * dY is staged two chunks ahead (its two-slot ring holds chunk q+1 being split and chunk q+2 in flight), X one;
//   * the 24 MFMAs of a chunk carry the vector work between them in program order, as in conv_igemm_x3_kernel.
// Same staging (LDS-DMA pieces, scalar pixel state, border-class masks) as conv_wgrad_uni_kernel; zero padding, 64-wide
// co tile only.  LDS: 48 KB staging + 2 x 12 KB planes = 72 KB (dynamic), two workgroups per CU.
namespace x3c {
constexpr unsigned D_ROW = 64 * 4, X_ROW = WG_K * 4;
constexpr unsigned D_BYTES = WG_MC * D_ROW, X_BYTES = WG_MC * X_ROW;          // 8 KB, 16 KB
constexpr unsigned P_BASE = 2 * D_BYTES + 2 * X_BYTES;                        // planes behind the staging tiles (48 KB)
constexpr unsigned P_BYTES = 3 * 4 * 64 * 16;                                 // [term][k-step][half][co][8 x bf16]
constexpr unsigned LDS_BYTES = P_BASE + 2 * P_BYTES;                          // 72 KB
}

// ROW8: the host guarantees Wo % 8 == 0 and slices of whole 8-pixel groups, so the eight pixels a wave stages per chunk
// never straddle an output row: one row class and one 32-pixel advance per chunk instead of a state update per pixel pair.
// REFLECT (ReflectionPad2d(1) + Conv3x3, the decoder): no tap is invalid; a lane whose tap leaves the image for the pair's
// row / column gets +-2 rows / columns added, as in conv_wgrad_uni_kernel.
template <bool BIAS, bool ROW8, bool REFLECT = false>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_x3c_kernel(const WgradUniArgs ua) {
    static_assert(!(ROW8 && REFLECT), "the lean walker has no reflect variant");
    using namespace x3c;
    const WgradArgs& a = ua.g;
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    float* smem_all = smem_dyn;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = a.ktiles * a.ctiles * a.S;
    int b = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (b >= nwg) return;
    const int kt = b % a.ktiles; b /= a.ktiles;
    const int ct = b % a.ctiles; b /= a.ctiles;
    const int s = b;
    const int co0 = ct * 64, k0 = kt * WG_K;
    const long mbeg = (long)s * a.mper;
    const long mend = (mbeg + a.mper < a.M) ? mbeg + a.mper : a.M;
    const int nrows = (int)(mend - mbeg);
    const int nchunks = nrows > 0 ? (nrows + WG_MC - 1) / WG_MC : 0;
    const int hw = a.Ho * a.Wo;

    const int img0 = (int)(mbeg / hw);
    const long shift = (long)a.pad * (a.sH + a.sW);
    const long rest = (((long)a.N - img0) * a.sN + shift) * 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)img0 * a.sN - shift, rest > 0x7fffffffL ? 0x7fffffffu : (unsigned)rest);
    const float* dy0 = a.dy + mbeg * a.ldd;

    // ---- per-lane invariants of the staging (see conv_wgrad_uni_kernel)
    const int xh = lane >> 5;
    const int kx0 = k0 + 4 * ((lane & 31) ^ (8 * xh));
    const int kx = REFLECT ? min(kx0, a.K - 4) : kx0;   // reflect: lanes past K re-read the last columns (never stored)
    const int tap = kx / a.C;
    const int xc = kx - tap * a.C, xkh = tap / a.KW, xkw = tap - xkh * a.KW;
    const unsigned vx = (unsigned)(xkh * (int)a.sH + (xkw + xh * a.stride) * (int)a.sW + xc) * 4u;
    const int ncw = 2 * ua.nbw + 1;
    const unsigned c_top = REFLECT && xkh == 0 ? (unsigned)(2 * (int)a.sH * 4) : 0u;
    const unsigned c_bot = REFLECT && xkh == 2 ? (unsigned)(-2 * (int)a.sH * 4) : 0u;
    const unsigned c_left = REFLECT && xkw == 0 && xh == 0 ? (unsigned)(2 * (int)a.sW * 4) : 0u;
    const unsigned c_right = REFLECT && xkw == 2 && xh == 1 ? (unsigned)(-2 * (int)a.sW * 4) : 0u;
    unsigned xmask = 0x80000000u;
    if (REFLECT) {
        xmask = 0;
    } else if (kx < a.K) {
        for (int ch = 0; ch <= 2 * ua.nb; ++ch) {
            const int oh = ch <= ua.nb ? ch : a.Ho - ua.nb + (ch - ua.nb - 1);
            const bool bad_h = (unsigned)(oh * a.stride - a.pad + xkh) >= (unsigned)a.H;
            for (int cw = 0; cw < ncw; ++cw) {
                const int ow = (cw <= ua.nbw ? 2 * cw : a.Wo - 2 * ua.nbw + 2 * (cw - ua.nbw - 1)) + xh;
                const bool bad_w = (unsigned)(ow * a.stride - a.pad + xkw) >= (unsigned)a.W;
                xmask |= (unsigned)(bad_h | bad_w) << (ch * ncw + cw);
            }
        }
    } else {
        xmask = 0xffffffffu;
    }
    const int dpi = lane >> 4;                                               // dY piece: lane -> (pixel of the quad, 16-byte slot)
    const int dco = co0 + 4 * ((lane & 15) ^ (8 * (dpi & 1)));
    const unsigned vd = dco < a.Co ? (unsigned)(dpi * (int)a.ldd + dco) * 4u : OOB;

    const int st_h4 = a.stride * (int)a.sH * 4, st_w4 = a.stride * (int)a.sW * 4, sn4 = (int)a.sN * 4;
    const int d_row4 = st_h4 - a.Wo * st_w4, d_img4 = sn4 - a.Ho * st_h4;
    int s_p = 8 * wave;
    int s_oh, s_ow, s_soff;
    {
        const int m = (int)(mbeg - (long)img0 * hw) + s_p;
        const int n = m / hw;
        const int rem = m - n * hw;
        s_oh = rem / a.Wo; s_ow = rem - s_oh * a.Wo;
        s_soff = n * sn4 + s_oh * st_h4 + s_ow * st_w4;
        s_oh = __builtin_amdgcn_readfirstlane(s_oh); s_ow = __builtin_amdgcn_readfirstlane(s_ow);
        s_soff = __builtin_amdgcn_readfirstlane(s_soff);
    }
    int s_qd = 0;                                                            // chunk the next dY pieces fetch
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t*)smem_all;
    const unsigned m0_d = lds0 + (unsigned)wave * 8 * D_ROW, m0_x = lds0 + 2 * D_BYTES + (unsigned)wave * 8 * X_ROW;

    // dY pieces of chunk s_qd (four pixels each; rows >= nrows read as zero through the per-instruction descriptor)
    auto load_d_piece = [&](auto dst_tag, auto p_tag) {
        constexpr unsigned DST = decltype(dst_tag)::value;
        constexpr int P = decltype(p_tag)::value;
        const int r0 = WG_MC * s_qd + 8 * wave + 4 * P, left = nrows - r0;
        const __amdgpu_buffer_rsrc_t rd = make_rsrc(dy0 + (long)r0 * a.ldd, left > 0 ? (unsigned)(left * (int)a.ldd) * 4u : 0u);
        dma16s(rd, m0_d + DST * D_BYTES + P * 4 * D_ROW, vd, 0u);
        if constexpr (P == 1) ++s_qd;
    };
    auto load_d = [&](auto dst_tag) {
        load_d_piece(dst_tag, std::integral_constant<int, 0>{});
        load_d_piece(dst_tag, std::integral_constant<int, 1>{});
    };
    // X piece J (one pixel pair) of the chunk the scalar pixel state points at
    auto load_x = [&](auto dst_tag, auto piece_tag) {
        constexpr unsigned DST = decltype(dst_tag)::value;
        constexpr int J = decltype(piece_tag)::value;
        const int ch = min(s_oh, ua.nb) + max(s_oh - (a.Ho - ua.nb) + 1, 0);
        const int pw = s_ow >> 1;
        const int cw = min(pw, ua.nbw) + max(pw - ((a.Wo >> 1) - ua.nbw) + 1, 0);
        if constexpr (REFLECT) {
            // pairs past the slice read an interior pair instead (their dY rows are zero): nothing leaves the tensor
            const bool live = s_p < nrows;
            const unsigned f_t = live && s_oh == 0 ? ~0u : 0u, f_b = live && s_oh == a.Ho - 1 ? ~0u : 0u;
            const unsigned f_l = live && s_ow == 0 ? ~0u : 0u, f_r = live && s_ow == a.Wo - 2 ? ~0u : 0u;
            const unsigned voff = vx + (f_t & c_top) + (f_b & c_bot) + ((f_l & c_left) + (f_r & c_right));
            dma16s(rx, m0_x + DST * X_BYTES + J * 2 * X_ROW, voff,
                   live ? (unsigned)s_soff : (unsigned)((int)a.sH + 2 * (int)a.sW) * 4u);
        } else {
        const unsigned cls = (unsigned)(ch * ncw + cw) | (s_p < nrows ? 0u : 31u);
        const unsigned bad = __builtin_amdgcn_ubfe(xmask, cls, 1u);
        dma16s(rx, m0_x + DST * X_BYTES + J * 2 * X_ROW, (bad << 31) + vx, (unsigned)s_soff);
        }
        constexpr int STEP = J < 3 ? 2 : WG_MC - 6;
        s_p += STEP; s_ow += STEP; s_soff += STEP * st_w4;
#ifdef PD_PROBE_NOWALK      // tools/build_probe.sh only: the cost of the row / image wrap bookkeeping (results meaningless)
        s_soff = s_soff >= (a.N - img0 - 1) * sn4 ? 0 : s_soff;      // (stay inside the tensor: the scalar offset is not range-checked)
        return;
#endif
#pragma unroll
        for (int w = 0; w < (J < 3 ? 1 : 2); ++w) {
            const int ow0 = s_ow;
            s_ow = ow0 >= a.Wo ? ow0 - a.Wo : ow0;
            s_soff += ow0 >= a.Wo ? d_row4 : 0;
            const int oh1 = ow0 >= a.Wo ? s_oh + 1 : s_oh;
            s_soff += oh1 == a.Ho ? d_img4 : 0;
            s_oh = oh1 == a.Ho ? 0 : oh1;
        }
    };

    // the four pairs of a chunk (ROW8: they share an output row: one row class, the pair's column class per piece, one
    // 32-pixel advance behind the last piece)
    auto load_x_piece = [&](auto dst_tag, auto j_tag) {
        constexpr int J = decltype(j_tag)::value;
        if constexpr (ROW8) {
            constexpr unsigned DST = decltype(dst_tag)::value;
            const int ch = min(s_oh, ua.nb) + max(s_oh - (a.Ho - ua.nb) + 1, 0);
            const unsigned dead = s_p < nrows ? 0u : 31u;                    // (slices end on 8-pixel boundaries)
            const int pw = (s_ow >> 1) + J, pwe = (a.Wo >> 1) - ua.nbw;
            const int cw = min(pw, ua.nbw) + max(pw - pwe + 1, 0);
            const unsigned bad = __builtin_amdgcn_ubfe(xmask, (unsigned)(ch * ncw + cw) | dead, 1u);
            dma16s(rx, m0_x + DST * X_BYTES + J * 2 * X_ROW, (bad << 31) + vx, (unsigned)(s_soff + J * 2 * st_w4));
            if constexpr (J == 3) {
                s_p += WG_MC; s_ow += WG_MC; s_soff += WG_MC * st_w4;
#pragma unroll
                for (int w = 0; w < 3; ++w) {          // Wo >= 14 (host check): 32 columns wrap at most three times
                    const int ow0 = s_ow;
                    s_ow = ow0 >= a.Wo ? ow0 - a.Wo : ow0;
                    s_soff += ow0 >= a.Wo ? d_row4 : 0;
                    const int oh1 = ow0 >= a.Wo ? s_oh + 1 : s_oh;
                    s_soff += oh1 == a.Ho ? d_img4 : 0;
                    s_oh = oh1 == a.Ho ? 0 : oh1;
                }
            }
        } else {
            load_x(dst_tag, j_tag);
        }
    };
    auto load_x_all = [&](auto dst_tag) {
        load_x_piece(dst_tag, std::integral_constant<int, 0>{}); load_x_piece(dst_tag, std::integral_constant<int, 1>{});
        load_x_piece(dst_tag, std::integral_constant<int, 2>{}); load_x_piece(dst_tag, std::integral_constant<int, 3>{});
    };

    // ---- fragment / split addresses: lane -> (index inside a 32-wide block = lane % 32, pixel half = lane / 32); the eight
    // pixels of (k-step g, half h) are 16 g + 2 e + h, e = 0..7 (even | odd pixels: the staging tiles keep odd pixels
    // XOR 32 floats, so the two half-waves of a ds_read_b32 sit on different banks).
    // A k tile with at most 64 valid columns (the last of K = 576: 4.5 tiles) has nothing for waves 2 and 3 to multiply:
    // they skip the MFMAs and the X split and only take part in staging and in the dY split.  (Giving them the second
    // k-step of waves 0 and 1, as conv_wgrad_uni_kernel does, was measured: 141 -> 128 TF on 5x5x64 @256x320 -- a second
    // copy of the chunk body costs more than it saves.)
    const bool active = k0 + 32 * wave < a.K;
    const int fi = lane & 31, fk = lane >> 5;
    unsigned fx_off = 2 * D_BYTES + fk * X_ROW + 4u * ((unsigned)(wave * 32 + fi) ^ (32u * fk));
    unsigned pa_off = P_BASE + (unsigned)fk * 1024u + (unsigned)fi * 16u;     // + ((term*2 + g)*2)*1024 + blk*512
    // dY split: thread -> (channel = lane, octet = wave: k-step wave / 2, half wave % 2)
    const int sg = wave >> 1, sh = wave & 1;
    unsigned sd_off = (unsigned)(16 * sg + sh) * D_ROW + 4u * ((unsigned)lane ^ (32u * sh));     // + e * 2 * D_ROW
    unsigned sp_off = P_BASE + (unsigned)(sg * 2 + sh) * 1024u + (unsigned)lane * 16u;          // + term * 4096
    asm volatile("" : "+v"(fx_off), "+v"(pa_off), "+v"(sd_off), "+v"(sp_off));
    char* lds_c = reinterpret_cast<char*>(smem_all);

    typedef float accv_t __attribute__((ext_vector_type(16)));
    accv_t acc[2], acc_sum[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acc_sum[t][r] = 0.f; }
    // The bf16 MFMA aligns its sixteen products and the accumulator by TRUNCATION: a bias of about -2^-32 of the largest
    // addend per instruction, negligible for a forward convolution (600 instructions per output) but proportional to
    // instructions x |accumulator| here, where a slice contracts thousands of pixels (12 instructions per chunk and
    // accumulator; 200 chunks per slice at batch 16).  Every FLUSH chunks the MFMA accumulators are added to a second
    // fp32 set (round to nearest: unbiased) and restart from zero, so the truncated quantity stays ~sqrt(FLUSH / chunks) of
    // its final size: 32 adds + 32 moves per 8 chunks next to their 864 split instructions.
    constexpr int FLUSH = 8;
    auto flush = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc_sum[t][r] += acc[t][r]; acc[t][r] = 0.f; }
    };
    float bsum = 0.f;
    const bool do_bias = BIAS && a.bpart != nullptr && kt == 0;
    auto bf = [](u32x4 v) { return __builtin_bit_cast(x3::bf16x8, v); };

    // dY staging slot SRC -> planes SRC (and the bias partial of this thread's channel and pixel octet): prologue only
    auto split_d = [&](auto src_tag) {
        constexpr unsigned SRC = decltype(src_tag)::value;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(lds_c + sd_off + (SRC * D_BYTES + (unsigned)e * 2 * D_ROW));
        if constexpr (BIAS) bsum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        const x3::Split sd = x3::split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
        *reinterpret_cast<u32x4*>(lds_c + sp_off + (SRC * P_BYTES + 0 * 4096u)) = sd.hi;
        *reinterpret_cast<u32x4*>(lds_c + sp_off + (SRC * P_BYTES + 1 * 4096u)) = sd.mid;
        *reinterpret_cast<u32x4*>(lds_c + sp_off + (SRC * P_BYTES + 2 * 4096u)) = sd.lo;
    };

    // One chunk: 24 MFMAs with the vector work threaded between them in program order (see conv_igemm_x3_kernel).  X is the
    // B operand: products in the order its split yields the terms -- (hi, mid, lo of dY) x hi, (hi, mid) x mid, hi x lo.
    auto chunk = [&](auto buf_tag) {
        constexpr unsigned BUF = decltype(buf_tag)::value, NXT = BUF ^ 1;
        const std::integral_constant<unsigned, NXT> nxt{};
        // The six LDS-DMA pieces of a chunk: at its head, or (SPREAD) one behind each of the first six MFMAs.  After the
        // barrier all eight waves of a CU stand at their chunk heads together: six back-to-back issues per wave are a phase
        // in which nobody feeds the matrix pipe; threaded in, each issue rides in the shadow of the wave's own MFMA.
        // (round 4, same box: 3x3x64 @256x320 114 -> 123 TF, @128x160 108 -> 126, 5x5x64 @256x320 144 -> 155, 5x5 256 -> 512
        //  @32x40 139 -> 148 together with the lean walker on every plane: profiles/r04_wgrad_variants.log)
        constexpr bool SPREAD = true;
        if (!SPREAD || !active) {
            load_x_all(nxt);
            load_d(buf_tag);                               // chunk q + 2 into the slot whose chunk q was split during chunk q - 1
        }
        if (!active) {                                     // (wave-uniform) no valid k column: only its share of the dY split
            split_d(nxt);
            dma_wait();
            __syncthreads();
            return;
        }
        float x0[8], x1[8], dv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x0[e] = *reinterpret_cast<const float*>(lds_c + fx_off + (BUF * X_BYTES + (unsigned)e * 2 * X_ROW));
        u32x4 fa[2][2][3];                                 // [k-step][co block][term]
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    fa[g][blk][t] = *reinterpret_cast<const u32x4*>(lds_c + pa_off + (BUF * P_BYTES + (unsigned)((t * 2 + g) * 2) * 1024u + blk * 512u));
#pragma unroll
        for (int e = 0; e < 8; ++e) x1[e] = *reinterpret_cast<const float*>(lds_c + fx_off + (BUF * X_BYTES + (unsigned)(8 + e) * 2 * X_ROW));
#pragma unroll
        for (int e = 0; e < 8; ++e) dv[e] = *reinterpret_cast<const float*>(lds_c + sd_off + (NXT * D_BYTES + (unsigned)e * 2 * D_ROW));
        x3::Terms t0, t1, td;
        // MFMA number N (0..11) of k-step G: term pair N / 2, co block N % 2
        auto mm = [&](const x3::Terms& t, auto g_tag, auto n_tag) {
            constexpr int G = decltype(g_tag)::value, N = decltype(n_tag)::value, T = N / 2, BLK = N % 2;
            const unsigned* bv = T < 3 ? t.h : T < 5 ? t.m : t.l;
            const u32x4 av = T == 0 ? fa[G][BLK][0] : T == 1 ? fa[G][BLK][1] : T == 2 ? fa[G][BLK][2] : T == 3 ? fa[G][BLK][0] : T == 4 ? fa[G][BLK][1] : fa[G][BLK][0];
            acc[BLK] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(av), bf(u32x4{bv[0], bv[1], bv[2], bv[3]}), acc[BLK], 0, 0, 0);
        };
#define PD_I(n) std::integral_constant<int, n>{}
#define PD_SB __builtin_amdgcn_sched_barrier(0);
        x3::sp_h<0>(x0, t0); x3::sp_h<1>(x0, t0); x3::sp_h<2>(x0, t0); x3::sp_h<3>(x0, t0);
        PD_SB
#define PD_XP(j) if constexpr (SPREAD) load_x_piece(nxt, PD_I(j));
#define PD_DP(j) if constexpr (SPREAD) load_d_piece(buf_tag, PD_I(j));
        mm(t0, PD_I(0), PD_I(0)); x3::sp_m<0>(x0, t0); PD_XP(0) PD_SB
        mm(t0, PD_I(0), PD_I(1)); x3::sp_m<1>(x0, t0); PD_XP(1) PD_SB
        mm(t0, PD_I(0), PD_I(2)); x3::sp_m<2>(x0, t0); PD_XP(2) PD_SB
        mm(t0, PD_I(0), PD_I(3)); x3::sp_m<3>(x0, t0); PD_XP(3) PD_SB
        mm(t0, PD_I(0), PD_I(4)); x3::sp_h<0>(dv, td); x3::sp_h<1>(dv, td); x3::sp_h<2>(dv, td); x3::sp_h<3>(dv, td); PD_DP(0) PD_SB
        mm(t0, PD_I(0), PD_I(5)); x3::sp_m<0>(dv, td); PD_DP(1) PD_SB
#undef PD_XP
#undef PD_DP
        mm(t0, PD_I(0), PD_I(6)); x3::sp_l<0>(t0); x3::sp_h<0>(x1, t1); PD_SB
        mm(t0, PD_I(0), PD_I(7)); x3::sp_l<1>(t0); x3::sp_h<1>(x1, t1); PD_SB
        mm(t0, PD_I(0), PD_I(8)); x3::sp_l<2>(t0); x3::sp_h<2>(x1, t1); PD_SB
        mm(t0, PD_I(0), PD_I(9)); x3::sp_l<3>(t0); x3::sp_h<3>(x1, t1); PD_SB
        mm(t0, PD_I(0), PD_I(10)); x3::sp_m<1>(dv, td); PD_SB
        mm(t0, PD_I(0), PD_I(11)); x3::sp_m<2>(dv, td); PD_SB
        mm(t1, PD_I(1), PD_I(0)); x3::sp_m<0>(x1, t1); PD_SB
        mm(t1, PD_I(1), PD_I(1)); x3::sp_m<1>(x1, t1); PD_SB
        mm(t1, PD_I(1), PD_I(2)); x3::sp_m<2>(x1, t1); PD_SB
        mm(t1, PD_I(1), PD_I(3)); x3::sp_m<3>(x1, t1); PD_SB
        mm(t1, PD_I(1), PD_I(4)); x3::sp_m<3>(dv, td); PD_SB
        mm(t1, PD_I(1), PD_I(5)); x3::sp_l<0>(td); x3::sp_l<1>(td); PD_SB
        mm(t1, PD_I(1), PD_I(6)); x3::sp_l<0>(t1); PD_SB
        mm(t1, PD_I(1), PD_I(7)); x3::sp_l<1>(t1); PD_SB
        mm(t1, PD_I(1), PD_I(8)); x3::sp_l<2>(t1); PD_SB
        mm(t1, PD_I(1), PD_I(9)); x3::sp_l<3>(t1); PD_SB
        mm(t1, PD_I(1), PD_I(10)); x3::sp_l<2>(td); x3::sp_l<3>(td); PD_SB
        mm(t1, PD_I(1), PD_I(11));
        if constexpr (BIAS) bsum += ((dv[0] + dv[1]) + (dv[2] + dv[3])) + ((dv[4] + dv[5]) + (dv[6] + dv[7]));
        *reinterpret_cast<u32x4*>(lds_c + sp_off + (NXT * P_BYTES + 0 * 4096u)) = u32x4{td.h[0], td.h[1], td.h[2], td.h[3]};
        *reinterpret_cast<u32x4*>(lds_c + sp_off + (NXT * P_BYTES + 1 * 4096u)) = u32x4{td.m[0], td.m[1], td.m[2], td.m[3]};
        *reinterpret_cast<u32x4*>(lds_c + sp_off + (NXT * P_BYTES + 2 * 4096u)) = u32x4{td.l[0], td.l[1], td.l[2], td.l[3]};
#undef PD_SB
#undef PD_I
        dma_wait();
        __syncthreads();
    };

    // ---- prologue: X of chunk 0, dY of chunks 0 and 1; planes of chunk 0
    {
        const std::integral_constant<unsigned, 0> d0{};
        const std::integral_constant<unsigned, 1> d1{};
        load_x_all(d0);
        load_d(d0);
        load_d(d1);
        dma_wait();
        __syncthreads();
        split_d(d0);
        __syncthreads();
    }
    for (int q = 0; q < nchunks; q += 2) {
        chunk(std::integral_constant<unsigned, 0>{});
        if (q + 1 < nchunks) chunk(std::integral_constant<unsigned, 1>{});
        if ((q & (FLUSH - 2)) == FLUSH - 2) flush();       // (wave-uniform; q is even)
    }
    flush();

    // C/D layout: col = lane % 32 -> k; row -> co: (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int k = k0 + wave * 32 + fi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * fk;
            if (co < a.Co && k < a.K) a.part[((long)s * a.Co + co) * a.K + k] = acc_sum[blk][r];
        }
    }
    if constexpr (BIAS) {
        if (do_bias) {        // (uniform per workgroup; the rings are drained)  The splits of the prologue and of every chunk saw
            // each pixel of the slice once -- plus one chunk past the end, which reads as zero
            float* red = smem_all;
            red[wave * 64 + lane] = bsum;
            __syncthreads();
            if (tid < 64 && co0 + tid < a.Co) a.bpart[(long)s * a.Co + co0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
        }
    }
}

#include "conv_wgrad_halo.hpp"
#include "conv_wgrad_roll.hpp"

// out[i] (+)= sum_s part[s][i]: 64 columns x 4 slice lanes per workgroup; every lane keeps four independent
// loads in flight; the lane partials are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int S, long n, int accumulate) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long i = blockIdx.x * 64L + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int j = sl;
        for (; j + 12 < S; j += 16) {
            a0 += part[(long)j * n + i];
            a1 += part[(long)(j + 4) * n + i];
            a2 += part[(long)(j + 8) * n + i];
            a3 += part[(long)(j + 12) * n + i];
        }
        for (; j < S; j += 4) a0 += part[(long)j * n + i];
    }
    red[sl][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && i < n) {
        const float s = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        out[i] = accumulate ? out[i] + s : s;
    }
}

// w [Co][T][Ci] -> wt [Ci][T][Co]  (T = KH*KW): operand layout of the data-gradient GEMM
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                               int Co, int T, int Ci) {
    const long n = (long)Co * T * Ci;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int co = (int)(i % Co);
        const long r = i / Co;
        const int t = (int)(r % T);
        const int ci = (int)(r / T);
        wt[i] = w[((long)co * T + t) * Ci + ci];
    }
}

// All convolution weights of a parameter store in one launch: entry e = {element offset in both flat buffers, Co, T, Ci},
// blk[e] = first workgroup of entry e, one workgroup per 32(co) x 32(ci) tile of one tap (LDS transpose: 128-byte
// rows on both sides), blk[n] = grid size.
__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                       const int* __restrict__ table,
                                                                       const int* __restrict__ blk, int n) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n;                         // last e with blk[e] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const int off = table[4 * lo], Co = table[4 * lo + 1], T = table[4 * lo + 2], Ci = table[4 * lo + 3];
    const float* w = src + off;
    float* wt = dst + off;
    int lb = (int)blockIdx.x - blk[lo];
    const int tiles_ci = (Ci + 31) >> 5, tiles_co = (Co + 31) >> 5;
    const int ci0 = (lb % tiles_ci) << 5; lb /= tiles_ci;
    const int co0 = (lb % tiles_co) << 5;
    const int t = lb / tiles_co;
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
#pragma unroll
    for (int r = y; r < 32; r += 8)
        if (co0 + r < Co && ci0 + x < Ci) tile[r][x] = w[((long)(co0 + r) * T + t) * Ci + ci0 + x];
    __syncthreads();
#pragma unroll
    for (int r = y; r < 32; r += 8)
        if (ci0 + r < Ci && co0 + x < Co) wt[((long)(ci0 + r) * T + t) * Co + co0 + x] = tile[x][r];
}

inline int wgrad_tco(int Co) { return Co > 32 ? 64 : (Co > 16 ? 32 : 16); }

// bf16-split weight gradient (conv_wgrad_x3c_kernel) unless the caller asks for the fp32 MFMA or for the in-register split
inline bool wgrad_x3c_enabled(unsigned flags) { return !(flags & (PD_CONV_FP32_MFMA | PD_CONV_WGRAD_SPLIT_IN_REGS)); }

void wgrad_plan(long M, int Co, int K, unsigned flags, int* S, long* mper) {
    const int tco = wgrad_tco(Co);
    const long tiles = (long)((Co + tco - 1) / tco) * ((K + WG_K - 1) / WG_K);
    // <= two full rounds of the 768 resident workgroups (floor: no ragged tail) when a slice count of one round would
    // be small (many tiles: 512-channel layers, S = 2 leaves a quarter of the slots empty); ONE round -- longer slices,
    // half the partial tiles to write and reduce -- when the tiles are few (measured +5 % on the 128x160 / 64x80 layers,
    // -7..-14 % on the many-tile layers, equal on the 256x320 ones)
    // The bf16-split kernel (64-wide co tile; 72 KB of LDS) has 512 resident workgroups, not 768: one round of those when
    // the tiles are few (3x3x128 @64x80: 144 -> 162 TF, 3x3x256 @32x40: 150 -> 163, 3x3x64 @128x160: 132 -> 137 against the
    // 768-workgroup plan), two for the million-pixel layers (5x5x64 @256x320: 153 -> 159), three when one round would leave
    // fewer than four slices per tile (3x3x512 @16x20, 288 tiles: 134 TF with 1536 workgroups, 92 with 512).
    const bool x3c = tco == 64 && wgrad_x3c_enabled(flags) && !(flags & PD_CONV_GENERAL_KERNELS);
    const long total = x3c ? (512 / tiles < 4 ? 1536 : M >= (1L << 20) ? 1024 : 512) : (tiles <= 32 ? 768 : 1536);
    long s = total / tiles;
    const long smax = (M + 511) / 512;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    long per = (M + s - 1) / s;
    per = (per + WG_MC - 1) / WG_MC * WG_MC;
    s = (M + per - 1) / per;
    if (s < 1) s = 1;
    *S = (int)s; *mper = per;
}

}  // namespace

extern "C" size_t pd_conv2d_wgrad_workspace(long M, int Co, int K, unsigned flags) {
    int S; long mper;
    wgrad_plan(M, Co, K, flags, &S, &mper);
    // (the halo-tile kernel cuts its own slices; Cout and K bound their number)
    if (wgrad_x3c_enabled(flags) && !(flags & (PD_CONV_GENERAL_KERNELS | PD_CONV_X3_IM2COL))) {
        const int sh = std::max(wgrad_halo_slices_bound(Co, K), wgrad_roll_slices_bound(Co, K));
        const long cap = (M + 63) / 64;
        S = S > (sh < cap ? sh : (int)cap) ? S : (sh < cap ? sh : (int)cap);
    }
    return ((size_t)S * Co * K + (size_t)S * Co) * sizeof(float);
}

// conv_wgrad_uni_kernel's X3 branch (both operands split in registers by every wave that needs them): 216 vector
// instructions per 24 MFMAs, measured 8 % slower than the fp32 kernel -- kept for its test, reached by flag only
static bool wgrad_x3_on(unsigned flags) { return (flags & PD_CONV_WGRAD_SPLIT_IN_REGS) && !(flags & PD_CONV_FP32_MFMA); }

// 1 when pd_conv2d_wgrad sends this zero-padded shape (16-byte aligned NHWC operands assumed) to conv_wgrad_x3c_kernel
extern "C" int pd_conv2d_wgrad_uses_x3(long M, int Co, int C, int KH, int KW, int stride, int pad, int mode, int H, int W, int Ho,
                                       int Wo, unsigned flags) {
    const bool refl_ok = mode == MODE_REFLECT && pad == 1 && KH == 3 && KW == 3 && stride == 1 && Ho == H && Wo == W && H >= 3;
    if (!(mode == MODE_ZERO || refl_ok)) return 0;
    const bool uni_on = !(flags & PD_CONV_GENERAL_KERNELS);
    int S = 0; long mper = 0;
    wgrad_plan(M, Co, KH * KW * C, flags, &S, &mper);
    const int nb = (pad + stride - 1) / stride, nbw = (nb + 1) / 2;
    const bool x3c = uni_on && wgrad_x3c_enabled(flags) && wgrad_tco(Co) == 64 && C % 4 == 0 && Co % 4 == 0 && Wo % 2 == 0 && Ho >= 2 * nb &&
           Wo >= 4 * nbw && Wo >= 14 && (2 * nb + 1) * (2 * nbw + 1) <= 31 && mper % WG_MC == 0 && M % 4 == 0 && KH * KW * C >= 4;
    if (wgrad_x3c_enabled(flags) && !(flags & (PD_CONV_GENERAL_KERNELS | PD_CONV_X3_IM2COL)) && C % 4 == 0) {
        // 2: the halo-tile kernel (both operands split once per tile, transposed LDS reads) -- the same test as pd_conv2d_wgrad's
        WgradArgs a{};
        a.mode = mode; a.stride = stride; a.KH = KH; a.KW = KW; a.pad = pad; a.C = C; a.Co = Co; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
        a.ldd = Co; a.sN = (long)H * W * C;
        if (wgrad_halo_eligible(a, true)) {
            // 3: the rolling-row kernel (all three filter rows per workgroup) -- with the workspace pd_conv2d_wgrad_workspace asks for
            a.N = (int)(M / ((long)Ho * Wo)); a.M = M; a.K = KH * KW * C;
            const size_t per_slice = ((size_t)Co * a.K + Co) * sizeof(float);
            const int s_cap = (int)std::min<size_t>(pd_conv2d_wgrad_workspace(M, Co, a.K, flags) / per_slice, 1 << 20);
            WgradRollPlan rp;
            return !(flags & PD_CONV_WGRAD_ROW_WORKGROUPS) && wgrad_roll_eligible(a, true, s_cap, rp) ? 3 : 2;
        }
    }
    return x3c;
}

extern "C" int pd_conv2d_wgrad(const void* x, const void* dy, void* dw, void* dbias, void* workspace, size_t ws_bytes,
                               int N, int H, int W, int C, long sN, long sH, long sW, long sC,
                               int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int mode,
                               int affine, float sub, float div, long ldd, int accumulate, unsigned flags, void* stream) {
    PD_REQUIRE(x && dy && dw && workspace, "pd_conv2d_wgrad: null tensor");
    PD_REQUIRE(pd::conv_flags_ok(flags), "pd_conv2d_wgrad: bad flags 0x%x", flags);
    PD_REQUIRE(mode == MODE_ZERO || mode == MODE_REFLECT, "pd_conv2d_wgrad: mode must be 0 or 1");
    PD_REQUIRE(Ho <= (H + 2 * pad - KH) / stride + 1 && Wo <= (W + 2 * pad - KW) / stride + 1,
               "pd_conv2d_wgrad: output grid does not match input/filter geometry");
    PD_REQUIRE(ldd >= Co && (ldd % 4 == 0 || Co < 4) , "pd_conv2d_wgrad: bad dy row stride");
    if (N == 0) return PD_OK;
    WgradArgs a;
    a.x = (const float*)x; a.dy = (const float*)dy;
    a.N = N; a.H = H; a.W = W; a.C = C; a.sN = sN; a.sH = sH; a.sW = sW; a.sC = sC;
    a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.mode = mode;
    a.affine = affine; a.sub = sub; a.div = div;
    a.K = KH * KW * C; a.M = (long)N * Ho * Wo; a.ldd = ldd;
    wgrad_plan(a.M, Co, a.K, flags, &a.S, &a.mper);
    const size_t need = pd_conv2d_wgrad_workspace(a.M, Co, a.K, flags);
    PD_REQUIRE(ws_bytes >= need, "pd_conv2d_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
    a.part = (float*)workspace;
    a.bpart = dbias ? a.part + (size_t)a.S * Co * a.K : nullptr;
    const int tco = wgrad_tco(Co);
    a.ktiles = (a.K + WG_K - 1) / WG_K; a.ctiles = (Co + tco - 1) / tco;
    const bool vec = (C % 4 == 0) && sC == 1 && (sW % 4 == 0) && (sH % 4 == 0) && (sN % 4 == 0) && !affine &&
                     pd::aligned16(x);
    PD_REQUIRE(pd::aligned16(dy) && (ldd % 4 == 0 || Co < 4), "pd_conv2d_wgrad: dy must be 16-byte aligned rows");
    PD_REQUIRE(a.mper * ldd * 4 < 0x7fffffffL, "pd_conv2d_wgrad: slice too large for 32-bit offsets");
    PD_REQUIRE((a.mper / ((long)Ho * Wo) + 2) * sN * 4 < 0x7fffffffL, "pd_conv2d_wgrad: image too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    if (wgrad_x3c_enabled(flags) && !(flags & (PD_CONV_GENERAL_KERNELS | PD_CONV_X3_IM2COL)) && wgrad_halo_eligible(a, vec)) {
        // halo-tile kernel: its slice count never exceeds what the workspace holds
        const size_t per_slice = ((size_t)Co * a.K + Co) * sizeof(float);
        const int s_cap = (int)std::min<size_t>(ws_bytes / per_slice, 1 << 20);
        WgradRollPlan rp;
        // 3x3 with long tile columns: all three filter rows per workgroup, input rows rolling through LDS
        const int S = !(flags & PD_CONV_WGRAD_ROW_WORKGROUPS) && wgrad_roll_eligible(a, vec, s_cap, rp) ? launch_wgrad_roll(a, rp, st, dbias != nullptr)
                                                                                             : launch_wgrad_halo(a, s_cap, st, dbias != nullptr);
        a.bpart = dbias ? a.part + (size_t)S * Co * a.K : nullptr;
        int rc = pd::check_launch("pd_conv2d_wgrad");
        if (rc) return rc;
        const long nw = (long)Co * a.K;
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((nw + 63) / 64)), dim3(256), 0, st, a.part, (float*)dw, S, nw, accumulate);
        if (dbias)
            hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((Co + 63) / 64)), dim3(256), 0, st, a.bpart, (float*)dbias, S, (long)Co, accumulate);
        return pd::check_launch("pd_conv2d_wgrad/reduce");
    }
    const dim3 grid((unsigned)(((long)a.ktiles * a.ctiles * a.S + 7) / 8 * 8)), block(NT);
#define PD_WG(T, V, MD) hipLaunchKernelGGL((conv_wgrad_kernel<T, V, MD>), grid, block, 0, st, a)
    // scalar-pixel variant: 16-byte path, zero padding, 64-wide co tile, even output rows (pixel pairs stay inside a
    // row), border classes that fit the 31-bit mask
    const bool uni_on = !(flags & PD_CONV_GENERAL_KERNELS);
    const int nb = (pad + stride - 1) / stride, nbw = (nb + 1) / 2;
    const bool refl_ok = mode == MODE_REFLECT && pad == 1 && KH == 3 && KW == 3 && stride == 1 && Ho == H && Wo == W && H >= 3;
    if (uni_on && (tco == 64 || tco == 32) && vec && (mode == MODE_ZERO || refl_ok) && Co % 4 == 0 && Wo % 2 == 0 && ldd % 4 == 0 &&
        Ho >= 2 * nb && Wo >= 4 * nbw && Wo >= 14 && (2 * nb + 1) * (2 * nbw + 1) <= 31 && a.mper % WG_MC == 0 &&
        a.M % 4 == 0 && a.K >= 4) {
        WgradUniArgs ua; ua.g = a; ua.nb = nb; ua.nbw = nbw;
        if (tco == 32) {              // 17..32 output channels (decoder 96->32, 64->32): reflect + bias in this network
            if (mode == MODE_ZERO) hipLaunchKernelGGL((conv_wgrad_uni_kernel<false, true, 32>), grid, block, 0, st, ua);
            else hipLaunchKernelGGL((conv_wgrad_uni_kernel<true, true, 32>), grid, block, 0, st, ua);
        } else if (wgrad_x3c_enabled(flags)) {   // bf16-split products, every element split once (zero or reflection padding)
            static const hipError_t lds_ok = [] {
                hipError_t e = hipSuccess;
                for (const void* f : {reinterpret_cast<const void*>(conv_wgrad_x3c_kernel<true, true>), reinterpret_cast<const void*>(conv_wgrad_x3c_kernel<false, true>),
                                      reinterpret_cast<const void*>(conv_wgrad_x3c_kernel<true, false>), reinterpret_cast<const void*>(conv_wgrad_x3c_kernel<false, false>),
                                      reinterpret_cast<const void*>(conv_wgrad_x3c_kernel<true, false, true>),
                                      reinterpret_cast<const void*>(conv_wgrad_x3c_kernel<false, false, true>)}) {
                    const hipError_t r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, x3c::LDS_BYTES);
                    if (r != hipSuccess) e = r;
                }
                return e;
            }();
            PD_REQUIRE(lds_ok == hipSuccess, "pd_conv2d_wgrad: cannot reserve %u bytes of LDS", x3c::LDS_BYTES);
            // The lean walker (one row class and one 32-pixel advance per chunk: 5 scalar instructions per MFMA instead of 8)
            // whenever the eight pixels a wave stages cannot straddle an output row.  (Round 3 kept it off the large planes,
            // where its four loads left back to back at the chunk head: 148 -> 134 TF on 5x5x64 @256x320; with the pieces
            // threaded behind the chunk's first MFMAs it wins there too: 144 -> 155.)
            const bool row8 = Wo % 8 == 0 && a.mper % 8 == 0 && a.M % 8 == 0;
            if (mode == MODE_REFLECT) {
                if (dbias) hipLaunchKernelGGL((conv_wgrad_x3c_kernel<true, false, true>), grid, block, x3c::LDS_BYTES, st, ua);
                else hipLaunchKernelGGL((conv_wgrad_x3c_kernel<false, false, true>), grid, block, x3c::LDS_BYTES, st, ua);
            } else if (row8) {
                if (dbias) hipLaunchKernelGGL((conv_wgrad_x3c_kernel<true, true>), grid, block, x3c::LDS_BYTES, st, ua);
                else hipLaunchKernelGGL((conv_wgrad_x3c_kernel<false, true>), grid, block, x3c::LDS_BYTES, st, ua);
            } else {
                if (dbias) hipLaunchKernelGGL((conv_wgrad_x3c_kernel<true, false>), grid, block, x3c::LDS_BYTES, st, ua);
                else hipLaunchKernelGGL((conv_wgrad_x3c_kernel<false, false>), grid, block, x3c::LDS_BYTES, st, ua);
            }
        } else if (wgrad_x3_on(flags)) {     // products on the bf16 matrix cores (three-way split, fp32 accuracy)
            if (mode == MODE_ZERO) {
                if (dbias) hipLaunchKernelGGL((conv_wgrad_uni_kernel<false, true, 64, true>), grid, block, 0, st, ua);
                else hipLaunchKernelGGL((conv_wgrad_uni_kernel<false, false, 64, true>), grid, block, 0, st, ua);
            } else {
                if (dbias) hipLaunchKernelGGL((conv_wgrad_uni_kernel<true, true, 64, true>), grid, block, 0, st, ua);
                else hipLaunchKernelGGL((conv_wgrad_uni_kernel<true, false, 64, true>), grid, block, 0, st, ua);
            }
        } else if (mode == MODE_ZERO) {
            if (dbias) hipLaunchKernelGGL((conv_wgrad_uni_kernel<false, true>), grid, block, 0, st, ua);
            else hipLaunchKernelGGL((conv_wgrad_uni_kernel<false, false>), grid, block, 0, st, ua);
        } else {
            if (dbias) hipLaunchKernelGGL((conv_wgrad_uni_kernel<true, true>), grid, block, 0, st, ua);
            else hipLaunchKernelGGL((conv_wgrad_uni_kernel<true, false>), grid, block, 0, st, ua);
        }
    } else
    if (tco == 64) {
        if (vec) { if (mode == MODE_ZERO) PD_WG(64, true, MODE_ZERO); else PD_WG(64, true, MODE_REFLECT); }
        else { if (mode == MODE_ZERO) PD_WG(64, false, MODE_ZERO); else PD_WG(64, false, MODE_REFLECT); }
    } else if (tco == 32) {
        if (vec) { if (mode == MODE_ZERO) PD_WG(32, true, MODE_ZERO); else PD_WG(32, true, MODE_REFLECT); }
        else { if (mode == MODE_ZERO) PD_WG(32, false, MODE_ZERO); else PD_WG(32, false, MODE_REFLECT); }
    } else {
        if (vec) { if (mode == MODE_ZERO) PD_WG(16, true, MODE_ZERO); else PD_WG(16, true, MODE_REFLECT); }
        else { if (mode == MODE_ZERO) PD_WG(16, false, MODE_ZERO); else PD_WG(16, false, MODE_REFLECT); }
    }
#undef PD_WG
    int rc = pd::check_launch("pd_conv2d_wgrad");
    if (rc) return rc;
    const long nw = (long)Co * a.K;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((nw + 63) / 64)), dim3(256), 0, st,
                       a.part, (float*)dw, a.S, nw, accumulate);
    if (dbias)
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((Co + 63) / 64)), dim3(256), 0, st,
                           a.bpart, (float*)dbias, a.S, (long)Co, accumulate);
    return pd::check_launch("pd_conv2d_wgrad/reduce");
}

extern "C" int pd_weight_transpose(const void* w, void* wt, int Co, int T, int Ci, void* stream) {
    PD_REQUIRE(w && wt && Co > 0 && T > 0 && Ci > 0, "pd_weight_transpose: bad arguments");
    const long n = (long)Co * T * Ci;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)w, (float*)wt, Co, T, Ci);
    return pd::check_launch("pd_weight_transpose");
}

extern "C" int pd_weight_transpose_batched(const void* src, void* dst, const void* table, const void* blk, int n,
                                           int nblocks, void* stream) {
    PD_REQUIRE(src && dst && table && blk && n > 0 && nblocks > 0, "pd_weight_transpose_batched: bad arguments");
    hipLaunchKernelGGL(weight_transpose_batched_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, (float*)dst, (const int*)table, (const int*)blk, n);
    return pd::check_launch("pd_weight_transpose_batched");
}

// ===================================================================== stride-2 data gradient by output parity
// dX of a 3x3 / stride-2 / pad-1 convolution: an input pixel (ih, iw) receives only the taps with (ih + 1 - kh) and
// (iw + 1 - kw) even -- 1, 2, 2 or 4 of the 9, by the parity of (ih, iw).  The masked transposed gather multiplies zeros
// for the other taps (three quarters of its MFMAs).  Per parity class (ph, pw) the gradient on the sub-grid
// (2i + ph, 2j + pw) is a stride-1 correlation of dY with a (1 + ph) x (1 + pw) sub-filter (odd parity: taps 0 and 2
// with padding 1, even parity: tap 1 alone): four uniform-tap launches (pd_conv2d_rect) of exactly the 9 tap-units
// instead of 36, then one interleave.
namespace {
// wt [Ci][3][3][Co] (data-gradient operand) -> wsub: the four class filters back to back, class (ph, pw) =
// [Ci][1 + ph][1 + pw][Co] at float offset Ci*Co*{0, 1, 3, 5}[2 ph + pw]  (9 Ci Co floats in all)
__global__ __launch_bounds__(256) void dgrad_s2_filters_kernel(const float* __restrict__ wt, float* __restrict__ wsub,
                                                               int Ci, int Co) {
    const long cc = (long)Ci * Co, total = 9 * cc;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cls = i < cc ? 0 : i < 3 * cc ? 1 : i < 5 * cc ? 2 : 3, ph = cls >> 1, pw = cls & 1;
        const long base = cc * (cls == 0 ? 0 : cls == 1 ? 1 : cls == 2 ? 3 : 5);
        long r = i - base;
        const int co = (int)(r % Co); r /= Co;
        const int kw2 = pw ? (int)(r & 1) : 0; r >>= pw;
        const int kh2 = ph ? (int)(r & 1) : 0; r >>= ph;
        const int ci = (int)r;
        // transposed-mode tap k' reads dY[i + pad - k']: odd parity (2 taps, pad 1): k' = 0 -> filter tap 0 (oh = i + 1),
        // k' = 1 -> tap 2 (oh = i); even parity (1 tap, pad 0): filter tap 1 (oh = i)
        const int kh = ph ? (kh2 ? 2 : 0) : 1, kw = pw ? (kw2 ? 2 : 0) : 1;
        wsub[i] = wt[(((long)ci * 3 + kh) * 3 + kw) * Co + co];
    }
}
// sub [4][N][Ho][Wo][C] -> dx [N][2 Ho][2 Wo][C]
__global__ __launch_bounds__(256) void interleave4_kernel(const float* __restrict__ sub, float* __restrict__ dx, int N, int Ho,
                                                          int Wo, int C) {
    const int cq = C >> 2;
    const long per = (long)N * Ho * Wo * cq, total = 4 * per;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % cq) * 4;
        long r = i / cq;
        const int pw = (int)(r & 1); r >>= 1;                      // output-major order: consecutive threads write consecutive pixels
        const int j = (int)(r % Wo); r /= Wo;
        const int ph = (int)(r & 1); r >>= 1;
        const int ii = (int)(r % Ho);
        const long n = r / Ho;
        const float4 v = ldg4(sub + ((long)(ph * 2 + pw) * N * Ho * Wo + (n * Ho + ii) * (long)Wo + j) * C + c4);
        *reinterpret_cast<float4*>(dx + ((n * 2 * Ho + 2 * ii + ph) * (long)(2 * Wo) + 2 * j + pw) * C + c4) = v;
    }
}
}  // namespace

extern "C" int pd_dgrad_s2_filters(const void* wt, void* wsub, int Ci, int Co, void* stream) {
    PD_REQUIRE(wt && wsub && Ci > 0 && Co > 0, "pd_dgrad_s2_filters: bad arguments");
    const long total = 9L * Ci * Co, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dgrad_s2_filters_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)wt, (float*)wsub, Ci, Co);
    return pd::check_launch("pd_dgrad_s2_filters");
}

extern "C" int pd_interleave4(const void* sub, void* dx, int N, int Ho, int Wo, int C, void* stream) {
    PD_REQUIRE(sub && dx && N > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, "pd_interleave4: bad arguments");
    const long total = 4L * N * Ho * Wo * (C / 4), blocks = (total + 255) / 256;
    hipLaunchKernelGGL(interleave4_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)sub, (float*)dx, N, Ho, Wo, C);
    return pd::check_launch("pd_interleave4");
}

// ===================================================================== 7x7 / stride-2 stems as 4x4 / stride-1
// A 7x7 stride-2 pad-3 convolution over [C,H,W] equals a 4x4 stride-1 pad-2 convolution over the
// space-to-depth tensor X2[H/2][W/2][4C] (channel q = (dy*2+dx)*C + c holds x[c][2i+dy][2j+dx]) with the
// filter taps regrouped: kh = 2a + dy - 1, kw = 2b + dx - 1 (taps -1 are zero).  4C is a multiple of 4, so the
// 2/3/9-channel stems (ShallowEncoder.Conv1, resnet conv1) take the 16-byte gather path of the implicit GEMM
// instead of the scalar one (K grows from 49C to 64C, the kernels get ~2x faster).
namespace {

// x: [N,C,H,W] with element strides; out: [N][H/2][W/2][4C]; optional (x - sub) / div
__global__ __launch_bounds__(256) void s2d_input_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                        int C, int H, int W, long sN, long sC, long sH, long sW,
                                                        int affine, float sub, float div) {
    const int H2 = H >> 1, W2 = W >> 1, C4 = 4 * C;
    const long total = (long)N * H2 * W2 * C4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4);
        long pix = i / C4;
        const int j = (int)(pix % W2); pix /= W2;
        const int ii = (int)(pix % H2);
        const long n = pix / H2;
        const int d = q / C, c = q - d * C;
        float v = x[n * sN + c * sC + (2 * ii + (d >> 1)) * sH + (2 * j + (d & 1)) * sW];
        if (affine) v = (v - sub) / div;
        out[i] = v;
    }
}

// w: [Co][7][7][C] -> w2: [Co][4][4][4C];  inverse (gradient): dw[co][kh][kw][c] (+)= dw2[co][a][b][(dy,dx,c)]
__global__ __launch_bounds__(256) void s2d_weight_kernel(const float* __restrict__ w, float* __restrict__ w2, int Co,
                                                         int C, int inverse, int accumulate, float* __restrict__ dw) {
    const int C4 = 4 * C;
    const long total = (long)Co * 16 * C4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4);
        long r = i / C4;
        const int b = (int)(r & 3); r >>= 2;
        const int a = (int)(r & 3);
        const long co = r >> 2;
        const int d = q / C, c = q - d * C;
        const int kh = 2 * a + (d >> 1) - 1, kw = 2 * b + (d & 1) - 1;
        const bool ok = kh >= 0 && kw >= 0;
        const long src = ((co * 7 + kh) * 7 + kw) * C + c;
        if (!inverse) w2[i] = ok ? w[src] : 0.f;
        else if (ok) dw[src] = (accumulate ? dw[src] : 0.f) + w2[i];
    }
}

// Row-contiguous planes (sW == 1): a workgroup reads the two source rows of 64 output pixels plane by plane (coalesced),
// regroups them in LDS and writes 64 x 4C contiguous floats.  (One thread per output element read 4-byte words a plane
// apart: 0.25 ms per step for 0.6 GB.)
constexpr int S2D_TW = 64;
__global__ __launch_bounds__(256) void s2d_input_tile_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                             int C, int H, int W, long sN, long sC, long sH, int affine,
                                                             float sub, float div, int tiles_w) {
    extern __shared__ float s2d_lds[];                     // [S2D_TW][4C + 1]
    const int H2 = H >> 1, W2 = W >> 1, C4 = 4 * C, LDP = C4 + 1;
    const int tw = blockIdx.x % tiles_w;
    long r = blockIdx.x / tiles_w;
    const int ii = (int)(r % H2);
    const long n = r / H2;
    const int j0 = tw * S2D_TW, nj = min(S2D_TW, W2 - j0);
    const float* xb = x + n * sN + (long)(2 * ii) * sH + 2 * j0;
    for (int e = threadIdx.x; e < 2 * C * 2 * S2D_TW; e += 256) {
        const int col = e & (2 * S2D_TW - 1), cd = e >> 7;          // 128 source columns per (c, dy)
        const int dy = cd & 1, c = cd >> 1;
        const int j = col >> 1, dx = col & 1;
        if (j < nj) {
            float v = xb[c * sC + dy * sH + col];
            if (affine) v = (v - sub) / div;
            s2d_lds[j * LDP + (dy * 2 + dx) * C + c] = v;
        }
    }
    __syncthreads();
    float* ob = out + ((n * H2 + ii) * (long)W2 + j0) * C4;
    for (int t = threadIdx.x; t < nj * C4; t += 256) {
        const int j = t / C4, q = t - j * C4;
        ob[t] = s2d_lds[j * LDP + q];
    }
}

}  // namespace

extern "C" int pd_stem_s2d_input(const void* x, void* out, int N, int C, int H, int W, long sN, long sC, long sH,
                                 long sW, int affine, float sub, float div, void* stream) {
    PD_REQUIRE(x && out && N >= 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "pd_stem_s2d_input: bad arguments");
    if (N == 0) return PD_OK;
    const long total = (long)N * H * W * C;
    const int tiles_w = (W / 2 + S2D_TW - 1) / S2D_TW;
    const long blocks = (long)N * (H / 2) * tiles_w;
    const size_t lds = (size_t)S2D_TW * (4 * C + 1) * sizeof(float);
    if (sW == 1 && lds <= 48 * 1024 && blocks < (1L << 31)) {
        hipLaunchKernelGGL(s2d_input_tile_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, (const float*)x,
                           (float*)out, N, C, H, W, sN, sC, sH, affine, sub, div, tiles_w);
        return pd::check_launch("pd_stem_s2d_input");
    }
    hipLaunchKernelGGL(s2d_input_kernel, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256),
                       0, (hipStream_t)stream, (const float*)x, (float*)out, N, C, H, W, sN, sC, sH, sW, affine, sub, div);
    return pd::check_launch("pd_stem_s2d_input");
}

extern "C" int pd_stem_s2d_weight(const void* w, void* w2, int Co, int C, void* stream) {
    PD_REQUIRE(w && w2 && Co > 0 && C > 0, "pd_stem_s2d_weight: bad arguments");
    const long total = (long)Co * 64 * C;
    hipLaunchKernelGGL(s2d_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)w, (float*)w2, Co, C, 0, 0, (float*)nullptr);
    return pd::check_launch("pd_stem_s2d_weight");
}

extern "C" int pd_stem_s2d_weight_grad(const void* dw2, void* dw, int Co, int C, int accumulate, void* stream) {
    PD_REQUIRE(dw2 && dw && Co > 0 && C > 0, "pd_stem_s2d_weight_grad: bad arguments");
    const long total = (long)Co * 64 * C;
    hipLaunchKernelGGL(s2d_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)nullptr, (float*)dw2, Co, C, 1, accumulate, (float*)dw);
    return pd::check_launch("pd_stem_s2d_weight_grad");
}
