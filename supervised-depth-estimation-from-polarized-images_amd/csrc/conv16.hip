// K2t -- the 16-channel decoder tail (upconv(0,0) 32->16 @256x320, upconv(0,1) 16->16 @512x640 and their data
// gradients): 3x3 convolutions with 16 output channels per 16-wide matrix tile.
//
// At 16 output channels every staged input byte feeds 8 FLOP, so the implicit-GEMM kernels (conv.hip), which
// re-stage the input once per filter tap, are bounded by the global -> LDS path (~59 TF measured ceiling, 51-57 TF
// achieved).  Here a workgroup stages the input pixels of an 8 x 32 output tile ONCE -- a 10 x 34 halo tile, ~1.3x the
// tile instead of 9x -- and reads all nine taps from LDS:
//   * halo[(row, col)][C] with a pixel stride of C + 4 floats: the 16 lanes of a fragment read (16 consecutive pixels,
//     16 bytes each) land on 16 different 4-bank groups without any XOR, so every tap / row / channel-chunk is an
//     IMMEDIATE offset from one per-lane base address -- no VALU in the main loop (the fp32 MFMA shares the FMA
//     lanes with the VALU, see conv.hip);
//   * the whole filter sits in LDS as [tap][ci/16][ci%16/4][cout] float4: a B fragment is one ds_read_b128;
//   * v_mfma_f32_16x16x4_f32: lane (pixel or cout = lane%16, kk = lane/16) holds four consecutive ci of a 16-channel
//     chunk; MFMA number c contracts ci = 4*kk + c over kk -- a permutation of ci shared by A and B.
// MODE 0: forward, reflection padding 1, bias + ELU/none (layers.py:364-380 Conv3x3 + ConvBlock's ELU).
// MODE 1: data gradient on the padded (H+2) x (W+2) grid (zero outside the image), folded by pd_reflect_fold.
// MODE 2: zero-padding (pad 1) data gradient on the H x W grid = the interior of MODE 1's result; the reflected border
//         strips are added by pd_reflect_dgrad_border.
#include "pd_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB16 = 0x80000000u;
constexpr int TR = 8, TC = 32;            // output tile
constexpr int HR = TR + 2, HC = TC + 2;   // halo tile

struct Conv16Args {
    const float* x;      // NHWC input (element strides sN, sH, sW; channel stride 1)
    const float* w;      // [NCO*16][3][3][NCI*16]
    const float* bias;   // [NCO*16] or null
    float* y;            // NHWC output, row stride ldy
    int N, H, W;         // input grid
    long sN, sH, sW;
    int Ho, Wo;          // output grid
    long ldy;
    int act;             // 0 none, 2 ELU
    int tiles_r, tiles_c, ntiles;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc16(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int NCI, int NCO, int MODE>
__global__ __launch_bounds__(256) void conv16_halo_kernel(const Conv16Args a) {
    constexpr int C = 16 * NCI, PS = C + 4, SP = C / 4;          // channels, pixel stride (floats), 16-byte slots per pixel
    constexpr int WL4 = 9 * NCI * 4 * 16 * NCO;                  // float4 entries of the filter image
    __shared__ __attribute__((aligned(16))) float halo[HR * HC * PS];
    __shared__ __attribute__((aligned(16))) float wl[WL4 * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int hmin = MODE == 1 ? -2 : -1;                    // source row of halo row 0 relative to r0 (same for columns)

    // ---- filter -> LDS: w[n][tap][k] -> wl[((tap*NCI + s)*4 + kk)*(16*NCO) + n][c], k = 16 s + 4 kk + c
    for (int i = tid; i < 9 * C * 16 * NCO / 4; i += 256) {      // one float4 of consecutive k per item
        const int k4 = i % (C / 4);
        const int t = (i / (C / 4)) % 9;
        const int nn = i / (C / 4) / 9;
        const float4 v = *reinterpret_cast<const float4*>(a.w + ((long)nn * 9 + t) * C + 4 * k4);
        const int s = k4 >> 2, kk = k4 & 3;
        *reinterpret_cast<float4*>(&wl[((((t * NCI + s) * 4 + kk) * (16 * NCO)) + nn) * 4]) = v;
    }

    // Persistent workgroups (the grid is what is resident at once) walk the 8 x 32 tiles with the filter image staying
    // in LDS; over time the workgroups of a CU drift out of phase, so the load phase of one overlaps the MFMAs of another
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int b = tile;
    const int tc = b % a.tiles_c; b /= a.tiles_c;
    const int tr = b % a.tiles_r;
    const int n = b / a.tiles_r;
    const int r0 = tr * TR, c0 = tc * TC;
    if (tile != (int)blockIdx.x) __syncthreads();                // the previous tile's staging reads are done
    // ---- halo tile -> LDS (each input pixel once; rows are uniform per iteration, columns fixed per thread)
    {
        const long img_bytes = ((long)a.H - 1) * a.sH * 4 + ((long)a.W - 1) * a.sW * 4 + C * 4;
        const __amdgpu_buffer_rsrc_t rx = rsrc16(a.x + (long)n * a.sN, (unsigned)img_bytes);
        constexpr int ITEMS = HC * SP;                           // 16-byte items per halo row: 136 or 272
        constexpr int PASSES = (ITEMS + 255) / 256;
        unsigned coff[PASSES];                                   // column part of the byte offset, or OOB
        int dsto[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int it = tid + 256 * p;
            const int px = it / SP, slot = it % SP;
            int sc = c0 + hmin + px;
            bool ok = it < ITEMS;
            if (MODE == 0) { sc = sc < 0 ? -sc : sc; sc = sc >= a.W ? 2 * a.W - 2 - sc : sc; ok = ok && sc >= 0 && sc < a.W; }
            else ok = ok && sc >= 0 && sc < a.W;
            coff[p] = ok ? (unsigned)(sc * (int)a.sW + 4 * slot) * 4u : OOB16;
            dsto[p] = it < ITEMS ? px * PS + 4 * slot : -1;
        }
        f32x4 hv[HR][PASSES];                                    // all loads in flight before the first LDS write
#pragma unroll
        for (int hr = 0; hr < HR; ++hr) {
            int sr = r0 + hmin + hr;
            bool rok = true;
            if (MODE == 0) { sr = sr < 0 ? -sr : sr; sr = sr >= a.H ? 2 * a.H - 2 - sr : sr; rok = sr >= 0 && sr < a.H; }
            else rok = sr >= 0 && sr < a.H;
            const unsigned roff = rok ? (unsigned)(sr * (int)a.sH) * 4u : OOB16;
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                hv[hr][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (roff | coff[p]) & OOB16 ? OOB16 : roff + coff[p], 0, 0));
        }
#pragma unroll
        for (int hr = 0; hr < HR; ++hr)
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                if (dsto[p] >= 0) *reinterpret_cast<f32x4*>(&halo[hr * HC * PS + dsto[p]]) = hv[hr][p];
    }
    __syncthreads();

    // ---- main loop: wave w owns output rows 2w, 2w+1 of the tile = four 16-pixel segments
    const int m = lane & 15, kk = lane >> 4;
    const float* abase = halo + (2 * wave * HC + m) * PS + 4 * kk;
    const float* bbase = wl + (kk * (16 * NCO) + m) * 4;          // (m doubles as the cout index of the B fragment)
    f32x4 acc[4][NCO];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NCO; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int th = MODE == 0 ? kh : 2 - kh, tw = MODE == 0 ? kw : 2 - kw;   // halo offset of the tap
            const int tap = kh * 3 + kw;
#pragma unroll
            for (int s = 0; s < NCI; ++s) {
                f32x4 bf[NCO], af[4];
#pragma unroll
                for (int j = 0; j < NCO; ++j)
                    bf[j] = *reinterpret_cast<const f32x4*>(bbase + ((tap * NCI + s) * 4 * (16 * NCO) + 16 * j) * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rowl = t >> 1, half = t & 1;                            // (the wave's rows are in abase)
                    af[t] = *reinterpret_cast<const f32x4*>(abase + ((rowl + th) * HC + 16 * half + tw) * PS + 16 * s);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < NCO; ++j) {
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t].x, bf[j].x, acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t].y, bf[j].y, acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t].z, bf[j].z, acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t].w, bf[j].w, acc[t][j], 0, 0, 0);
                    }
            }
        }

    // ---- epilogue: D layout of 16x16x4: column (cout) = lane % 16, rows (pixels) 4*(lane/16) + r.  Bias + activation,
    // then the wave transposes its 64 pixels through LDS (the halo tile is dead once every wave is past its MFMAs) and
    // stores 16 bytes per lane: whole 128-byte lines instead of sixteen 64-byte halves per lane.
    constexpr int OS = 20;                                        // pixel stride of the staging tile (floats), 16 channels at a time
    static_assert(256 * OS <= HR * HC * PS, "output staging fits the halo array");
    __syncthreads();
    float* ot = halo + wave * 64 * OS;
    const __amdgpu_buffer_rsrc_t ry = rsrc16(a.y + (long)n * a.Ho * a.Wo * a.ldy, (unsigned)((long)a.Ho * a.Wo * a.ldy * 4));
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
        const float bv = a.bias ? a.bias[16 * j + m] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[t][j][r] + bv;
                // ELU as torch evaluates it, exp(x) - 1 (Activation.cpp elu: (std::exp(x) - 1) * negcoef), on the hardware
                // exp2 (|err| < 2e-7 absolute): expm1f is ~40 VALU instructions per element, a fifth of this kernel's time
                if (a.act == 2) v = v > 0.f ? v : __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.f;
                ot[((t >> 1) * 32 + 16 * (t & 1) + 4 * kk + r) * OS + m] = v;
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = lane + 64 * it;
            const int px = i >> 2, slot = i & 3;                 // pixel of the wave's 64 (2 rows x 32), 16-byte slot
            const int oh = r0 + 2 * wave + (px >> 5), ow = c0 + (px & 31);
            const f32x4 v = *reinterpret_cast<const f32x4*>(&ot[px * OS + 4 * slot]);
            const bool ok = oh < a.Ho && ow < a.Wo;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ry,
                                                   ok ? (unsigned)(((long)oh * a.Wo + ow) * a.ldy + 16 * j + 4 * slot) * 4u : OOB16, 0, 0);
        }
    }
    }   // tile loop
}

// ---------------------------------------------------------------- weight (+ bias) gradient of the same layers
// dW[co][tap][ci] = sum_p dz[p][co] * x[reflect(p + tap - 1)][ci]: a 16 x (9 C) result contracted over millions of
// pixels.  Persistent workgroups walk the 8 x 32 tiles; per tile the x halo (reflect) and the dz tile are staged once;
// MFMA 16x16x4: A = dz^T (lane: co = lane%16, pixel = lane/16 of the 4-pixel step), B = x shifted by the tap (lane:
// ci = lane%16, same pixel), one accumulator per tap and 16-channel chunk -- 9 NCI MFMAs per (1 + 9 NCI) ds_read_b32.
// Pixel stride 16 floats in LDS: the 4 pixels x 16 channels of a fragment read cover the 64 banks exactly.
// Every workgroup leaves ONE partial [16][9][C] (+ [16] bias sums); reduce_rows sums them in a fixed order.
struct Wgrad16Args {
    const float* x;      // NHWC input of the forward convolution (strides sN, sH, sW)
    const float* dz;     // NHWC [N,H,W,16] gradient of the pre-activation output, row stride ldd
    float* part;         // [G][16*9*C]
    float* bpart;        // [G][16] or null
    int N, H, W;
    long sN, sH, sW, ldd;
    int tiles_r, tiles_c, ntiles;
};

template <int NCI>
__global__ __launch_bounds__(256) void conv16_wgrad_kernel(const Wgrad16Args a) {
    constexpr int C = 16 * NCI, SP = C / 4;
    __shared__ __attribute__((aligned(16))) float halo[HR * HC * C];     // [row][col][C]
    __shared__ __attribute__((aligned(16))) float dzt[TR * TC * 16];     // [row][col][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, pk = lane >> 4;                              // channel index, pixel of the step
    f32x4 acc[9][NCI];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < NCI; ++s) acc[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int b = tile;
        const int tc = b % a.tiles_c; b /= a.tiles_c;
        const int tr = b % a.tiles_r;
        const int n = b / a.tiles_r;
        const int r0 = tr * TR, c0 = tc * TC;
        __syncthreads();                                         // the previous tile's fragment reads are done
        {   // x halo, reflection padding (rows uniform per iteration, columns fixed per thread)
            const long img_bytes = ((long)a.H - 1) * a.sH * 4 + ((long)a.W - 1) * a.sW * 4 + C * 4;
            const __amdgpu_buffer_rsrc_t rx = rsrc16(a.x + (long)n * a.sN, (unsigned)img_bytes);
            constexpr int ITEMS = HC * SP, PASSES = (ITEMS + 255) / 256;
            unsigned coff[PASSES];
            int dsto[PASSES];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int it = tid + 256 * p;
                const int px = it / SP, slot = it % SP;
                int sc = c0 - 1 + px;
                sc = sc < 0 ? -sc : sc; sc = sc >= a.W ? 2 * a.W - 2 - sc : sc;
                const bool ok = it < ITEMS && sc >= 0 && sc < a.W;
                coff[p] = ok ? (unsigned)(sc * (int)a.sW + 4 * slot) * 4u : OOB16;
                dsto[p] = it < ITEMS ? px * C + 4 * slot : -1;
            }
            f32x4 hv[HR][PASSES], dv[TR * TC * 4 / 256];              // all loads in flight before the first LDS write
#pragma unroll
            for (int hr = 0; hr < HR; ++hr) {
                int sr = r0 - 1 + hr;
                sr = sr < 0 ? -sr : sr; sr = sr >= a.H ? 2 * a.H - 2 - sr : sr;
                const unsigned roff = (sr >= 0 && sr < a.H) ? (unsigned)(sr * (int)a.sH) * 4u : OOB16;
#pragma unroll
                for (int p = 0; p < PASSES; ++p)
                    hv[hr][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (roff | coff[p]) & OOB16 ? OOB16 : roff + coff[p], 0, 0));
            }
            // dz tile: pixels beyond the image read as zero (they contribute nothing)
            const __amdgpu_buffer_rsrc_t rd = rsrc16(a.dz + (long)n * a.H * a.W * a.ldd, (unsigned)((long)a.H * a.W * a.ldd * 4));
#pragma unroll
            for (int p = 0; p < TR * TC * 4 / 256; ++p) {
                const int it = tid + 256 * p;
                const int slot = it & 3, px = (it >> 2) % TC, row = (it >> 2) / TC;
                const bool ok = r0 + row < a.H && c0 + px < a.W;
                dv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rd, ok ? (unsigned)(((long)(r0 + row) * a.W + c0 + px) * a.ldd + 4 * slot) * 4u : OOB16, 0, 0));
            }
#pragma unroll
            for (int hr = 0; hr < HR; ++hr)
#pragma unroll
                for (int p = 0; p < PASSES; ++p)
                    if (dsto[p] >= 0) *reinterpret_cast<f32x4*>(&halo[hr * HC * C + dsto[p]]) = hv[hr][p];
#pragma unroll
            for (int p = 0; p < TR * TC * 4 / 256; ++p) {
                const int it = tid + 256 * p;
                *reinterpret_cast<f32x4*>(&dzt[(((it >> 2) / TC) * TC + (it >> 2) % TC) * 16 + 4 * (it & 3)]) = dv[p];
            }
        }
        __syncthreads();
        // wave w: rows 2w, 2w+1 of the tile, 8 steps of 4 pixels per row
        const float* abase = dzt + ((2 * wave) * TC + pk) * 16 + q;
        const float* bbase = halo + ((2 * wave) * HC + pk) * C + q;
#pragma unroll
        for (int rl = 0; rl < 2; ++rl)
#pragma unroll
            for (int st = 0; st < TC / 4; ++st) {
                const float av = abase[(rl * TC + 4 * st) * 16];
                bsum += av;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int s = 0; s < NCI; ++s) {
                            const float bv = bbase[((rl + kh) * HC + 4 * st + kw) * C + 16 * s];
                            acc[kh * 3 + kw][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[kh * 3 + kw][s], 0, 0, 0);
                        }
            }
    }
    // ---- workgroup reduction of the four waves' accumulators through LDS in a fixed order, (w0 + w2) + (w1 + w3);
    // wave 0 then writes the workgroup's partial straight from its registers
    __shared__ float bred[256];
    bred[tid] = bsum;
    float* red = halo;                                            // 2 waves x 9 NCI x 64 lanes x 4 floats fit the halo array
    constexpr int PER_WAVE = 9 * NCI * 64 * 4;
    static_assert(2 * PER_WAVE <= HR * HC * C, "reduction scratch");
    auto put = [&](int slot) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s = 0; s < NCI; ++s)
                *reinterpret_cast<f32x4*>(&red[slot * PER_WAVE + ((t * NCI + s) * 64 + lane) * 4]) = acc[t][s];
    };
    auto take = [&](int slot) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s = 0; s < NCI; ++s) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(&red[slot * PER_WAVE + ((t * NCI + s) * 64 + lane) * 4]);
                acc[t][s].x += u.x; acc[t][s].y += u.y; acc[t][s].z += u.z; acc[t][s].w += u.w;
            }
    };
    __syncthreads();                                              // (all fragment reads of the last tile are done)
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) take(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0) {
        take(0);
        float* out = a.part + (long)blockIdx.x * (16 * 9 * C);
        // D layout: column (ci) = lane % 16, rows (co) = 4 * (lane / 16) + r
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s = 0; s < NCI; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[((4 * pk + r) * 9 + t) * C + 16 * s + q] = acc[t][s][r];
    }
    if (a.bpart && tid < 16) {
        float sb = 0.f;
        for (int i = tid; i < 256; i += 16) sb += bred[i];
        a.bpart[(long)blockIdx.x * 16 + tid] = sb;
    }
}

}  // namespace

extern "C" int pd_conv16(const void* x, const void* w, const void* bias, void* y, int N, int H, int W, int C,
                         long sN, long sH, long sW, int Ho, int Wo, int Cout, long ldy, int mode, int act, void* stream) {
    PD_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "pd_conv16: bad arguments");
    PD_REQUIRE(mode >= 0 && mode <= 2, "pd_conv16: mode 0 (reflect forward), 1 (data gradient on the padded grid), 2 (pad-1 data gradient)");
    PD_REQUIRE(act == 0 || act == 2, "pd_conv16: activation none or ELU");
    PD_REQUIRE((mode == 0 && Cout == 16 && (C == 16 || C == 32)) || (mode != 0 && C == 16 && (Cout == 16 || Cout == 32)),
               "pd_conv16: unsupported channel counts %d -> %d (mode %d)", C, Cout, mode);
    PD_REQUIRE(mode == 1 || (H >= 2 && W >= 2 && Ho == H && Wo == W), "pd_conv16: modes 0 and 2 keep the grid (H, W >= 2)");
    PD_REQUIRE(mode != 1 || (Ho == H + 2 && Wo == W + 2), "pd_conv16: the data gradient lands on the (H+2) x (W+2) grid");
    PD_REQUIRE(sN % 4 == 0 && sH % 4 == 0 && sW % 4 == 0 && pd::aligned16(x) && pd::aligned16(w) && ldy >= Cout,
               "pd_conv16: 16-byte aligned NHWC operands");
    PD_REQUIRE(((long)H - 1) * sH * 4 + ((long)W - 1) * sW * 4 + C * 4 < 0x7fffffffL && (long)Ho * Wo * ldy * 4 < 0x7fffffffL,
               "pd_conv16: image too large for 32-bit offsets");
    Conv16Args a;
    a.x = (const float*)x; a.w = (const float*)w; a.bias = (const float*)bias; a.y = (float*)y;
    a.N = N; a.H = H; a.W = W; a.sN = sN; a.sH = sH; a.sW = sW; a.Ho = Ho; a.Wo = Wo; a.ldy = ldy; a.act = act;
    a.tiles_r = (Ho + TR - 1) / TR; a.tiles_c = (Wo + TC - 1) / TC;
    a.ntiles = N * a.tiles_r * a.tiles_c;
    const int resident = 256 * (C == 16 && Cout == 16 ? 4 : (mode == 1 ? 3 : 2));   // workgroups per CU by LDS: 36 / 46 / 67 KB
    const dim3 grid((unsigned)(a.ntiles < resident ? a.ntiles : resident)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) {
        if (C == 16) hipLaunchKernelGGL((conv16_halo_kernel<1, 1, 0>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv16_halo_kernel<2, 1, 0>), grid, block, 0, st, a);
    } else if (mode == 1) {
        if (Cout == 16) hipLaunchKernelGGL((conv16_halo_kernel<1, 1, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv16_halo_kernel<1, 2, 1>), grid, block, 0, st, a);
    } else {
        if (Cout == 16) hipLaunchKernelGGL((conv16_halo_kernel<1, 1, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv16_halo_kernel<1, 2, 2>), grid, block, 0, st, a);
    }
    return pd::check_launch("pd_conv16");
}

namespace {
// out[i] (+)= sum_g part[g][i], fixed order (deterministic): 32 columns x 32 row lanes per workgroup -- the result has
// only 2304-4608 columns, so the rows must supply the parallelism (64 x 4 lanes left 36 workgroups chasing 1024 rows:
// 0.3 ms of latency)
__global__ __launch_bounds__(1024) void reduce16_kernel(const float* __restrict__ part, float* __restrict__ out, int G, int n,
                                                        int accumulate) {
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int g = sl;
        for (; g + 96 < G; g += 128) {
            a0 += part[(long)g * n + i]; a1 += part[(long)(g + 32) * n + i];
            a2 += part[(long)(g + 64) * n + i]; a3 += part[(long)(g + 96) * n + i];
        }
        for (; g < G; g += 32) a0 += part[(long)g * n + i];
    }
    red[sl][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += red[r][col];
        out[i] = accumulate ? out[i] + s : s;
    }
}
}  // namespace

extern "C" size_t pd_conv16_wgrad_workspace(int C) { return (size_t)1024 * (16 * 9 * (size_t)C + 16) * sizeof(float); }

extern "C" int pd_conv16_wgrad(const void* x, const void* dz, void* dw, void* dbias, void* workspace, size_t ws_bytes,
                               int N, int H, int W, int C, long sN, long sH, long sW, long ldd, int accumulate,
                               void* stream) {
    PD_REQUIRE(x && dz && dw && workspace && N > 0 && H >= 2 && W >= 2, "pd_conv16_wgrad: bad arguments");
    PD_REQUIRE(C == 16 || C == 32, "pd_conv16_wgrad: 16 or 32 input channels");
    PD_REQUIRE(ws_bytes >= pd_conv16_wgrad_workspace(C), "pd_conv16_wgrad: workspace too small");
    PD_REQUIRE(sN % 4 == 0 && sH % 4 == 0 && sW % 4 == 0 && ldd % 4 == 0 && pd::aligned16(x) && pd::aligned16(dz),
               "pd_conv16_wgrad: 16-byte aligned NHWC operands");
    PD_REQUIRE(((long)H - 1) * sH * 4 + ((long)W - 1) * sW * 4 + C * 4 < 0x7fffffffL && (long)H * W * ldd * 4 < 0x7fffffffL,
               "pd_conv16_wgrad: image too large for 32-bit offsets");
    Wgrad16Args a;
    a.x = (const float*)x; a.dz = (const float*)dz;
    a.N = N; a.H = H; a.W = W; a.sN = sN; a.sH = sH; a.sW = sW; a.ldd = ldd;
    a.tiles_r = (H + TR - 1) / TR; a.tiles_c = (W + TC - 1) / TC;
    a.ntiles = N * a.tiles_r * a.tiles_c;
    // persistent grid = the workgroups resident at once (3 per CU at 39 KB of LDS and 152 registers): loads of one
    // workgroup overlap the MFMAs of its neighbours (loads alone 0.15 ms, MFMAs alone 0.19 ms, together 0.25 ms at 16->16)
    const int G = a.ntiles < 768 ? a.ntiles : 768;
    const int nw = 16 * 9 * C;
    a.part = (float*)workspace;
    a.bpart = dbias ? a.part + (size_t)1024 * nw : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) hipLaunchKernelGGL((conv16_wgrad_kernel<1>), dim3(G), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv16_wgrad_kernel<2>), dim3(G), dim3(256), 0, st, a);
    int rc = pd::check_launch("pd_conv16_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(reduce16_kernel, dim3((nw + 31) / 32), dim3(1024), 0, st, a.part, (float*)dw, G, nw, accumulate);
    if (dbias) hipLaunchKernelGGL(reduce16_kernel, dim3(1), dim3(1024), 0, st, a.bpart, (float*)dbias, G, 16, accumulate);
    return pd::check_launch("pd_conv16_wgrad/reduce");
}
