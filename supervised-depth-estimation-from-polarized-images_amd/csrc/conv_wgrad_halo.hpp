// conv_wgrad_halo_x3_kernel -- bf16-split weight gradient with BOTH operands split once per tile and read through the
// hardware transpose (included by conv.hip).
//
//   dW[co][kh][kw][ci] = sum_p dY[p][co] * X[p + (kh, kw) - pad][ci]
// contracts over PIXELS, but both tensors are stored pixel-major (NHWC): an MFMA operand wants eight consecutive
// contraction indices per lane, i.e. eight pixels of ONE channel.  conv_wgrad_x3c_kernel gets them with eight ds_read_b32 per
// fragment from an im2col image that is re-gathered (LDS-DMA) and re-split for every 128 (tap, ci) columns: every input
// element is fetched 9 (25) times and split 9 (25) times per 64 output channels, 4.5 vector + 5-8 scalar instructions per MFMA.
// gfx950's ds_read_b64_tr_b16 reads a 4 x 16 block of 16-bit elements per 16-lane group and hands each lane a COLUMN: from a
// plain [pixel][channel] bf16 image a lane receives four consecutive pixels of its channel -- the operand layout, for free,
// starting at ANY pixel.  So here a workgroup owns one filter row kh, 64 output and 64 input channels and walks 2 x 32-pixel
// tiles of its slice (4 x 16 / 8 x 8 where 32 does not divide the width):
//   * dY tile (64 px x 64 co) and X row pair (2 x (32 + K - 1) px x 64 ci, shifted by kh) go global -> registers one tile
//     ahead, are split ONCE into three bf16 planes [term][channel block][pixel][32 channels] and written to LDS;
//   * wave (co block, ci block) reads its A fragments (dY) once per 16-pixel step and the B fragment of tap kw at pixel
//     offset kw -- two ds_read_b64_tr_b16 per term -- and issues 6 MFMAs per (step, kw) into the accumulator tile of kw;
//   * nothing is gathered per tap, no pixel walker, no masks in the loop: ~2 vector instructions per MFMA (3x3), 1.3 (5x5);
//     52 KB of LDS, three (3x3) or two (5x5: registers) workgroups per CU.
// Partial tiles [slice][co][K] and the ordered reduction are those of the other weight-gradient kernels.  Zero or reflection padding,
// stride 1, K = 3 | 5, C % 64 == 0, Cout % 64 == 0, the output grid a whole number of 2 x 32, 4 x 16 or 8 x 8 tiles.
// CO32 (round 4: the decoder's 96 -> 32 and 64 -> 32 layers): a workgroup owns 32 output channels; its wave pairs take the
// two 16-pixel steps of a tile instead of two output-channel blocks and write two partial slices (2 s, 2 s + 1).  C % 32 == 0:
// the upper 32-channel block of the last 64-channel input block may be empty (its wave idles, its loads are masked).
namespace wgh {
constexpr int NPX = 32;                                                  // pixels per tile: TR rows x TW columns, 2 MFMA k-steps
// TW = 32 | 16 | 8 | 4 (the widest that divides Wo): 1 x 32, 2 x 16, 4 x 8 or 8 x 4 pixel tiles -- a 16-pixel step is then half a
// row, a row, two or four rows; either way the step's pixels are the tile's row-major pixels 16 ks .. 16 ks + 15, in runs of four
// consecutive columns (what one transposed read takes).  (8 x 4: the 16x20 planes of the 512-channel layers.)
template <int KS, int TW, bool CO32> struct Geo {
    static constexpr int TR = NPX / TW;
    static constexpr int HW = TW + KS - 1;                               // X columns per tile row
    static constexpr int XPX = TR * HW;                                  // X pixels per tile (one filter row: no vertical halo)
    static constexpr int NIX = (XPX * 16 + NT - 1) / NT;                 // 16-byte X items per thread (64 channels = 16 quads per pixel)
    static constexpr int NCOB = CO32 ? 1 : 2;                            // 32-channel blocks of dY per workgroup
    static constexpr unsigned DP_BYTES = 3 * NCOB * NPX * 64;            // dY planes of one tile: [term][co block][pixel][32 ch x 2 B]
    static constexpr unsigned XP_BYTES = 3 * 2 * XPX * 64;               // X planes of one tile: [term][ci block][pixel][32 ch x 2 B]
    static constexpr unsigned BUF_BYTES = DP_BYTES + XP_BYTES;           // two tiles live in LDS: the one multiplied, the one being written
    static constexpr unsigned LDS_BYTES = 2 * BUF_BYTES;
};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) char lds_char;
// 4 consecutive pixels (rows of 64 bytes) x the lane's channel: ds_read_b64_tr_b16 (all 64 lanes active); `imm` is a
// compile-time constant at every call site and lands in the instruction's offset field
__device__ __forceinline__ uint2 tr_read(lds_char* base, unsigned imm) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + imm));
    return __builtin_bit_cast(uint2, v);
}
}  // namespace wgh

struct WgradHaloArgs {
    WgradArgs g;
    int ncb;              // 64-channel input blocks
    int tiles_w, tiles_h, ntiles_total, tiles_per_slice;
};

template <int KS, int TW, bool BIAS, bool CO32>
__global__ __launch_bounds__(NT, KS == 3 ? 3 : 2) void conv_wgrad_halo_x3_kernel(const WgradHaloArgs ha) {
    using namespace wgh;
    using G = Geo<KS, TW, CO32>;
    constexpr int HW = G::HW, XPX = G::XPX, NIX = G::NIX, TR = G::TR, NCOB = G::NCOB, ND = CO32 ? 1 : 2, BCO = 32 * NCOB;
    constexpr unsigned DP_BYTES = G::DP_BYTES, BUF_BYTES = G::BUF_BYTES;
    static_assert(NIX <= 3 && 16 * (NIX - 1) < XPX, "X items per thread");
    const WgradArgs& a = ha.g;
    __shared__ __attribute__((aligned(16))) float smem_all[G::LDS_BYTES / 4];
    char* lds_c = reinterpret_cast<char*>(smem_all);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = KS * ha.ncb * a.ctiles * a.S;
    int b = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (b >= nwg) return;
    const int kh = b % KS; b /= KS;
    const int cb = b % ha.ncb; b /= ha.ncb;
    const int cob = b % a.ctiles;
    const int s = b / a.ctiles;
    const int co0 = cob * BCO, ci0 = cb * 64;
    // wave -> (output-channel block, input-channel block); CO32: (16-pixel step, input-channel block)
    const int w_cob = CO32 ? 0 : (wave & 1), w_cib = CO32 ? (wave & 1) : (wave >> 1), w_ks = CO32 ? (wave >> 1) : 0;
    const bool w_live = ci0 + 32 * w_cib < a.C;                        // (uniform) C % 64 == 32: the last block's upper half is empty
    const int t_beg = s * ha.tiles_per_slice;
    const int t_end = min(t_beg + ha.tiles_per_slice, ha.ntiles_total);

    // ---- items of this thread (tile-invariant part): a tile is 2 (CO32: 1) dY items and NIX (<= 3) X items of 16 bytes per thread
    const int quad = tid & 15;                                          // 4 channels: block quad >> 3, unit quad & 7
    unsigned d_voff[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int p = CO32 ? (tid >> 3) : (tid >> 4) + 16 * i, r = p / TW, c = p % TW;      // pixel p of the TR x TW tile
        d_voff[i] = (unsigned)((r * a.Wo + c) * (int)a.ldd + co0 + 4 * (CO32 ? (tid & 7) : quad)) * 4u;
    }
    const unsigned d_dst = CO32 ? (unsigned)((tid >> 3) * 64 + (tid & 7) * 8)
                                : (unsigned)((((quad >> 3) * NPX + (tid >> 4)) * 64) + (quad & 7) * 8);                 // + buf, term, 1024 i
    int x_hr[NIX], x_hx[NIX];
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        const int p = (tid >> 4) + 16 * i;                              // pixel p of the TR x HW strip
        x_hr[i] = p / HW; x_hx[i] = p - x_hr[i] * HW;
    }
    const bool x_last_ok = (tid >> 4) + 16 * (NIX - 1) < XPX;
    const unsigned x_dst = DP_BYTES + (unsigned)((((quad >> 3) * XPX + (tid >> 4)) * 64) + (quad & 7) * 8);          // + buf, term, 1024 i
    const unsigned x_cq = (unsigned)(ci0 + 4 * quad) * 4u;
    const bool x_ch_ok = ci0 + 4 * quad < a.C;

    // ---- the tile being loaded (two ahead of the one multiplied): walked incrementally, no divisions in the loop
    int l_t = t_beg, l_n, l_ty, l_tx;
    {
        const int per_img = ha.tiles_h * ha.tiles_w;
        l_n = t_beg / per_img;
        const int rem = t_beg - l_n * per_img;
        l_ty = rem / ha.tiles_w; l_tx = rem - l_ty * ha.tiles_w;
        l_n = __builtin_amdgcn_readfirstlane(l_n); l_ty = __builtin_amdgcn_readfirstlane(l_ty); l_tx = __builtin_amdgcn_readfirstlane(l_tx);
    }
    __amdgpu_buffer_rsrc_t rd, rx;
    int l_iy0 = 0, l_ix0 = 0;
    auto tile_begin = [&]() {                                          // descriptors / origin of tile l_t (scalar)
        const int oy0 = l_ty * TR, ox0 = l_tx * TW;
        const long pix0 = ((long)l_n * a.Ho + oy0) * a.Wo + ox0;
        const bool live = l_t < t_end;                                   // past the slice: empty descriptors, every load returns zero
        rd = make_rsrc(a.dy + pix0 * a.ldd, live ? (unsigned)(((long)(TR - 1) * a.Wo + TW) * a.ldd * 4) : 0u);
        rx = make_rsrc(a.x + (long)l_n * a.sN, live ? (unsigned)((long)a.sN * 4) : 0u);
        l_iy0 = oy0 - a.pad + kh; l_ix0 = ox0 - a.pad;
    };
    auto tile_next = [&]() {
        ++l_t;
        if (++l_tx == ha.tiles_w) { l_tx = 0; if (++l_ty == ha.tiles_h) { l_ty = 0; ++l_n; } }
    };
    float4 dv[ND], xv[NIX];
    auto load_d = [&](auto i_tag) { constexpr int I = decltype(i_tag)::value; dv[I] = buf_ld4(rd, d_voff[I]); };
    auto load_x = [&](auto i_tag) {
        constexpr int I = decltype(i_tag)::value;
        int iy = l_iy0 + x_hr[I], ix = l_ix0 + x_hx[I];
        if (a.mode == MODE_REFLECT) {                   // (uniform) ReflectionPad2d: the strip holds the mirrored pixels
            iy = iy < 0 ? -iy : iy; iy = iy >= a.H ? 2 * a.H - 2 - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= a.W ? 2 * a.W - 2 - ix : ix;
        }
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && (I < NIX - 1 || x_last_ok) && x_ch_ok;
        xv[I] = buf_ld4(rx, ok ? (unsigned)(iy * (int)a.sH + ix * (int)a.sW) * 4u + x_cq : OOB);
    };
    // Odd slices accumulate the NEGATED gradient (dY planes with flipped sign bits) and negate their partial tile at the end:
    // the bf16 MFMA's truncation bias (toward -infinity whatever the signs, ~2^-32 of the accumulator per instruction, thousands
    // of instructions per slice) then points the other way in half of the slices and cancels in their sum.
    const unsigned dsign = (s & 1) ? 0x80008000u : 0u;
    float4 bs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_bias = BIAS && a.bpart != nullptr && kh == 0 && cb == 0;
    auto put3 = [&](const float4 v, unsigned off, unsigned plane_stride, unsigned sg) {
        uint2 h, m, l;
        x3::split2(v.x, v.y, h.x, m.x, l.x);
        x3::split2(v.z, v.w, h.y, m.y, l.y);
        *reinterpret_cast<uint2*>(lds_c + off) = uint2{h.x ^ sg, h.y ^ sg};
        *reinterpret_cast<uint2*>(lds_c + off + plane_stride) = uint2{m.x ^ sg, m.y ^ sg};
        *reinterpret_cast<uint2*>(lds_c + off + 2 * plane_stride) = uint2{l.x ^ sg, l.y ^ sg};
    };
    // item K (0 .. ND - 1: dY; ND .. ND + NIX - 1: X) of the tile in the registers -> planes of buffer DST; its registers then
    // take the same item of the tile two ahead
    auto item = [&](auto dst_tag, auto k_tag) {
        constexpr unsigned DST = decltype(dst_tag)::value;
        constexpr int K = decltype(k_tag)::value;
        if constexpr (K < ND) {
            put3(dv[K], DST * BUF_BYTES + d_dst + 1024u * K, NCOB * NPX * 64, dsign);
            if constexpr (BIAS) { if (do_bias) { bs4.x += dv[K].x; bs4.y += dv[K].y; bs4.z += dv[K].z; bs4.w += dv[K].w; } }
            load_d(k_tag);
        } else if constexpr (K - ND < NIX) {
            constexpr int I = K - ND;
            if (I < NIX - 1 || x_last_ok) put3(xv[I], DST * BUF_BYTES + x_dst + 1024u * I, 2 * XPX * 64, 0u);
            load_x(std::integral_constant<int, I>{});
        }
    };

    // ---- fragment addresses: lane -> (half h: pixels 8h..8h+7 of the step; group gq: channels 16 gq..; row q, unit p of the block)
    const int fh = lane >> 5, gq = (lane >> 4) & 1, fq = (lane & 15) >> 2, fp = lane & 3;
    lds_char* const lds_a = (lds_char*)lds_c + (unsigned)(((w_cob * NPX + 16 * w_ks + 8 * fh + fq) * 64) + 32 * gq + 8 * fp);
    // X strip (row-major, HW columns): pixel 8 fh + 4 e + q of step ks sits at (row0(ks) + hrow, col0(ks) + hcol + 4 e + q + kw)
    // the upper half-wave (pixels 8 .. 15 of a step): 8 columns on (TW >= 16), the next row (TW = 8) or two rows down (TW = 4);
    // a lane's second run of four pixels: 4 columns on, or (TW = 4) the next row
    constexpr int HROW = TW >= 16 ? 0 : TW == 8 ? 1 : 2, HCOL = TW >= 16 ? 8 : 0, RUN2 = TW == 4 ? HW : 4;
    // (CO32: the wave's own step w_ks -- row / column origin of its 16 pixels -- is part of the base)
    const int ks_row = TW == 32 ? 0 : TW == 16 ? w_ks : TW == 8 ? 2 * w_ks : 4 * w_ks, ks_col = TW == 32 ? 16 * w_ks : 0;
    lds_char* const lds_b = (lds_char*)lds_c + (DP_BYTES + (unsigned)(((w_cib * XPX + ks_row * HW + ks_col + fh * (HROW * HW + HCOL) + fq) * 64) + 32 * gq + 8 * fp));

    typedef float accv_t __attribute__((ext_vector_type(16)));
    accv_t acc[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    auto bf = [](uint2 lo, uint2 hi) { return __builtin_bit_cast(x3::bf16x8, u32x4{lo.x, lo.y, hi.x, hi.y}); };

#define PD_I(n) std::integral_constant<int, n>{}
#define PD_U(n) std::integral_constant<unsigned, n>{}
    // ---- prologue: tile t_beg -> planes 0, tile t_beg + 1 -> registers
    tile_begin();
    load_d(PD_I(0)); if constexpr (!CO32) load_d(PD_I(1));
    load_x(PD_I(0)); load_x(PD_I(1)); load_x(PD_I(2));
    tile_next(); tile_begin();
    item(PD_U(0), PD_I(0)); item(PD_U(0), PD_I(1)); item(PD_U(0), PD_I(2)); item(PD_U(0), PD_I(3)); item(PD_U(0), PD_I(4));
    tile_next(); tile_begin();
    __syncthreads();

    // One tile: 2 steps x KS taps x 6 MFMAs on buffer BUF; between them the tile in the registers (t + 1) is split into buffer
    // BUF ^ 1 item by item, and each item's registers are refilled from tile t + 2 (l_*: its descriptors).
    auto tile = [&](auto buf_tag) {
        constexpr unsigned BUF = decltype(buf_tag)::value;
        const std::integral_constant<unsigned, BUF ^ 1> nxt{};
#pragma unroll
        for (int ks = 0; ks < (CO32 ? 1 : 2); ++ks) {
            x3::bf16x8 fa[3];
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
                fa[tm] = bf(tr_read(lds_a, BUF * BUF_BYTES + (unsigned)(tm * NCOB * NPX * 64 + ks * 16 * 64)),
                            tr_read(lds_a, BUF * BUF_BYTES + (unsigned)(tm * NCOB * NPX * 64 + ks * 16 * 64 + 4 * 64)));
            x3::bf16x8 fb[KS][3];
#pragma unroll
            for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) {
                    const int row0 = TW == 32 ? 0 : TW == 16 ? ks : TW == 8 ? 2 * ks : 4 * ks, col0 = TW == 32 ? 16 * ks : 0;
                    const unsigned o = BUF * BUF_BYTES + (unsigned)(tm * 2 * XPX * 64 + (row0 * HW + col0 + kw) * 64);
                    fb[kw][tm] = bf(tr_read(lds_b, o), tr_read(lds_b, o + RUN2 * 64));
                }
            // products largest first, the taps interleaved so that consecutive MFMAs never share an accumulator; one item of
            // the next tile behind every second group of KS MFMAs
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                const int ta = pr == 0 ? 0 : pr == 1 ? 0 : pr == 2 ? 1 : pr == 3 ? 0 : pr == 4 ? 1 : 2;
                const int tb = pr == 0 ? 0 : pr == 1 ? 1 : pr == 2 ? 0 : pr == 3 ? 2 : pr == 4 ? 1 : 0;
                if (w_live) {
#pragma unroll
                    for (int kw = 0; kw < KS; ++kw)
                        acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta], fb[kw][tb], acc[kw], 0, 0, 0);
                }
                const int g = 6 * ks + pr;                                  // MFMA group 0 .. 11 (CO32: 0 .. 5) of the tile
                if constexpr (CO32) {
                    if (g == 0) item(nxt, PD_I(0));
                    else if (g == 1) item(nxt, PD_I(1));
                    else if (g == 2) item(nxt, PD_I(2));
                    else if (g == 3) item(nxt, PD_I(3));
                } else {
                    if (g == 1) item(nxt, PD_I(0));
                    else if (g == 3) item(nxt, PD_I(1));
                    else if (g == 5) item(nxt, PD_I(2));
                    else if (g == 7) item(nxt, PD_I(3));
                    else if (g == 9) item(nxt, PD_I(4));
                }
            }
        }
        tile_next(); tile_begin();
        __syncthreads();                        // buffer BUF ^ 1 is complete, buffer BUF is free
    };
    for (int t = t_beg; t < t_end; t += 2) {
        tile(PD_U(0));
        if (t + 1 < t_end) tile(PD_U(1));
    }
#undef PD_I
#undef PD_U

    // ---- partial tile of this slice: C/D layout col = lane % 32 -> ci, row -> co: (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (w_live) {
        const int ci = ci0 + w_cib * 32 + (lane & 31);
        const long srow = CO32 ? 2L * s + w_ks : s;                      // CO32: the two pixel steps of a tile are two partial slices
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const long k = (long)(kh * KS + kw) * a.C + ci;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + w_cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                a.part[(srow * a.Co + co) * a.K + k] = dsign ? -acc[kw][r] : acc[kw][r];
            }
        }
    }
    if constexpr (BIAS) {
        if (do_bias) {                                                   // (uniform per workgroup; the planes are dead)
            float4* red = reinterpret_cast<float4*>(smem_all);
            red[tid] = bs4;                                              // thread -> quad tid & 15, pixel lane tid >> 4 (CO32: tid & 7, tid >> 3)
            __syncthreads();
            if (tid < BCO) {
                const int q4 = tid >> 2, e = tid & 3;
                float sum = 0.f;
                for (int j = 0; j < (CO32 ? 32 : 16); ++j) sum += reinterpret_cast<const float*>(&red[j * (CO32 ? 8 : 16) + q4])[e];
                if constexpr (CO32) { a.bpart[(2L * s) * a.Co + co0 + tid] = sum; a.bpart[(2L * s + 1) * a.Co + co0 + tid] = 0.f; }
                else a.bpart[(long)s * a.Co + co0 + tid] = sum;
            }
        }
    }
}

// tile width: the widest of 32 | 16 | 8 whose 32-pixel tile (1 | 2 | 4 rows) divides the output grid; 0: none
static int wgrad_halo_tw(int Ho, int Wo) {
    return Wo % 32 == 0 ? 32 : (Wo % 16 == 0 && Ho % 2 == 0) ? 16 : (Wo % 8 == 0 && Ho % 4 == 0) ? 8 : (Wo % 4 == 0 && Ho % 8 == 0) ? 4 : 0;
}
static bool wgrad_halo_co32(const WgradArgs& a) { return a.Co % 64 != 0; }
static bool wgrad_halo_eligible(const WgradArgs& a, bool vec) {
    if (!(vec && (a.mode == MODE_ZERO || (a.mode == MODE_REFLECT && a.pad < a.H && a.pad < a.W)) && a.stride == 1 && a.KH == a.KW &&
          (a.KH == 3 || a.KH == 5) && a.pad < a.KH && wgrad_halo_tw(a.Ho, a.Wo) != 0 && a.ldd % 4 == 0 &&
          (long)a.sN * 4 < 0x7fffffffL && (8L * a.Wo + 32) * a.ldd * 4 < 0x7fffffffL && a.Ho <= a.H + 2 * a.pad - a.KH + 1))
        return false;
    if (a.Co % 64 == 0) return a.C % 64 == 0 && (wgrad_halo_tw(a.Ho, a.Wo) != 4 || a.KH == 3);       // (8 x 4 tiles: 3x3 only)
    // 32 output channels per workgroup: 3x3 on 1 x 32 tiles, whole 32-channel input blocks
    return a.Co % 32 == 0 && a.C % 32 == 0 && a.KH == 3 && wgrad_halo_tw(a.Ho, a.Wo) == 32;
}

// Slices of the halo kernel for a [Cout][K] gradient: as many as keep every resident workgroup slot busy once (3x3: three per
// CU, 5x5: two).  pd_conv2d_wgrad_workspace knows Cout and K only: the larger of the two filter sizes K is a multiple of.
static int wgrad_halo_slices(int KS, int C, int Co) {
    const int per_slice_wgs = KS * ((C + 63) / 64) * (Co % 64 == 0 ? Co / 64 : Co / 32);
    const int S = (KS == 3 ? 768 : 512) / (per_slice_wgs > 0 ? per_slice_wgs : 1);
    return S < 1 ? 1 : S;
}
// partial rows the workspace must hold (a 32-channel workgroup writes two per slice)
static int wgrad_halo_slices_bound(int Co, int K) {
    int best = 0;
    if (Co % 64 != 0) return (Co % 32 == 0 && K % (9 * 32) == 0) ? 2 * wgrad_halo_slices(3, K / 9, Co) : 0;
    for (int KS : {3, 5})
        if (K % (KS * KS * 64) == 0) { const int S = wgrad_halo_slices(KS, K / (KS * KS), Co); best = S > best ? S : best; }
    return best;
}

// s_cap: partial rows the caller's workspace holds; returns the partial rows written
static int launch_wgrad_halo(WgradArgs a, int s_cap, hipStream_t st, bool bias) {
    WgradHaloArgs ha;
    const bool co32 = wgrad_halo_co32(a);
    ha.ncb = (a.C + 63) / 64;
    a.ctiles = co32 ? a.Co / 32 : a.Co / 64;
    const int tw = wgrad_halo_tw(a.Ho, a.Wo);
    ha.tiles_w = a.Wo / tw;
    ha.tiles_h = a.Ho / (wgh::NPX / tw);
    ha.ntiles_total = a.N * ha.tiles_h * ha.tiles_w;
    const int per_slice_wgs = a.KH * ha.ncb * a.ctiles;
    const int rows_per_slice = co32 ? 2 : 1;
    int S = wgrad_halo_slices(a.KH, a.C, a.Co);
    if (S > s_cap / rows_per_slice) S = s_cap / rows_per_slice;
    if (S < 1) S = 1;
    if (S > ha.ntiles_total) S = ha.ntiles_total;
    ha.tiles_per_slice = (ha.ntiles_total + S - 1) / S;
    S = (ha.ntiles_total + ha.tiles_per_slice - 1) / ha.tiles_per_slice;
    a.S = S;
    if (a.bpart) a.bpart = a.part + (size_t)S * rows_per_slice * a.Co * a.K;          // bias partials behind the weight tiles
    ha.g = a;
    const long nwg = (long)per_slice_wgs * S;
    const dim3 grid((unsigned)((nwg + 7) / 8 * 8)), block(NT);
#define PD_WGH(KSV, TWV, C32) do { if (bias) hipLaunchKernelGGL((conv_wgrad_halo_x3_kernel<KSV, TWV, true, C32>), grid, block, 0, st, ha); \
                                else hipLaunchKernelGGL((conv_wgrad_halo_x3_kernel<KSV, TWV, false, C32>), grid, block, 0, st, ha); } while (0)
    if (co32) PD_WGH(3, 32, true);
    else if (a.KH == 3) { if (tw == 32) PD_WGH(3, 32, false); else if (tw == 16) PD_WGH(3, 16, false); else if (tw == 8) PD_WGH(3, 8, false); else PD_WGH(3, 4, false); }
    else { if (tw == 32) PD_WGH(5, 32, false); else if (tw == 16) PD_WGH(5, 16, false); else PD_WGH(5, 8, false); }
#undef PD_WGH
    return S * rows_per_slice;
}
