// conv_wgrad_roll_x3_kernel -- the 3x3 weight gradient with ALL THREE filter rows in one workgroup and the input rows rolling
// through LDS (included by conv.hip behind conv_wgrad_halo.hpp, whose helpers it uses).
//
// conv_wgrad_halo_x3_kernel gives a workgroup ONE filter row kh: the three workgroups of a (channel block, slice) stage and
// split the same dY tile three times, and every input row three times (as the row above, at and below an output row) -- 5 staged
// items per 36 MFMAs of a wave, ~3.5 vector instructions per MFMA, which is what bounds it (matrix pipe 64 % busy).  Here a
// workgroup owns a COLUMN of tiles (1 x 32 or 2 x 16 pixels each, walked top to bottom), 64 output x 64 input channels and nine
// accumulator tiles per wave (kh x kw).  Output tile t needs the input row groups t - 1, t, t + 1 (a group = the 1 | 2 rows of
// a tile, 2 halo columns): they sit in a ring of four groups in LDS, so per tile ONE new group and one dY tile are staged and
// split -- 5 items per 108 MFMAs.  230 registers (two workgroups per CU, as the 5x5 kernel), 77 KB of LDS.
// Slices are row ranges of one tile column; a slice starts with three groups (7 % more staging at 43 tiles per slice).
namespace wgr {
using namespace wgh;
template <int TW> struct RGeo {
    static constexpr int TR = NPX / TW;                                  // rows per tile: 1 | 2
    static constexpr int HW = TW + 2;
    static constexpr int GPX = TR * HW;                                  // pixels of one input row group
    static constexpr int NIX = (GPX * 16 + NT - 1) / NT;                 // 16-byte X items per thread and group
    static constexpr unsigned DP_BYTES = 3 * 2 * NPX * 64;               // dY planes of one tile: [term][co block][pixel][32 ch x 2 B]
    static constexpr unsigned GS_BYTES = 3 * 2 * GPX * 64;               // X planes of one group: [term][ci block][pixel][32 ch x 2 B]
    static constexpr unsigned X_BASE = 2 * DP_BYTES;
    static constexpr unsigned LDS_BYTES = X_BASE + 4 * GS_BYTES;
};
}  // namespace wgr

struct WgradRollArgs {
    WgradArgs g;
    int ncb;                      // 64-channel input blocks
    int tiles_w, tiles_h;         // tiles per image row / column
    int spc, rps;                 // slices per tile column, tile rows per slice
};

template <int TW, bool BIAS>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_roll_x3_kernel(const WgradRollArgs ra) {
    using namespace wgr;
    using G = RGeo<TW>;
    constexpr int HW = G::HW, GPX = G::GPX, NIX = G::NIX, TR = G::TR;
    constexpr unsigned DP_BYTES = G::DP_BYTES, GS_BYTES = G::GS_BYTES, X_BASE = G::X_BASE;
    static_assert(NIX == 3 && 16 * (NIX - 1) < GPX, "X items per thread");
    const WgradArgs& a = ra.g;
    __shared__ __attribute__((aligned(16))) float smem_all[G::LDS_BYTES / 4];
    char* lds_c = reinterpret_cast<char*>(smem_all);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = ra.ncb * a.ctiles * a.S;
    int b = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (b >= nwg) return;
    const int cb = b % ra.ncb; b /= ra.ncb;
    const int cob = b % a.ctiles;
    const int s = b / a.ctiles;
    const int co0 = cob * 64, ci0 = cb * 64;
    const int col = s / ra.spc, part = s - col * ra.spc;
    const int n = col / ra.tiles_w, tx = col - n * ra.tiles_w;
    const int ty0 = part * ra.rps, ty1 = min(ty0 + ra.rps, ra.tiles_h);           // tile rows of this slice (never empty)
    const int ox0 = tx * TW;

    // ---- items of this thread: a tile is 2 dY items, an input row group 3 X items of 16 bytes
    const int quad = tid & 15;                                          // 4 channels: block quad >> 3, unit quad & 7
    unsigned d_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = (tid >> 4) + 16 * i, r = p / TW, c = p % TW;      // pixel p of the TR x TW tile
        d_voff[i] = (unsigned)((r * a.Wo + ox0 + c) * (int)a.ldd + co0 + 4 * quad) * 4u;
    }
    const unsigned d_dst = (unsigned)((((quad >> 3) * NPX + (tid >> 4)) * 64) + (quad & 7) * 8);                       // + buf, term, 1024 i
    int x_hr[NIX], x_hx[NIX];
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        const int p = (tid >> 4) + 16 * i;                              // pixel p of the TR x HW strip
        x_hr[i] = p / HW; x_hx[i] = p - x_hr[i] * HW;
    }
    const bool x_last_ok = (tid >> 4) + 16 * (NIX - 1) < GPX;
    const unsigned x_dst = X_BASE + (unsigned)((((quad >> 3) * GPX + (tid >> 4)) * 64) + (quad & 7) * 8);              // + ring slot, term, 1024 i
    const unsigned x_cq = (unsigned)(ci0 + 4 * quad) * 4u;

    const long img_px = (long)a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(a.dy + (long)n * img_px * a.ldd, (unsigned)(img_px * a.ldd * 4));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)n * a.sN, (unsigned)((long)a.sN * 4));
    const unsigned d_row_bytes = (unsigned)((long)TR * a.Wo * a.ldd * 4);                // dY bytes per tile row
    // what the next load_d / load_x fetch: dY tile l_t (dead from ty1 on), input row group l_g (dead beyond ty1: never read)
    int l_t = ty0, l_g = ty0 - 1;
    float4 dv[2], xv[NIX];
    auto ld_d = [&](int i) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rd, l_t < ty1 ? d_voff[i] : OOB, (unsigned)l_t * d_row_bytes, 0));
    };
    auto ld_x = [&](int i) {
        int iy = l_g * TR + x_hr[i], ix = ox0 - 1 + x_hx[i];            // (pad == 1: group g = input rows g TR ..; outside: zeros)
        if (a.mode == MODE_REFLECT) {                   // (uniform) ReflectionPad2d: the strip holds the mirrored pixels
            iy = iy < 0 ? -iy : iy; iy = iy >= a.H ? 2 * a.H - 2 - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= a.W ? 2 * a.W - 2 - ix : ix;
        }
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && (i < NIX - 1 || x_last_ok) && l_g <= ty1;
        return buf_ld4(rx, ok ? (unsigned)(iy * (int)a.sH + ix * (int)a.sW) * 4u + x_cq : OOB);
    };
    const unsigned dsign = (s & 1) ? 0x80008000u : 0u;                  // odd slices accumulate the negated gradient (conv_wgrad_halo.hpp)
    float4 bs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_bias = BIAS && a.bpart != nullptr && cb == 0;
    auto put3 = [&](const float4 v, unsigned off, unsigned plane_stride, unsigned sg) {
        uint2 h, m, l;
        x3::split2(v.x, v.y, h.x, m.x, l.x);
        x3::split2(v.z, v.w, h.y, m.y, l.y);
        *reinterpret_cast<uint2*>(lds_c + off) = uint2{h.x ^ sg, h.y ^ sg};
        *reinterpret_cast<uint2*>(lds_c + off + plane_stride) = uint2{m.x ^ sg, m.y ^ sg};
        *reinterpret_cast<uint2*>(lds_c + off + 2 * plane_stride) = uint2{l.x ^ sg, l.y ^ sg};
    };
    auto put_d = [&](int i, unsigned buf) {
        put3(dv[i], buf * DP_BYTES + d_dst + 1024u * i, 2 * NPX * 64, dsign);
        if constexpr (BIAS) { if (do_bias) { bs4.x += dv[i].x; bs4.y += dv[i].y; bs4.z += dv[i].z; bs4.w += dv[i].w; } }
    };
    auto put_x = [&](const float4 v, int i, unsigned slot_off) {
        if (i < NIX - 1 || x_last_ok) put3(v, slot_off + x_dst + 1024u * i, 2 * GPX * 64, 0u);
    };

    // ---- fragment addresses (as conv_wgrad_halo_x3_kernel): lane -> (half fh: pixels 8 fh .. of the step; group gq: channels 16 gq ..;
    // row fq, unit fp of the transposed 4 x 16 block)
    const int fh = lane >> 5, gq = (lane >> 4) & 1, fq = (lane & 15) >> 2, fp = lane & 3;
    const int w_cob = wave & 1, w_cib = wave >> 1;
    lds_char* const lds_a = (lds_char*)lds_c + (unsigned)(((w_cob * NPX + 8 * fh + fq) * 64) + 32 * gq + 8 * fp);
    // a 16-pixel step is half a row (TW = 32) or a row (TW = 16): its upper half-wave sits 8 columns on
    const unsigned lds_b0 = X_BASE + (unsigned)(((w_cib * GPX + fh * 8 + fq) * 64) + 32 * gq + 8 * fp);

    typedef float accv_t __attribute__((ext_vector_type(16)));
    accv_t acc[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kh][kw][r] = 0.f;
    auto bf = [](uint2 lo, uint2 hi) { return __builtin_bit_cast(x3::bf16x8, u32x4{lo.x, lo.y, hi.x, hi.y}); };

    // ---- prologue: groups ty0 - 1, ty0, ty0 + 1 -> ring slots 0, 1, 2; dY tile ty0 -> buffer 0; tile ty0 + 1 and group ty0 + 2
    // -> registers.  (Ring slot of group g: (g - ty0 + 1) & 3; dY buffer of tile t: (t - ty0) & 1.)
    {
        float4 xa[NIX], xb[NIX];
#pragma unroll
        for (int i = 0; i < 2; ++i) dv[i] = ld_d(i);
#pragma unroll
        for (int i = 0; i < NIX; ++i) xv[i] = ld_x(i);
        ++l_g;
#pragma unroll
        for (int i = 0; i < NIX; ++i) xa[i] = ld_x(i);
        ++l_g;
#pragma unroll
        for (int i = 0; i < NIX; ++i) xb[i] = ld_x(i);
        ++l_g; ++l_t;
#pragma unroll
        for (int i = 0; i < 2; ++i) { put_d(i, 0u); dv[i] = ld_d(i); }
#pragma unroll
        for (int i = 0; i < NIX; ++i) { put_x(xv[i], i, 0u); xv[i] = ld_x(i); }
#pragma unroll
        for (int i = 0; i < NIX; ++i) { put_x(xa[i], i, GS_BYTES); put_x(xb[i], i, 2 * GS_BYTES); }
        ++l_g; ++l_t;                                   // the items of the first tile refill from tile ty0 + 2 / group ty0 + 3
        __syncthreads();
    }

    // One tile: 2 steps x 3 filter rows x 3 taps x 6 MFMAs on dY buffer BUF and the ring slots (j, j + 1, j + 2) & 3; between them
    // the tile / group in the registers are split into buffer BUF ^ 1 / slot (j + 3) & 3, and the registers refilled.
    int j = 0;                                          // tiles done (uniform)
    auto tile = [&](auto buf_tag) {
        constexpr unsigned BUF = decltype(buf_tag)::value;
        const unsigned slot_w = (unsigned)((j + 3) & 3) * GS_BYTES;
        lds_char* bb[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bb[d] = (lds_char*)lds_c + (lds_b0 + (unsigned)((j + d) & 3) * GS_BYTES);
        int it = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x3::bf16x8 fa[3];
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
                fa[tm] = bf(tr_read(lds_a, BUF * DP_BYTES + (unsigned)(tm * 2 * NPX * 64 + ks * 16 * 64)),
                            tr_read(lds_a, BUF * DP_BYTES + (unsigned)(tm * 2 * NPX * 64 + ks * 16 * 64 + 4 * 64)));
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // input row of (step ks, filter row kh): TW = 32: the group kh - 1 of the ring, columns 16 ks ..; TW = 16: row
                // e = ks + kh - 1 relative to the tile's first row: group -1 | 0 | +1, row e & 1 of it
                const int e = ks + kh - 1;
                const int d = TW == 32 ? kh : (e < 0 ? 0 : e > 1 ? 2 : 1);
                const int row = TW == 32 ? 0 : (e & 1), col0 = TW == 32 ? 16 * ks : 0;
                x3::bf16x8 fb[3][3];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm) {
                        const unsigned o = (unsigned)(tm * 2 * GPX * 64 + (row * HW + col0 + kw) * 64);
                        fb[kw][tm] = bf(tr_read(bb[d], o), tr_read(bb[d], o + 4 * 64));
                    }
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    const int ta = pr == 0 ? 0 : pr == 1 ? 0 : pr == 2 ? 1 : pr == 3 ? 0 : pr == 4 ? 1 : 2;
                    const int tb = pr == 0 ? 0 : pr == 1 ? 1 : pr == 2 ? 0 : pr == 3 ? 2 : pr == 4 ? 1 : 0;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc[kh][kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta], fb[kw][tb], acc[kh][kw], 0, 0, 0);
                    if (pr == 2) {                      // one item of the next tile / group behind the first half of every (step, filter row)
                        if (it < 2) { put_d(it, BUF ^ 1u); dv[it] = ld_d(it); }
                        else if (it < 2 + NIX) { put_x(xv[it - 2], it - 2, slot_w); xv[it - 2] = ld_x(it - 2); }
                        ++it;
                    }
                }
            }
        }
        ++l_t; ++l_g; ++j;
        __syncthreads();                        // buffer BUF ^ 1 and the new ring slot are complete, buffer BUF and the oldest slot free
    };
    for (int t = ty0; t < ty1; t += 2) {
        tile(std::integral_constant<unsigned, 0>{});
        if (t + 1 < ty1) tile(std::integral_constant<unsigned, 1>{});
    }

    // ---- partial tile of this slice: C/D layout col = lane % 32 -> ci, row -> co: (r&3) + 8*(r>>2) + 4*(lane>>5)
    {
        const int ci = ci0 + w_cib * 32 + (lane & 31);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const long k = (long)(kh * 3 + kw) * a.C + ci;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + w_cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    a.part[((long)s * a.Co + co) * a.K + k] = dsign ? -acc[kh][kw][r] : acc[kh][kw][r];
                }
            }
    }
    if constexpr (BIAS) {
        if (do_bias) {                                                   // (uniform per workgroup; the planes are dead)
            float4* red = reinterpret_cast<float4*>(smem_all);
            red[tid] = bs4;                                              // thread -> quad tid & 15, pixel lane tid >> 4
            __syncthreads();
            if (tid < 64) {
                const int q4 = tid >> 2, e = tid & 3;
                float sum = 0.f;
                for (int jj = 0; jj < 16; ++jj) sum += reinterpret_cast<const float*>(&red[jj * 16 + q4])[e];
                a.bpart[(long)s * a.Co + co0 + tid] = sum;
            }
        }
    }
}

// Plan: slices = (tile column) x (row range).  As many slices per column as fill the 512 resident workgroup slots once.
struct WgradRollPlan { int tw, tiles_w, tiles_h, spc, rps, S; };
static bool wgrad_roll_plan(const WgradArgs& a, int s_cap, WgradRollPlan& p) {
    p.tw = a.Wo % 32 == 0 ? 32 : (a.Wo % 16 == 0 && a.Ho % 2 == 0) ? 16 : 0;
    if (p.tw == 0) return false;
    p.tiles_w = a.Wo / p.tw;
    p.tiles_h = a.Ho / (32 / p.tw);
    const int per_slice_wgs = (a.C / 64) * (a.Co / 64);
    const long ncols = (long)a.N * p.tiles_w;
    long want = 512 / (per_slice_wgs > 0 ? per_slice_wgs : 1);
    if (want > s_cap) want = s_cap;
    long spc = want / ncols;
    if (spc < 1) spc = 1;
    if (spc > p.tiles_h) spc = p.tiles_h;
    p.rps = (int)((p.tiles_h + spc - 1) / spc);
    p.spc = (p.tiles_h + p.rps - 1) / p.rps;
    p.S = (int)(ncols * p.spc);
    // worth it when a slice is long enough to amortise its three-group start and the slices fill the chip about once
    return p.S <= s_cap && p.rps >= 12 && (long)p.S * per_slice_wgs >= 384;
}
static bool wgrad_roll_eligible(const WgradArgs& a, bool vec, int s_cap, WgradRollPlan& p) {
    return wgrad_halo_eligible(a, vec) && a.KH == 3 && a.Co % 64 == 0 && a.C % 64 == 0 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W &&
           (long)a.Ho * a.Wo * a.ldd * 4 < 0x7fffffffL && wgrad_roll_plan(a, s_cap, p);
}
// partial rows the workspace must hold for a [Cout][K] gradient on this kernel (K = 9 C): at most 512 / (blocks) + one per column...
// bounded by 640 / blocks (the plan never exceeds its target by more than the rounding of slices per column)
static int wgrad_roll_slices_bound(int Co, int K) {
    if (Co % 64 != 0 || K % (9 * 64) != 0) return 0;
    const int per = (K / 9 / 64) * (Co / 64);
    return 512 / (per > 0 ? per : 1);
}

static int launch_wgrad_roll(WgradArgs a, const WgradRollPlan& p, hipStream_t st, bool bias) {
    WgradRollArgs ra;
    ra.ncb = a.C / 64;
    a.ctiles = a.Co / 64;
    ra.tiles_w = p.tiles_w; ra.tiles_h = p.tiles_h; ra.spc = p.spc; ra.rps = p.rps;
    a.S = p.S;
    if (a.bpart) a.bpart = a.part + (size_t)p.S * a.Co * a.K;          // bias partials behind the weight tiles
    ra.g = a;
    const long nwg = (long)ra.ncb * a.ctiles * p.S;
    const dim3 grid((unsigned)((nwg + 7) / 8 * 8)), block(NT);
#define PD_WGR(TWV) do { if (bias) hipLaunchKernelGGL((conv_wgrad_roll_x3_kernel<TWV, true>), grid, block, 0, st, ra); \
                         else hipLaunchKernelGGL((conv_wgrad_roll_x3_kernel<TWV, false>), grid, block, 0, st, ra); } while (0)
    if (p.tw == 32) PD_WGR(32); else PD_WGR(16);
#undef PD_WGR
    return p.S;
}
