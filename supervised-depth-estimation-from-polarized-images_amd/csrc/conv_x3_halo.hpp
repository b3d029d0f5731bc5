// conv_halo_x3_kernel -- the bf16-split convolution with the ACTIVATION split out of the MFMA loop (included by conv.hip).
//
// conv_igemm_x3_kernel gathers the im2col rows of every filter tap from L2 (LDS-DMA), and every wave splits its 64 x 16
// activation block into bf16 hi / mid / lo terms once PER TAP: the same input element is fetched and split 9 (3x3) or 25
// (5x5) times per tile -- 3.75 vector instructions and ~680 B of L2 -> LDS traffic per MFMA.  On a chip that runs this
// kernel at its power limit (1.29 GHz) both are energy that buys no products.  Here a workgroup owns an
// 8 x 32 OUTPUT tile and, per 16-channel group, stages the (8 + K - 1) x (32 + K - 1) input HALO of that tile ONCE:
//   * global -> registers (16-byte bounds-checked loads, issued one channel group ahead, in flight under a whole group
//     of MFMAs), split once in registers, written as bf16 planes to LDS: [term][k half][halo pixel] x 16 bytes;
//   * the A fragment of tap (kh, kw) for output row y is the plane row y + kh shifted by kw pixels: 32 consecutive
//     16-byte slots per half-wave -- one conflict-free ds_read_b128 per term at  lane base + tap offset  (one v_add per
//     tap), no gather, no masks, no per-tap split: the loop is 24 MFMAs, 12 ds_read_b128, the weight chunk's split
//     (0.75 vector instructions per MFMA instead of 3.75) and one barrier;
//   * an input element crosses L2 -> CU 1.33x (3x3) / 1.69x (5x5) per tile instead of 9x / 25x.
// The weight chunk (64 x 16 per tap) is staged through REGISTERS as well: every thread splits its 16 bytes of chunk q + 1
// between the first MFMAs of chunk q, requests chunk q + 2 into the same registers behind them (16 MFMAs x the waves sharing
// the SIMD to land) and writes the three bf16 planes -- no fp32 staging ring in LDS, no LDS-DMA, no hand-counted vmcnt: 45 KB (3x3) / 52.5 KB (5x5) of LDS, both
// filter sizes fit a CU three times.  The halo of the next channel group is
// split and written between the MFMAs of the current group's LAST tap, behind a barrier that follows that tap's fragment
// reads (the planes are free from there on): no serial split phase.  Epilogue (bias, ELU, addend, BatchNorm partial sums, LDS transposition
// to full 128-byte lines) as there.  Zero padding or the stride-1 data gradient, K = 3 | 5, C % 16 == 0, Cout % 64 == 0,
// the output grid a whole number of 8 x 32, 16 x 16 or 32 x 8 tiles.
// Two more shapes ride on the same body (round 4):
//   * NCB = 1: a workgroup owns 32 output channels instead of 64 (wave tile 64 pixels x 32 channels, 12 MFMAs per chunk) --
//     the decoder's 32-channel layers (96 -> 32 @256x320 forward, its data gradient as three 32-column tiles);
//   * ROWWIN: a KH x KW filter over C channels read as a KH x 1 filter over "KW * C channels" whose pixels OVERLAP (pixel
//     stride C): the KW * C floats under a filter row are contiguous in NHWC memory whatever C is, so the 4 x 4 space-to-depth
//     stems (C = 36 / 12 / 8) get whole 16-channel groups (144 / 48 / 32 floats per filter row) where the per-tap form would
//     leave their last group 1/4 .. 3/4 empty.  Only the first / last columns of the IMAGE see a window that leaves its row:
//     per halo item two 2-bit counts (pixels missing on the left / right) mask those quads, in the border tiles only.
namespace x3h {
constexpr int CK = 16;
// TW = 32 | 16 | 8: the 256-pixel output tile is 8 x 32, 16 x 16 or 32 x 8; an MFMA row block (32 pixels) is then one row,
// two rows of 16 or four rows of 8 -- the widest tile that divides the output grid
template <int KH, int KW, int TW, int NCB> struct Geo {
    static constexpr int TR = 256 / TW, BR = 32 / TW;
    static constexpr int HW = TW + KW - 1, HH = TR + KH - 1, HP = HW * HH;
    static constexpr unsigned BP_BYTES = 32 * NCB * CK * 2;          // one weight plane: (32 NCB) x 16 bf16
    static constexpr int NI = (HP * 4 + NT - 1) / NT;                 // 16-byte halo items per thread and channel group
    static constexpr unsigned AP_BYTES = 3 * 2 * HP * 16;             // A planes: [term][half][halo pixel] x 16 B
    static constexpr unsigned BP_BASE = AP_BYTES;                     // weight planes: [buffer][term] x 64 x 16 bf16
    static constexpr unsigned LDS_BYTES = BP_BASE + 2 * 3 * BP_BYTES;
};
}  // namespace x3h

template <int MODE, int KH, int KW, int TW, int NCB, bool ROWWIN>
__global__ __launch_bounds__(NT, 3) void conv_halo_x3_kernel(const ConvArgs a) {
    using namespace x3h;
    using G = Geo<KH, KW, TW, NCB>;
    constexpr int HW = G::HW, HP = G::HP, NI = G::NI, T = KH * KW, TR = G::TR, BR = G::BR, BNW = 32 * NCB;
    constexpr unsigned BP_BASE = G::BP_BASE, BP_BYTES = G::BP_BYTES;
    static_assert(!ROWWIN || (KW == 1 && MODE == MODE_ZERO), "row windows: zero padding, the filter row folded into the channels");
    static_assert(G::LDS_BYTES >= 4 * 64 * 32 * 4 + 4 * 64 * 2 * 4, "the epilogue's transposition tiles reuse the ring");
    __shared__ __attribute__((aligned(16))) float smem_all[G::LDS_BYTES / 4];

    // ---- tile: XCD-aware order as everywhere (consecutive logical tiles -- neighbours in x, then y -- share an L2)
    const int tiles_w = a.Wo / TW, tiles_h = a.Ho / TR;
    const int ntile_m = a.N * tiles_h * tiles_w;
    const int nblk = ntile_m * a.ntiles;
    const int per_xcd = (int)gridDim.x >> 3;
    const int logical = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (logical >= nblk) return;
    const int mt = a.nmajor ? logical % ntile_m : logical / a.ntiles, nt = a.nmajor ? logical / ntile_m : logical - mt * a.ntiles;
    const int n0 = nt * BNW;
    const int img = mt / (tiles_h * tiles_w);
    const int trem = mt - img * (tiles_h * tiles_w);
    const int ty = trem / tiles_w, tx = trem - ty * tiles_w;
    const int oy0 = ty * TR, ox0 = tx * TW;
    // The bf16 MFMA truncates where it aligns its addends (about -2^-32 of the largest per instruction, toward -infinity whatever
    // the signs): per output that is below one fp32 rounding, but it has ONE sign, and sums over many outputs -- BatchNorm
    // statistics, weight gradients -- collect it.  Every other tile therefore computes the NEGATED convolution (weight planes with
    // flipped sign bits, 6 v_xor per chunk) and negates back in the epilogue: its bias points the other way and the sums cancel.
    const unsigned wsign = (mt & 1) ? 0x80008000u : 0u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int padh = MODE == MODE_TRANSPOSED ? KH - 1 - a.pad : a.pad, padw = MODE == MODE_TRANSPOSED ? KW - 1 - a.pad_w : a.pad_w;

    // The input descriptor starts at the halo's first pixel (which may lie above / left of the image: only addresses of valid
    // items are ever formed) and ends 4 MB - 64 B behind it or at the image's end: a halo item's offset is then a 16-bit count
    // of 64-byte units, two items per register, and the count 0xffff -- an item outside the image (zero padding) or beyond
    // the halo -- is out of the descriptor's range by construction: the hardware returns zeros.
    const int by = oy0 - padh, bx = ox0 - padw;
    const long base_f = (long)by * a.sH + (long)bx * a.sW;
    const long left_bytes = ((long)a.sN - base_f) * 4;
    // (row windows: 16-byte units, 1 MB - 16 B)
    constexpr long XSPAN = ROWWIN ? 0xffff0L : 0x3fffc0L;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)img * a.sN + base_f, (unsigned)(left_bytes < XSPAN ? left_bytes : XSPAN));
    const __amdgpu_buffer_rsrc_t rw_ = make_rsrc(a.w, a.w_bytes);

    // ---- halo items of this thread: item = pixel * 4 + channel quad
    // (item i of a thread is pixel (tid >> 2) + 64 i, quad tid & 3: its plane slot is hdst0 + 1024 i, and only the LAST item
    //  of a thread can lie beyond the halo)
    unsigned hpk[(NI + 1) / 2];
    unsigned hmask = 0;                          // row windows: 4 bits per item -- pixels of the window left / right of the image row
    const int hquad = tid & 3;
    const unsigned hdst0 = (unsigned)(((hquad >> 1) * HP + (tid >> 2)) * 16 + (hquad & 1) * 8);        // + term * 2 * HP * 16 + 1024 * i
    const bool hlast_ok = (tid >> 2) + 64 * (NI - 1) < HP;
    static_assert(64 * (NI - 1) < HP, "only the last halo item of a thread may be empty");
    const int sH16 = (int)a.sH >> (ROWWIN ? 2 : 4), sW16 = (int)a.sW >> (ROWWIN ? 2 : 4);   // (eligibility: both strides are whole units)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = (tid >> 2) + 64 * i;
        const int hy = p / HW, hx = p - hy * HW;
        int iy = by + hy, ix = bx + hx;
        if constexpr (MODE == MODE_REFLECT) {        // ReflectionPad2d: the halo holds the mirrored pixels, every tap is in range
            iy = iy < 0 ? -iy : iy; iy = iy >= a.H ? 2 * a.H - 2 - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= a.W ? 2 * a.W - 2 - ix : ix;
        }
        bool ok = p < HP && (unsigned)iy < (unsigned)a.H;
        if constexpr (ROWWIN) {                      // the window [ix, ix + a.KW) may leave the row on either side
            const int lo = ix < 0 ? -ix : 0, hi = ix + a.KW > a.W ? ix + a.KW - a.W : 0;
            ok = ok && lo < a.KW && hi < a.KW;
            hmask |= (unsigned)((lo & 3) | ((hi & 3) << 2)) << (4 * i);
        } else {
            ok = ok && (unsigned)ix < (unsigned)a.W;
        }
        const unsigned u = ok ? (unsigned)((iy - by) * sH16 + (ix - bx) * sW16) : 0xffffu;
        if (i & 1) hpk[i >> 1] |= u << 16; else hpk[i >> 1] = u;
    }
    const unsigned hq16 = 16u * (unsigned)hquad;
    const bool border_tile = ROWWIN && (tx == 0 || tx == tiles_w - 1);      // (uniform)
    // weights: NCB = 2: thread -> row tid / 4 of the 64 x 16 chunk, 16-byte slot tid % 4; NCB = 1: row tid / 8 of the 32 x 16
    // chunk, 8-byte slot tid % 8 (every thread splits its share: no idle waves, no branches between the MFMAs)
    const int srow = NCB == 2 ? tid >> 2 : tid >> 3, sls = NCB == 2 ? tid & 3 : tid & 7;
    const unsigned vb = (unsigned)((n0 + srow) * a.K) * 4u + (NCB == 2 ? 16u : 8u) * (unsigned)sls;

    const int ngroups = (ROWWIN ? a.KW * a.C : a.C) / CK;
    const unsigned tap_bytes = (unsigned)(ROWWIN ? a.KW * a.C : a.C) * 4u;
    const int nchunks = ngroups * T;
    int s_qb = 0, sb_tap = 0;
    unsigned s_boff = 0, sb_c4 = 0;
    float4 wq;                                                    // this thread's 16 bytes of the chunk in flight
    // NCB = 1: a chunk is 12 MFMAs per wave -- half a chunk is less than an L2 round trip when two or three waves share a SIMD --
    // so the 8-byte requests run TWO chunks ahead, in two register sets by the parity of the chunk they are for
    float2 wq2[2];
    auto load_b = [&](auto set_tag) {                             // chunk s_qb (past the end: chunk 0 again, never used)
        if constexpr (NCB == 2) {
            wq = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw_, vb, s_qb < nchunks ? s_boff : 0u, 0));
        } else {
            wq2[decltype(set_tag)::value] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rw_, vb, s_qb < nchunks ? s_boff : 0u, 0));
        }
        ++s_qb;
        s_boff += tap_bytes;                                      // next tap of the same channel group
        if (++sb_tap == T) { sb_tap = 0; sb_c4 += CK * 4; s_boff = sb_c4; }
    };

    // ---- fragment addresses
    const int frow = lane & 31, fh = lane >> 5;
    // row block 2 wave + i of the tile = BR output rows of TW pixels: lane -> (row frow / TW, column frow % TW) of the block
    const unsigned fa_base = (unsigned)((fh * HP + (2 * wave * BR + frow / TW) * HW + frow % TW) * 16);   // + tap offset + (term * 2 * HP + i * BR * HW) * 16
    unsigned fb_off = BP_BASE + (unsigned)frow * (CK * 2) + 16u * (fh ^ ((frow >> 3) & 1));
    asm volatile("" : "+v"(fb_off));
    const unsigned sp_off = NCB == 2 ? BP_BASE + (unsigned)srow * (CK * 2) + 16u * ((sls >> 1) ^ ((srow >> 3) & 1)) + 8u * (sls & 1)
                                     : BP_BASE + (unsigned)srow * (CK * 2) + 16u * ((sls >> 2) ^ ((srow >> 3) & 1)) + 4u * (sls & 3);
    char* lds_c = reinterpret_cast<char*>(smem_all);

    typedef float accv_t __attribute__((ext_vector_type(16)));
    accv_t acc[2][NCB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto bf = [](u32x4 v) { return __builtin_bit_cast(x3::bf16x8, v); };

    // ---- halo pipeline: loads of channel group g + 1 fly under the MFMAs of group g
    float4 hv[NI];
    auto halo_load = [&](int g) {
        if constexpr (ROWWIN) {
            if (border_tile) {                  // first / last tile of an image row: quads of a window outside the row read zeros
                const int f = g * CK + 4 * hquad;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const unsigned u = (i & 1) ? hpk[i >> 1] >> 16 : hpk[i >> 1] & 0xffffu;
                    const int lo = (int)((hmask >> (4 * i)) & 3u) * a.C, hi = (a.KW - (int)((hmask >> (4 * i + 2)) & 3u)) * a.C;
                    hv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        rx, (f >= lo && f < hi) ? (u << 4) + hq16 : OOB, g * (CK * 4), 0));
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
            hv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                rx, (((i & 1) ? hpk[i >> 1] >> 16 : hpk[i >> 1] & 0xffffu) << (ROWWIN ? 4 : 6)) + hq16, g * (CK * 4), 0));
    };

    auto split_b0 = [&]() {                     // chunk 0 -> planes 0 (prologue only; the loop threads it)
        const float4 w4 = NCB == 2 ? wq : make_float4(wq2[0].x, wq2[0].y, 0.f, 0.f);
        uint2 h, m, l;
        x3::split2(w4.x, w4.y, h.x, m.x, l.x);
        if constexpr (NCB == 2) {
            x3::split2(w4.z, w4.w, h.y, m.y, l.y);
            *reinterpret_cast<uint2*>(lds_c + sp_off + 0 * BP_BYTES) = uint2{h.x ^ wsign, h.y ^ wsign};
            *reinterpret_cast<uint2*>(lds_c + sp_off + 1 * BP_BYTES) = uint2{m.x ^ wsign, m.y ^ wsign};
            *reinterpret_cast<uint2*>(lds_c + sp_off + 2 * BP_BYTES) = uint2{l.x ^ wsign, l.y ^ wsign};
        } else {
            *reinterpret_cast<unsigned*>(lds_c + sp_off + 0 * BP_BYTES) = h.x ^ wsign;
            *reinterpret_cast<unsigned*>(lds_c + sp_off + 1 * BP_BYTES) = m.x ^ wsign;
            *reinterpret_cast<unsigned*>(lds_c + sp_off + 2 * BP_BYTES) = l.x ^ wsign;
        }
    };
    // one halo item: registers -> three plane slots (every element split once per tile)
    auto halo_item = [&](auto i_tag) {
        constexpr int I = decltype(i_tag)::value;
        if constexpr (I < NI) {
            uint2 h, m, l;
            x3::split2(hv[I].x, hv[I].y, h.x, m.x, l.x);
            x3::split2(hv[I].z, hv[I].w, h.y, m.y, l.y);
            if (I < NI - 1 || hlast_ok) {
                *reinterpret_cast<uint2*>(lds_c + hdst0 + (1024u * I + 0 * (2 * HP * 16))) = h;
                *reinterpret_cast<uint2*>(lds_c + hdst0 + (1024u * I + 1 * (2 * HP * 16))) = m;
                *reinterpret_cast<uint2*>(lds_c + hdst0 + (1024u * I + 2 * (2 * HP * 16))) = l;
            }
        }
    };
    auto halo_split = [&]() {                                    // prologue: all items at once
#define PD_I(n) std::integral_constant<int, n>{}
        halo_item(PD_I(0)); halo_item(PD_I(1)); halo_item(PD_I(2)); halo_item(PD_I(3)); halo_item(PD_I(4)); halo_item(PD_I(5)); halo_item(PD_I(6));
#undef PD_I
    };

    // scalar tap state of the chunk being multiplied
    int s_kh = 0, s_kw = 0, s_g = 0;

    // One chunk = one tap of one channel group: 24 (NCB = 1: 12) MFMAs; the next chunk's weights are split between them.  LAST (the last
    // tap of a group that has a successor): the next group's halo is split and written behind the MFMAs as well.
    // (LAST is a run-time, wave-uniform condition: ONE copy of the MFMA stream per weight-plane parity, so that the
    //  accumulators never move between register sets; the halo items sit behind scalar branches.)
    auto chunk = [&](auto buf_tag) {
        constexpr unsigned BUF = decltype(buf_tag)::value, NXT = BUF ^ 1;
        const bool LAST = s_kh == KH - 1 && s_kw == KW - 1 && s_g + 1 < ngroups;
        const bool halo_issue = (s_kh | s_kw) == 0 && s_g + 1 < ngroups;    // (uniform) first tap of a group that has a successor
        // next chunk's weights (fp32), requested during the previous chunk: the ONLY vmcnt wait of the loop is taken here, in
        // front of the halo requests (vmcnt counts in order: behind them it would wait for the halo as well; and a branch in
        // the middle of the chunk lets LLVM sink the whole split below it)
        float ws[4];
        if constexpr (NCB == 2) {
            asm volatile("" : "+v"(wq.x), "+v"(wq.y), "+v"(wq.z), "+v"(wq.w));
            ws[0] = wq.x; ws[1] = wq.y; ws[2] = wq.z; ws[3] = wq.w;
        } else {
            asm volatile("" : "+v"(wq2[NXT].x), "+v"(wq2[NXT].y));
            ws[0] = wq2[NXT].x; ws[1] = wq2[NXT].y; ws[2] = 0.f; ws[3] = 0.f;
        }
        if (halo_issue) halo_load(s_g + 1);
        const int ey = MODE == MODE_TRANSPOSED ? KH - 1 - s_kh : s_kh, ex = MODE == MODE_TRANSPOSED ? KW - 1 - s_kw : s_kw;
        const unsigned fa = fa_base + (unsigned)((ey * HW + ex) * 16);
        u32x4 av[2][3], fb[NCB][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                av[i][t] = *reinterpret_cast<const u32x4*>(lds_c + fa + (unsigned)((t * 2 * HP + i * BR * HW) * 16));
        // (the weights' lo planes are read behind the last use of the activations' lo fragments, into their registers)
        auto load_fb = [&](int j, int t) { fb[j][t] = *reinterpret_cast<const u32x4*>(lds_c + fb_off + ((BUF * 3 + t) * BP_BYTES + (unsigned)j * 32 * CK * 2)); };
        load_fb(0, 0); load_fb(0, 1);
        if constexpr (NCB == 2) { load_fb(1, 0); load_fb(1, 1); }
        if (LAST) __syncthreads();                  // every wave holds its fragments of this group's last tap: the planes are free
        x3::Terms tw;
        // MFMA N (0..11) of row block I: products hi*hi, hi*mid, lo*hi, mid*hi, mid*mid, hi*lo; column block N % 2
        auto mm = [&](auto i_tag, auto n_tag) {
            constexpr int I = decltype(i_tag)::value, N = decltype(n_tag)::value, TT = N / NCB, J = N % NCB;
            constexpr int TA = TT < 2 ? 0 : TT == 2 ? 2 : TT < 5 ? 1 : 0, TB = TT == 0 ? 0 : TT == 1 ? 1 : TT == 2 ? 0 : TT == 3 ? 0 : TT == 4 ? 1 : 2;
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(av[I][TA]), bf(fb[J][TB]), acc[I][J], 0, 0, 0);
        };
#define PD_I(n) std::integral_constant<int, n>{}
#define PD_SB __builtin_amdgcn_sched_barrier(0);
        if constexpr (NCB == 2) {
        PD_SB
        mm(PD_I(0), PD_I(0)); x3::sp_h<0, true>(ws, tw); PD_SB
        mm(PD_I(0), PD_I(1)); x3::sp_h<1, true>(ws, tw); PD_SB
        mm(PD_I(1), PD_I(0)); x3::sp_m<0, true>(ws, tw); PD_SB
        mm(PD_I(1), PD_I(1)); PD_SB
        mm(PD_I(0), PD_I(2)); x3::sp_m<1, true>(ws, tw); PD_SB
        mm(PD_I(0), PD_I(3)); PD_SB
        mm(PD_I(1), PD_I(2)); x3::sp_l<0, true>(tw); PD_SB
        mm(PD_I(1), PD_I(3)); x3::sp_l<1, true>(tw); PD_SB
        load_b(std::integral_constant<int, 0>{});   // chunk q + 2's weights: their registers are free, two thirds of a chunk (x 3 waves per SIMD) to land
        mm(PD_I(0), PD_I(4)); PD_SB
        mm(PD_I(0), PD_I(5));
        *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 0) * BP_BYTES) = uint2{tw.h[0] ^ wsign, tw.h[1] ^ wsign};
        *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 1) * BP_BYTES) = uint2{tw.m[0] ^ wsign, tw.m[1] ^ wsign};
        *reinterpret_cast<uint2*>(lds_c + sp_off + (NXT * 3 + 2) * BP_BYTES) = uint2{tw.l[0] ^ wsign, tw.l[1] ^ wsign};
        PD_SB
        // LAST: one halo item (18 vector instructions, 3 ds_write_b64) per two MFMAs
        mm(PD_I(1), PD_I(4)); mm(PD_I(1), PD_I(5)); load_fb(0, 2); load_fb(1, 2); if (LAST) halo_item(PD_I(0)); PD_SB
        mm(PD_I(0), PD_I(6)); mm(PD_I(0), PD_I(7)); if (LAST) halo_item(PD_I(1)); PD_SB
        mm(PD_I(1), PD_I(6)); mm(PD_I(1), PD_I(7)); if (LAST) halo_item(PD_I(2)); PD_SB
        mm(PD_I(0), PD_I(8)); mm(PD_I(0), PD_I(9)); if (LAST) halo_item(PD_I(3)); PD_SB
        mm(PD_I(1), PD_I(8)); mm(PD_I(1), PD_I(9)); if (LAST) halo_item(PD_I(4)); PD_SB
        mm(PD_I(0), PD_I(10)); mm(PD_I(0), PD_I(11)); if (LAST) halo_item(PD_I(5)); PD_SB
        mm(PD_I(1), PD_I(10)); mm(PD_I(1), PD_I(11)); if (LAST) halo_item(PD_I(6));
        } else {                                    // 32 output channels: 12 MFMAs, half the weight chunk per thread
        PD_SB
        mm(PD_I(0), PD_I(0)); x3::sp_h<0, true>(ws, tw); PD_SB
        mm(PD_I(1), PD_I(0)); PD_SB
        mm(PD_I(0), PD_I(1)); x3::sp_m<0, true>(ws, tw); PD_SB
        mm(PD_I(1), PD_I(1)); PD_SB
        mm(PD_I(0), PD_I(2)); x3::sp_l<0, true>(tw); PD_SB
        mm(PD_I(1), PD_I(2)); PD_SB
        load_b(std::integral_constant<int, (int)NXT>{});          // chunk q + 3 into the set chunk q + 1's weights have just left
        mm(PD_I(0), PD_I(3));
        *reinterpret_cast<unsigned*>(lds_c + sp_off + (NXT * 3 + 0) * BP_BYTES) = tw.h[0] ^ wsign;
        *reinterpret_cast<unsigned*>(lds_c + sp_off + (NXT * 3 + 1) * BP_BYTES) = tw.m[0] ^ wsign;
        *reinterpret_cast<unsigned*>(lds_c + sp_off + (NXT * 3 + 2) * BP_BYTES) = tw.l[0] ^ wsign;
        PD_SB
        mm(PD_I(1), PD_I(3)); load_fb(0, 2); if (LAST) halo_item(PD_I(0)); PD_SB
        mm(PD_I(0), PD_I(4)); if (LAST) { halo_item(PD_I(1)); halo_item(PD_I(2)); } PD_SB
        mm(PD_I(1), PD_I(4)); if (LAST) { halo_item(PD_I(3)); halo_item(PD_I(4)); } PD_SB
        mm(PD_I(0), PD_I(5)); if (LAST) halo_item(PD_I(5)); PD_SB
        mm(PD_I(1), PD_I(5)); if (LAST) halo_item(PD_I(6));
        }
        static_assert(NI <= 7, "halo items per thread");
#undef PD_SB
#undef PD_I
        if (++s_kw == KW) { s_kw = 0; if (++s_kh == KH) { s_kh = 0; ++s_g; } }
        __syncthreads();                            // the next chunk's weight planes (and, behind LAST, the next halo) are written
    };

    // ---- prologue: halo of group 0, weights of chunks 0 and 1 (register sets 0, 1); planes of the halo and of chunk 0
    {
        halo_load(0);
        load_b(std::integral_constant<int, 0>{});
        halo_split();
        split_b0();
        load_b(std::integral_constant<int, 1>{});                 // chunk 1 (NCB = 2: the one set)
        if constexpr (NCB == 1) load_b(std::integral_constant<int, 0>{});       // ... and chunk 2
        __syncthreads();
    }
    for (int q = 0; q < nchunks; q += 2) {
        chunk(std::integral_constant<unsigned, 0>{});
        if (q + 1 < nchunks) chunk(std::integral_constant<unsigned, 1>{});
    }
    __syncthreads();                                              // (the epilogue's transposition tiles reuse the planes)

    // ---- epilogue: per column block the wave transposes its 64 x 32 block (two output rows of 32 pixels) through LDS and
    // leaves with full 128-byte lines; BatchNorm partial sums: one row of `stats` per 128 output pixels (waves 0-1 | 2-3)
    {
        float* Tt = smem_all + wave * (64 * 32);
        float (*red)[64][2] = reinterpret_cast<float (*)[64][2]>(smem_all + 4 * 64 * 32);
        const int col_l = lane & 31, rbase = 4 * (lane >> 5);
        const bool elu = a.act == ACT_ELU;
        const long img_m = (long)img * a.Ho * a.Wo;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y + img_m * a.ldy, (unsigned)((long)a.Ho * a.Wo * a.ldy * 4));
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(a.add ? a.add + img_m * a.ld_add : a.y, a.add ? (unsigned)((long)a.Ho * a.Wo * a.ld_add * 4) : 0u);
        // pixel index (inside the image) of row (t & 3) * 8 + lane / 8 of row block t >> 2 of this wave (t: the 8-row store group)
        auto pix_of = [&](int t) {
            const int r = (t & 3) * 8 + (lane >> 3);
            return (oy0 + (2 * wave + (t >> 2)) * BR + r / TW) * a.Wo + ox0 + r % TW;
        };
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            float4 addv[8];
            if (a.add) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int pix = pix_of(t);
                    addv[t] = buf_ld4(ra, (unsigned)(pix * (int)a.ld_add + n0 + 32 * j + 4 * (lane & 7)) * 4u);
                }
            }
            float s1 = 0.f, s2 = 0.f;
            const float bv = a.bias ? a.bias[n0 + 32 * j + col_l] : 0.f;     // (the statistics are those of conv + bias)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (wsign ? -acc[i][j][r] : acc[i][j][r]) + bv;
                    Tt[(32 * i + (r & 3) + 8 * (r >> 2) + rbase) * 32 + col_l] =
                        elu ? (v > 0.f ? v : __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.f) : v;
                    s1 += v;
                    s2 = __builtin_fmaf(v, v, s2);
                }
            const float4* Tq = reinterpret_cast<const float4*>(Tt) + lane;
            float4 v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v[t] = Tq[t * 64];
                if (a.add) { v[t].x += addv[t].x; v[t].y += addv[t].y; v[t].z += addv[t].z; v[t].w += addv[t].w; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int pix = pix_of(t);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[t]), ry,
                                                      (unsigned)(pix * (int)a.ldy + n0 + 32 * j + 4 * (lane & 7)) * 4u, 0, 0);
            }
            asm volatile("s_nop 1");
            __builtin_amdgcn_sched_barrier(0);
            if (a.stats) {
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (lane < 32) { red[wave][32 * j + col_l][0] = s1; red[wave][32 * j + col_l][1] = s2; }
            }
        }
        if (a.stats) {
            __syncthreads();
            if (a.stats_rows == 128) {                         // two rows of `stats` per tile: waves 0-1 | 2-3
                if (tid < 2 * 64 && (tid & 63) < BNW) {
                    const int row_l = tid >> 6, cl = tid & 63;
                    const float t1 = red[2 * row_l][cl][0] + red[2 * row_l + 1][cl][0];
                    const float t2 = red[2 * row_l][cl][1] + red[2 * row_l + 1][cl][1];
                    float* o = a.stats + ((long)(mt * 2 + row_l) * a.Co + n0 + cl) * 2;
                    o[0] = t1; o[1] = t2;
                }
            } else {                                           // M < 65536: one row per 64 output pixels = per wave
                const int row_l = tid >> 6, cl = tid & 63;
                if (cl < BNW) {
                    float* o = a.stats + ((long)(mt * 4 + row_l) * a.Co + n0 + cl) * 2;
                    o[0] = red[row_l][cl][0]; o[1] = red[row_l][cl][1];
                }
            }
        }
    }
}

// Shapes the halo kernel takes (the caller has established the x3 preconditions: vector path, alignment, no folded scale,
// activation none | ELU): stride 1, 128- or 64-row BatchNorm statistics, an image within 32-bit byte offsets, the output grid
// a whole number of 8 x 32, 16 x 16 or 32 x 8 tiles, at least 512 workgroups (two per CU: with 320 the gather kernel's
// 128-row tiles fill the chip better -- 3x3x256 @32x40 151 vs 132 TF, the data gradient of 5x5 256 -> 512 @32x40 165 vs 151), and
//   * a square 3x3 / 5x5 filter, zero padding | the stride-1 data gradient | reflection padding, whole 16-channel groups,
//     Cout % 64 == 0 with 64-column workgroups -- or, where those would be fewer than 512 or Cout % 64 == 32, 32-column
//     workgroups (3x3 on every tile shape, 5x5 on 32 x 8 tiles); or
//   * a 4x4 zero-padded filter over contiguous pixels (sW == C: the space-to-depth stems) with 4 C % 16 == 0, Cout % 64 == 0,
//     8 x 32 tiles: the row-window form.
// tile width: the widest of 32 | 16 | 8 whose 256-pixel tile (8 | 16 | 32 rows) divides the output grid; 0: none
static int x3_halo_tw(int Ho, int Wo) {
    return (Wo % 32 == 0 && Ho % 8 == 0) ? 32 : (Wo % 16 == 0 && Ho % 16 == 0) ? 16 : (Wo % 8 == 0 && Ho % 32 == 0) ? 8 : 0;
}
struct HaloPlan { int tw, ncb; bool rowwin; };
static bool x3_halo_plan(const ConvArgs& a, HaloPlan& p) {
    p.tw = x3_halo_tw(a.Ho, a.Wo);
    p.rowwin = a.KH == 4 && a.KW == 4;
    // 64-column workgroups where they are at least 512; otherwise 32-column ones (3x3x256 @32x40: 320 -> 640 workgroups)
    p.ncb = (a.Co % 64 == 0 && (a.M / 256) * (a.Co / 64) >= 512) ? 2 : 1;
    if (p.tw == 0 || a.stride != 1 || a.sC != 1 || !(a.stats_rows == 128 || a.stats_rows == 64) || a.Co % 32 != 0) return false;
    if (!((long)a.sN * 4 < 0x7fffffffL && (long)a.Ho * a.Wo * a.ldy * 4 < 0x7fffffffL && (!a.add || (long)a.Ho * a.Wo * a.ld_add * 4 < 0x7fffffffL)))
        return false;
    if ((a.M / 256) * (a.Co / (32 * p.ncb)) < 512) return false;
    // halo offsets are 16-bit counts of 64-byte (row windows: 16-byte) units from the halo's first pixel, two per register
    const long unit = p.rowwin ? 4 : 16;
    const long hh = 256 / p.tw + a.KH - 1, hw = p.tw + a.KW - 1;
    if (a.sH % unit != 0 || a.sW % unit != 0 || (hh + 1) * (a.sH / unit) + (hw + 1) * (a.sW / unit) >= 0xffffL) return false;
    if (p.rowwin)
        return a.mode == MODE_ZERO && a.sW == a.C && (a.KW * a.C) % x3h::CK == 0 && p.ncb == 2 && p.tw == 32 && a.pad < a.KH && a.pad_w < a.KW;
    if (!((a.KH == 3 || a.KH == 5) && a.KW == a.KH && a.C % x3h::CK == 0)) return false;
    if (!(a.mode == MODE_ZERO || (a.mode == MODE_TRANSPOSED && a.sshift == 0) || (a.mode == MODE_REFLECT && a.pad < a.H && a.pad_w < a.W))) return false;
    // 32-column workgroups: 3x3 on 8 x 32 / 16 x 16 tiles, 5x5 on the 32 x 8 tiles of the 32x40 planes (same box, vs the gather
    // kernel's 128-row tiles: 64 -> 64 @64x80 127 vs 122 TF, the data gradient of 5x5 256 -> 512 @32x40 176 vs 161; 3x3x256 @32x40
    // 146 vs 150: not taken)
    return p.ncb == 2 || (a.KH == 3 && p.tw != 8) || (a.KH == 5 && p.tw == 8);
}
static bool x3_halo_eligible(const ConvArgs& a) { HaloPlan p; return x3_halo_plan(a, p); }

static int launch_conv_x3_halo(ConvArgs& a, hipStream_t st) {
    HaloPlan p;
    if (!x3_halo_plan(a, p)) return pd::fail(PD_EINVAL, "pd_conv2d: internal: halo plan");
    const int tw = p.tw;
    a.mtiles = a.N * (a.Ho / (256 / tw)) * (a.Wo / tw);
    a.ntiles = a.Co / (32 * p.ncb);
    a.nmajor = x3_nmajor(a);
    const long nblk = (long)a.mtiles * a.ntiles;
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8)), block(NT);
    // Occupancy history (round 4, same box): a three-chunk LDS-DMA weight ring at two workgroups per CU -> a two-chunk ring at
    // three (3x3 only: 53 KB, 166 registers; 3x3x64 @256x320 forward 154 -> 185 TF, data gradient 185 -> 200) -> weights staged
    // through registers, no ring: 45 KB / 52.5 KB, three workgroups per CU for 5x5 as well.
#define PD_HALO(KHV, KWV, TWV, NCBV, RW) do { \
        if (a.mode == MODE_ZERO) hipLaunchKernelGGL((conv_halo_x3_kernel<MODE_ZERO, KHV, KWV, TWV, NCBV, RW>), grid, block, 0, st, a); \
        else if constexpr (!RW) { \
            if (a.mode == MODE_REFLECT) hipLaunchKernelGGL((conv_halo_x3_kernel<MODE_REFLECT, KHV, KWV, TWV, NCBV, false>), grid, block, 0, st, a); \
            else hipLaunchKernelGGL((conv_halo_x3_kernel<MODE_TRANSPOSED, KHV, KWV, TWV, NCBV, false>), grid, block, 0, st, a); } } while (0)
    if (p.rowwin) PD_HALO(4, 1, 32, 2, true);
    else if (p.ncb == 1 && a.KH == 3) { if (tw == 32) PD_HALO(3, 3, 32, 1, false); else PD_HALO(3, 3, 16, 1, false); }
    else if (p.ncb == 1) PD_HALO(5, 5, 8, 1, false);
    else if (a.KH == 3) { if (tw == 32) PD_HALO(3, 3, 32, 2, false); else if (tw == 16) PD_HALO(3, 3, 16, 2, false); else PD_HALO(3, 3, 8, 2, false); }
    else { if (tw == 32) PD_HALO(5, 5, 32, 2, false); else if (tw == 16) PD_HALO(5, 5, 16, 2, false); else PD_HALO(5, 5, 8, 2, false); }
#undef PD_HALO
    return pd::check_launch("pd_conv2d");
}
