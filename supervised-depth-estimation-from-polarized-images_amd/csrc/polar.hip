// K1 -- fused Stokes / DoLP / AoLP / physical-normals kernel for gfx950 (MI355X).
//
// One pass over the four uint8 polarizer planes produces, per pixel,
//   DoLP rho, AoLP phi                       (polarisation/xolp.py:8-34, canonical closed form)
//   standardised XOLP                        (manydepth/networks/pre_encoders.py:78-79)
//   theta_diffuse, theta_spec1, theta_spec2  (manydepth/normals_vec.py:11-50, scipy _call_linear)
//   the 9-channel physical normals           (manydepth/normals_vec.py:53-60, pre_encoders.py:99-113)
// Memory-bound by design: 4 B/px read, 8..80 B/px written, every access a full
// 4-byte (loads) or 16-byte (stores) per-lane vector on planar NCHW tensors.
//
// Exactness strategy
//   * rho: the reference's fp64 op sequence (xolp.py:22-27) is executed literally in
//     fp64 (IEEE sqrt / div / add, no FMA contraction in this file) and rounded once
//     to fp32  ->  bit-equal to the CPU restatement.
//   * phi: depends only on the integer pair (d1,d2) = (I0-I90, I45-I135) in
//     [-255,255]^2; a 511x511 fp32 LUT built on the host in fp64 (L2-resident, 1 MB)
//     gives the exactly-rounded value through integer indexing.
//   * theta tables: searchsorted-left on fp32 keys floor32(x[i]) is exact for an fp32
//     query.  The three tables are searched at once: ONE binary search over the merged,
//     sorted key array gives p = #{merged keys < rho}; a rank table maps p to the three
//     per-table counts (= the three searchsorted results).  The per-bin (x_lo, y_lo, slope)
//     (x_lo, slope) pairs are precomputed in fp64 with the operations scipy performs, next to
//     sin/cos(y_lo).  Keys, a sqrt(rho) bucket index, ranks and bins (96 KB) live in LDS.
//   * normals: cos/sin(phi) in fp32 (torch CPU computes them on the fp32 tensor),
//     promoted and multiplied with the fp64 sin/cos(theta), rounded to fp32.  Inside the tables
//     sin/cos(theta) = angle addition of the tabulated sin/cos(y_lo) with a Taylor series of the
//     sub-step delta (|delta| <= 1.6e-3 rad); extrapolated rho uses a 3-term Cody-Waite reduction
//     + fdlibm kernels (< 1 ulp for |theta| < 1.6e6; LS mode reaches 1.4e2, Stokes mode 2.5e4).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include "pd_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x50444c34u;  // "PDL4"
constexpr int kBuckets = 4096;            // sqrt(rho) buckets of the guided search (<= 4 keys in three buckets
                                          // for 99.6 % of the default tables' range)
constexpr int kLutSide = 511;
constexpr int kLutCount = kLutSide * kLutSide;

struct PolarHeader {   // 64 bytes, little endian
    uint32_t magic;
    int32_t n_d, n_s1, n_s2;
    uint32_t off_lut;    // float[511*511]
    uint32_t off_lds;    // start of the LDS image (keys, then bins)
    uint32_t lds_bytes;  // size of the LDS image (multiple of 16)
    uint32_t total_bytes;
    uint32_t off_bins32; // float[nk][8] fp32 image of the bins (fast normals), absolute offset
    uint32_t img_bins;   // offset of the fp64 bins inside the LDS image (= bytes of the common part)
    uint32_t pad[6];
};
static_assert(sizeof(PolarHeader) == 64, "header size");

// LDS image layout (offsets relative to its start), nk = n_d + n_s1 + n_s2, nkp = nk rounded up to 4:
//   float    mkeys[nkp]           all keys floor32(x) of the three tables, sorted ascending (+inf padding)
//   uint16_t bstart[kBuckets + 4] bstart[b] = #{merged keys k : sqrt(k) < b * smax / kBuckets}; [kBuckets+1..] = nk
//   float    bscale, pad          kBuckets / smax, smax = sqrt(largest finite key)   (8 bytes)
//   uint16_t rank[nk + 1][4]      rank[p] = (#diffuse, #spec1, #spec2, 0) among the first p merged keys
//   double   bins[nk][4]          per table (diffuse | spec1 | spec2), entry i describes bin idx == i:
//                                 x_lo, slope, sin(y_lo), cos(y_lo)
// After the image the blob carries the fp32 form of the bins used by the fast normals path, which replaces
// bins[] in LDS:  float fbins[nk][8] = x_lo32 (= floor32 x_lo), (float)(x_lo - x_lo32), (float)slope,
//                                      s = (float)sin, (float)(sin - s), c = (float)cos, (float)(cos - c), 0
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int keys_padded(int nk) { return (nk + 3) / 4 * 4 + 4; }   // +inf padding: four keys are read at once
inline size_t lds_off_bstart(int nk) { return size_t(keys_padded(nk)) * 4; }
inline size_t lds_off_rank(int nk) { return lds_off_bstart(nk) + size_t(kBuckets + 4) * 2 + 8; }
inline size_t lds_off_bins(int nk) { return round_up(lds_off_rank(nk) + size_t(nk + 1) * 8, 16); }
inline size_t lds_image_bytes(int nk) { return round_up(lds_off_bins(nk) + size_t(nk) * 32, 16); }

float floor32(double x) {  // largest fp32 <= x
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) f = nextafterf(f, -INFINITY);
    return f;
}

}  // namespace

extern "C" size_t pd_polar_tables_bytes(int n_d, int n_s1, int n_s2) {
    if (n_d < 2 || n_s1 < 2 || n_s2 < 2) return 0;
    return sizeof(PolarHeader) + round_up(size_t(kLutCount) * 4, 16) + lds_image_bytes(n_d + n_s1 + n_s2) +
           size_t(n_d + n_s1 + n_s2) * 32;
}

extern "C" int pd_polar_tables_pack(const double* x_d, const double* y_d, int n_d,
                                    const double* x_s1, const double* y_s1, int n_s1,
                                    const double* x_s2, const double* y_s2, int n_s2,
                                    void* host_blob, size_t blob_bytes) {
    PD_REQUIRE(x_d && y_d && x_s1 && y_s1 && x_s2 && y_s2 && host_blob, "pd_polar_tables_pack: null pointer");
    PD_REQUIRE(n_d >= 2 && n_s1 >= 2 && n_s2 >= 2, "pd_polar_tables_pack: each table needs >= 2 nodes");
    PD_REQUIRE(n_d < 4096 && n_s1 < 4096 && n_s2 < 4096, "pd_polar_tables_pack: table too large");
    size_t need = pd_polar_tables_bytes(n_d, n_s1, n_s2);
    PD_REQUIRE(blob_bytes >= need, "pd_polar_tables_pack: blob too small (%zu < %zu)", blob_bytes, need);
    char* base = static_cast<char*>(host_blob);
    memset(base, 0, need);
    PolarHeader h{};
    h.magic = kMagic; h.n_d = n_d; h.n_s1 = n_s1; h.n_s2 = n_s2;
    h.off_lut = sizeof(PolarHeader);
    h.off_lds = h.off_lut + uint32_t(round_up(size_t(kLutCount) * 4, 16));
    const int nk = n_d + n_s1 + n_s2;
    PD_REQUIRE(lds_image_bytes(nk) <= 160 * 1024, "pd_polar_tables_pack: tables exceed the 160 KB LDS of a CU");
    h.lds_bytes = uint32_t(lds_image_bytes(nk));
    h.off_bins32 = h.off_lds + h.lds_bytes;
    h.img_bins = uint32_t(lds_off_bins(nk));
    h.total_bytes = uint32_t(need);
    memcpy(base, &h, sizeof(h));

    // AoLP LUT: phi = 0.5 * atan2(x2, x1), x1 = d1/2, x2 = d2/2 (xolp.py:30), fp64 -> fp32.
    float* lut = reinterpret_cast<float*>(base + h.off_lut);
    for (int d2 = -255; d2 <= 255; ++d2)
        for (int d1 = -255; d1 <= 255; ++d1)
            lut[(d2 + 255) * kLutSide + (d1 + 255)] =
                static_cast<float>(0.5 * atan2(d2 / 2.0, d1 / 2.0));

    const int nkp = keys_padded(nk);
    char* img = base + h.off_lds;
    float* mkeys = reinterpret_cast<float*>(img);
    uint16_t* bstart = reinterpret_cast<uint16_t*>(img + lds_off_bstart(nk));
    float* bscale = reinterpret_cast<float*>(img + lds_off_bstart(nk) + size_t(kBuckets + 4) * 2);
    uint16_t* rank = reinterpret_cast<uint16_t*>(img + lds_off_rank(nk));
    double* bins = reinterpret_cast<double*>(img + lds_off_bins(nk));
    const double* xs[3] = {x_d, x_s1, x_s2};
    const double* ys[3] = {y_d, y_s1, y_s2};
    const int ns[3] = {n_d, n_s1, n_s2};
    std::vector<std::pair<float, int>> merged;   // (key, table)
    merged.reserve(nk);
    int o = 0;
    for (int t = 0; t < 3; ++t) {
        for (int i = 0; i < ns[t]; ++i) {
            PD_REQUIRE(i == 0 || xs[t][i] >= xs[t][i - 1], "pd_polar_tables_pack: x not ascending (table %d)", t);
            PD_REQUIRE(xs[t][i] >= 0.0, "pd_polar_tables_pack: negative node (table %d)", t);
            merged.emplace_back(floor32(xs[t][i]), t);
            if (i >= 1) {  // scipy _call_linear: slope = (y_hi - y_lo) / (x_hi - x_lo); y = slope*(x - x_lo) + y_lo
                double* bin = bins + size_t(o + i) * 4;
                bin[0] = xs[t][i - 1];
                bin[1] = (ys[t][i] - ys[t][i - 1]) / (xs[t][i] - xs[t][i - 1]);
                bin[2] = sin(ys[t][i - 1]);
                bin[3] = cos(ys[t][i - 1]);
                float* fb = reinterpret_cast<float*>(base + h.off_bins32) + size_t(o + i) * 8;
                fb[0] = floor32(bin[0]); fb[1] = (float)(bin[0] - (double)fb[0]); fb[2] = (float)bin[1];
                fb[3] = (float)bin[2]; fb[4] = (float)(bin[2] - (double)fb[3]);
                fb[5] = (float)bin[3]; fb[6] = (float)(bin[3] - (double)fb[5]); fb[7] = 0.f;
            }
        }
        o += ns[t];
    }
    std::stable_sort(merged.begin(), merged.end(),
                     [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    uint16_t cnt[3] = {0, 0, 0};
    for (int p = 0; p <= nk; ++p) {
        rank[4 * p] = cnt[0]; rank[4 * p + 1] = cnt[1]; rank[4 * p + 2] = cnt[2]; rank[4 * p + 3] = 0;
        if (p < nk) { mkeys[p] = merged[p].first; ++cnt[merged[p].second]; }
    }
    for (int p = nk; p < nkp; ++p) mkeys[p] = INFINITY;
    // guided search: buckets uniform in sqrt(rho) (the tables are ~quadratic in theta at both ends)
    const double smax = sqrt((double)merged[nk - 1].first);
    PD_REQUIRE(smax > 0.0, "pd_polar_tables_pack: all nodes are zero");
    int p = 0;
    for (int bkt = 0; bkt <= kBuckets; ++bkt) {
        const double edge = bkt * smax / kBuckets;
        while (p < nk && sqrt((double)merged[p].first) < edge) ++p;
        bstart[bkt] = (uint16_t)p;
    }
    for (int bkt = kBuckets + 1; bkt < kBuckets + 4; ++bkt) bstart[bkt] = (uint16_t)nk;
    bscale[0] = (float)(kBuckets / smax);
    bscale[1] = 0.f;
    return PD_OK;
}

extern "C" int pd_polar_tables_build(double n, void* host_blob, size_t blob_bytes, size_t* used) {
    PD_REQUIRE(n > 1.0, "pd_polar_tables_build: refractive index must be > 1");
    const int N = 1000;  // normals_vec.py:13,27
    std::vector<double> th(N), rd(N), rs(N);
    const double step = (M_PI / 2 - 0.0) / (N - 1);  // numpy.linspace
    for (int i = 0; i < N; ++i) th[i] = i * step + 0.0;
    th[N - 1] = M_PI / 2;
    for (int i = 0; i < N; ++i) {
        double s = sin(th[i]), c = cos(th[i]);
        double s2 = s * s;
        double nm = n - 1 / n, np_ = n + 1 / n;
        // normals_vec.py:14-19
        rd[i] = ((nm * nm) * s2) / (2 + 2 * (n * n) - (np_ * np_) * s2 + 4 * c * sqrt(n * n - s2));
        // normals_vec.py:28-38
        rs[i] = (2 * s2 * c * sqrt(n * n - s2)) / (n * n - s2 - (n * n) * s2 + 2 * (s2 * s2));
    }
    int imax = 0;
    for (int i = 1; i < N; ++i) if (rs[i] > rs[imax]) imax = i;  // np.argmax: first maximum
    PD_REQUIRE(imax >= 2 && N - imax >= 2, "pd_polar_tables_build: degenerate specular split (imax=%d)", imax);
    // spec2 is descending: scipy sorts it ascending (stable argsort) -> reverse.
    int n2 = N - imax;
    std::vector<double> x2(n2), y2(n2);
    for (int i = 0; i < n2; ++i) { x2[i] = rs[N - 1 - i]; y2[i] = th[N - 1 - i]; }
    for (int i = 1; i < n2; ++i) PD_REQUIRE(x2[i] >= x2[i - 1], "pd_polar_tables_build: spec2 not monotone");
    for (int i = 1; i < imax; ++i) PD_REQUIRE(rs[i] >= rs[i - 1], "pd_polar_tables_build: spec1 not monotone");
    for (int i = 1; i < N; ++i) PD_REQUIRE(rd[i] >= rd[i - 1], "pd_polar_tables_build: diffuse not monotone");
    if (used) *used = pd_polar_tables_bytes(N, imax, n2);
    return pd_polar_tables_pack(rd.data(), th.data(), N, rs.data(), th.data(), imax, x2.data(), y2.data(), n2,
                                host_blob, blob_bytes);
}

// ------------------------------------------------------------------ device side
namespace {

struct Tabs {  // LDS-resident view of the interpolation tables
    const float* mkeys;
    const uint16_t* bstart;
    const uint2* rank;          // ushort4 packed in 8 bytes
    const double* bins;         // [nk][4]: x_lo, slope, sin(y_lo), cos(y_lo)  (LDS when PRECISE, else global)
    const float* fbins;         // [nk][8] fp32 image (LDS, fast path only)
    float bscale;
    int nk, n_d, n_s1, n_s2;
};

template <bool PRECISE>
__device__ __forceinline__ void stage_tables(const char* __restrict__ blob, char* smem, int nthreads, Tabs& t) {
    const PolarHeader* h = reinterpret_cast<const PolarHeader*>(blob);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint4* src = reinterpret_cast<const uint4*>(blob + h->off_lds);
    const int ncommon = h->img_bins / 16, n16 = h->lds_bytes / 16;
    for (int i = threadIdx.x; i < ncommon; i += nthreads) dst[i] = src[i];
    const uint4* bsrc = PRECISE ? src + ncommon : reinterpret_cast<const uint4*>(blob + h->off_bins32);
    for (int i = threadIdx.x; i < n16 - ncommon; i += nthreads) dst[ncommon + i] = bsrc[i];
    t.n_d = h->n_d; t.n_s1 = h->n_s1; t.n_s2 = h->n_s2;
    t.nk = t.n_d + t.n_s1 + t.n_s2;
    const int nkp = (t.nk + 3) / 4 * 4 + 4;
    const int off_rank = nkp * 4 + (kBuckets + 4) * 2 + 8;
    t.mkeys = reinterpret_cast<const float*>(smem);
    t.bstart = reinterpret_cast<const uint16_t*>(smem + nkp * 4);
    t.rank = reinterpret_cast<const uint2*>(smem + off_rank);
    t.fbins = reinterpret_cast<const float*>(smem + h->img_bins);
    t.bins = PRECISE ? reinterpret_cast<const double*>(smem + h->img_bins)
                     : reinterpret_cast<const double*>(blob + h->off_lds + h->img_bins);
    __syncthreads();
    t.bscale = *reinterpret_cast<const float*>(smem + nkp * 4 + (kBuckets + 4) * 2);
}

// fp64 sin/cos: Cody-Waite with pi/2 = C1 + C2 + C3 (33 + 33 + 53 bits, fdlibm constants) and the
// fdlibm polynomial kernels on [-pi/4, pi/4]; FMAs are explicit (this file is built with -ffp-contract=off).
__device__ __forceinline__ void sincos_f64(double x, double& s, double& c) {
    // Valid (< 1 ulp) while j * C1 is exact, |x| < 1.6e6; uint8 inputs give |theta| <= 2.5e4 (Stokes mode,
    // rho <= 361).  Beyond that precision degrades gradually; inf / NaN propagate as NaN like numpy.
    const double j = rint(x * 6.36619772367581382433e-01);
    double r = fma(-j, 1.57079632673412561417e+00, x);
    r = fma(-j, 6.07710050630396597660e-11, r);
    r = fma(-j, 2.02226624879595063154e-21, r);
    const double z = r * r;
    // __kernel_sin
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    const double sn = fma(r * z, fma(z, ps, -1.66666666666666324348e-01), r);
    // __kernel_cos
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cs = w + (((1.0 - w) - hz) + z * (z * pc));
    const int q = static_cast<int>(j) & 3;
    const double a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// fp32 sin/cos for |x| <= 4 (AoLP in [-pi/2, pi/2], AoLP + pi/2 in [0, pi]); ~1 ulp
__device__ __forceinline__ void sincos_f32(float x, float& s, float& c) {
    const float j = rintf(x * 0.63661977236758134f);
    float r = fmaf(-j, 1.5707962512969971f, x);       // pi/2 hi (24 bits)
    r = fmaf(-j, 7.5497894158615964e-08f, r);         // pi/2 lo
    const float z = r * r;
    float ps = fmaf(z, 2.7183114939898219e-06f, -1.9839334836096632e-04f);
    ps = fmaf(z, ps, 8.3333095718939529e-03f);
    const float sn = fmaf(r * z, fmaf(z, ps, -1.6666665459843126e-01f), r);
    float pc = fmaf(z, 2.4390448796277409e-05f, -1.3887316255677255e-03f);
    pc = fmaf(z, pc, 4.1666645683222281e-02f);
    const float cs = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int q = static_cast<int>(j) & 3;
    const float a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// sin/cos of theta = y_lo + delta from the tabulated sin/cos(y_lo): delta = slope * (rho - x_lo) is at most
// one table step (1.6e-3 rad) inside the tables, where a short Taylor series is exact to 1e-22; extrapolated
// rho (beyond the table ends) takes the general reduction.  Equals sin/cos(fl64(theta)) to ~1e-16.
__device__ __forceinline__ void sincos_bin(const double* __restrict__ bin, double x, double& s, double& c) {
    const double d = bin[1] * (x - bin[0]);
    double sd, cd;
    if (fabs(d) < 4.0e-3) {
        const double d2 = d * d;
        sd = d * fma(d2, fma(d2, 8.33333333333333322e-03, -1.66666666666666657e-01), 1.0);
        cd = fma(d2, fma(d2, fma(d2, -1.38888888888888894e-03, 4.16666666666666644e-02), -0.5), 1.0);
    } else {
        sincos_f64(d, sd, cd);
    }
    s = fma(bin[2], cd, bin[3] * sd);
    c = fma(bin[3], cd, -(bin[2] * sd));
}

// fp32 form of the same angle addition for the fast path: (rho - x_lo32) is exact (neighbouring floats), the
// residuals of x_lo / sin / cos restore the bits the fp32 table entries drop; error ~1e-7 absolute.
// Extrapolated rho falls back to the fp64 routine (bins read from global memory).
__device__ __forceinline__ void sincos_bin_fast(const float* __restrict__ fb, const double* __restrict__ gbin,
                                                float rho, float& s, float& c) {
    const float4 a = *reinterpret_cast<const float4*>(fb);       // x_lo32, dx, slope, sin
    const float4 b = *reinterpret_cast<const float4*>(fb + 4);   // sin residual, cos, cos residual
    const float d = a.z * ((rho - a.x) - a.y);
    if (fabsf(d) < 4.0e-3f) {
        const float d2 = d * d;
        const float sd = d * fmaf(d2, fmaf(d2, 8.3333338e-03f, -1.6666667e-01f), 1.0f);
        const float cd = fmaf(d2, fmaf(d2, 4.1666668e-02f, -0.5f), 1.0f);
        s = fmaf(a.w, cd, fmaf(b.y, sd, b.x));
        c = fmaf(b.y, cd, fmaf(-a.w, sd, b.z));
    } else {
        double sd, cd;
        sincos_bin(gbin, static_cast<double>(rho), sd, cd);
        s = static_cast<float>(sd);
        c = static_cast<float>(cd);
    }
}

// theta lookups + the three physical normals of one pixel (normals_vec.py:11-60, pre_encoders.py:99-113)
template <bool PRECISE>
__device__ __forceinline__ void normals9(float rho, float phi, const Tabs& t, float (&v)[9], int (&bi)[3]) {
    const float kHalfPi = static_cast<float>(1.5707963267948966);
    // pos = number of merged keys strictly below rho (searchsorted-left on the merged key array).
    // A sqrt(rho) bucket table narrows the range to a few keys (+-1 bucket of slack covers the rounding of
    // the fp32 sqrt), then an exact binary search finishes.
    const float u = sqrtf(fmaxf(rho, 0.f)) * t.bscale;
    const int bk = (int)fminf(u, (float)kBuckets);
    int lo = t.bstart[max(bk - 1, 0)], hi = t.bstart[bk + 2];
    if (hi - lo <= 4) {
        // the usual case: at most four candidate keys -> four independent LDS reads instead of a dependent
        // binary-search chain (keys at or beyond hi are > rho, the array is padded with +inf)
        lo += (t.mkeys[lo] < rho) + (t.mkeys[lo + 1] < rho) + (t.mkeys[lo + 2] < rho) + (t.mkeys[lo + 3] < rho);
    } else {
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (t.mkeys[mid] < rho) lo = mid + 1; else hi = mid;
        }
    }
    const uint2 rk = t.rank[lo];
    const bool isnan_ = rho != rho;   // NaN sorts last (numpy)
    bi[0] = isnan_ ? t.n_d - 1 : min(max((int)(rk.x & 0xffffu), 1), t.n_d - 1);
    bi[1] = isnan_ ? t.n_s1 - 1 : min(max((int)(rk.x >> 16), 1), t.n_s1 - 1);
    bi[2] = isnan_ ? t.n_s2 - 1 : min(max((int)(rk.y & 0xffffu), 1), t.n_s2 - 1);
    const int i0 = bi[0], i1 = t.n_d + bi[1], i2 = t.n_d + t.n_s1 + bi[2];
    float sp, cp;
    sincos_f32(phi, sp, cp);             // torch.cos/sin on the fp32 AoLP
    if (PRECISE) {
        const double x = static_cast<double>(rho);
        double sd, cd, s1, c1, s2, c2;
        sincos_bin(t.bins + 4 * i0, x, sd, cd);
        sincos_bin(t.bins + 4 * i1, x, s1, c1);
        sincos_bin(t.bins + 4 * i2, x, s2, c2);
        float sq, cq;
        sincos_f32(phi + kHalfPi, sq, cq);   // phi + np.pi/2 evaluated in fp32 (pre_encoders.py:108-109)
        v[0] = static_cast<float>(static_cast<double>(cp) * sd);
        v[1] = static_cast<float>(static_cast<double>(sp) * sd);
        v[2] = static_cast<float>(cd);
        v[3] = static_cast<float>(static_cast<double>(cq) * s1);
        v[4] = static_cast<float>(static_cast<double>(sq) * s1);
        v[5] = static_cast<float>(c1);
        v[6] = static_cast<float>(static_cast<double>(cq) * s2);
        v[7] = static_cast<float>(static_cast<double>(sq) * s2);
        v[8] = static_cast<float>(c2);
    } else {
        float sd, cd, s1, c1, s2, c2;
        sincos_bin_fast(t.fbins + 8 * i0, t.bins + 4 * i0, rho, sd, cd);
        sincos_bin_fast(t.fbins + 8 * i1, t.bins + 4 * i1, rho, s1, c1);
        sincos_bin_fast(t.fbins + 8 * i2, t.bins + 4 * i2, rho, s2, c2);
        // cos(phi + pi/2) = -sin(phi), sin(phi + pi/2) = cos(phi): within 2e-7 of the reference's fp32
        // evaluation of cos/sin(fl32(phi + pi/2))
        v[0] = cp * sd; v[1] = sp * sd; v[2] = cd;
        v[3] = -sp * s1; v[4] = cp * s1; v[5] = c1;
        v[6] = -sp * s2; v[7] = cp * s2; v[8] = c2;
    }
}

struct Px {
    float rho, phi;
    float d1, d2;    // I0 - I90, I45 - I135 (exact small integers)
};

// The reference's fp64 operation sequence, executed literally (IEEE sqrt / div, one rounding to fp32).
// S = I0 + I45 + I90 + I135 (LS) or I0 + I90 (Stokes); all arguments are exact small integers.
template <int MODE>
__device__ __noinline__ float rho_ieee(float S, float d1, float d2) {
    double rho;
    if (MODE == PD_POLAR_LS) {
        // x = closed-form least-squares solution; then xolp.py:22-29 literally, in fp64.
        const double x0 = static_cast<double>(S) * 0.25;
        const double x1 = static_cast<double>(d1) * 0.5;
        const double x2 = static_cast<double>(d2) * 0.5;
        const double r = sqrt(x1 * x1 + x2 * x2);
        const double imax = x0 + r;
        const double imin = x0 - r;
        rho = (imax - imin) / (imax + imin);
        if (isinf(rho) || isnan(rho)) rho = 0.0;  // rho[rho == inf] = 0; nan_to_num
    } else {
        // physical_normals_channels.py:21-26: rho = sqrt(s1^2 + s2^2) / s0, no guard.
        const double s1 = static_cast<double>(d1), s2 = static_cast<double>(d2);
        rho = sqrt(s1 * s1 + s2 * s2) / static_cast<double>(S);
    }
    return static_cast<float>(rho);
}

// DoLP without the IEEE sqrt/div sequences (Ziv's rounding test).  Mathematically rho = sqrt(s4) / den with the
// integers s4 = d1^2 + d2^2 and den = S/2 (LS) or S (Stokes).  The reference's fp64 chain differs from that
// value by < 2^-42 relative (its largest term: the roundings of x0 +- r, 2^-53 * x0/r <= 2^-44 for uint8 data).
// Here: hardware rsq/rcp seeds (1 ulp fp32) and one Newton step each in fp64, relative error < 2^-43.  Both
// therefore round to the same fp32 unless the value lies within 2^-42 of a rounding midpoint; whenever the
// low 29 mantissa bits of q are within 2^14 fp64-ulps (2^-38 relative) of the midpoint pattern the pixel takes
// the literal sequence instead (about one pixel in 16k).  tests/test_polar_gpu.py checks the equality against
// the literal sequence over all 2^32 uint8 quadruples.
template <int MODE>
__device__ __forceinline__ float rho_pixel(float S, float d1, float d2, float s4, bool ieee) {
    const float den = MODE == PD_POLAR_LS ? 0.5f * S : S;
    const float y = __builtin_amdgcn_rsqf(s4);
    const float z = __builtin_amdgcn_rcpf(den);
    const double A = static_cast<double>(s4), D = static_cast<double>(den);
    const double Y = static_cast<double>(y), Yh = static_cast<double>(-0.5f * y);
    double Z = static_cast<double>(z);
    double g = A * Y;                          // sqrt(s4) (1 + d),        |d| < 2^-22
    g = fma(g, fma(Yh, g, 0.5), g);            // sqrt(s4) (1 - 1.5 d^2)
    Z = fma(Z, fma(-D, Z, 1.0), Z);            // 1 / den  (1 - d'^2)
    const double q = g * Z;
    const unsigned lo = static_cast<unsigned>(__double2loint(q)) & 0x1fffffffu;
    bool slow = (lo - 0x0fffc000u) < 0x8000u;  // within 2^14 ulps of the fp32 rounding midpoint
    if (MODE == PD_POLAR_STOKES) slow |= (S == 0.f);   // x / 0 -> inf, 0 / 0 -> NaN like numpy
    float rho = static_cast<float>(q);
    if (s4 == 0.f) { rho = 0.f; if (MODE == PD_POLAR_LS || S != 0.f) slow = false; }
    if (slow | ieee) rho = rho_ieee<MODE>(S, d1, d2);
    return rho;
}

template <int MODE>
__device__ __forceinline__ Px xolp_pixel(float f0, float f45, float f90, float f135, const float* __restrict__ lut, bool ieee) {
    Px p;
    p.d1 = f0 - f90;
    p.d2 = f45 - f135;
    const float s4 = fmaf(p.d1, p.d1, p.d2 * p.d2);                      // <= 130050: exact in fp32
    const float S = MODE == PD_POLAR_LS ? (f0 + f90) + (f45 + f135) : f0 + f90;
    p.rho = rho_pixel<MODE>(S, p.d1, p.d2, s4, ieee);
    // (d2 + 255) * 511 + (d1 + 255), exact in fp32
    const unsigned idx = static_cast<unsigned>(static_cast<int>(fmaf(p.d2, 511.f, p.d1 + 130560.f)));
    p.phi = lut[idx];
    return p;
}

// (x - mean) / std with the correctly rounded quotient (Markstein: q = RN(a*y), r = a - q*b exactly, RN(q + r*y)
// equals RN(a / b) when y = RN(1/b) and b's significand is not all ones; checked over every fp32 in [-4, 4] by
// tests/test_polar_gpu.py).  Replaces the ten-instruction IEEE division sequence.
__device__ __forceinline__ float standardise(float x) {
    const float kMean = static_cast<float>(0.08693199701957657);
    const float kStd = static_cast<float>(0.44430732785457433);
    const float kInv = 1.0f / kStd;
    const float a = x - kMean;
    const float q = a * kInv;
    return fmaf(fmaf(-q, kStd, a), kInv, q);
}

constexpr int kThreads = 512;    // XOLP-only kernels: 4 workgroups per CU
constexpr int kThreadsN = 768;   // kernels with the fp64 normals: one 12-wave workgroup per CU (3 waves/SIMD, <= 168 VGPRs);
                                 // its 96 KB LDS image of the tables is staged once per CU

constexpr int kThreadsF = 1024;  // fast (fp32) normals: 112 VGPRs -> one 16-wave workgroup per CU (4 waves/SIMD)

// Launch geometry: every index is 32-bit (the host splits a batch whose planes would exceed 2^32 bytes).
// The output of one image is Hrows rows of wq_out quads (4 pixels); with a pitched output (Wout > W) a row is
// an image row, otherwise the whole plane is one row.  A thread walks (image, row, quad) by a constant step.
struct PolarGeo {
    int B, Hrows, wq_in, wq_out;
    int drow, dcq;                 // persistent-loop step (grid * block quads) as rows + quads
    unsigned P, Pout;              // pixels per input / output plane
    int flags;
};

template <int MODE, bool NORMALS, bool PRECISE>
__global__ __launch_bounds__(NORMALS ? (PRECISE ? kThreadsN : kThreadsF) : kThreads) void polar_kernel(
    const uint8_t* __restrict__ pol, const uint8_t* __restrict__ mask, float* __restrict__ xolp,
    float* __restrict__ xolp_std, float* __restrict__ normals, int* __restrict__ ints,
    const char* __restrict__ blob, const PolarGeo g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PolarHeader* h = reinterpret_cast<const PolarHeader*>(blob);
    const float* lut = reinterpret_cast<const float*>(blob + h->off_lut);
    constexpr int NTH = NORMALS ? (PRECISE ? kThreadsN : kThreadsF) : kThreads;
    Tabs tabs;
    if (NORMALS) stage_tables<PRECISE>(blob, smem, NTH, tabs);
    const bool ieee = (g.flags & PD_POLAR_IEEE_RHO) != 0;

    const unsigned q0 = blockIdx.x * (unsigned)NTH + threadIdx.x;
    unsigned row = q0 / (unsigned)g.wq_out;
    int cq = (int)(q0 - row * (unsigned)g.wq_out);
    int b = (int)(row / (unsigned)g.Hrows);
    row -= (unsigned)b * (unsigned)g.Hrows;
    const unsigned in_row = 4u * g.wq_in, out_row = 4u * g.wq_out;

    while (b < g.B) {
        const unsigned po = row * out_row + 4u * cq;     // first output pixel of the quad inside its plane
        if (cq >= g.wq_in) {                             // right padding columns of a pitched output: zeros
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xolp) { *reinterpret_cast<float4*>(xolp + ((unsigned)b * 2u * g.Pout + po)) = z; *reinterpret_cast<float4*>(xolp + ((unsigned)b * 2u * g.Pout + g.Pout + po)) = z; }
            if (xolp_std) { *reinterpret_cast<float4*>(xolp_std + ((unsigned)b * 2u * g.Pout + po)) = z; *reinterpret_cast<float4*>(xolp_std + ((unsigned)b * 2u * g.Pout + g.Pout + po)) = z; }
            if (NORMALS && normals)
                for (unsigned c = 0; c < 9; ++c) *reinterpret_cast<float4*>(normals + (((unsigned)b * 9u + c) * g.Pout + po)) = z;
            if (ints)
                for (unsigned c = 0; c < (NORMALS ? 5u : 2u); ++c) *reinterpret_cast<int4*>(ints + (((unsigned)b * 5u + c) * g.Pout + po)) = make_int4(0, 0, 0, 0);
        } else {
            const unsigned p4 = (unsigned)b * 4u * g.P + row * in_row + 4u * cq;   // first input pixel (plane 0)
            const uint32_t w0 = *reinterpret_cast<const uint32_t*>(pol + p4);
            const uint32_t w45 = *reinterpret_cast<const uint32_t*>(pol + (p4 + g.P));
            const uint32_t w90 = *reinterpret_cast<const uint32_t*>(pol + (p4 + 2u * g.P));
            const uint32_t w135 = *reinterpret_cast<const uint32_t*>(pol + (p4 + 3u * g.P));
            uint32_t wm = 0x01010101u;
            if (MODE == PD_POLAR_STOKES && mask) wm = *reinterpret_cast<const uint32_t*>(mask + ((unsigned)b * g.P + row * in_row + 4u * cq));

            float o_rho[4], o_phi[4], o_n[9][4];
            int o_i[5][4];
            // With the normals the four pixels are processed in pairs (scheduling barrier between them):
            // interleaving four fp64 trig chains quadruples the live registers.
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (NORMALS && (j & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                const int sh = 8 * j;
                const bool on = MODE != PD_POLAR_STOKES || ((wm >> sh) & 0xffu) != 0;
                // v_cvt_f32_ubyteN: byte -> float in one instruction
                float f0 = static_cast<float>((w0 >> sh) & 0xffu), f45 = static_cast<float>((w45 >> sh) & 0xffu);
                float f90 = static_cast<float>((w90 >> sh) & 0xffu), f135 = static_cast<float>((w135 >> sh) & 0xffu);
                Px p;
                if (MODE == PD_POLAR_STOKES && !on) {   // images are masked first (:117-121); outputs are zero outside
                    p.rho = 0.f; p.phi = 0.f; p.d1 = 0.f; p.d2 = 0.f;
                } else {
                    p = xolp_pixel<MODE>(f0, f45, f90, f135, lut, ieee);
                }
                o_rho[j] = p.rho;
                o_phi[j] = p.phi;
                o_i[0][j] = static_cast<int>(p.d1);
                o_i[1][j] = static_cast<int>(p.d2);
                if (NORMALS) {
                    int bi[3];
                    float v[9];
                    normals9<PRECISE>(p.rho, p.phi, tabs, v, bi);
                    o_i[2][j] = bi[0]; o_i[3][j] = bi[1]; o_i[4][j] = bi[2];
#pragma unroll
                    for (int c = 0; c < 9; ++c) o_n[c][j] = on ? v[c] : 0.f;
                }
            }
            if (xolp) {
                float* o = xolp + ((unsigned)b * 2u * g.Pout + po);
                *reinterpret_cast<float4*>(o) = make_float4(o_rho[0], o_rho[1], o_rho[2], o_rho[3]);
                *reinterpret_cast<float4*>(o + g.Pout) = make_float4(o_phi[0], o_phi[1], o_phi[2], o_phi[3]);
            }
            if (xolp_std) {
                float* o = xolp_std + ((unsigned)b * 2u * g.Pout + po);
                *reinterpret_cast<float4*>(o) = make_float4(standardise(o_rho[0]), standardise(o_rho[1]),
                                                            standardise(o_rho[2]), standardise(o_rho[3]));
                *reinterpret_cast<float4*>(o + g.Pout) = make_float4(standardise(o_phi[0]), standardise(o_phi[1]),
                                                                     standardise(o_phi[2]), standardise(o_phi[3]));
            }
            if (NORMALS && normals) {
                float* o = normals + ((unsigned)b * 9u * g.Pout + po);
#pragma unroll
                for (unsigned c = 0; c < 9; ++c)
                    *reinterpret_cast<float4*>(o + c * g.Pout) = make_float4(o_n[c][0], o_n[c][1], o_n[c][2], o_n[c][3]);
            }
            if (ints) {
                int* o = ints + ((unsigned)b * 5u * g.Pout + po);
                const unsigned nch = NORMALS ? 5 : 2;
#pragma unroll
                for (unsigned c = 0; c < 5; ++c)
                    if (c < nch) *reinterpret_cast<int4*>(o + c * g.Pout) = make_int4(o_i[c][0], o_i[c][1], o_i[c][2], o_i[c][3]);
            }
        }
        // next quad of this thread
        cq += g.dcq;
        row += (unsigned)g.drow;
        if (cq >= g.wq_out) { cq -= g.wq_out; ++row; }
        if (row >= (unsigned)g.Hrows) {
            if (row < 2u * (unsigned)g.Hrows) { row -= (unsigned)g.Hrows; ++b; }
            else { const unsigned k = row / (unsigned)g.Hrows; row -= k * (unsigned)g.Hrows; b += (int)k; }
        }
    }
}

// get_normals() on an existing fp32 XOLP tensor (pre_encoders.py:99-113): [B,2,H,W] -> [B,9,H,W]
template <bool PRECISE>
__global__ __launch_bounds__(PRECISE ? kThreadsN : kThreadsF) void normals_from_xolp_kernel(const float* __restrict__ xolp,
                                                                     float* __restrict__ normals,
                                                                     const char* __restrict__ blob, long P,
                                                                     long quads_per_img, long total_quads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Tabs tabs;
    constexpr int NTH = PRECISE ? kThreadsN : kThreadsF;
    stage_tables<PRECISE>(blob, smem, NTH, tabs);
    for (long q = blockIdx.x * (long)NTH + threadIdx.x; q < total_quads; q += (long)gridDim.x * NTH) {
        const long b = q / quads_per_img;
        const long p4 = (q - b * quads_per_img) * 4;
        const float4 r4 = *reinterpret_cast<const float4*>(xolp + (b * 2) * P + p4);
        const float4 f4 = *reinterpret_cast<const float4*>(xolp + (b * 2 + 1) * P + p4);
        const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, ff[4] = {f4.x, f4.y, f4.z, f4.w};
        float o_n[9][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            float v[9]; int bi[3];
            normals9<PRECISE>(rr[j], ff[j], tabs, v, bi);
#pragma unroll
            for (int c = 0; c < 9; ++c) o_n[c][j] = v[c];
        }
        float* o = normals + (b * 9) * P + p4;
#pragma unroll
        for (int c = 0; c < 9; ++c)
            *reinterpret_cast<float4*>(o + c * P) = make_float4(o_n[c][0], o_n[c][1], o_n[c][2], o_n[c][3]);
    }
}

// LDS image size implied by the blob size: blob = header + LUT + image(nk) + nk * 32 (monotone in nk)
inline size_t lds_from_blob_bytes(size_t tables_bytes) {
    const size_t fixed = sizeof(PolarHeader) + round_up(size_t(kLutCount) * 4, 16);
    for (int nk = 6; nk < 3 * 4096; ++nk)
        if (fixed + lds_image_bytes(nk) + size_t(nk) * 32 == tables_bytes) return lds_image_bytes(nk);
    return 0;
}

template <typename K>
int set_lds_limit(K kernel, size_t lds) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return pd::fail(PD_ELAUNCH, "pd_polar: cannot reserve %zu bytes of LDS", lds);
    return PD_OK;
}

}  // namespace

extern "C" int pd_polar_normals_from_xolp(const void* xolp, void* normals, const void* tables, size_t tables_bytes,
                                          int B, int H, int W, int flags, void* stream) {
    PD_REQUIRE(B >= 0 && H > 0 && W > 0, "pd_polar_normals_from_xolp: bad shape");
    const long P = (long)H * W;
    PD_REQUIRE(P % 4 == 0, "pd_polar_normals_from_xolp: H*W=%ld must be a multiple of 4", P);
    if (B == 0) return PD_OK;
    PD_REQUIRE(xolp && normals && tables, "pd_polar_normals_from_xolp: null pointer");
    PD_REQUIRE(pd::aligned16(xolp) && pd::aligned16(normals) && pd::aligned16(tables), "pd_polar_normals_from_xolp: unaligned");
    PD_REQUIRE(tables_bytes >= sizeof(PolarHeader) + size_t(kLutCount) * 4, "pd_polar_normals_from_xolp: tables blob too small");
    const size_t lds = lds_from_blob_bytes(tables_bytes);
    PD_REQUIRE(lds > 0 && lds <= 160 * 1024, "pd_polar_normals_from_xolp: tables blob has an unexpected size");
    const bool precise = (flags & PD_POLAR_FAST_NORMALS) == 0;
    {
        int rc = precise ? set_lds_limit(normals_from_xolp_kernel<true>, lds) : set_lds_limit(normals_from_xolp_kernel<false>, lds);
        if (rc) return rc;
    }
    const long qpi = P / 4, total = qpi * B;
    const int nth = precise ? kThreadsN : kThreadsF;
    long blocks = (total + nth - 1) / nth;
    if (blocks > 256) blocks = 256;
    if (precise)
        hipLaunchKernelGGL(normals_from_xolp_kernel<true>, dim3((unsigned)blocks), dim3(kThreadsN), lds, (hipStream_t)stream,
                           (const float*)xolp, (float*)normals, (const char*)tables, P, qpi, total);
    else
        hipLaunchKernelGGL(normals_from_xolp_kernel<false>, dim3((unsigned)blocks), dim3(kThreadsF), lds, (hipStream_t)stream,
                           (const float*)xolp, (float*)normals, (const char*)tables, P, qpi, total);
    return pd::check_launch("pd_polar_normals_from_xolp");
}

extern "C" int pd_polar_fwd(const void* pol, const void* mask, void* xolp, void* xolp_std, void* normals,
                            void* ints, const void* tables, size_t tables_bytes, int B, int H, int W, int Wout,
                            int mode, int flags, void* stream) {
    PD_REQUIRE(B >= 0 && H > 0 && W > 0, "pd_polar_fwd: bad shape B=%d H=%d W=%d", B, H, W);
    if (B == 0) return PD_OK;  // empty batch: nothing to do (pointers may be null)
    PD_REQUIRE(pol && tables, "pd_polar_fwd: pol and tables must not be null");
    PD_REQUIRE(mode == PD_POLAR_LS || mode == PD_POLAR_STOKES, "pd_polar_fwd: unknown mode %d", mode);
    const long P = (long)H * W;
    if (Wout <= 0) Wout = W;
    PD_REQUIRE(Wout >= W, "pd_polar_fwd: output pitch %d < W %d", Wout, W);
    PD_REQUIRE(Wout == W ? P % 4 == 0 : (W % 4 == 0 && Wout % 4 == 0),
               "pd_polar_fwd: H*W (or W and the output pitch, when they differ) must be multiples of 4");
    PD_REQUIRE(tables_bytes >= sizeof(PolarHeader) + size_t(kLutCount) * 4, "pd_polar_fwd: tables blob too small");
    PD_REQUIRE(pd::aligned16(pol) && pd::aligned16(xolp) && pd::aligned16(xolp_std) && pd::aligned16(normals) &&
                   pd::aligned16(ints) && pd::aligned16(tables) && (!mask || pd::aligned16(mask)),
               "pd_polar_fwd: pointers must be 16-byte aligned");
    PD_REQUIRE(xolp || xolp_std || normals || ints, "pd_polar_fwd: no output requested");
    const long Pout = (long)H * Wout;
    const bool pitched = Wout != W;
    const bool need_normals = normals != nullptr || ints != nullptr;
    const bool precise = (flags & PD_POLAR_FAST_NORMALS) == 0;
    // LDS image size is fixed by the table node counts (96,232 bytes for the default 1000/625/375 nodes)
    const size_t lds = need_normals ? lds_from_blob_bytes(tables_bytes) : 0;
    PD_REQUIRE(!need_normals || (lds > 0 && lds <= 160 * 1024), "pd_polar_fwd: tables blob has an unexpected size");
    const int nth = need_normals ? (precise ? kThreadsN : kThreadsF) : kThreads;
    // 32-bit addressing inside the kernel: at most 2^30 elements per output tensor and 2^31 quads per launch
    PD_REQUIRE(9 * Pout < (1L << 30), "pd_polar_fwd: image too large (%ld output pixels)", Pout);
    const long max_b = (1L << 30) / (9 * Pout);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = PD_OK;
    for (long b0 = 0; b0 < B && rc == PD_OK; b0 += max_b) {
        const int nb = (int)std::min<long>(max_b, B - b0);
        PolarGeo g;
        g.B = nb; g.Hrows = pitched ? H : 1;
        g.wq_in = pitched ? W / 4 : (int)(P / 4); g.wq_out = pitched ? Wout / 4 : (int)(P / 4);
        g.P = (unsigned)P; g.Pout = (unsigned)Pout; g.flags = flags;
        const long total = (long)nb * (Pout / 4);
        long blocks = (total + nth - 1) / nth;
        // with normals: persistent, one workgroup per CU (the 96 KB table image is staged once per CU);
        // XOLP only: no tables to amortise -> one quad per thread, hardware-scheduled (measured 1.4x faster)
        if (need_normals && blocks > 256) blocks = 256;
        const long step = blocks * nth;
        g.drow = (int)(step / g.wq_out); g.dcq = (int)(step % g.wq_out);
        const uint8_t* pol_b = static_cast<const uint8_t*>(pol) + b0 * 4 * P;
        const uint8_t* mask_b = mask ? static_cast<const uint8_t*>(mask) + b0 * P : nullptr;
        float* xolp_b = xolp ? static_cast<float*>(xolp) + b0 * 2 * Pout : nullptr;
        float* std_b = xolp_std ? static_cast<float*>(xolp_std) + b0 * 2 * Pout : nullptr;
        float* nrm_b = normals ? static_cast<float*>(normals) + b0 * 9 * Pout : nullptr;
        int* ints_b = ints ? static_cast<int*>(ints) + b0 * 5 * Pout : nullptr;
        auto go = [&](auto kern) -> int {
            int r = set_lds_limit(kern, lds);
            if (r) return r;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(nth), lds, st, pol_b, mask_b, xolp_b, std_b, nrm_b,
                               ints_b, static_cast<const char*>(tables), g);
            return PD_OK;
        };
        if (mode == PD_POLAR_LS) {
            if (!need_normals) rc = go(polar_kernel<PD_POLAR_LS, false, false>);
            else rc = precise ? go(polar_kernel<PD_POLAR_LS, true, true>) : go(polar_kernel<PD_POLAR_LS, true, false>);
        } else {
            if (!need_normals) rc = go(polar_kernel<PD_POLAR_STOKES, false, false>);
            else rc = precise ? go(polar_kernel<PD_POLAR_STOKES, true, true>) : go(polar_kernel<PD_POLAR_STOKES, true, false>);
        }
    }
    if (rc) return rc;
    return pd::check_launch("pd_polar_fwd");
}
