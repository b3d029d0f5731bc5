// K1 -- fused Stokes / DoLP / AoLP / physical-normals kernel for gfx950 (MI355X).
//
// One pass over the four uint8 polarizer planes produces, per pixel,
//   DoLP rho, AoLP phi                       (polarisation/xolp.py:8-34, canonical closed form)
//   standardised XOLP                        (manydepth/networks/pre_encoders.py:78-79)
//   theta_diffuse, theta_spec1, theta_spec2  (manydepth/normals_vec.py:11-50, scipy _call_linear)
//   the 9-channel physical normals           (manydepth/normals_vec.py:53-60, pre_encoders.py:99-113)
// Memory-bound by design: 4 B/px read, 8..80 B/px written, every access a full 4-byte (loads) or 16-byte
// (stores) per-lane vector on planar NCHW tensors; outputs leave with nontemporal stores (a write-once
// stream: +20 % on this access shape, tools/membench2.hip), the next quad's planes are loaded before the
// current quad is computed.
//
// Exactness strategy
//   * rho: the reference's fp64 op sequence (xolp.py:22-27) rounded once to fp32 -> bit-equal to the CPU
//     restatement (Newton-refined hardware seeds + Ziv's rounding test, literal IEEE sequence near midpoints).
//   * phi: depends only on the integer pair (d1,d2) = (I0-I90, I45-I135) in [-255,255]^2; LUTs built on the
//     host in fp64 (L2-resident) give the exactly rounded value through integer indexing: a 1 MB fp32 LUT
//     (phi) and a 4 MB float4 LUT (phi, cos(fl32 phi), sin(fl32 phi)) for the fast normals.
//   * theta tables: searchsorted-left on fp32 keys floor32(x[i]) is exact for an fp32 query.  All three tables
//     are searched with ONE 16-byte LDS read: a monotone, exactly reproducible bucket function of rho (integer
//     approximation of sqrt(rho): the tables are ~quadratic in theta) selects a record holding, per table, the
//     number of keys in lower buckets and the single key inside the bucket; index = base + (key < rho), with
//     scipy's clip(1, n-1) folded in by the host.  Buckets holding two or more keys of one table (rho within
//     5e-3 of the specular maximum and a sliver near 0.004) are flagged and take exact binary searches.
//   * normals, default (fast) path: theta - j*pi/2 = slope*(rho - x_lo) + c in fp32 with a two-float constant
//     per bin (the bin knows its quadrant j), sin/cos by degree-9/8 polynomials on |r| < 0.79, products in fp32;
//     |err| <= ~3e-7.  rho beyond the tables (extrapolation, |r| >= 0.79) and PD_POLAR_PRECISE_NORMALS take
//     the fp64 path: angle addition of tabulated sin/cos(y_lo) with a Taylor series of the sub-step, or a
//     3-term Cody-Waite reduction + fdlibm kernels (< 1 ulp for |theta| < 1.6e6).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include "pd_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x50444c36u;   // "PDL6"
constexpr int kNB = 4096;                  // buckets of the one-read search
constexpr uint32_t kSqrtMagic = 0x1fbd1df5u;   // as_float((bits(x) >> 1) + magic) ~ sqrt(x), monotone, integer-exact
constexpr int kLutSide = 511;
constexpr int kLutCount = kLutSide * kLutSide;
constexpr int kMaxNodes = 1023;            // 10-bit bases in the bucket record

struct PolarHeader {   // 64 bytes, little endian
    uint32_t magic;
    int32_t n_d, n_s1, n_s2;
    uint32_t off_lut;      // float[511*511]            phi
    uint32_t off_lut4;     // float[511*511][4]         phi, cos(fl32 phi), sin(fl32 phi), 0
    uint32_t off_img_fast;     // LDS image of the fast kernels:    buckets | keys | float4 fbins[nk] | pad to 1 KiB
    uint32_t img_fast_bytes;
    uint32_t off_img_precise;  // LDS image of the precise kernels: buckets | keys | double bins[nk][4] | pad to 1 KiB
    uint32_t img_precise_bytes;
    uint32_t common_bytes;     // buckets + keys = offset of the bins inside either image
    uint32_t total_bytes;
    float bscale;          // bucket = min(int(approx_sqrt(rho) * bscale), kNB - 1)
    uint32_t n_buckets;
    uint32_t off_ylo;          // double[nk]: y_lo of every bin (entry i = y[i-1]), for the theta outputs of pd_polar_theta
    uint32_t pad;
};
static_assert(sizeof(PolarHeader) == 64, "header size");

// Bucket record (16 bytes): x, y, z = the key (fp32 bits) of the diffuse / spec1 / spec2 table that lies inside the
// bucket (+inf when none, or when clip(1, n-1) makes the comparison irrelevant); w = base_d | base_s1 << 10 |
// base_s2 << 20 | multi << 30, base_T = clip(#{keys of T in lower buckets}, 1, n_T - 1).
// keys[]: the three tables' keys floor32(x) back to back (diffuse | spec1 | spec2), for the flagged buckets.
// Bin entry i of a table describes scipy's bin idx == i (x_lo = x[i-1]); entry 0 is unused:
//   fast bins    float[nk][4]   x_lo32, slope32, c_hi, c_lo (bit 0 = quadrant j)
//   precise bins double[nk][4]  x_lo, slope, sin(y_lo), cos(y_lo)
// Each image is one contiguous, 1-KiB-padded range of the blob: a workgroup copies it to LDS with direct-to-LDS
// loads (1 KiB per wave instruction).
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline size_t keys_padded(int nk) { return round_up(size_t(nk), 4); }
inline size_t common_bytes_for(int nk) { return size_t(kNB) * 16 + keys_padded(nk) * 4; }
inline size_t lut_bytes() { return round_up(size_t(kLutCount) * 4, 16); }
inline size_t lut4_bytes() { return size_t(kLutCount) * 16; }
inline size_t img_bytes_for(int nk, bool precise) { return round_up(common_bytes_for(nk) + size_t(nk) * (precise ? 32 : 16), 1024); }
inline size_t ylo_bytes_for(int nk) { return round_up(size_t(nk) * 8, 16); }

float floor32(double x) {  // largest fp32 <= x
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) f = nextafterf(f, -INFINITY);
    return f;
}

// The bucket function, bit-identical on host and device (integer shift/add, one IEEE multiply, truncation).
__host__ __device__ inline float approx_sqrt_bits(float rho) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(rho);
#else
    memcpy(&u, &rho, 4);
#endif
    if (static_cast<int32_t>(u) < 0) u = 0;          // negative values (and -0) sort below every key
    u = (u >> 1) + kSqrtMagic;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float t; memcpy(&t, &u, 4); return t;
#endif
}
__host__ __device__ inline int bucket_of(float rho, float bscale) {
    const float v = approx_sqrt_bits(rho) * bscale;
    return static_cast<int>(v < static_cast<float>(kNB - 1) ? v : static_cast<float>(kNB - 1));
}

}  // namespace

extern "C" size_t pd_polar_tables_bytes(int n_d, int n_s1, int n_s2) {
    if (n_d < 2 || n_s1 < 2 || n_s2 < 2) return 0;
    const int nk = n_d + n_s1 + n_s2;
    return sizeof(PolarHeader) + lut_bytes() + lut4_bytes() + img_bytes_for(nk, false) + img_bytes_for(nk, true) + ylo_bytes_for(nk);
}

extern "C" int pd_polar_tables_pack(const double* x_d, const double* y_d, int n_d,
                                    const double* x_s1, const double* y_s1, int n_s1,
                                    const double* x_s2, const double* y_s2, int n_s2,
                                    void* host_blob, size_t blob_bytes) {
    PD_REQUIRE(x_d && y_d && x_s1 && y_s1 && x_s2 && y_s2 && host_blob, "pd_polar_tables_pack: null pointer");
    PD_REQUIRE(n_d >= 2 && n_s1 >= 2 && n_s2 >= 2, "pd_polar_tables_pack: each table needs >= 2 nodes");
    PD_REQUIRE(n_d <= kMaxNodes && n_s1 <= kMaxNodes && n_s2 <= kMaxNodes,
               "pd_polar_tables_pack: table too large (at most %d nodes each)", kMaxNodes);
    size_t need = pd_polar_tables_bytes(n_d, n_s1, n_s2);
    PD_REQUIRE(blob_bytes >= need, "pd_polar_tables_pack: blob too small (%zu < %zu)", blob_bytes, need);
    char* base = static_cast<char*>(host_blob);
    memset(base, 0, need);
    const int nk = n_d + n_s1 + n_s2;
    PolarHeader h{};
    h.magic = kMagic; h.n_d = n_d; h.n_s1 = n_s1; h.n_s2 = n_s2;
    h.off_lut = sizeof(PolarHeader);
    h.off_lut4 = h.off_lut + uint32_t(lut_bytes());
    h.off_img_fast = h.off_lut4 + uint32_t(lut4_bytes());
    h.img_fast_bytes = uint32_t(img_bytes_for(nk, false));
    h.off_img_precise = h.off_img_fast + h.img_fast_bytes;
    h.img_precise_bytes = uint32_t(img_bytes_for(nk, true));
    h.common_bytes = uint32_t(common_bytes_for(nk));
    h.off_ylo = h.off_img_precise + h.img_precise_bytes;
    h.total_bytes = uint32_t(need);
    h.n_buckets = kNB;
    PD_REQUIRE(h.img_precise_bytes <= 160 * 1024, "pd_polar_tables_pack: tables exceed the 160 KB LDS of a CU");

    // AoLP LUTs: phi = 0.5 * atan2(x2, x1), x1 = d1/2, x2 = d2/2 (xolp.py:30), fp64 -> fp32; the float4 LUT adds
    // cos/sin of the fp32 AoLP (torch evaluates them on the fp32 tensor, normals_vec.py:56-57), correctly rounded.
    float* lut = reinterpret_cast<float*>(base + h.off_lut);
    float* lut4 = reinterpret_cast<float*>(base + h.off_lut4);
    for (int d2 = -255; d2 <= 255; ++d2)
        for (int d1 = -255; d1 <= 255; ++d1) {
            const int i = (d2 + 255) * kLutSide + (d1 + 255);
            const float phi = static_cast<float>(0.5 * atan2(d2 / 2.0, d1 / 2.0));
            lut[i] = phi;
            lut4[4 * i] = phi;
            lut4[4 * i + 1] = static_cast<float>(cos(static_cast<double>(phi)));
            lut4[4 * i + 2] = static_cast<float>(sin(static_cast<double>(phi)));
            lut4[4 * i + 3] = 0.f;
        }

    const double* xs[3] = {x_d, x_s1, x_s2};
    const double* ys[3] = {y_d, y_s1, y_s2};
    const int ns[3] = {n_d, n_s1, n_s2};
    uint32_t* buckets = reinterpret_cast<uint32_t*>(base + h.off_img_fast);
    float* keys = reinterpret_cast<float*>(base + h.off_img_fast + size_t(kNB) * 16);
    float* fbins = reinterpret_cast<float*>(base + h.off_img_fast + h.common_bytes);
    double* bins = reinterpret_cast<double*>(base + h.off_img_precise + h.common_bytes);
    double* ylo = reinterpret_cast<double*>(base + h.off_ylo);
    float kmax = 0.f;
    int o = 0;
    for (int t = 0; t < 3; ++t) {
        for (int i = 0; i < ns[t]; ++i) {
            PD_REQUIRE(i == 0 || xs[t][i] >= xs[t][i - 1], "pd_polar_tables_pack: x not ascending (table %d)", t);
            PD_REQUIRE(xs[t][i] >= 0.0, "pd_polar_tables_pack: negative node (table %d)", t);
            keys[o + i] = floor32(xs[t][i]);
            if (std::isfinite(keys[o + i])) kmax = std::max(kmax, keys[o + i]);
            if (i >= 1) {  // scipy _call_linear: slope = (y_hi - y_lo) / (x_hi - x_lo); y = slope*(x - x_lo) + y_lo
                double* bin = bins + size_t(o + i) * 4;
                const double x_lo = xs[t][i - 1], y_lo = ys[t][i - 1];
                const double slope = (ys[t][i] - y_lo) / (xs[t][i] - x_lo);
                bin[0] = x_lo; bin[1] = slope; bin[2] = sin(y_lo); bin[3] = cos(y_lo);
                ylo[o + i] = y_lo;
                // fast bin: theta - j*pi/2 = slope*(rho - x_lo32) + c,  c = y_lo - slope*(x_lo - x_lo32) - j*pi/2
                float* fb = fbins + size_t(o + i) * 4;
                const float x32 = floor32(x_lo);
                const int j = 0.5 * (y_lo + ys[t][i]) > M_PI / 4 ? 1 : 0;
                const double c = y_lo - slope * (x_lo - static_cast<double>(x32)) - j * (M_PI / 2);
                const float c_hi = static_cast<float>(c);
                float c_lo = static_cast<float>(c - static_cast<double>(c_hi));
                uint32_t lo_bits; memcpy(&lo_bits, &c_lo, 4);
                lo_bits = (lo_bits & ~1u) | uint32_t(j);        // the quadrant rides in the last bit of a ~1e-9 term
                memcpy(&c_lo, &lo_bits, 4);
                fb[0] = x32; fb[1] = static_cast<float>(slope); fb[2] = c_hi; fb[3] = c_lo;
            }
        }
        o += ns[t];
    }
    PD_REQUIRE(kmax > 0.f, "pd_polar_tables_pack: all nodes are zero");
    for (size_t p = size_t(nk); p < keys_padded(nk); ++p) keys[p] = INFINITY;
    // bucket scale: the largest finite key lands in bucket kNB - 2, everything above it in kNB - 1
    float bscale = static_cast<float>((kNB - 2) / static_cast<double>(approx_sqrt_bits(kmax)));
    while (bucket_of(kmax, bscale) > kNB - 2) bscale = nextafterf(bscale, 0.f);
    h.bscale = bscale;
    memcpy(base, &h, sizeof(h));

    std::vector<int> cnt(size_t(3) * kNB, 0);          // keys per (table, bucket)
    std::vector<float> first(size_t(3) * kNB, INFINITY);
    o = 0;
    for (int t = 0; t < 3; ++t) {
        for (int i = 0; i < ns[t]; ++i) {
            const float k = keys[o + i];
            const int b = std::isfinite(k) ? bucket_of(k, bscale) : kNB - 1;
            if (cnt[size_t(t) * kNB + b]++ == 0) first[size_t(t) * kNB + b] = k;
        }
        o += ns[t];
    }
    int below[3] = {0, 0, 0};
    for (int b = 0; b < kNB; ++b) {
        uint32_t rec[4] = {0, 0, 0, 0};
        for (int t = 0; t < 3; ++t) {
            const int c = cnt[size_t(t) * kNB + b];
            float key = c >= 1 ? first[size_t(t) * kNB + b] : INFINITY;
            if (c >= 2) rec[3] |= 1u << 30;
            // index = clip(below + (key < rho), 1, n - 1): fold the clip into base / key
            const int lo = std::min(std::max(below[t], 1), ns[t] - 1);
            const int hi = std::min(std::max(below[t] + (c >= 1 ? 1 : 0), 1), ns[t] - 1);
            if (hi == lo) key = INFINITY;
            memcpy(&rec[t], &key, 4);
            rec[3] |= uint32_t(lo) << (10 * t);
            below[t] += c;
        }
        memcpy(buckets + size_t(b) * 4, rec, 16);
    }
    memcpy(base + h.off_img_precise, base + h.off_img_fast, h.common_bytes);   // the precise image starts with the same part
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

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

// write-once output stream: nontemporal 16-byte stores (tools/membench2.hip: 6.0-6.6 TB/s vs 4.9-5.4 TB/s plain)
template <bool NT>
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    const f4 v = {a, b, c, d};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
    else *reinterpret_cast<f4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void store4i(int* p, int a, int b, int c, int d) {
    const i4 v = {a, b, c, d};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<i4*>(p));
    else *reinterpret_cast<i4*>(p) = v;
}

struct Tabs {  // view of the interpolation tables
    const uint4* buckets;       // LDS
    const float* keys;          // LDS
    const float4* fbins;        // LDS (fast) -- [nk] x_lo32, slope32, c_hi, c_lo|j
    const double* bins;         // [nk][4]: x_lo, slope, sin(y_lo), cos(y_lo); LDS when PRECISE, else global
    float bscale;
    int n_d, n_s1, n_s2;
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_void_t;

// The ~100 KB table image of a workgroup, L2 -> LDS by direct-to-LDS loads: one wave instruction moves 1 KiB
// (wave-uniform LDS base + lane * 16), nothing passes through registers and all pieces of a wave are in flight
// together (a copy loop through registers ran as a chain of L2 round trips: 7 us per launch).
template <bool PRECISE>
__device__ __forceinline__ void stage_tables_issue(const char* __restrict__ blob, char* smem, int nthreads, Tabs& t) {
    const PolarHeader* h = reinterpret_cast<const PolarHeader*>(blob);
    t.n_d = h->n_d; t.n_s1 = h->n_s1; t.n_s2 = h->n_s2;
    const char* src = blob + (PRECISE ? h->off_img_precise : h->off_img_fast);
    const int npieces = static_cast<int>((PRECISE ? h->img_precise_bytes : h->img_fast_bytes) >> 10);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = nthreads >> 6;
    for (int p = wave; p < npieces; p += nwaves)
        __builtin_amdgcn_global_load_lds((global_void_t*)(src + (static_cast<size_t>(p) << 10) + lane * 16),
                                         (lds_void_t*)(smem + (static_cast<size_t>(p) << 10)), 16, 0, 0);
    t.buckets = reinterpret_cast<const uint4*>(smem);
    t.keys = reinterpret_cast<const float*>(smem + size_t(kNB) * 16);
    t.fbins = reinterpret_cast<const float4*>(smem + h->common_bytes);
    t.bins = PRECISE ? reinterpret_cast<const double*>(smem + h->common_bytes)
                     : reinterpret_cast<const double*>(blob + h->off_img_precise + h->common_bytes);
    t.bscale = h->bscale;
}
template <bool PRECISE>
__device__ __forceinline__ void stage_tables(const char* __restrict__ blob, char* smem, int nthreads, Tabs& t) {
    stage_tables_issue<PRECISE>(blob, smem, nthreads, t);
    __syncthreads();
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

// fp32 polynomial kernels on |r| <= pi/4 (+ a table step): ~1 ulp
__device__ __forceinline__ void sincos_kernel_f32(float r, float& sn, float& cs) {
    const float z = r * r;
    float ps = fmaf(z, 2.7183114939898219e-06f, -1.9839334836096632e-04f);
    ps = fmaf(z, ps, 8.3333095718939529e-03f);
    sn = fmaf(r * z, fmaf(z, ps, -1.6666665459843126e-01f), r);
    float pc = fmaf(z, 2.4390448796277409e-05f, -1.3887316255677255e-03f);
    pc = fmaf(z, pc, 4.1666645683222281e-02f);
    cs = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
}

// fp32 sin/cos for |x| <= 4 (AoLP in [-pi/2, pi/2], AoLP + pi/2 in [0, pi]); ~1 ulp
__device__ __forceinline__ void sincos_f32(float x, float& s, float& c) {
    const float j = rintf(x * 0.63661977236758134f);
    float r = fmaf(-j, 1.5707962512969971f, x);       // pi/2 hi (24 bits)
    r = fmaf(-j, 7.5497894158615964e-08f, r);         // pi/2 lo
    float sn, cs;
    sincos_kernel_f32(r, sn, cs);
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

// Out of line: the fp64 evaluation for the (rare in real frames) pixels whose rho lies beyond a table.
// (returns by value: reference outputs of a non-inlined function live in scratch memory, and the caller's reloads
// -- on the hot path, behind every table -- wait for every older load of the wave)
__device__ __noinline__ float2 sincos_bin_slow(const double* __restrict__ gbin, float rho) {
    double sd, cd;
    sincos_bin(gbin, static_cast<double>(rho), sd, cd);
    return make_float2(static_cast<float>(sd), static_cast<float>(cd));
}

// Fast path: r = theta - j*pi/2 in fp32 from the two-float bin constant ((rho - x_lo32) is exact or carries a
// 6e-8 relative error on a term <= 1.6e-3), polynomial kernels, quadrant swap.  |err| ~1e-7.
__device__ __forceinline__ void sincos_bin_fast(const float4 fb, const double* __restrict__ gbin, float rho, float& s, float& c) {
    const float r = fmaf(fb.y, rho - fb.x, fb.z) + fb.w;
    if (fabsf(r) < 0.79f) {
        float sn, cs;
        sincos_kernel_f32(r, sn, cs);
        const bool j = (__float_as_uint(fb.w) & 1u) != 0;
        s = j ? cs : sn;
        c = j ? -sn : cs;
    } else {
        const float2 sc = sincos_bin_slow(gbin, rho);
        s = sc.x;
        c = sc.y;
    }
}

__device__ __forceinline__ int search_left(const float* __restrict__ keys, int n, float rho) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < rho) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The three scipy bin indices clip(searchsorted(x, rho, 'left'), 1, n-1) of one pixel: one 16-byte LDS read.
// GENERIC: rho may be NaN (sorts last, numpy) -- Stokes mode (0/0) and caller-supplied XOLP tensors.
template <bool GENERIC>
__device__ __forceinline__ void lookup3(float rho, const Tabs& t, int (&bi)[3]) {
    const uint4 rec = t.buckets[bucket_of(rho, t.bscale)];
    bi[0] = static_cast<int>(rec.w & 1023u) + (__uint_as_float(rec.x) < rho ? 1 : 0);
    bi[1] = static_cast<int>((rec.w >> 10) & 1023u) + (__uint_as_float(rec.y) < rho ? 1 : 0);
    bi[2] = static_cast<int>((rec.w >> 20) & 1023u) + (__uint_as_float(rec.z) < rho ? 1 : 0);
    if (rec.w >> 30) {   // two or more keys of one table in this bucket: exact binary searches
        bi[0] = min(max(search_left(t.keys, t.n_d, rho), 1), t.n_d - 1);
        bi[1] = min(max(search_left(t.keys + t.n_d, t.n_s1, rho), 1), t.n_s1 - 1);
        bi[2] = min(max(search_left(t.keys + t.n_d + t.n_s1, t.n_s2, rho), 1), t.n_s2 - 1);
    }
    if (GENERIC && rho != rho) { bi[0] = t.n_d - 1; bi[1] = t.n_s1 - 1; bi[2] = t.n_s2 - 1; }
}

// theta lookups + the three physical normals of one pixel (normals_vec.py:11-60, pre_encoders.py:99-113).
// PRECISE: fp64 theta trig, cos/sin(fl32(phi + pi/2)) evaluated like the reference.  Otherwise (cp, sp) =
// cos/sin(phi) are given (LUT) or computed, and cos(phi + pi/2) = -sin(phi), sin(phi + pi/2) = cos(phi)
// (within 1.2e-7 of the reference's evaluation at the rounded sum).
template <bool PRECISE, bool GENERIC>
__device__ __forceinline__ void normals9(float rho, float phi, float cp, float sp, const Tabs& t, float (&v)[9], int (&bi)[3]) {
    lookup3<GENERIC>(rho, t, bi);
    const int i0 = bi[0], i1 = t.n_d + bi[1], i2 = t.n_d + t.n_s1 + bi[2];
    if (PRECISE) {
        const float kHalfPi = static_cast<float>(1.5707963267948966);
        const double x = static_cast<double>(rho);
        double sd, cd, s1, c1, s2, c2;
        sincos_bin(t.bins + 4 * i0, x, sd, cd);
        sincos_bin(t.bins + 4 * i1, x, s1, c1);
        sincos_bin(t.bins + 4 * i2, x, s2, c2);
        float sq, cq;
        sincos_f32(phi, sp, cp);             // torch.cos/sin on the fp32 AoLP
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
        sincos_bin_fast(t.fbins[i0], t.bins + 4 * i0, rho, sd, cd);
        sincos_bin_fast(t.fbins[i1], t.bins + 4 * i1, rho, s1, c1);
        sincos_bin_fast(t.fbins[i2], t.bins + 4 * i2, rho, s2, c2);
        v[0] = cp * sd; v[1] = sp * sd; v[2] = cd;
        v[3] = -sp * s1; v[4] = cp * s1; v[5] = c1;
        v[6] = -sp * s2; v[7] = cp * s2; v[8] = c2;
    }
}

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

// Output variants of the fused kernel
constexpr int OUT_XOLP = 0;      // DoLP / AoLP (+ standardised, + d1/d2) only: no tables in LDS, one quad per thread
constexpr int OUT_FAST = 1;      // + normals, fp32 path (default)
constexpr int OUT_PRECISE = 2;   // + normals, fp64 theta trig (PD_POLAR_PRECISE_NORMALS)

constexpr int kThreads = 512;    // XOLP-only kernels: 4 workgroups per CU
constexpr int kThreadsP = 512;   // fp64 normals: one 8-wave workgroup per CU (2 waves/SIMD, <= 256 VGPRs: no spills)

// Launch geometry: every index is 32-bit (the host splits a batch whose planes would exceed 2^32 bytes).
// One image is Hrows rows of wq_in quads (4 pixels) in and wq_out >= wq_in quads out; with a pitched output
// (Wout > W) a row is an image row, otherwise the whole plane is one row.  A thread walks the OUTPUT quads
// (image, row, quad) by a constant step, so that a wave's 64 quads are one 1-KiB-aligned segment of every
// output plane (full 128-byte lines: walking the input quads instead left two partial lines per store and
// cost 30 % on the 612 -> 640 case); lanes in the padding columns load nothing and store zeros.
struct PolarGeo {
    int B, Hrows, wq_in, wq_out;
    int drow, dcq;                 // persistent-loop step (grid * block quads) as rows + quads
    unsigned P, Pout;              // pixels per input / output plane
    int flags;
#ifdef PD_POLAR_TRACE
    unsigned long long* trace;     // [grid][8] 100 MHz timestamps of wave 0 (tools/k1_trace.py builds this variant)
#endif
};

#ifdef PD_POLAR_TRACE
#define PD_TRACE(slot) do { if (g.trace && threadIdx.x == 0) g.trace[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PD_TRACE(slot) do { } while (0)
#endif

struct QuadPos {
    int b, cq;
    unsigned row;
};

__device__ __forceinline__ void advance(QuadPos& q, const PolarGeo& g) {
    q.cq += g.dcq;
    q.row += static_cast<unsigned>(g.drow);
    if (q.cq >= g.wq_out) { q.cq -= g.wq_out; ++q.row; }
    if (q.row >= static_cast<unsigned>(g.Hrows)) {
        if (q.row < 2u * static_cast<unsigned>(g.Hrows)) { q.row -= static_cast<unsigned>(g.Hrows); ++q.b; }
        else { const unsigned k = q.row / static_cast<unsigned>(g.Hrows); q.row -= k * static_cast<unsigned>(g.Hrows); q.b += static_cast<int>(k); }
    }
}

// HOT: the training step's output set (xolp + normals, nothing else) known at compile time -- no pointer tests
// around the stores, and a store count the compiler can put into its s_waitcnt vmcnt(N).
// LUT4: AoLP, cos, sin from the 4 MB float4 LUT (no trigonometry for phi); otherwise AoLP from the 1 MB LUT + fp32
// polynomials.  The small LUT fits every XCD's L2 and is pulled in by a prologue prefetch -- inside the training step K1
// starts with cold caches, and the scattered first-touch misses of the big LUT (every XCD fetches every line it needs
// from HBM, behind K1's own write stream) stretched every iteration of the loop, not just the first.
template <int MODE, int OUT, int NTH, bool NT, bool HOT = false, bool LUT4 = false, bool NTL = false>
__global__ __launch_bounds__(NTH) void polar_kernel(
    const uint8_t* __restrict__ pol, const uint8_t* __restrict__ mask, float* __restrict__ xolp,
    float* __restrict__ xolp_std, float* __restrict__ normals, int* __restrict__ ints,
    const char* __restrict__ blob, const PolarGeo g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool NORMALS = OUT != OUT_XOLP;
    constexpr bool PRECISE = OUT == OUT_PRECISE;
    constexpr bool FAST = OUT == OUT_FAST;
    const PolarHeader* h = reinterpret_cast<const PolarHeader*>(blob);
    const float* lut = reinterpret_cast<const float*>(blob + h->off_lut);
    const float4* lut4 = reinterpret_cast<const float4*>(blob + h->off_lut4);
    const bool ieee = (g.flags & PD_POLAR_IEEE_RHO) != 0;
    const bool w_xolp = HOT || xolp != nullptr, w_std = !HOT && xolp_std != nullptr;
    const bool w_normals = HOT || normals != nullptr, w_ints = !HOT && ints != nullptr;

    // Software pipeline: while quad i is computed, the AoLP gathers of quad i+1 and the plane loads of quads i+2 .. i+4
    // are in flight.  Inside the training step the LUT is never cache-resident when K1 starts (the
    // step streams gigabytes between two launches) and different image regions need different LUT lines, so a
    // wave meets HBM-latency gathers in every iteration: issued one iteration ahead they cost nothing
    // (tools/membench2.hip d, tools/k1_instep_probe.py).  Queue order per iteration: gathers(i+1), loads(i+2),
    // stores(i) -- nothing waits for a store.  The gathers are unconditional (quads past the end and padding
    // lanes read the centre entry): a conditional load would be waited for at the join.
    struct Words { uint32_t w0, w45, w90, w135; };
    const unsigned in_row = 4u * g.wq_in, out_row = 4u * g.wq_out;
    auto load_words = [&](const QuadPos& p) -> Words {
        Words w = {0u, 0u, 0u, 0u};
        if (p.b < g.B && p.cq < g.wq_in) {
            const unsigned p4 = static_cast<unsigned>(p.b) * 4u * g.P + p.row * in_row + 4u * p.cq;
            // NTL: the nontemporal hint on this read-once stream (no L2 / Infinity-Cache allocation that would push out the LUT and
            // tables).  Measured both ways (bench.py, tools/bench_polar.py): with the hint the kernel starts faster from the cache
            // state the training step leaves behind (B = 16 inside the step: 57-64 us against 68-73) but runs a third slower in
            // steady state (B = 128 back to back: 3.4-3.7 against 5.1-5.9 TB/s) -- the host picks it for short launches only.
            if (NTL) {
                w.w0 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pol + p4));
                w.w45 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pol + (p4 + g.P)));
                w.w90 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pol + (p4 + 2u * g.P)));
                w.w135 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pol + (p4 + 3u * g.P)));
            } else {
                w.w0 = *reinterpret_cast<const uint32_t*>(pol + p4);
                w.w45 = *reinterpret_cast<const uint32_t*>(pol + (p4 + g.P));
                w.w90 = *reinterpret_cast<const uint32_t*>(pol + (p4 + 2u * g.P));
                w.w135 = *reinterpret_cast<const uint32_t*>(pol + (p4 + 3u * g.P));
            }
        }
        return w;
    };
    struct Gather { float phi[4], cp[4], sp[4]; };
    auto gather = [&](const Words& w) -> Gather {
        Gather G;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sh = 8 * j;
            const float e1 = static_cast<float>((w.w0 >> sh) & 0xffu) - static_cast<float>((w.w90 >> sh) & 0xffu);
            const float e2 = static_cast<float>((w.w45 >> sh) & 0xffu) - static_cast<float>((w.w135 >> sh) & 0xffu);
            // (d2 + 255) * 511 + (d1 + 255), exact in fp32
            const unsigned idx = static_cast<unsigned>(static_cast<int>(fmaf(e2, 511.f, e1 + 130560.f)));
            if (FAST && LUT4) { const float4 L = lut4[idx]; G.phi[j] = L.x; G.cp[j] = L.y; G.sp[j] = L.z; }
            else { G.phi[j] = lut[idx]; G.cp[j] = 1.f; G.sp[j] = 0.f; }
        }
        return G;
    };

    // first quad of this thread; its planes are requested before the tables are staged
    QuadPos q;
    {
        const unsigned q0 = blockIdx.x * static_cast<unsigned>(NTH) + threadIdx.x;
        q.row = q0 / static_cast<unsigned>(g.wq_out);
        q.cq = static_cast<int>(q0 - q.row * static_cast<unsigned>(g.wq_out));
        q.b = static_cast<int>(q.row / static_cast<unsigned>(g.Hrows));
        q.row -= static_cast<unsigned>(q.b) * static_cast<unsigned>(g.Hrows);
    }
    PD_TRACE(0);
    // Prologue: every read the first iterations need is requested before anything waits -- the planes of quads 0..3 (at
    // the bench size, B = 16, a thread has five quads: a read burst, then a pure write stream; tools/membench3.hip:
    // interleaving reads with the stores costs 3-7 us of 50 when the inputs come from HBM, as inside the training step),
    // the LUT prefetch, the table image.  (Requesting the table image first changed nothing: the barrier waits for all.)
    Words wc = load_words(q);
    QuadPos qn = q;
    advance(qn, g);
    Words wn = load_words(qn);
    QuadPos qa = qn;
    advance(qa, g);
    Words wa = load_words(qa);
    QuadPos qb = qa;
    advance(qb, g);
    Words wb = load_words(qb);
    Tabs tabs;
    if (NORMALS && !LUT4) {
        // L2 prefetch of the AoLP LUT: the workgroups of an XCD (blockIdx.x % 8, round-robin dispatch) share its L2; workgroup
        // k of the XCD pulls slice k of the 1 MB table through direct-to-LDS loads into a junk slot behind the table image
        // (1 KiB per wave; nothing reads it)
        const unsigned lut_kib = (static_cast<unsigned>(kLutCount) * 4u + 1023u) >> 10;
        const unsigned nslices = (gridDim.x + 7u) >> 3, slice = blockIdx.x >> 3;
        const unsigned per = (lut_kib + nslices - 1u) / nslices;
        const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = NTH >> 6;
        char* junk = smem + (PRECISE ? h->img_precise_bytes : h->img_fast_bytes) + (wave << 10);
        const char* lsrc = reinterpret_cast<const char*>(lut);
        for (unsigned p = slice * per + wave; p < min((slice + 1u) * per, lut_kib - 1u); p += nwaves)   // (last partial KiB skipped)
            __builtin_amdgcn_global_load_lds((global_void_t*)(lsrc + (static_cast<size_t>(p) << 10) + lane * 16),
                                             (lds_void_t*)junk, 16, 0, 0);
    }
    if (NORMALS) stage_tables<PRECISE>(blob, smem, NTH, tabs);
    PD_TRACE(1);                                 // table image landed in LDS (barrier)
    Gather Gc = gather(wc);
#ifdef PD_POLAR_TRACE
    { float sink = Gc.phi[0] + Gc.phi[3]; asm volatile("" :: "v"(sink)); }
    PD_TRACE(2);                                 // first planes + first LUT gathers landed
    int it_ = 0;
#endif

    while (q.b < g.B) {
        const unsigned po = q.row * out_row + 4u * q.cq;     // first output pixel of the quad inside its plane
        const unsigned pin = static_cast<unsigned>(q.b) * g.P + q.row * in_row + 4u * q.cq;
        const int b = q.b;
        const bool pad = q.cq >= g.wq_in;                    // padding columns of a pitched output
        uint32_t wm = 0x01010101u;
        if (MODE == PD_POLAR_STOKES && mask && !pad) wm = *reinterpret_cast<const uint32_t*>(mask + pin);

        const Gather Gn = gather(wn);
        __builtin_amdgcn_sched_barrier(0);
        QuadPos qnn = qb;
        advance(qnn, g);
        const Words wnn = load_words(qnn);
        __builtin_amdgcn_sched_barrier(0);

        float d1[4], d2[4], S[4], s4[4], o_phi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sh = 8 * j;
            // v_cvt_f32_ubyteN: byte -> float in one instruction
            const float f0 = static_cast<float>((wc.w0 >> sh) & 0xffu), f45 = static_cast<float>((wc.w45 >> sh) & 0xffu);
            const float f90 = static_cast<float>((wc.w90 >> sh) & 0xffu), f135 = static_cast<float>((wc.w135 >> sh) & 0xffu);
            d1[j] = f0 - f90;
            d2[j] = f45 - f135;
            s4[j] = fmaf(d1[j], d1[j], d2[j] * d2[j]);                      // <= 130050: exact in fp32
            S[j] = MODE == PD_POLAR_LS ? (f0 + f90) + (f45 + f135) : f0 + f90;
        }

        float o_rho[4], o_n[9][4];
        int o_i[5][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (PRECISE && (j & 1) == 0) __builtin_amdgcn_sched_barrier(0);   // fp64 chains in pairs: register pressure
            const bool on = MODE != PD_POLAR_STOKES || ((wm >> (8 * j)) & 0xffu) != 0;
            float rho, phi = Gc.phi[j], cp = Gc.cp[j], sp = Gc.sp[j];
            if (FAST && !LUT4) sincos_f32(phi, sp, cp);          // ~1 ulp; the 1 MB LUT supplies phi only
            if (MODE == PD_POLAR_STOKES && !on) {   // images are masked first (:117-121); outputs are zero outside
                rho = 0.f; phi = 0.f; cp = 1.f; sp = 0.f;
                o_i[0][j] = 0; o_i[1][j] = 0;
            } else {
                rho = rho_pixel<MODE>(S[j], d1[j], d2[j], s4[j], ieee);
                o_i[0][j] = static_cast<int>(d1[j]);
                o_i[1][j] = static_cast<int>(d2[j]);
            }
            o_rho[j] = rho;
            o_phi[j] = phi;
            if (NORMALS) {
                int bi[3];
                float v[9];
                normals9<PRECISE, MODE == PD_POLAR_STOKES>(rho, phi, cp, sp, tabs, v, bi);
                o_i[2][j] = bi[0]; o_i[3][j] = bi[1]; o_i[4][j] = bi[2];
#pragma unroll
                for (int c = 0; c < 9; ++c) o_n[c][j] = on ? v[c] : 0.f;
            }
        }
        if (pad) {   // all-zero planes gave rho = phi = 0 already; the normals of (0, 0) are (0, 0, 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (NORMALS) {
#pragma unroll
                    for (int c = 0; c < 9; ++c) o_n[c][j] = 0.f;
                    o_i[2][j] = 0; o_i[3][j] = 0; o_i[4][j] = 0;
                }
            }
        }
        if (w_xolp) {
            float* o = xolp + (static_cast<unsigned>(b) * 2u * g.Pout + po);
            store4<NT>(o, o_rho[0], o_rho[1], o_rho[2], o_rho[3]);
            store4<NT>(o + g.Pout, o_phi[0], o_phi[1], o_phi[2], o_phi[3]);
        }
        if (w_std) {
            float* o = xolp_std + (static_cast<unsigned>(b) * 2u * g.Pout + po);
            float sr[4], sf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { sr[j] = pad ? 0.f : standardise(o_rho[j]); sf[j] = pad ? 0.f : standardise(o_phi[j]); }
            store4<NT>(o, sr[0], sr[1], sr[2], sr[3]);
            store4<NT>(o + g.Pout, sf[0], sf[1], sf[2], sf[3]);
        }
        if (NORMALS && w_normals) {
            float* o = normals + (static_cast<unsigned>(b) * 9u * g.Pout + po);
#pragma unroll
            for (unsigned c = 0; c < 9; ++c) store4<NT>(o + c * g.Pout, o_n[c][0], o_n[c][1], o_n[c][2], o_n[c][3]);
        }
        if (w_ints) {
            int* o = ints + (static_cast<unsigned>(b) * 5u * g.Pout + po);
            const unsigned nch = NORMALS ? 5 : 2;
#pragma unroll
            for (unsigned c = 0; c < 5; ++c)
                if (c < nch) store4i<NT>(o + c * g.Pout, o_i[c][0], o_i[c][1], o_i[c][2], o_i[c][3]);
        }
        q = qn; wc = wn; Gc = Gn; qn = qa; wn = wa; qa = qb; wa = wb; qb = qnn; wb = wnn;
#ifdef PD_POLAR_TRACE
        if (it_ < 3) PD_TRACE(3 + it_);          // end of iterations 0, 1, 2 (stores issued, next operands landed)
        ++it_;
#endif
    }
    PD_TRACE(6);
}

// get_normals() on an existing fp32 XOLP tensor (pre_encoders.py:99-113): [B,2,H,W] -> [B,9,H,W]
template <bool PRECISE, int NTH>
__global__ __launch_bounds__(NTH) void normals_from_xolp_kernel(const float* __restrict__ xolp, float* __restrict__ normals,
                                                                const char* __restrict__ blob, long P,
                                                                long quads_per_img, long total_quads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Tabs tabs;
    stage_tables<PRECISE>(blob, smem, NTH, tabs);
    for (long q = blockIdx.x * (long)NTH + threadIdx.x; q < total_quads; q += (long)gridDim.x * NTH) {
        const long b = q / quads_per_img;
        const long p4 = (q - b * quads_per_img) * 4;
        const float4 r4 = *reinterpret_cast<const float4*>(xolp + (b * 2) * P + p4);
        const float4 f4_ = *reinterpret_cast<const float4*>(xolp + (b * 2 + 1) * P + p4);
        const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, ff[4] = {f4_.x, f4_.y, f4_.z, f4_.w};
        float o_n[9][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (PRECISE) __builtin_amdgcn_sched_barrier(0);
            float v[9]; int bi[3];
            float sp = 0.f, cp = 1.f;
            if (!PRECISE) sincos_f32(ff[j], sp, cp);     // torch.cos/sin on the fp32 AoLP
            normals9<PRECISE, true>(rr[j], ff[j], cp, sp, tabs, v, bi);
#pragma unroll
            for (int c = 0; c < 9; ++c) o_n[c][j] = v[c];
        }
        float* o = normals + (b * 9) * P + p4;
#pragma unroll
        for (int c = 0; c < 9; ++c) store4<true>(o + c * P, o_n[c][0], o_n[c][1], o_n[c][2], o_n[c][3]);
    }
}

// rho_diffuse / rho_spec of the reference as they stand (manydepth/normals_vec.py:11-50): fp32 rho -> the three fp64
// angles scipy's interp1d(fill_value="extrapolate") returns, theta = slope * (rho - x_lo) + y_lo evaluated in scipy's
// operation order (_call_linear; no fused multiply-add: this file is built with -ffp-contract=off), so that rho far
// outside a table extrapolates to the same +41 / -135 rad the reference produces.  Optional int32 bin indices.
template <int NTH>
__global__ __launch_bounds__(NTH) void theta_kernel(const float* __restrict__ rho, double* __restrict__ th_d,
                                                    double* __restrict__ th_s1, double* __restrict__ th_s2,
                                                    int* __restrict__ bins_out, const char* __restrict__ blob, long n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Tabs tabs;
    stage_tables<true>(blob, smem, NTH, tabs);
    const PolarHeader* h = reinterpret_cast<const PolarHeader*>(blob);
    const double* ylo = reinterpret_cast<const double*>(blob + h->off_ylo);
    for (long i = blockIdx.x * (long)NTH + threadIdx.x; i < n; i += (long)gridDim.x * NTH) {
        const float r = rho[i];
        int bi[3];
        lookup3<true>(r, tabs, bi);
        const int idx[3] = {bi[0], tabs.n_d + bi[1], tabs.n_d + tabs.n_s1 + bi[2]};
        const double x = static_cast<double>(r);
        double* outs[3] = {th_d, th_s1, th_s2};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (outs[t]) {
                const double* bin = tabs.bins + 4 * idx[t];
                outs[t][i] = bin[1] * (x - bin[0]) + ylo[idx[t]];
            }
            if (bins_out) bins_out[t * n + i] = bi[t];
        }
    }
}

// node count implied by the blob size (monotone in nk)
inline int nk_from_blob_bytes(size_t tables_bytes) {
    const size_t fixed = sizeof(PolarHeader) + lut_bytes() + lut4_bytes();
    for (int nk = 6; nk <= 3 * kMaxNodes; ++nk)
        if (fixed + img_bytes_for(nk, false) + img_bytes_for(nk, true) + ylo_bytes_for(nk) == tables_bytes) return nk;
    return 0;
}

// (once per kernel and size: the attribute call is not allowed while a stream is being captured into a hipGraph)
template <typename K>
int set_lds_limit(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return PD_OK;
    static std::mutex mu;
    static std::map<const void*, size_t> done;
    const void* key = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find(key);
    if (it != done.end() && it->second >= lds) return PD_OK;
    if (hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return pd::fail(PD_ELAUNCH, "pd_polar: cannot reserve %zu bytes of LDS", lds);
    done[key] = lds;
    return PD_OK;
}

// workgroup size of the fast-normals kernel: 1024 = 4 waves per SIMD on one workgroup per CU (256 and 512 were measured
// and dropped: profiles/r02_*, r03_polar_kernel_gbps.log)
constexpr int fast_threads() { return 1024; }

}  // namespace

#ifdef PD_POLAR_TRACE
static unsigned long long* pd_polar_trace_buffer = nullptr;
extern "C" void pd_polar_set_trace(void* buf) { pd_polar_trace_buffer = static_cast<unsigned long long*>(buf); }
#endif

extern "C" int pd_polar_normals_from_xolp(const void* xolp, void* normals, const void* tables, size_t tables_bytes,
                                          int B, int H, int W, int flags, void* stream) {
    PD_REQUIRE(B >= 0 && H > 0 && W > 0, "pd_polar_normals_from_xolp: bad shape");
    const long P = (long)H * W;
    PD_REQUIRE(P % 4 == 0, "pd_polar_normals_from_xolp: H*W=%ld must be a multiple of 4", P);
    if (B == 0) return PD_OK;
    PD_REQUIRE(xolp && normals && tables, "pd_polar_normals_from_xolp: null pointer");
    PD_REQUIRE(pd::aligned16(xolp) && pd::aligned16(normals) && pd::aligned16(tables), "pd_polar_normals_from_xolp: unaligned");
    const int nk = nk_from_blob_bytes(tables_bytes);
    PD_REQUIRE(nk > 0, "pd_polar_normals_from_xolp: tables blob has an unexpected size");
    const bool precise = (flags & PD_POLAR_PRECISE_NORMALS) != 0;
    const size_t lds = img_bytes_for(nk, precise);
    {
        int rc = precise ? set_lds_limit(normals_from_xolp_kernel<true, kThreadsP>, lds)
                         : set_lds_limit(normals_from_xolp_kernel<false, 1024>, lds);
        if (rc) return rc;
    }
    const long qpi = P / 4, total = qpi * B;
    const int nth = precise ? kThreadsP : 1024;
    long blocks = (total + nth - 1) / nth;
    if (blocks > 256) blocks = 256;
    if (precise)
        hipLaunchKernelGGL((normals_from_xolp_kernel<true, kThreadsP>), dim3((unsigned)blocks), dim3(kThreadsP), lds, (hipStream_t)stream,
                           (const float*)xolp, (float*)normals, (const char*)tables, P, qpi, total);
    else
        hipLaunchKernelGGL((normals_from_xolp_kernel<false, 1024>), dim3((unsigned)blocks), dim3(1024), lds, (hipStream_t)stream,
                           (const float*)xolp, (float*)normals, (const char*)tables, P, qpi, total);
    return pd::check_launch("pd_polar_normals_from_xolp");
}

namespace {
// normals_vec.py:53-60 with torch's type promotion: a factor computed from an fp32 operand is evaluated in fp32 and then
// promoted (cos(phi_f32) * sin(theta_f64) = double(cosf(phi)) * sin(theta))
template <typename TP, typename TT, typename TO>
__global__ __launch_bounds__(256) void calc_normals_kernel(const TP* __restrict__ phi, const TT* __restrict__ theta,
                                                          TO* __restrict__ out, int B, long P) {
    const long total = (long)B * P;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / P, p = i - b * P;
        const TP ph = phi[i];
        const TT th = theta[i];
        const TO c = (TO)cos(ph), s = (TO)sin(ph), st = (TO)sin(th), ct = (TO)cos(th);
        TO* o = out + b * 3 * P + p;
        o[0] = c * st; o[P] = s * st; o[2 * P] = ct;
    }
}
}  // namespace

extern "C" int pd_polar_calc_normals(const void* phi, const void* theta, void* out, int B, long P, int phi_f64, int theta_f64,
                                     void* stream) {
    PD_REQUIRE(B >= 0 && P >= 0, "pd_polar_calc_normals: bad shape");
    if ((long)B * P == 0) return PD_OK;
    PD_REQUIRE(phi && theta && out, "pd_polar_calc_normals: null pointer");
    const long total = (long)B * P;
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 4096);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (phi_f64 && theta_f64)
        hipLaunchKernelGGL((calc_normals_kernel<double, double, double>), dim3(grid), dim3(256), 0, st, (const double*)phi, (const double*)theta, (double*)out, B, P);
    else if (phi_f64)
        hipLaunchKernelGGL((calc_normals_kernel<double, float, double>), dim3(grid), dim3(256), 0, st, (const double*)phi, (const float*)theta, (double*)out, B, P);
    else if (theta_f64)
        hipLaunchKernelGGL((calc_normals_kernel<float, double, double>), dim3(grid), dim3(256), 0, st, (const float*)phi, (const double*)theta, (double*)out, B, P);
    else
        hipLaunchKernelGGL((calc_normals_kernel<float, float, float>), dim3(grid), dim3(256), 0, st, (const float*)phi, (const float*)theta, (float*)out, B, P);
    return pd::check_launch("pd_polar_calc_normals");
}

extern "C" int pd_polar_theta(const void* rho, void* theta_d, void* theta_s1, void* theta_s2, void* bins,
                              const void* tables, size_t tables_bytes, long n, void* stream) {
    PD_REQUIRE(n >= 0, "pd_polar_theta: bad element count");
    if (n == 0) return PD_OK;
    PD_REQUIRE(rho && tables && (theta_d || theta_s1 || theta_s2 || bins), "pd_polar_theta: null pointer / no output");
    const int nk = nk_from_blob_bytes(tables_bytes);
    PD_REQUIRE(nk > 0, "pd_polar_theta: tables blob has an unexpected size");
    const size_t lds = img_bytes_for(nk, true);
    int rc = set_lds_limit(theta_kernel<kThreadsP>, lds);
    if (rc) return rc;
    long blocks = (n + kThreadsP - 1) / kThreadsP;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL((theta_kernel<kThreadsP>), dim3((unsigned)blocks), dim3(kThreadsP), lds, (hipStream_t)stream,
                       (const float*)rho, (double*)theta_d, (double*)theta_s1, (double*)theta_s2, (int*)bins,
                       (const char*)tables, n);
    return pd::check_launch("pd_polar_theta");
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
    const int nk = nk_from_blob_bytes(tables_bytes);
    PD_REQUIRE(nk > 0, "pd_polar_fwd: tables blob too small or of an unexpected size");
    PD_REQUIRE(pd::aligned16(pol) && pd::aligned16(xolp) && pd::aligned16(xolp_std) && pd::aligned16(normals) &&
                   pd::aligned16(ints) && pd::aligned16(tables) && (!mask || pd::aligned16(mask)),
               "pd_polar_fwd: pointers must be 16-byte aligned");
    PD_REQUIRE(xolp || xolp_std || normals || ints, "pd_polar_fwd: no output requested");
    const long Pout = (long)H * Wout;
    const bool pitched = Wout != W;
    const bool need_normals = normals != nullptr || ints != nullptr;
    const bool precise = (flags & PD_POLAR_PRECISE_NORMALS) != 0;
    // (every instantiation stores nontemporally: plain stores are -20 % on this access shape, r02_membench2_*)
    const int nth = need_normals ? (precise ? kThreadsP : fast_threads()) : kThreads;
    // LDS image: 64 KB of bucket records + the keys + 16 (fast) or 32 (precise) bytes per bin, + 1 KiB per wave for the
    // LUT prefetch sink
    const size_t lds = need_normals ? img_bytes_for(nk, precise) + (size_t)(nth / 64) * 1024 : 0;
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
#ifdef PD_POLAR_TRACE
        g.trace = pd_polar_trace_buffer;
#endif
        const long total = (long)nb * (Pout / 4);
        long blocks = (total + nth - 1) / nth;
        // with normals: persistent, one workgroup per CU (the table image is staged once per CU);
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
            if (!need_normals) rc = go(polar_kernel<PD_POLAR_LS, OUT_XOLP, kThreads, true>);
            else if (precise) rc = go(polar_kernel<PD_POLAR_LS, OUT_PRECISE, kThreadsP, true>);
            else if (xolp && normals && !xolp_std && !ints) {
                // the training step's output set; nontemporal plane loads for launches of up to 32 frames' worth of 512x640
                // output (see load_words)
                // (PD_POLAR_NT_LOADS / PD_POLAR_PLAIN_LOADS in `flags` override the size rule: measurement)
                const bool ntl = (flags & PD_POLAR_NT_LOADS) ? true : (flags & PD_POLAR_PLAIN_LOADS) ? false : total * 4 <= 32L * 512 * 640;
                rc = ntl ? go(polar_kernel<PD_POLAR_LS, OUT_FAST, 1024, true, true, false, true>)
                         : go(polar_kernel<PD_POLAR_LS, OUT_FAST, 1024, true, true, false, false>);
            }
            else rc = go(polar_kernel<PD_POLAR_LS, OUT_FAST, 1024, true>);
        } else {
            if (!need_normals) rc = go(polar_kernel<PD_POLAR_STOKES, OUT_XOLP, kThreads, true>);
            else if (precise) rc = go(polar_kernel<PD_POLAR_STOKES, OUT_PRECISE, kThreadsP, true>);
            else rc = go(polar_kernel<PD_POLAR_STOKES, OUT_FAST, 1024, true>);
        }
    }
    if (rc) return rc;
    return pd::check_launch("pd_polar_fwd");
}
