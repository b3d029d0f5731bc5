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
//     query; the per-bin (x_lo, y_lo, slope) triples are precomputed in fp64 with the
//     same operations scipy performs, and live in LDS together with the keys.
//   * normals: cos/sin(phi) in fp32 (torch CPU computes them on the fp32 tensor),
//     promoted and multiplied with the fp64 sin/cos(theta), rounded to fp32.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include "pd_common.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x50444c54u;  // "PDLT"
constexpr int kLutSide = 511;
constexpr int kLutCount = kLutSide * kLutSide;

struct PolarHeader {   // 64 bytes, little endian
    uint32_t magic;
    int32_t n_d, n_s1, n_s2;
    uint32_t off_lut;    // float[511*511]
    uint32_t off_lds;    // start of the LDS image (keys, then bins)
    uint32_t lds_bytes;  // size of the LDS image (multiple of 16)
    uint32_t total_bytes;
    uint32_t pad[8];
};
static_assert(sizeof(PolarHeader) == 64, "header size");

// LDS image layout (all offsets relative to its start):
//   float  keys[nk]                     nk = n_d + n_s1 + n_s2, padded to a multiple of 4
//   double xlo[nk], ylo[nk], slope[nk]  entry i of a table describes bin idx == i (i >= 1)
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int keys_padded(int nk) { return (nk + 3) / 4 * 4; }

float floor32(double x) {  // largest fp32 <= x
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) f = nextafterf(f, -INFINITY);
    return f;
}

}  // namespace

extern "C" size_t pd_polar_tables_bytes(int n_d, int n_s1, int n_s2) {
    if (n_d < 2 || n_s1 < 2 || n_s2 < 2) return 0;
    int nk = n_d + n_s1 + n_s2;
    size_t lds = round_up(size_t(keys_padded(nk)) * 4 + size_t(nk) * 24, 16);
    return sizeof(PolarHeader) + round_up(size_t(kLutCount) * 4, 16) + lds;
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
    int nk = n_d + n_s1 + n_s2;
    h.lds_bytes = uint32_t(round_up(size_t(keys_padded(nk)) * 4 + size_t(nk) * 24, 16));
    h.total_bytes = uint32_t(need);
    memcpy(base, &h, sizeof(h));

    // AoLP LUT: phi = 0.5 * atan2(x2, x1), x1 = d1/2, x2 = d2/2 (xolp.py:30), fp64 -> fp32.
    float* lut = reinterpret_cast<float*>(base + h.off_lut);
    for (int d2 = -255; d2 <= 255; ++d2)
        for (int d1 = -255; d1 <= 255; ++d1)
            lut[(d2 + 255) * kLutSide + (d1 + 255)] =
                static_cast<float>(0.5 * atan2(d2 / 2.0, d1 / 2.0));

    float* keys = reinterpret_cast<float*>(base + h.off_lds);
    double* xlo = reinterpret_cast<double*>(base + h.off_lds + size_t(keys_padded(nk)) * 4);
    double* ylo = xlo + nk;
    double* slope = ylo + nk;
    const double* xs[3] = {x_d, x_s1, x_s2};
    const double* ys[3] = {y_d, y_s1, y_s2};
    const int ns[3] = {n_d, n_s1, n_s2};
    int o = 0;
    for (int t = 0; t < 3; ++t) {
        for (int i = 0; i < ns[t]; ++i) {
            PD_REQUIRE(i == 0 || xs[t][i] >= xs[t][i - 1], "pd_polar_tables_pack: x not ascending (table %d)", t);
            keys[o + i] = floor32(xs[t][i]);
            if (i >= 1) {  // scipy _call_linear: slope = (y_hi - y_lo) / (x_hi - x_lo)
                xlo[o + i] = xs[t][i - 1];
                ylo[o + i] = ys[t][i - 1];
                slope[o + i] = (ys[t][i] - ys[t][i - 1]) / (xs[t][i] - xs[t][i - 1]);
            }
        }
        o += ns[t];
    }
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

struct Tab {  // LDS-resident view of one theta table
    const float* keys;
    const double* xlo;
    const double* ylo;
    const double* slope;
    int n;
};

// searchsorted(x, v, side='left').clip(1, n-1) on fp32 keys (exact for fp32 v).
__device__ __forceinline__ int bin_index(const Tab& t, float v) {
    int pos = 0;  // number of keys strictly less than v
#pragma unroll
    for (int step = 2048; step > 0; step >>= 1) {
        int c = pos + step;
        if (c <= t.n && t.keys[c - 1] < v) pos = c;
    }
    if (v != v) pos = t.n;  // NaN sorts last (numpy)
    return min(max(pos, 1), t.n - 1);
}

__device__ __forceinline__ double interp(const Tab& t, int idx, float v) {
    // y_new = slope * (x_new - x_lo) + y_lo   (two roundings, -ffp-contract=off)
    return t.slope[idx] * (static_cast<double>(v) - t.xlo[idx]) + t.ylo[idx];
}

// theta lookups + the three physical normals of one pixel (normals_vec.py:11-60, pre_encoders.py:99-113)
__device__ __forceinline__ void normals9(float rho, float phi, const Tab& td, const Tab& t1, const Tab& t2,
                                         float (&v)[9], int (&bi)[3]) {
    const float kHalfPi = static_cast<float>(1.5707963267948966);
    bi[0] = bin_index(td, rho); bi[1] = bin_index(t1, rho); bi[2] = bin_index(t2, rho);
    const double thd = interp(td, bi[0], rho);
    const double th1 = interp(t1, bi[1], rho);
    const double th2 = interp(t2, bi[2], rho);
    double sd, cd, s1, c1, s2, c2;
    sincos(thd, &sd, &cd);
    sincos(th1, &s1, &c1);
    sincos(th2, &s2, &c2);
    float sp, cp, sq, cq;
    sincosf(phi, &sp, &cp);             // torch.cos/sin on the fp32 AoLP
    sincosf(phi + kHalfPi, &sq, &cq);   // phi + np.pi/2 evaluated in fp32 (pre_encoders.py:108-109)
    v[0] = static_cast<float>(static_cast<double>(cp) * sd);
    v[1] = static_cast<float>(static_cast<double>(sp) * sd);
    v[2] = static_cast<float>(cd);
    v[3] = static_cast<float>(static_cast<double>(cq) * s1);
    v[4] = static_cast<float>(static_cast<double>(sq) * s1);
    v[5] = static_cast<float>(c1);
    v[6] = static_cast<float>(static_cast<double>(cq) * s2);
    v[7] = static_cast<float>(static_cast<double>(sq) * s2);
    v[8] = static_cast<float>(c2);
}

// stage keys + bins of the three tables into LDS (16-byte copies) and build the views
__device__ __forceinline__ void stage_tables(const char* __restrict__ blob, char* smem, int nthreads, Tab& td, Tab& t1,
                                             Tab& t2) {
    const PolarHeader* h = reinterpret_cast<const PolarHeader*>(blob);
    const uint4* src = reinterpret_cast<const uint4*>(blob + h->off_lds);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const int n16 = h->lds_bytes / 16;
    for (int i = threadIdx.x; i < n16; i += nthreads) dst[i] = src[i];
    const int nk = h->n_d + h->n_s1 + h->n_s2;
    const float* keys = reinterpret_cast<const float*>(smem);
    const double* xlo = reinterpret_cast<const double*>(smem + ((nk + 3) / 4 * 4) * 4);
    const double* ylo = xlo + nk;
    const double* slope = ylo + nk;
    td = Tab{keys, xlo, ylo, slope, h->n_d};
    int o = h->n_d;
    t1 = Tab{keys + o, xlo + o, ylo + o, slope + o, h->n_s1};
    o += h->n_s1;
    t2 = Tab{keys + o, xlo + o, ylo + o, slope + o, h->n_s2};
    __syncthreads();
}

struct Px {
    float rho, phi;
    int d1, d2;
};

template <int MODE>
__device__ __forceinline__ Px xolp_pixel(int i0, int i45, int i90, int i135, const float* __restrict__ lut) {
    Px p;
    p.d1 = i0 - i90;
    p.d2 = i45 - i135;
    double rho;
    if (MODE == PD_POLAR_LS) {
        // x = closed-form least-squares solution; then xolp.py:22-29 literally, in fp64.
        double x0 = static_cast<double>(i0 + i45 + i90 + i135) * 0.25;
        double x1 = static_cast<double>(p.d1) * 0.5;
        double x2 = static_cast<double>(p.d2) * 0.5;
        double r = sqrt(x1 * x1 + x2 * x2);
        double imax = x0 + r;
        double imin = x0 - r;
        rho = (imax - imin) / (imax + imin);
        if (isinf(rho) || isnan(rho)) rho = 0.0;  // rho[rho == inf] = 0; nan_to_num
    } else {
        // physical_normals_channels.py:21-26: rho = sqrt(s1^2 + s2^2) / s0, no guard.
        double s0 = static_cast<double>(i0 + i90);
        double s1 = static_cast<double>(p.d1);
        double s2 = static_cast<double>(p.d2);
        rho = sqrt(s1 * s1 + s2 * s2) / s0;
    }
    p.rho = static_cast<float>(rho);
    p.phi = lut[(p.d2 + 255) * kLutSide + (p.d1 + 255)];
    return p;
}

constexpr int kThreads = 512;

template <int MODE, bool NORMALS>
__global__ __launch_bounds__(kThreads) void polar_kernel(
    const uint8_t* __restrict__ pol, const uint8_t* __restrict__ mask, float* __restrict__ xolp,
    float* __restrict__ xolp_std, float* __restrict__ normals, int* __restrict__ ints,
    const char* __restrict__ blob, long P, long Pout, int wq_in, int wq_out, long quads_per_img, long total_quads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PolarHeader* h = reinterpret_cast<const PolarHeader*>(blob);
    const float* lut = reinterpret_cast<const float*>(blob + h->off_lut);
    Tab td, t1, t2;
    if (NORMALS) stage_tables(blob, smem, kThreads, td, t1, t2);
    const float kMean = static_cast<float>(0.08693199701957657);
    const float kStd = static_cast<float>(0.44430732785457433);

    for (long q = blockIdx.x * (long)kThreads + threadIdx.x; q < total_quads; q += (long)gridDim.x * kThreads) {
        const long b = q / quads_per_img;
        const long ro = q - b * quads_per_img;          // quad index inside the (pitched) output plane
        const long row = ro / wq_out;
        const int cq = (int)(ro - row * wq_out);
        const long po = ro * 4;                          // first output pixel of the quad
        if (cq >= wq_in) {                               // right padding columns of a pitched output: zeros
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xolp) { *reinterpret_cast<float4*>(xolp + (b * 2) * Pout + po) = z; *reinterpret_cast<float4*>(xolp + (b * 2 + 1) * Pout + po) = z; }
            if (xolp_std) { *reinterpret_cast<float4*>(xolp_std + (b * 2) * Pout + po) = z; *reinterpret_cast<float4*>(xolp_std + (b * 2 + 1) * Pout + po) = z; }
            if (NORMALS && normals)
                for (int c = 0; c < 9; ++c) *reinterpret_cast<float4*>(normals + (b * 9 + c) * Pout + po) = z;
            if (ints)
                for (int c = 0; c < (NORMALS ? 5 : 2); ++c) *reinterpret_cast<int4*>(ints + (b * 5 + c) * Pout + po) = make_int4(0, 0, 0, 0);
            continue;
        }
        const long p4 = (row * wq_in + cq) * 4;          // first input pixel of the quad inside its plane
        const uint8_t* pb = pol + (b * 4) * P + p4;
        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(pb);
        const uint32_t w45 = *reinterpret_cast<const uint32_t*>(pb + P);
        const uint32_t w90 = *reinterpret_cast<const uint32_t*>(pb + 2 * P);
        const uint32_t w135 = *reinterpret_cast<const uint32_t*>(pb + 3 * P);
        uint32_t wm = 0x01010101u;
        if (MODE == PD_POLAR_STOKES && mask) wm = *reinterpret_cast<const uint32_t*>(mask + b * P + p4);

        float o_rho[4], o_phi[4], o_n[9][4];
        int o_i[5][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sh = 8 * j;
            const bool on = ((wm >> sh) & 0xffu) != 0;
            int i0 = (w0 >> sh) & 0xff, i45 = (w45 >> sh) & 0xff, i90 = (w90 >> sh) & 0xff, i135 = (w135 >> sh) & 0xff;
            if (MODE == PD_POLAR_STOKES && !on) { i0 = i45 = i90 = i135 = 0; }  // images are masked first (:117-121)
            Px p = xolp_pixel<MODE>(i0, i45, i90, i135, lut);
            if (MODE == PD_POLAR_STOKES && !on) { p.rho = 0.f; p.phi = 0.f; }
            o_rho[j] = p.rho;
            o_phi[j] = p.phi;
            o_i[0][j] = p.d1;
            o_i[1][j] = p.d2;
            if (NORMALS) {
                int bi[3];
                float v[9];
                normals9(p.rho, p.phi, td, t1, t2, v, bi);
                o_i[2][j] = bi[0]; o_i[3][j] = bi[1]; o_i[4][j] = bi[2];
#pragma unroll
                for (int c = 0; c < 9; ++c) o_n[c][j] = (MODE == PD_POLAR_STOKES && !on) ? 0.f : v[c];
            }
        }
        if (xolp) {
            float* o = xolp + (b * 2) * Pout + po;
            *reinterpret_cast<float4*>(o) = make_float4(o_rho[0], o_rho[1], o_rho[2], o_rho[3]);
            *reinterpret_cast<float4*>(o + Pout) = make_float4(o_phi[0], o_phi[1], o_phi[2], o_phi[3]);
        }
        if (xolp_std) {
            float* o = xolp_std + (b * 2) * Pout + po;
            *reinterpret_cast<float4*>(o) = make_float4((o_rho[0] - kMean) / kStd, (o_rho[1] - kMean) / kStd,
                                                        (o_rho[2] - kMean) / kStd, (o_rho[3] - kMean) / kStd);
            *reinterpret_cast<float4*>(o + Pout) = make_float4((o_phi[0] - kMean) / kStd, (o_phi[1] - kMean) / kStd,
                                                            (o_phi[2] - kMean) / kStd, (o_phi[3] - kMean) / kStd);
        }
        if (NORMALS && normals) {
            float* o = normals + (b * 9) * Pout + po;
#pragma unroll
            for (int c = 0; c < 9; ++c)
                *reinterpret_cast<float4*>(o + c * Pout) = make_float4(o_n[c][0], o_n[c][1], o_n[c][2], o_n[c][3]);
        }
        if (ints) {
            int* o = ints + (b * 5) * Pout + po;
            const int nch = NORMALS ? 5 : 2;
#pragma unroll
            for (int c = 0; c < 5; ++c)
                if (c < nch) *reinterpret_cast<int4*>(o + c * Pout) = make_int4(o_i[c][0], o_i[c][1], o_i[c][2], o_i[c][3]);
        }
    }
}

// get_normals() on an existing fp32 XOLP tensor (pre_encoders.py:99-113): [B,2,H,W] -> [B,9,H,W]
__global__ __launch_bounds__(kThreads) void normals_from_xolp_kernel(const float* __restrict__ xolp,
                                                                     float* __restrict__ normals,
                                                                     const char* __restrict__ blob, long P,
                                                                     long quads_per_img, long total_quads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Tab td, t1, t2;
    stage_tables(blob, smem, kThreads, td, t1, t2);
    for (long q = blockIdx.x * (long)kThreads + threadIdx.x; q < total_quads; q += (long)gridDim.x * kThreads) {
        const long b = q / quads_per_img;
        const long p4 = (q - b * quads_per_img) * 4;
        const float4 r4 = *reinterpret_cast<const float4*>(xolp + (b * 2) * P + p4);
        const float4 f4 = *reinterpret_cast<const float4*>(xolp + (b * 2 + 1) * P + p4);
        const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, ff[4] = {f4.x, f4.y, f4.z, f4.w};
        float o_n[9][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[9]; int bi[3];
            normals9(rr[j], ff[j], td, t1, t2, v, bi);
#pragma unroll
            for (int c = 0; c < 9; ++c) o_n[c][j] = v[c];
        }
        float* o = normals + (b * 9) * P + p4;
#pragma unroll
        for (int c = 0; c < 9; ++c)
            *reinterpret_cast<float4*>(o + c * P) = make_float4(o_n[c][0], o_n[c][1], o_n[c][2], o_n[c][3]);
    }
}

}  // namespace

extern "C" int pd_polar_normals_from_xolp(const void* xolp, void* normals, const void* tables, size_t tables_bytes,
                                          int B, int H, int W, void* stream) {
    PD_REQUIRE(B >= 0 && H > 0 && W > 0, "pd_polar_normals_from_xolp: bad shape");
    const long P = (long)H * W;
    PD_REQUIRE(P % 4 == 0, "pd_polar_normals_from_xolp: H*W=%ld must be a multiple of 4", P);
    if (B == 0) return PD_OK;
    PD_REQUIRE(xolp && normals && tables, "pd_polar_normals_from_xolp: null pointer");
    PD_REQUIRE(pd::aligned16(xolp) && pd::aligned16(normals) && pd::aligned16(tables), "pd_polar_normals_from_xolp: unaligned");
    PD_REQUIRE(tables_bytes >= sizeof(PolarHeader), "pd_polar_normals_from_xolp: tables blob too small");
    const size_t lds = tables_bytes - (sizeof(PolarHeader) + ((size_t(kLutCount) * 4 + 15) / 16 * 16));
    PD_REQUIRE(lds <= 64 * 1024, "pd_polar_normals_from_xolp: theta tables need %zu bytes of LDS", lds);
    const long qpi = P / 4, total = qpi * B;
    long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(normals_from_xolp_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, (hipStream_t)stream,
                       (const float*)xolp, (float*)normals, (const char*)tables, P, qpi, total);
    return pd::check_launch("pd_polar_normals_from_xolp");
}

extern "C" int pd_polar_fwd(const void* pol, const void* mask, void* xolp, void* xolp_std, void* normals,
                            void* ints, const void* tables, size_t tables_bytes, int B, int H, int W, int Wout,
                            int mode, void* stream) {
    PD_REQUIRE(B >= 0 && H > 0 && W > 0, "pd_polar_fwd: bad shape B=%d H=%d W=%d", B, H, W);
    if (B == 0) return PD_OK;  // empty batch: nothing to do (pointers may be null)
    PD_REQUIRE(pol && tables, "pd_polar_fwd: pol and tables must not be null");
    PD_REQUIRE(mode == PD_POLAR_LS || mode == PD_POLAR_STOKES, "pd_polar_fwd: unknown mode %d", mode);
    const long P = (long)H * W;
    if (Wout <= 0) Wout = W;
    PD_REQUIRE(Wout >= W, "pd_polar_fwd: output pitch %d < W %d", Wout, W);
    PD_REQUIRE(Wout == W ? P % 4 == 0 : (W % 4 == 0 && Wout % 4 == 0),
               "pd_polar_fwd: H*W (or W and the output pitch, when they differ) must be multiples of 4");
    PD_REQUIRE(tables_bytes >= sizeof(PolarHeader), "pd_polar_fwd: tables blob too small");
    PD_REQUIRE(pd::aligned16(pol) && pd::aligned16(xolp) && pd::aligned16(xolp_std) && pd::aligned16(normals) &&
                   pd::aligned16(ints) && pd::aligned16(tables) && (!mask || pd::aligned16(mask)),
               "pd_polar_fwd: pointers must be 16-byte aligned");
    PD_REQUIRE(xolp || xolp_std || normals || ints, "pd_polar_fwd: no output requested");
    const long Pout = (long)H * Wout;
    const int wq_in = Wout == W ? (int)(P / 4) : W / 4, wq_out = Wout == W ? (int)(P / 4) : Wout / 4;
    const long qpi = Pout / 4, total = qpi * B;
    const bool need_normals = normals != nullptr || ints != nullptr;
    // LDS image size is fixed by the table node counts; validated against the blob size on the
    // host by the caller's pd_polar_tables_bytes(); the default tables need 56,000 bytes.
    const size_t lds = need_normals ? tables_bytes - (sizeof(PolarHeader) + ((size_t(kLutCount) * 4 + 15) / 16 * 16)) : 0;
    PD_REQUIRE(lds <= 64 * 1024, "pd_polar_fwd: theta tables need %zu bytes of LDS (max 65536)", lds);
    long blocks = (total + kThreads - 1) / kThreads;
    const long cap = need_normals ? 256L * 2 : 256L * 4;  // persistent: 2 (normals) / 4 (xolp only) blocks per CU
    if (blocks > cap) blocks = cap;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto args = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kThreads), lds, st,
                           static_cast<const uint8_t*>(pol), static_cast<const uint8_t*>(mask),
                           static_cast<float*>(xolp), static_cast<float*>(xolp_std), static_cast<float*>(normals),
                           static_cast<int*>(ints), static_cast<const char*>(tables), P, Pout, wq_in, wq_out, qpi, total);
    };
    if (mode == PD_POLAR_LS) {
        if (need_normals) args(polar_kernel<PD_POLAR_LS, true>); else args(polar_kernel<PD_POLAR_LS, false>);
    } else {
        if (need_normals) args(polar_kernel<PD_POLAR_STOKES, true>); else args(polar_kernel<PD_POLAR_STOKES, false>);
    }
    return pd::check_launch("pd_polar_fwd");
}
