/*
 * polardepth.h -- C ABI of libpolardepth.so (MI355X / gfx950 hot path).
 *
 * Drop-in boundary for the polarimetric depth hot path.  The reference
 * (kkaytekin/Supervised-Depth-Estimation-from-Polarized-Images) is pure Python on
 * PyTorch and has no FFI of its own; the entry points below are what a ctypes
 * binding inside the reference's modules would call (INTEGRATION.md shows the
 * stubs).  Each entry point cites the reference code it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only, no framework types;
 *   - device pointers come from the caller (e.g. tensor.data_ptr()); the caller
 *     owns every buffer including workspaces; nothing is allocated or freed here;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no entry point
 *     synchronises the device;
 *   - return 0 on success, a negative PD_E* code on failure; the message is
 *     available from pd_last_error() (thread-local);
 *   - re-entrant: no global mutable state (the only process-wide data are read-only lookup tables and idempotent
 *     hipFuncSetAttribute bookkeeping); the library reads NO environment variable -- every choice a caller can make
 *     (kernel family / arithmetic, measurement variants) is an argument of the entry point it concerns.
 */
#ifndef POLARDEPTH_H
#define POLARDEPTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_OK 0
#define PD_EINVAL (-22)  /* bad argument (shape, alignment, null pointer) */
#define PD_ELAUNCH (-5)  /* HIP reported a launch error */

const char* pd_last_error(void);
int pd_version(void);

/* ------------------------------------------------------------------------- K1
 * Fused Stokes / DoLP / AoLP / physical-normals kernel.
 *
 * Replaces, in one pass over the four polarizer planes:
 *   polarisation/xolp.py:8-34            Iun_and_xolp   (mode PD_POLAR_LS)
 *   manydepth/datasets/indoor_dataset.py:430-442  IndoorDataset.get_xolp
 *   ppp_code/physical_normals_channels.py:15-36   PolarisationImage_channel (mode PD_POLAR_STOKES)
 *   manydepth/normals_vec.py:11-60       rho_diffuse / rho_spec / calc_normals
 *   manydepth/networks/pre_encoders.py:78-79,99-113  normalizeInput('XOLP') / get_normals
 *
 * Tables: a packed, position-independent blob (AoLP LUT over (I0-I90, I45-I135)
 * plus the three theta(rho) interpolation tables with scipy interp1d
 * 'linear'/'extrapolate' semantics).  Build it on the host with one of the two
 * functions below, copy it to the device once, and pass the device pointer.
 */
#define PD_POLAR_LS 0
#define PD_POLAR_STOKES 1
/* flags: default (0) = fp32 theta trig (two-float bin constants, |err| <= ~3e-7; cos/sin(phi) from a LUT,
 * cos/sin(phi+pi/2) by identity); PD_POLAR_PRECISE_NORMALS = fp64 sin/cos(theta) rounded once and
 * cos/sin(fl32(phi+pi/2)) like the reference (~1e-7, slower: ALU-bound).  DoLP / AoLP / index maps are
 * bit-exact in both. */
#define PD_POLAR_PRECISE_NORMALS 1
/* PD_POLAR_IEEE_RHO = evaluate the reference's fp64 sqrt/div sequence for every pixel instead of the
 * Newton-refined hardware seeds with a rounding test (same bits, ~1.3x the arithmetic); the exhaustive test
 * compares the two over all 2^32 uint8 quadruples. */
#define PD_POLAR_IEEE_RHO 2
/* measurement only (tools/bench_polar.py): force the nontemporal hint on / off the plane loads of the training step's
 * instantiation; by default launches of up to 32 frames' worth of 512x640 output carry it. */
#define PD_POLAR_NT_LOADS 4
#define PD_POLAR_PLAIN_LOADS 8

/* bytes needed for the table blob with n_d / n_s1 / n_s2 table nodes */
size_t pd_polar_tables_bytes(int n_d, int n_s1, int n_s2);
/* Pack caller-computed tables (x ascending, y), e.g. the NumPy ones of normals_vec.py:13-47. */
int pd_polar_tables_pack(const double* x_d, const double* y_d, int n_d,
                         const double* x_s1, const double* y_s1, int n_s1,
                         const double* x_s2, const double* y_s2, int n_s2,
                         void* host_blob, size_t blob_bytes);
/* Compute the tables with libm for refractive index n (1000 theta nodes, normals_vec.py:13,27)
 * and pack them.  *blob_bytes_out receives the size actually used. */
int pd_polar_tables_build(double n, void* host_blob, size_t blob_bytes, size_t* blob_bytes_used);

/*
 * pol        [B,4,H,W] uint8 device, planes in 0/45/90/135 degree order (pol00, pol01, pol10, pol11)
 * mask       [B,H,W] uint8 device or NULL (STOKES mode: outputs are 0 where mask == 0)
 * xolp       [B,2,H,W] fp32 or NULL   ch0 DoLP, ch1 AoLP      == inputs[("xolp",0,0)].float()
 * xolp_std   [B,2,H,W] fp32 or NULL   (xolp-0.08693199701957657)/0.44430732785457433
 * normals    [B,9,H,W] fp32 or NULL   cat(N_diff, N_spec1, N_spec2)   == get_normals(xolp).float()
 * ints       [B,5,H,W] int32 or NULL  exact by-products: d1, d2, idx_diffuse, idx_spec1, idx_spec2
 * tables     device copy of the blob
 * Wout    row pitch (in pixels) of every OUTPUT plane, Wout >= W (0 means W).  Wout > W writes the
 *         W real columns and zero-fills the Wout-W padding columns: the 512x612 HAMMER planes land
 *         directly in the 512x640 tensors the network needs (multiples of 32, trainer.py:107-108).
 * H*W must be a multiple of 4 (W and Wout multiples of 4 when they differ); pointers 16-byte aligned.
 */
int pd_polar_fwd(const void* pol, const void* mask, void* xolp, void* xolp_std, void* normals,
                 void* ints, const void* tables, size_t tables_bytes,
                 int B, int H, int W, int Wout, int mode, int flags, void* stream);

/* get_normals() on an existing fp32 XOLP tensor (pre_encoders.py:99-113, the path taken when a
 * data loader already produced inputs[("xolp",0,0)]): xolp [B,2,H,W] -> normals [B,9,H,W]. */
int pd_polar_normals_from_xolp(const void* xolp, void* normals, const void* tables, size_t tables_bytes,
                               int B, int H, int W, int flags, void* stream);

/* calc_normals (manydepth/normals_vec.py:53-60): out [B,3,P] = (cos(phi) sin(theta), sin(phi) sin(theta), cos(theta)) for
 * phi, theta [B,P].  Element types as torch promotes them there: *_f64 != 0 marks an fp64 operand; the output is fp64 when
 * either operand is (the reference's get_normals hands over an fp32 phi and an fp64 theta: cos(phi) is evaluated in fp32
 * and promoted, normals_vec.py:56-58), fp32 otherwise. */
int pd_polar_calc_normals(const void* phi, const void* theta, void* out, int B, long P, int phi_f64, int theta_f64,
                          void* stream);

/* rho_diffuse / rho_spec as functions of their own (manydepth/normals_vec.py:11-22, 25-50): rho fp32 [n] ->
 * theta_d, theta_s1, theta_s2 fp64 [n] (each may be NULL) = scipy interp1d(fill_value="extrapolate") of the three
 * tables, evaluated in scipy's operation order (slope * (rho - x_lo) + y_lo, fp64), so rho beyond a table extrapolates
 * exactly like the reference (theta_s1 up to +41 rad, theta_s2 down to -135 rad for DoLP ~ 2).  bins: optional int32
 * [3][n], the clip(searchsorted(x, rho, 'left'), 1, n-1) index of each table.  NaN rho sorts last like numpy. */
int pd_polar_theta(const void* rho, void* theta_d, void* theta_s1, void* theta_s2, void* bins,
                   const void* tables, size_t tables_bytes, long n, void* stream);

/* ------------------------------------------------------------------------- K2
 * Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 *
 * Replaces torch.nn.Conv2d (+ ReflectionPad2d) and its autograd on the hot path:
 *   manydepth/networks/pre_encoders.py:15-25       ConvBlock.conv (bias, zero padding, stride 1/2)
 *   manydepth/networks/resnet_encoder.py:813-818   torchvision resnet18 conv1 / layer1 / layer2
 *   manydepth/layers.py:364-380                    Conv3x3 = ReflectionPad2d(1) + Conv2d(3)  (decoder)
 *
 * Layouts: x is fp32 with explicit element strides (sN,sH,sW,sC) -- NHWC (sC == 1) takes the
 * 16-byte gather path, anything else (e.g. the NCHW 2/3/9-channel stem inputs) the scalar one;
 * w is [Cout][KH][KW][Cin] (torch channels_last storage of the [Cout,Cin,KH,KW] parameter);
 * y is NHWC with row stride ldy >= Cout (ldy > Cout writes into a channel slice of a wider buffer).
 * (Ho, Wo) may be smaller than the full output grid: only the leading Ho x Wo outputs are computed.
 *
 * mode  0 zero padding | 1 reflection padding | 2 transposed (data gradient: x is dY on the
 *       forward output grid [N,H,W,C=Cout_fwd], (Ho,Wo) is the forward INPUT grid, w is the
 *       transposed weight [Cin_fwd][KH][KW][Cout_fwd] from pd_weight_transpose)
 * act   0 none | 1 ReLU | 2 ELU(alpha=1) | 3 sigmoid      (applied after bias)
 * out_scale  NULL or fp32 [Cout]: y = act(conv * out_scale + bias) -- an inference-mode BatchNorm folded into the
 *       epilogue (scale = gamma / sqrt(var + eps), bias = beta - mean * scale + conv_bias * scale)
 * affine != 0: every in-bounds input tap is replaced by (x - sub) / div before the product
 *       (ShallowEncoder.normalizeInput pre_encoders.py:76-83, resnet_encoder.py:812), scalar path only.
 * stats  NULL or fp32 [pd_conv2d_stats_rows(M,Cout)][Cout][2]: per-workgroup column sums and sums
 *       of squares of (conv + bias) -- training-mode BatchNorm statistics (reduced by pd_bn_finalize).
 * flags  kernel family / arithmetic of the products, the same word for pd_conv2d, pd_conv2d_add, pd_conv2d_rect,
 *       pd_conv2d_wgrad and their query functions (a query answers for the flags it is given):
 *         PD_CONV_AUTO       the library's choice (bf16-split products where they are faster, fp32 MFMA elsewhere);
 *         PD_CONV_FP32_MFMA  every product on v_mfma_f32_32x32x2_f32 / 16x16x4 (the reference's own arithmetic:
 *                            plain fp32 nn.Conv2d, pre_encoders.py:8-34);
 *         PD_CONV_BF16X3     bf16-split products whenever the shape fits those kernels, whatever the tile count (tests);
 *         PD_CONV_WGRAD_SPLIT_IN_REGS  weight gradient: conv_wgrad_uni_kernel's in-register split instead of
 *                            conv_wgrad_x3c_kernel (kept for its test; slower than either);
 *         PD_CONV_GENERAL_KERNELS      the general gather kernels instead of the uniform-tap / scalar-pixel ones (tests).
 *       PD_CONV_FP32_MFMA and PD_CONV_BF16X3 exclude each other; unknown bits are an error.
 */
#define PD_CONV_AUTO 0u
#define PD_CONV_FP32_MFMA 1u
#define PD_CONV_BF16X3 2u
#define PD_CONV_WGRAD_SPLIT_IN_REGS 4u
#define PD_CONV_GENERAL_KERNELS 8u
#define PD_CONV_X3_IM2COL 16u   /* bf16-split forward / data gradient through the per-tap gather kernel (conv_igemm_x3_kernel)
                                 * also where the halo-tile kernel (conv_halo_x3_kernel) fits: A/B measurement and tests */
#define PD_CONV_WGRAD_ROW_WORKGROUPS 32u /* bf16-split 3x3 weight gradient with one filter row per workgroup (conv_wgrad_halo_x3_kernel)
                                         * also where the rolling-row kernel (conv_wgrad_roll_x3_kernel) fits: A/B measurement and tests */
#define PD_CONV_FLAGS_ALL 63u
int pd_conv2d_tile_m(long M, int Cout);
long pd_conv2d_stats_rows(long M, int Cout);
/* Non-zero (3: the halo-tile kernel conv_halo_x3_kernel -- stride 1, the output grid Ho x Wo (0 x 0: unknown, never 3) a whole
 * number of 8 x 32, 16 x 16 or 32 x 8 tiles, at least 512 workgroups; 3x3 / 5x5 with zero padding, the stride-1 data gradient or
 * reflection padding, C % 16 == 0, Cout % 32 == 0 (64-column workgroups, or 32-column ones where those would be fewer than 512
 * or Cout % 64 == 32), or the 4x4 zero-padded space-to-depth stems (contiguous pixels, 4 C % 16 == 0, Cout % 64 == 0) in their
 * row-window form --; 2 | 1: the per-tap gather kernel with 256- | 128-row tiles) when pd_conv2d / pd_conv2d_add send this shape (16-byte aligned NHWC operands assumed) to the kernel that forms the fp32
 * products on the bf16 matrix cores (conv_igemm_x3_kernel: x = hi + mid + lo in bf16, six MFMAs per 32x32x16 block, fp32
 * accumulation; PD_CONV_FP32_MFMA in `flags` keeps every layer on the fp32 MFMA): zero padding, the stride-1 data gradient or 3x3 reflection padding (a same-size layer is assumed), C % 4 == 0 and >= 8 (16-channel groups, the last may be partly empty),
 * Cout % 64 == 0, at least 512 tiles of 256 x 64 (M % 256 == 0) or 320 of 128 x 64 (M % 128 == 0), no out_scale, activation none or ELU.  The profiler label of a launch
 * (ops._igemm_label) and bench.py's roofline object use it. */
int pd_conv2d_uses_x3(long M, int Cout, int C, int KH, int KW, int stride, int pad, int mode, int act, int has_out_scale,
                      int Ho, int Wo, unsigned flags);
int pd_conv2d(const void* x, const void* w, const void* bias, const void* out_scale, void* y, void* stats,
              int N, int H, int W, int C, long sN, long sH, long sW, long sC,
              int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int mode, int act,
              int affine, float sub, float div, long ldy, unsigned flags, void* stream);

/* The 16-channel decoder tail (depth_decoder.py upconv(0,0) 32->16, upconv(0,1) 16->16; layers.py:329-380): 3x3
 * convolutions with 16 output channels from an input halo tile staged once in LDS (the implicit GEMM above re-stages
 * the input per tap and is bound by that path at 16 output channels).
 *   mode 0: y = act(conv3x3(reflection_pad1(x), w) + bias), x NHWC [N,H,W,C] (C = 16 or 32, strides sN/sH/sW, channel
 *           stride 1), w [16][3][3][C], y NHWC [N,H,W,16] with row stride ldy, act 0 none | 2 ELU;
 *   mode 1: data gradient on the reflection-PADDED grid: x = dz [N,H,W,16], w = the transposed filter
 *           [Cout][3][3][16] (pd_weight_transpose), y [N,H+2,W+2,Cout] (Cout = 16 or 32), zero outside the image;
 *           pd_reflect_fold folds the border afterwards;
 *   mode 2: the zero-padding (pad 1) data gradient on the H x W grid (same operands as mode 1, y [N,H,W,Cout]) = the
 *           interior of mode 1's result; pd_reflect_dgrad_border adds the folded border strips. */
int pd_conv16(const void* x, const void* w, const void* bias, void* y, int N, int H, int W, int C,
              long sN, long sH, long sW, int Ho, int Wo, int Cout, long ldy, int mode, int act, void* stream);

/* Weight (+ bias) gradient of the mode-0 convolution above: dw [16][3][3][C] (+)= sum_p dz[p] (x) x[reflect(p+tap-1)],
 * dbias [16] (+)= sum_p dz[p] (or NULL).  Persistent workgroups, one deterministic partial per workgroup in
 * `workspace` (>= pd_conv16_wgrad_workspace(C) bytes), summed in a fixed order. */
size_t pd_conv16_wgrad_workspace(int C);
int pd_conv16_wgrad(const void* x, const void* dz, void* dw, void* dbias, void* workspace, size_t ws_bytes,
                    int N, int H, int W, int C, long sN, long sH, long sW, long ldd, int accumulate, void* stream);

/* Stride-2 data gradient by output parity (3x3 / stride 2 / pad 1, even input grid; resnet18 layer2.0.conv1,
 * resnet_encoder.py / torchvision BasicBlock): pd_dgrad_s2_filters turns the transposed filter wt [Cin][3][3][Cout] into
 * the four exact class filters, back to back in wsub (9*Cin*Cout floats): class c = 2*(ih%2) + iw%2 is
 * [Cin][1 + ih%2][1 + iw%2][Cout] at float offset Cin*Cout*{0,1,3,5}[c]; each class is pd_conv2d_rect(mode 2,
 * KH = 1 + ih%2, KW = 1 + iw%2, pad_h = ih%2, pad_w = iw%2) of dY onto the [N,Ho,Wo,Cin] sub-grid; pd_interleave4
 * scatters the four sub-grids [4][N][Ho][Wo][C] into dX [N][2Ho][2Wo][C].  9 tap-units instead of the 36 of the masked
 * transposed gather.
 * pd_conv2d_rect: stride-1 convolution (mode 0) / data gradient (mode 2) with a KH x KW filter and separate row /
 * column padding, no bias / activation; needs C % 32 == 0 and 16-byte aligned NHWC operands (uniform-tap kernel). */
int pd_dgrad_s2_filters(const void* wt, void* wsub, int Cin, int Cout, void* stream);
int pd_conv2d_rect(const void* x, const void* w, void* y, int N, int H, int W, int C, long sN, long sH, long sW, long sC,
                   int Ho, int Wo, int Cout, int KH, int KW, int pad_h, int pad_w, int mode, long ldy, unsigned flags,
                   void* stream);
int pd_interleave4(const void* sub, void* dx, int N, int Ho, int Wo, int C, void* stream);

/* y = conv(x, w) + addend: the same convolution (no bias / scale / activation / statistics) with an NHWC tensor of the
 * output's shape (row stride ld_add, may alias y) added in the epilogue.  Used for the data gradient of the first
 * convolution of a residual block, which autograd would otherwise sum with the gradient of the skip connection in a
 * separate pass (the reference: torch's `out += identity` backward, pre_encoders.py:46, torchvision BasicBlock). */
int pd_conv2d_add(const void* x, const void* w, const void* addend, long ld_add, void* y,
                  int N, int H, int W, int C, long sN, long sH, long sW, long sC,
                  int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int mode, long ldy, unsigned flags,
                  void* stream);

/* Weight gradient dW[Cout][KH][KW][Cin] (+ optional dbias[Cout]) of the convolution above
 * (modes 0 and 1).  dy is NHWC on the output grid with row stride ldd.  Partial tiles go to
 * `workspace` (pd_conv2d_wgrad_workspace bytes) and are summed deterministically; accumulate != 0
 * adds to dw/dbias instead of overwriting.  Replaces autograd's conv weight/bias gradient. */
size_t pd_conv2d_wgrad_workspace(long M, int Cout, int K, unsigned flags);
int pd_conv2d_wgrad(const void* x, const void* dy, void* dw, void* dbias, void* workspace, size_t ws_bytes,
                    int N, int H, int W, int C, long sN, long sH, long sW, long sC,
                    int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, int mode,
                    int affine, float sub, float div, long ldd, int accumulate, unsigned flags, void* stream);
/* 3 (conv_wgrad_roll_x3_kernel: 3x3, stride 1, same-size output, C % 64 == 0, Cout % 64 == 0, Wo % 16 == 0, tile columns long and
 * many enough for >= 12 tile rows per slice on >= 384 workgroups: all three filter rows per workgroup, input rows rolling
 * through LDS; PD_CONV_WGRAD_ROW_WORKGROUPS in `flags` declines it),
 * 2 (conv_wgrad_halo_x3_kernel: 3x3 / 5x5, stride 1, zero or reflection padding, C % 64 == 0 and Cout % 64 == 0 -- or 3x3 with
 * Cout % 32 == 0, C % 32 == 0 on 1 x 32 tiles --, the output grid a whole number of 1 x 32, 2 x 16, 4 x 8 or (3x3) 8 x 4 pixel tiles: both
 * operands split once per tile and read through ds_read_b64_tr_b16) or
 * 1 when pd_conv2d_wgrad sends this shape (zero or reflection padding; 16-byte aligned NHWC operands assumed) to the kernel that forms the
 * fp32 products on the bf16 matrix cores, every element split once (conv_wgrad_x3c_kernel; PD_CONV_FP32_MFMA: fp32 MFMA) --
 * the profiler label of a launch and bench.py's roofline object use it. */
int pd_conv2d_wgrad_uses_x3(long M, int Cout, int C, int KH, int KW, int stride, int pad, int mode, int H, int W, int Ho, int Wo,
                            unsigned flags);

/* 7x7 / stride-2 / pad-3 stems (pre_encoders.py:54 ShallowEncoder.Conv1, torchvision resnet conv1) executed as a
 * 4x4 / stride-1 / pad-2 convolution over the space-to-depth input [N][H/2][W/2][4C] (4C is a multiple of 4 ->
 * 16-byte gather path): input transform (with the optional (x-sub)/div normalisation), weight regrouping
 * [Cout][7][7][C] -> [Cout][4][4][4C] and the inverse mapping of the weight gradient.  pd_conv2d /
 * pd_conv2d_wgrad are then called with KH=KW=4, stride 1, pad 2 and the (H/2, W/2) output grid. */
int pd_stem_s2d_input(const void* x, void* out, int N, int C, int H, int W, long sN, long sC, long sH, long sW,
                      int affine, float sub, float div, void* stream);
int pd_stem_s2d_weight(const void* w, void* w2, int Cout, int C, void* stream);
int pd_stem_s2d_weight_grad(const void* dw2, void* dw, int Cout, int C, int accumulate, void* stream);

/* w [Cout][T][Cin] -> wt [Cin][T][Cout], T = KH*KW: operand of the mode-2 (data gradient) GEMM. */
int pd_weight_transpose(const void* w, void* wt, int Cout, int T, int Cin, void* stream);
/* Every convolution weight of a flat parameter buffer in one launch (the data-gradient operands of a whole step):
 * entry e of `table` (device int32[n][4]) = {element offset in src and dst, Cout, T, Cin}; blk (device int32[n+1]) =
 * first workgroup of entry e at T * ceil(Cout/32) * ceil(Cin/32) workgroups per entry, blk[n] = nblocks.  Replaces one pd_weight_transpose per
 * layer and step (the reference keeps no such copy: cuDNN's backward-data reads the forward filter, trainer.py:430). */
int pd_weight_transpose_batched(const void* src, void* dst, const void* table, const void* blk, int n, int nblocks,
                                void* stream);

/* ------------------------------------------------------------------------- K3
 * Memory-bound kernels fused around the convolutions (NHWC fp32, 16 bytes per lane).
 *
 * pd_bn_fwd_finalize: BatchNorm2d statistics.  training != 0: reduces the conv epilogue partials
 *   [R][C][2] in fp64 (acc_ws: 2*C + 1 doubles -- the sums and an arrival ticket -- that must be ZERO on entry
 *   and are left zero by the call: a long-lived accumulator per stream saves a memset per layer; the workgroup
 *   that arrives last finalises, so the whole call is ONE launch; same contract in pd_bn_bwd_finalize),
 *   updates running_mean/var with torch semantics
 *   (momentum, unbiased variance) and emits scale = gamma*invstd, shift = beta - mean*scale plus
 *   the saved mean / invstd; training == 0: coefficients from the running statistics.
 *   Replaces nn.BatchNorm2d's statistics pass (pre_encoders.py:19,29; torchvision BasicBlock.bn*).
 * pd_chain_fwd: out = relu_post( dropout( pool2x2( relu_pre( x*scale + shift ) ) ) + res )
 *   i.e. the tail of pre_encoders.ConvBlock (:28-34) + the ResidualBlock add (:46), or the
 *   torchvision BasicBlock tail (bn -> +identity -> relu).  scale == NULL means identity.
 *   Dropout masks come from Philox4x32-10(seed, offset, element) and are regenerated in backward.
 * pd_chain_bwd_reduce / pd_bn_bwd_finalize / pd_chain_bwd_apply: the two-pass backward
 *   (per-channel sum g, sum g*xhat -> dgamma, dbeta -> dx of the raw conv output; optional dres
 *   = dy * (out > 0) for post-add ReLU blocks).
 */
int pd_bn_fwd_finalize(const void* partial, long R, int C, double count, const void* gamma, const void* beta,
                       void* running_mean, void* running_var, float momentum, float eps, void* acc_ws, long acc_len,
                       void* scale, void* shift, void* save_mean, void* save_invstd, int training, void* stream);
/* acc_len: doubles in acc_ws, checked against 2*C + 1 (the ticket word lives at acc_ws[2*C]). */
int pd_bn_bwd_finalize(const void* partial, long R, int C, double count, void* acc_ws, long acc_len, void* dgamma,
                       void* dbeta, void* coef, int accumulate, void* stream);
long pd_chain_bwd_rows(int N, int H, int W, int C);
int pd_chain_fwd(const void* x, const void* scale, const void* shift, const void* res, void* out,
                 int N, int H, int W, int C, long ld_res, long ld_out, int relu_pre, int pool,
                 float drop_p, uint64_t seed, uint64_t offset, const void* step_state, int relu_post, void* stream);
int pd_chain_bwd_reduce(const void* dy, long ld_dy, const void* x, const void* out, long ld_out,
                        const void* scale, const void* shift, const void* mean, const void* invstd,
                        void* partial, int N, int H, int W, int C, int relu_pre, int pool, float drop_p,
                        uint64_t seed, uint64_t offset, const void* step_state, int relu_post, void* stream);
int pd_chain_bwd_apply(const void* dy, long ld_dy, const void* x, const void* out, long ld_out,
                       const void* scale, const void* shift, const void* mean, const void* invstd,
                       const void* coef, void* dx, void* dres, int N, int H, int W, int C, int relu_pre,
                       int pool, float drop_p, uint64_t seed, uint64_t offset, const void* step_state, int relu_post,
                       void* stream);

/* nn.MaxPool2d(3, 2, 1) of the ResNet stem (resnet_encoder.py:814) and its gradient. */
/* idx (optional in fwd): uint8 [N,Ho,Wo,C] window position (dh*3+dw) of the first maximum, consumed by bwd. */
int pd_maxpool3s2_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream);
int pd_maxpool3s2_bwd(const void* idx, const void* dy, void* dx, int N, int H, int W, int C, void* stream);
/* ... + addend (optional, NHWC [N,H,W,C] with row stride ld_add): the gradient a second consumer of the pooled tensor
 * delivers (the decoder's skip connection, depth_decoder.py:65-66), summed in the same pass. */
int pd_maxpool3s2_bwd_add(const void* idx, const void* dy, const void* addend, long ld_add, void* dx, int N, int H, int W,
                          int C, void* stream);

/* Decoder glue: out[N,2H,2W,Ca+Cs] = cat(bilinear_x2(a), skip)  (layers.py:446-449 upsample,
 * depth_decoder.py:64-67 cat) and the gradient of the upsampled part (gather form). */
int pd_upcat_fwd(const void* a, const void* skip, long ld_skip, void* out, int N, int H, int W, int Ca, int Cs,
                 void* stream);
int pd_up_bwd(const void* dout, long ld_d, void* da, int N, int H, int W, int Ca, void* stream);
/* same with the ELU derivative folded in: a = elu_y [N,H,W,Ca] is the output of a ConvBlock (layers.py:329-342), da
 * leaves multiplied by (a > 0 ? 1 : a + 1), i.e. as the gradient of the convolution output. elu_y NULL = pd_up_bwd. */
int pd_up_bwd_elu(const void* dout, long ld_d, const void* elu_y, void* da, int N, int H, int W, int Ca, void* stream);

/* DPT decoder glue (reference manydepth/dpt/blocks.py; --train_dpt, trainer.py:147-171):
 * pd_up2x_ac_fwd / _bwd: nn.functional.interpolate(scale_factor=2, mode="bilinear", align_corners=True) of an NHWC
 *   tensor [N,H,W,C] -> [N,2H,2W,C] (blocks.py:138-172 Interpolate, :375-377 in FeatureFusionBlock_custom) and its
 *   gradient in gather form (deterministic);
 * pd_relu_add: out = (relu ? max(x, 0) : x) + (res ? res : 0) over n floats -- the activation in front of the first
 *   convolution and the skip sum of ResidualConvUnit_custom (blocks.py:289-307). */
int pd_up2x_ac_fwd(const void* a, void* out, int N, int H, int W, int C, void* stream);
int pd_up2x_ac_bwd(const void* dout, void* da, int N, int H, int W, int C, void* stream);
int pd_relu_add(const void* x, const void* res, void* out, long n, int relu, void* stream);

/* dz = dy * f'(.) through the activation OUTPUT y: act 1 ReLU, 2 ELU (layers.py:337), 3 sigmoid. */
int pd_act_bwd(const void* dy, const void* y, void* dz, long n, int act, void* stream);
/* Gradient of ReflectionPad2d(1): dxp [N,H+2,W+2,C] -> dx [N,H,W,C]  (layers.py:372). */
int pd_reflect_fold(const void* dxp, void* dx, int N, int H, int W, int C, void* stream);
/* ... of ReflectionPad2d(pad), 1 <= pad < min(H, W): dxp [N,H+2pad,W+2pad,C] -> dx [N,H,W,C]  (layers.py:352, Conv5x5). */
int pd_reflect_fold_pad(const void* dxp, void* dx, int N, int H, int W, int C, int pad, void* stream);
/* The same gradient without the padded intermediate: dx (in/out, NHWC [N,H,W,Cin]) already holds the zero-padding
 * (pad 1) data gradient -- the interior of the padded-grid gradient -- and receives the four folded border strips
 * (1x3 / 3x1 filter slices applied to the first / last row and column of dz [N,H,W,Cout], row stride ldd;
 * w [Cout][3][3][Cin], the forward filter).  Touches 2(H+W) pixels per image instead of two full-tensor passes. */
int pd_reflect_dgrad_border(const void* dz, long ldd, const void* w, void* dx, int N, int H, int W, int Cout, int Cin,
                            void* stream);
/* torch.optim.Adam step (trainer.py:238,442) over one flat fp32 buffer; grads are pre-multiplied
 * by grad_scale (1/world_size after the RCCL sum).  zero_grad != 0 also clears g in the same pass
 * (trainer.py:436 zero_grad of the NEXT iteration, without a separate 85 MB memset).
 * step_state (optional): the step words below; Adam's t is then read from step_state[1] and lr / grad_scale from
 * step_state[2] on the device instead of the `step`, `lr`, `grad_scale` arguments, so that a hipGraph of the training step
 * can be replayed with frozen arguments and still follow a learning-rate schedule (trainer.py:238-240,467 StepLR).  Bias
 * corrections use beta^t by repeated squaring in double on either side (identical bits). */
int pd_adam_step(void* p, void* g, void* m, void* v, long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, long step, const void* step_state, float grad_scale, int zero_grad, void* stream);
/* Step words in device memory, int64[4]: [0] training steps begun -- the dropout sites of pd_chain_* add
 * step_state[0] << 12 to their Philox offset (step_state NULL: the offset argument alone) --, [1] optimizer steps,
 * [2] the fp32 bit patterns of lr (low word) and grad_scale (high word).  pd_step_tick increments the selected counters
 * (one thread); pd_step_set_hyper writes word [2] (call it whenever the scheduler changed lr: it is NOT part of a captured
 * step).  The only per-step state of a captured training step. */
int pd_step_tick(void* step_state, int bump_dropout, int bump_adam, void* stream);
int pd_step_set_hyper(void* step_state, float lr, float grad_scale, void* stream);

/* ------------------------------------------------------------------------- K5
 * Multi-scale supervised loss (trainer.py:1126-1150,1241-1265,1298-1309; layers.py:62-71,452-465).
 * Per scale: pd_disp_to_depth (bilinear to H x W + disp_to_depth), pd_sup_loss_fwd (masked L1 and
 * the kornia depth_to_normals cosine term as wavefront-reduced partial sums), pd_smooth_fwd;
 * pd_loss_finalize turns the partials into vals = [loss, (loss/s, supervised_depth_loss/s,
 * normals_loss/s) per scale] on the device.  Backward: pd_loss_weights, pd_sup_loss_bwd,
 * pd_up_gather_bwd, pd_smooth_bwd.  No host synchronisation anywhere.
 */
int pd_loss_rows(long n);
int pd_disp_to_depth(const void* disp, void* depth, void* updisp, int N, int hs, int ws, int H, int W,
                     float min_depth, float max_depth, void* stream);
/* generic != 0: the any-ratio kernel instead of the one specialised for zooms 1 / 2 / 4 / 8 (tests compare the two) */
int pd_up_gather_bwd(const void* gup, void* gdisp, int N, int hs, int ws, int H, int W, int accumulate, int generic,
                     void* stream);
/* gt_normals (optional, NULL = recompute per call): [N,H,W,4] floats written by pd_gt_normals -- the unit normal of the
 * ground-truth depth at every in-range pixel (trainer.py:1298-1309 evaluates it once per scale; it does not depend on
 * the scale, so a step computes it once and its eight consumers read it back). */
int pd_gt_normals(const void* gt, const void* K, void* gt_normals, int N, int H, int W, float min_depth, float max_depth,
                  void* stream);
int pd_sup_loss_fwd(const void* pred, const void* gt, const void* K, const void* gt_normals, void* partial, int N, int H,
                    int W, float min_depth, float max_depth, int with_normals, void* stream);
/* Trainer.compute_supervised_normals_losses as a function of its own (trainer.py:1298-1309) with the CALLER's mask:
 * out[0] = sum((2 - cos(n_gt, n_pred)) * mask) / sum(mask), normals = kornia depth_to_normals of the two depth maps at
 * every pixel (no depth-range gate), mask fp32 [N,H,W] (any values: it multiplies).  partial_ws: pd_loss_rows(N*H*W) * 2
 * floats; the ratio is formed on the device from an ordered fp64 sum of the partials. */
int pd_normals_loss_masked(const void* pred, const void* gt, const void* K, const void* mask, void* partial_ws, void* out,
                           int N, int H, int W, void* stream);
/* Loss of PREDICTED normals -- the `arch1++_separate_normals_dec` variant (reference README.md:54: "the decoder directly
 * predicts normals; these normals are compared with normals calculated from ground truth"; its source is not in the reference
 * checkout, the formula is trainer.py:1298-1309 with the network's 3-channel output in place of depth_to_normals(pred)):
 * out[0] = sum((2 - cos(pred, n_gt)) m) / sum(m), out[1] = sum(m); pred pixel-major (NHWC) fp32 with pixel stride ld >= 3,
 * gt_normals from pd_gt_normals, m = gt depth inside [min_depth, max_depth].  partial_ws: pd_loss_rows(N*H*W) * 2 floats.
 * _bwd: dpred = gout[0] / out[1] * m * -(d cos / d pred), written for every pixel (zeros outside the mask). */
int pd_normals_pred_loss_fwd(const void* pred, long ld, const void* gt_normals, const void* gt, void* partial_ws, void* out,
                             int N, int H, int W, float min_depth, float max_depth, void* stream);
int pd_normals_pred_loss_bwd(const void* pred, long ld, const void* gt_normals, const void* gt, const void* gout,
                             const void* loss_count, void* dpred, long ld_dpred, int N, int H, int W, float min_depth,
                             float max_depth, void* stream);
/* The loop over scales of trainer.py:1134-1265 in one call: the per-scale kernels above dispatched by blockIdx.y = scale
 * with each scale's own grid (identical partial sums, hence identical bits), the supervised forward pass reading the
 * ground truth once for all scales: 4 launches forward, 4 + a memset backward, whatever S <= 8.
 * Arrays of S device pointers (host arrays): disps [N,1,hs,ws], colors [N,3,hs,ws], depths [N,1,H,W] (out), means [N] (out),
 * edge_ws [N,hs,ws,2] (out, entries may be NULL); sup_part [S][part_stride][3], sm_part [S][part_stride][2] floats with
 * part_stride >= pd_loss_rows(N*H*W); then pd_loss_finalize as before.
 * Backward: wts [S][3] (pd_loss_weights), sums [S][5]; workspaces gup_ws S*N*H*W floats, g_ws sum_s N*hs*ws floats,
 * gd_acc S*N doubles; gdisps [N,1,hs,ws] (out) = d loss / d disp_s. */
int pd_multiscale_loss_fwd(const void* const* disps, const void* const* colors, const int* hs, const int* ws, int S,
                           const void* gt, const void* K, const void* gt_normals, void* const* depths, void* const* means,
                           void* const* edge_ws, void* sup_part, void* sm_part, int part_stride, int N, int H, int W,
                           float min_depth, float max_depth, int with_normals, void* stream);
int pd_multiscale_loss_bwd(const void* const* disps, const void* const* colors, const void* const* depths,
                           const void* const* means, const void* const* edge_ws, const int* hs, const int* ws, int S,
                           const void* gt, const void* K, const void* gt_normals, const void* wts, const void* sums,
                           void* gup_ws, void* g_ws, void* gd_acc, void* const* gdisps, int N, int H, int W,
                           float min_depth, float max_depth, void* stream);
/* pd_sup_loss_bwd: ab_ws ([N,H,W,6] floats) is only read by the two-pass form (two_pass_form != 0, kept for its test);
 * by default one kernel evaluates the per-pixel normal gradients for the halo of an 8 x 64 tile into LDS and gathers
 * from there, so ab_ws may be NULL. */
int pd_sup_loss_bwd(const void* pred, const void* gt, const void* K, const void* gt_normals, const void* wts,
                    const void* sums, void* ab_ws, void* gout, int N, int H, int W, float min_depth, float max_depth,
                    int with_normals, int to_disp, int two_pass_form, void* stream);
/* Edge-aware smoothness (layers.py:452-465 get_smooth_loss) of the mean-normalised disparity of trainer.py:1256-1258:
 * partial[block][2] = (sum |dx (disp / (mean + 1e-7))| e^{-|dx I|}, same for y); the caller divides by the element
 * counts N h (w-1) and N (h-1) w.  mean [N] receives the per-image means; mean == NULL evaluates the term on disp as it
 * is -- layers.get_smooth_loss as a function of its own (the facade's manydepth.layers.get_smooth_loss).
 * edge_w (optional, NULL = recompute in the backward pass): [N,h,w,2] floats, the image-only edge weights
 * e^{-|dx I|}, e^{-|dy I|} written by the forward pass and read by the backward pass.
 * pd_smooth_bwd: wts[2] = d loss / d (smooth_x + smooth_y) (the two terms are averaged separately, as in the reference);
 * g_ws [N,h,w] floats, gd_acc N doubles. */
int pd_smooth_fwd(const void* disp, const void* img, void* mean, void* partial, void* edge_w, int N, int h, int w,
                  void* stream);
int pd_smooth_bwd(const void* disp, const void* img, const void* mean, const void* wts, const void* edge_w, void* g_ws,
                  void* gd_acc, void* gdisp, int N, int h, int w, int accumulate, void* stream);
int pd_loss_finalize(const void* sup_part, const int* sup_rows, const void* sm_part, const int* sm_rows,
                     const int* dims, const int* scale_ids, int S, int part_stride, float w_normals,
                     float w_smooth, void* sums, void* vals, void* stream);
/* vals from given sums [S][5] fp64 (sum|.|m, sum(2-cos)m, sum m, smooth_x, smooth_y): the division step of
 * pd_loss_finalize alone, for data-parallel runs that all-reduce the three masked sums per scale first
 * (trainer.py:1247,1308 normalise by the mask count of the whole batch). */
int pd_loss_from_sums(const void* sums, const int* dims, const int* scale_ids, int S, float w_normals,
                      float w_smooth, void* vals, void* stream);
int pd_loss_weights(const void* gvals, const int* scale_ids, int S, float w_normals, float w_smooth, void* wts,
                    void* stream);

/* SSIM map (mode 0, layers.py:468-499) or the photometric reprojection loss of trainer.py:1069-1081 (mode 1:
 * out [N,1,H,W] = 0.85 * mean_c SSIM + 0.15 * mean_c |y - x|, or the L1 part alone when no_ssim).  x, y planar
 * NCHW fp32.  Inactive under --depth_supervision_only; forward only. */
int pd_ssim_fwd(const void* x, const void* y, void* out, int N, int C, int H, int W, int mode, int no_ssim,
                void* stream);
/* Gradient of pd_ssim_fwd w.r.t. both images: gout [N,C,H,W] (mode 0) or [N,1,H,W] (mode 1), gx / gy [N,C,H,W] (gy may be
 * NULL), coef_ws 5*N*C*H*W floats.  Two launches: per-pixel coefficients of the five window means (with the clamp mask), then
 * the gather over the (reflected) 3x3 windows that contain each pixel. */
int pd_ssim_bwd(const void* x, const void* y, const void* gout, void* coef_ws, void* gx, void* gy, int N, int C, int H, int W,
                int mode, int no_ssim, void* stream);
/* compute_depth_errors (layers.py:539-557) per image on the device: metrics [N][8] = abs_rel, sq_rel, rmse,
 * rmse_log, a1, a2, a3, pixel count over min < gt < max (and mask == mask_value when mask != NULL, the
 * per-material selection of trainer.py:1385-1411); pred is clamped to [min, max] (trainer.py:1422-1423).
 * partial_ws: N * 64 * 8 floats. */
int pd_depth_metrics(const void* gt, const void* pred, const void* mask, int mask_value, void* partial_ws,
                     void* metrics, int N, long P, float min_depth, float max_depth, void* stream);

/* Row softmax of the attention variant (BASELINE config 5; SURVEY A17 -- defined by this build, the reference
 * branch is absent): in place x[r][:] = softmax(scale * x[r][:]) and its backward
 * dp[r][:] <- scale * p * (dp - sum(dp * p)).  The score GEMMs (Q K^T, P V and their gradients) are pd_conv2d /
 * pd_conv2d_wgrad calls with 1x1 filters. */
int pd_softmax_rows_fwd(void* x, long R, long L, float scale, void* stream);
int pd_softmax_rows_bwd(const void* p, void* dp, long R, long L, float scale, void* stream);

/* ---- Pillow-compatible 8-bit LANCZOS resampling of uint8 planes (Image.resize(..., Image.ANTIALIAS),
 * indoor_dataset.py:335-349): one pass of ImagingResample's 8bpc branch.
 * src [P][Hs][Ws] uint8; vertical == 0: dst [P][Hs][out_size] (horizontal pass), else dst [P][out_size][Ws].
 * coeffs int32 [out_size][ksize] (22-bit fixed point), bounds int32 [out_size][2] = (first input index, count):
 * Pillow's precompute_coeffs + normalize_coeffs_8bpc, built by polardepth/resize.py on the host. */
int pd_resize_u8_pass(const void* src, void* dst, const void* coeffs, const void* bounds, int ksize, int P, int Hs,
                      int Ws, int out_size, int vertical, void* stream);

/* ---- fused attention (flash-attention recurrence on the fp32 matrix cores).
 * q, k, v, o, do, dq, dk, dv: [N][T][128] fp32 (token-major = NHWC), lse / delta: [N][T] fp32; T % 32 == 0.
 * pd_attn_fwd:  o = softmax(q k^T * scale) v,  lse = log sum exp of the scaled scores (saved for the backward). */
int pd_attn_fwd(const void* q, const void* k, const void* v, void* o, void* lse, int N, int T, int C, float scale,
                void* stream);
/* pd_attn_bwd: dq, dk, dv from do (recomputing the scores block by block); delta [N][T] is scratch (sum_c do*o). */
int pd_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const void* lse,
                void* delta, void* dq, void* dk, void* dv, int N, int T, int C, float scale, void* stream);

/* The same attention on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16; BASELINE configs[4] "bf16, MFMA attention
 * path"): identical tensors and semantics (fp32 [N][T][128] in and out), operands rounded to bf16 when staged,
 * softmax statistics / exp2 / accumulators in fp32.  Selected by PD_ATTENTION_BF16=1 / bench.py --attention --bf16.
 * workspace (caller-owned, pd_attn_bf16_workspace bytes, 16-byte aligned): the operands every workgroup re-reads are packed
 * once per call into bf16 images of the kernels' LDS tiles (forward: K rows + V^T; backward: K rows, K^T, V rows, Q rows,
 * Q^T, dO rows, dO^T), so that staging a block is a verbatim 16-byte copy. */
size_t pd_attn_bf16_workspace(int N, int T, int C, int backward);
int pd_attn_bf16_fwd(const void* q, const void* k, const void* v, void* o, void* lse, void* workspace, size_t ws_bytes,
                     int N, int T, int C, float scale, void* stream);
int pd_attn_bf16_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const void* lse,
                     void* delta, void* dq, void* dk, void* dv, void* workspace, size_t ws_bytes, int N, int T, int C,
                     float scale, void* stream);
/* The same in parts (1 = delta + operand packing, 2 = dK / dV, 4 = dQ; 7 = pd_attn_bf16_bwd): the two gradient kernels are
 * independent, so a caller may enqueue parts 2 and 4 on two streams behind part 1 -- the workgroups of one kernel then fill
 * the last, partly empty round of the other. */
int pd_attn_bf16_bwd_parts(const void* q, const void* k, const void* v, const void* o, const void* d_o, const void* lse,
                           void* delta, void* dq, void* dk, void* dv, void* workspace, size_t ws_bytes, int N, int T, int C,
                           float scale, int parts, void* stream);

/* ---- disparity heads: sigmoid(Conv3x3(x)) with one output channel
 * (manydepth/networks/depth_decoder.py:52-53,69-71; layers.py:364-380 Conv3x3 = ReflectionPad2d(1) + Conv2d(C,1,3)).
 * Direct memory-bound kernels instead of a 32-wide MFMA tile with one useful column.
 *   x  [N,H,W,C] NHWC fp32, C in {16,32,64,128};  w [3][3][C] (= channels_last storage of the [1,C,3,3] weight);
 *   y / dy [N,H,W] (= [N,1,H,W]);  bias [1] or NULL.
 * pd_disphead_fwd:        y = sigmoid(bias + conv)
 * pd_disphead_bwd_data:   dx [N,H,W,C] = gradient w.r.t. x given dy (gradient w.r.t. y) and y; the fold of the
 *                         reflection padding is built in
 * pd_disphead_bwd_weight: dw [3][3][C] (+)= , dbias [1] (+)= (NULL to skip); workspace >= pd_disphead_workspace(C)
 *                         bytes, 16-byte aligned; deterministic two-stage summation
 * pd_disphead_bwd:        both gradients in one pass over x (dx may be NULL): what the training step calls.
 *                         add [N,H,W,C] or NULL is summed into dx (the gradient x receives from its other consumer,
 *                         depth_decoder.py:64 reads the same x); elu != 0: x is the ELU output of a ConvBlock
 *                         (layers.py:329-342) and dx leaves multiplied by ELU'(.) = (x > 0 ? 1 : x + 1) */
int pd_disphead_fwd(const void* x, const void* w, const void* bias, void* y, int N, int H, int W, int C, void* stream);
int pd_disphead_bwd_data(const void* dy, const void* y, const void* w, void* dx, int N, int H, int W, int C,
                         void* stream);
size_t pd_disphead_workspace(int C);
int pd_disphead_bwd_weight(const void* dy, const void* y, const void* x, void* dw, void* dbias, void* workspace,
                           size_t ws_bytes, int N, int H, int W, int C, int accumulate, void* stream);
int pd_disphead_bwd(const void* dy, const void* y, const void* x, const void* w, const void* add, int elu, void* dx,
                    void* dw, void* dbias, void* workspace, size_t ws_bytes, int N, int H, int W, int C, int accumulate,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POLARDEPTH_H */
