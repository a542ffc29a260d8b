/* libdeepsee_hip.so — C ABI of the MI355X-native DeepSEE train-step hot path.
 *
 * The reference (mcbuehler/DeepSEE) has no FFI: its "kernels" are ATen ops behind torch.nn
 *  and its plugin boundary is the Python object protocol SRModel / BaseManager /
 * TrainerManager.  This header is the boundary a maintainer binds instead of those
 * ATen calls; every entry point cites the reference call site(s) it replaces (paths relative to the
 * reference repo root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc / torch caching allocator); the caller owns all
 *    buffers including workspaces (query *_workspace()); the library never allocates or syncs.
 *  - activations are fp32 NHWC with the channel count padded to a multiple of 4 (pad channels = 0).
 *  - label maps are uint8 [N][H][W] at the HR resolution; kernels index lower resolutions with
 *    src = dst << shift (== F.interpolate(mode='nearest') for power-of-two ratios, SURVEY B-3).
 *  - every function takes the hipStream_t to enqueue on and returns 0 (DSEE_OK) or a negative
 *    DSEE_E* code; dsee_last_error() returns a thread-local message.
 */
#ifndef DEEPSEE_HIP_H
#define DEEPSEE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_INCLUDE_HIP_HIP_RUNTIME_API_H__
typedef struct ihipStream_t* hipStream_t;
#endif

#define DSEE_OK 0
#define DSEE_EINVAL (-1)
#define DSEE_ELAUNCH (-2)
#define DSEE_EUNSUPPORTED (-3)

#define DSEE_ACT_NONE 0
#define DSEE_ACT_LRELU 1
#define DSEE_ACT_RELU 2
#define DSEE_ACT_TANH 3
#define DSEE_ACT_MASK 4 /* conv epilogue only: out = residual > 0 ? v : 0 (ReLU backward fused into a dgrad) */

int dsee_version(void);
const char* dsee_last_error(void);

/* Geometry of one implicit-GEMM convolution.  Output pixel o and tap k read source position
 * p = o*mul + off + k*kdir; the tap contributes iff p >= 0, p is a multiple of 2^dshift and
 * (p >> dshift) < (Hi << ups); the stored pixel is (p >> dshift) >> ups.
 *   forward conv, stride s, padding pd:        mul = s, off = -pd, kdir = +1, dshift = 0
 *   its data gradient (roles of in/out swap):  mul = 1, off = +pd, kdir = -1, dshift = log2(s)
 *   input nearest-upsampled x2 on the fly:     ups = 1 (nn.Upsample, sr.py:57; encoder.py:95,154) */
typedef struct dsee_conv_geom {
  int32_t N, Hi, Wi, Cin; /* stored input  [N][Hi][Wi][Cin], Cin % 4 == 0 */
  int32_t Ho, Wo, Cout;   /* output        [N][Ho][Wo][Cout], Cout % 4 == 0 */
  int32_t KH, KW;
  int32_t mul, off, kdir, dshift, ups;
  int32_t korder;         /* K order of the packed weight this conv reads: 0 tap-major k = tap*Cin + c (any geometry);
                             1 chunk-major k = (c/32)*(taps*32) + tap*32 + c%32 (needs dshift == 0, ups == 0,
                             Cin % 32 == 0): the 9 taps of a 32-channel chunk are adjacent slabs -> L2 reuse */
} dsee_conv_geom;

/* packed weight shape helpers: rows = round_up(Cout,128), row length = round_up(KH*KW*Cin_stored,32) */
int dsee_conv_kpad(int KH, int KW, int Cin_stored);
int dsee_conv_wrows(int Cout);

/* OIHW fp32 -> GEMM-B layouts.  Optional device scalars: value multiplied by (*scale_num / *scale_den)
 * (scale_den = sigma of spectral norm: W = W_orig / sigma, torch.nn.utils.spectral_norm). */
int dsee_pack_weight_fwd(const float* w_oihw, const float* scale_num, const float* scale_den, float* packed, int Cout,
                         int Cin, int KH, int KW, int Cin_stored, int korder, hipStream_t stream);
int dsee_pack_weight_dgrad(const float* w_oihw, const float* scale_num, const float* scale_den, float* packed,
                           int Cout, int Cin, int KH, int KW, int Cout_stored, int korder, hipStream_t stream);

/* out = act(conv(in, W) + bias + residual).  Replaces nn.Conv2d / F.conv2d (+ the following
 * LeakyReLU/ReLU/tanh and the resblock's `x_s + dx`): architecture.py:98,122,127,146-147; sr.py:65,94-95;
 * discriminator.py:78-96; encoder.py:83-99,142-158; architecture.py:151-181 (VGG19).
 * With the dgrad geometry + dsee_pack_weight_dgrad it is the data gradient of the same convs. */
int dsee_conv2d_fwd(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                    const float* residual, int residual_ld, float* out, int act, float slope, hipStream_t stream);
/* The same convolution with both operands scaled by powers of two (device maxima amax_in = max |in|, amax_w =
 * max |w_packed|, 2048-float arrays as written by dsee_absmax) and split into two fp16 terms inside the kernel:
 * 3 fp16 MFMA products per multiply-add instead of the fp32 MFMA, error vs float64 equal to an sgemm's. */
int dsee_conv2d_fwd_f16x2(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                          const float* residual, int residual_ld, float* out, int act, float slope,
                          const float* amax_in, const float* amax_w, hipStream_t stream);

/* round 6: both forms with the operand bound of the NEXT direct layer written by this layer's epilogue: amax_out (optional;
 * 2048 floats, zeroed by the caller) receives max |out| -- no dsee_absmax pass over the activation between two direct layers
 * (VGG conv -> conv / pool -> conv, `architecture.py:151-181`; the discriminator's first layer, `discriminator.py:78-96`). */
int dsee_conv2d_fwd_amax(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                         const float* residual, int residual_ld, float* out, int act, float slope, float* amax_out,
                         hipStream_t stream);
/* flags: DSEE_CONV_NO_HALO keeps a 3 x 3 / stride 1 layer off the halo kernel (which stages an 8 x 16 patch and converts it to split
 * fp16 once per 32-channel chunk instead of once per tap) and a 4-channel input off the LDS-free K = 36 kernel -- for A/B
 * measurements; both issue the same MFMA sequence on the same fragments as the implicit-GEMM kernel: bit-identical results. */
#define DSEE_CONV_NO_HALO 1
int dsee_conv2d_fwd_f16x2_amax(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                               const float* residual, int residual_ld, float* out, int act, float slope,
                               const float* amax_in, const float* amax_w, float* amax_out, int flags, hipStream_t stream);

/* Winograd F(4x4,3x3) path for 3x3 / stride 1 / pad 1 convolutions (same call sites as dsee_conv2d_fwd):
 *   V = dsee_wino43_input(x)                                [36][T][Cin],  T = N*(H/4)*(W/4)
 *   M = dsee_conv2d_fwd_grouped(V viewed [36][T][1][Cin], U, group_stride = wrows*Kpad)   36 GEMMs, 2.25 MAC/px/ch pair
 *   y = dsee_wino43_output(M, bias, residual, act)
 * with U = dsee_wino43_weights(w, transpose_flip = 0) (forward) or 1 (data gradient of the same conv). */
int dsee_wino43_input(const float* x, float* V, int N, int H, int W, int C, float* amax, hipStream_t stream);
/* noise_w != NULL: y += noise_w[c] * eps, eps = the Philox N(0,1) stream (noise_seed, noise_offset) of dsee_rng_fill in
 * NHWC element order -- the NoiseInjection that follows the conv (architecture.py:111-112) fused into its epilogue.
 * One transform plane of M (N * H/4 * W/4 * C elements) must stay below 4 GB (32-bit plane offsets; DSEE_EINVAL otherwise).
 * res_noise_w != NULL: the residual is residual + res_noise_w[c] * eps' (stream res_noise_seed / _offset) -- the resblock
 * shortcut x_s = noise_skip(x) (architecture.py:133-134,127) regenerated from x instead of read from its own tensor. */
int dsee_wino43_output(const float* M, const float* bias, const float* residual, int residual_ld, float* y, int N,
                       int H, int W, int C, int act, float slope, const float* noise_w, uint64_t noise_seed,
                       uint64_t noise_offset, const float* res_noise_w, uint64_t res_noise_seed,
                       uint64_t res_noise_offset, const float* mscale, hipStream_t stream);
/* The fp32 form with the BatchNorm statistics rows of y written in the same pass (see dsee_norm_stats_finalize_parts). */
int dsee_wino43_output_stats(const float* M, const float* bias, const float* residual, int residual_ld, float* y, int N,
                             int H, int W, int C, int act, float slope, const float* noise_w, uint64_t noise_seed,
                             uint64_t noise_offset, const float* res_noise_w, uint64_t res_noise_seed,
                             uint64_t res_noise_offset, float* stats_part, hipStream_t stream);
int dsee_wino43_weights(const float* w_oihw, float* U, int Cout, int Cin, int transpose_flip, int split,
                        const float* amax_w, hipStream_t stream);
/* ... of `layers` equally shaped weights in ONE launch (the generator's ten 512 -> 512 convolutions: their normalised
 * weights sit in the flat output of dsee_spectral_norm_group_fwd, w_stride floats apart; amax_w [layers][2048]) */
int dsee_wino43_weights_batch(const float* w_oihw, float* U, int layers, long w_stride, long u_stride, int Cout, int Cin,
                              int transpose_flip, int split, const float* amax_w, hipStream_t stream);
/* fp32-accurate GEMM on the bf16 matrix cores (operands split into three bf16 terms by their producers, six MFMA
 * products accumulated in fp32; see deepsee_amd/csrc/gemm_bf16x3.hip).  With `split` the producers write slab-major
 * [K/16][rows][3][16] bf16 (1.5x the fp32 bytes) instead of fp32 rows:
 *   C[m][n] = sum_k A[m][k] * B[m / rows_per_group][n][k],  A3 [K/16][M][3][16], B3 [groups][K/16][b_rows][3][16]. */
int dsee_wino43_input_split(const float* x, void* V3, int N, int H, int W, int C, hipStream_t stream);
/* dsee_gemm_bf16x3_af32: A stays fp32 [M][K] (4 instead of 6 bytes per element) and is split inside the kernel */
int dsee_gemm_bf16x3_af32(const float* A, const void* B3, float* C, long M, int N, int K, long rows_per_group, int b_rows,
                          int tile, hipStream_t stream);
int dsee_gemm_bf16x3(const void* A3, const void* B3, float* C, long M, int N, int K, long rows_per_group, int b_rows,
                     int tile, hipStream_t stream);
/* fp16x2: the same fp32-accurate GEMMs with HALF the matrix-core work.  An operand is scaled by the power of two
 * s = 2^(13 - floor(log2(max |x|))) (largest element into [2^13, 2^14)) and split into two fp16 terms h0 + h1 (residual
 * <= 2^-22 |x|, rms 2^-24: below the rounding of the fp32 accumulator); a*b = (a1*b0 + a0*b1 + a0*b0) / (s_a s_b), 3 MFMA products instead of 6.
 * The maxima travel as device scalars: dsee_wino43_input / dsee_wino43_dout / dsee_modulate_bwd_reduce_wino write max |V|,
 * max |dM| into `amax` (atomic max, the caller zeroes it); weights use dsee_absmax(w) (|G g G^T| <= max |g|), and
 * dsee_wino43_weights(..., split = 2, amax_w) writes B2 [K/16][rows][2][16] fp16 scaled by the same function. */
int dsee_absmax(const float* x, long n, float* amax, hipStream_t stream);
/* Half-precision compute mode (BASELINE configs[2]'s 16-bit data-parallel training; opt.precision = "fp16"): the same
 * GEMMs with ONE 16-bit MFMA product per multiply-add -- operands scaled by powers of two and rounded to fp16 (fp32
 * accumulate, fp32 master weights) -- and the Winograd-domain products M / dV stored as scaled fp16: the GEMM writes
 * the inverse scale to *cscale and the output transforms take it as `mscale` / `dvscale` (non-NULL = the tensor is fp16).
 * dsee_wino43_weights split = 3 writes B1 [K/16][rows][16] fp16.  fp16 and not bf16 operands because F(4x4,3x3) amplifies
 * operand rounding ~10x (2.6 % per layer with bf16, 0.33 % with scaled fp16; a direct bf16 conv: 0.24 %).  Checked against
 * the fp32 path, not against the CPU reference (<= 3e-2 on fake). */
int dsee_gemm_f16_af32(const float* A, const void* B1, void* C, long M, int N, int K, long rows_per_group, int b_rows,
                       int tile, const float* amax_a, const float* amax_b, int c_f16, float* cscale, hipStream_t stream);
int dsee_gemm_f16_tn_f32(const float* P, const float* Q, float* C, int groups, long T, int rows_p, int rows_q, int ldc,
                         int splits, const float* amax_p, const float* amax_q, hipStream_t stream);
int dsee_gemm_f16x2_af32(const float* A, const void* B2, float* C, long M, int N, int K, long rows_per_group, int b_rows,
                         int tile, const float* amax_a, const float* amax_b, hipStream_t stream);
/* ... and with the A operand pre-split by its producer (round 3): A2 [K/16][M][2][16] fp16 written by
 * dsee_wino43_input_f16x2 with the power-of-two scale of a_bound x *amax_a fixed BEFORE the transform runs (a_bound >= 100
 * bounds |B^T d B| / max|d|), so the GEMM streams both operands global -> LDS without staging or conversion.  256 x 256
 * (256 x 128 for N an odd multiple of 128) tiles: rows_per_group % 256 == 0, N % 128 == 0.  Layers: architecture.py:98,122
 * (forward convolutions), the adjoint data gradients. */
int dsee_gemm_f16x2_pre(const void* A2, const void* B2, float* C, long M, int N, int K, long rows_per_group, int b_rows,
                        const float* amax_a, float a_bound, const float* amax_b, hipStream_t stream);
/* Round 6: the same two GEMMs (dsee_gemm_f16x2_pre / dsee_gemm_f16p_pre: same operand images, scales, products and output) on
 * the one-wave-per-SIMD kernel of csrc/gemm_w4.hip -- 4 waves x (128 x 128) accumulator tiles per 256 x 256 block (8 instead of
 * 12 KB of LDS fragment reads per 24 MFMAs), one barrier per slab, fragment reads and LDS-DMA requests issued between the
 * MFMAs, a five-stage XOR-swizzled LDS ring (160 KB).  N % 256 == 0 and an even number of 64-byte-row slabs (K % 32 == 0
 * two-term, K % 64 == 0 packed); the hosts fall back to the 8-wave kernels otherwise. */
int dsee_gemm_f16x2_pre_w4(const void* A2, const void* B2, float* C, long M, int N, int K, long rows_per_group, int b_rows,
                           const float* amax_a, float a_bound, const float* amax_b, hipStream_t stream);
int dsee_gemm_f16p_pre_w4(const void* A1, const void* B1, void* C16, long M, int N, int K, long rows_per_group, int b_rows,
                          const float* amax_a, float a_bound, const float* amax_b, float* cscale, hipStream_t stream);
int dsee_gemm_f16x2_tn_f32(const float* P, const float* Q, float* C, int groups, long T, int rows_p, int rows_q, int ldc,
                           int splits, const float* amax_p, const float* amax_q, hipStream_t stream);
/* ... and with Q pre-split (round 3): Q2 = the V2 [rows_q/16][groups*T][2][16] fp16 dsee_wino43_input_f16x2 wrote for the
 * forward pass (power-of-two scale of q_bound x *amax_x); the kernel reads its fragments -- 8 consecutive tiles of a channel --
 * with ds_read_b64_tr_b16 from the landed rows, so the forward's split V is the weight gradient's operand and no fp32 V is kept.
 * rows_p % 256 == 0, rows_q == 160 or rows_q % 128 == 0.  Weight gradients of architecture.py:98,122, normalization.py:107-120. */
int dsee_gemm_f16x2_tn_qpre(const float* P, const void* Q2, float* C, int groups, long T, int rows_p, int rows_q, int ldc,
                            int splits, const float* amax_p, const float* amax_x, float q_bound, hipStream_t stream);
/* ... and with P pre-split too: P2 = dsee_wino43_dout_f16x2's dM2 [rows_p/16][groups*T][2][16] (scale of p_bound x *amax_dy). */
int dsee_gemm_f16x2_tn_pqpre(const void* P2, const void* Q2, float* C, int groups, long T, int rows_p, int rows_q, int ldc,
                             int splits, const float* amax_dy, float p_bound, const float* amax_x, float q_bound,
                             hipStream_t stream);
/* ---- 16-bit STORAGE mode (round 4; opt.precision = "fp16", BASELINE.json configs[2]'s 16-bit arithmetic for the layers of
 * architecture.py:98,122 and normalization.py:107-120,167-213): the Winograd-domain operands are written by their producers as
 * ONE scaled fp16 term per element in the "packed one-term" image -- [K/32][rows][32] fp16, i.e. the 64-byte rows of the
 * fp16x2 image holding 32 k's of one term instead of 2 terms x 16 k's -- so every pre-split kernel streams them through the
 * same LDS-DMA path with half the bytes and one MFMA product per multiply-add (fp32 accumulate); the products M / dV leave
 * the GEMM as scaled fp16.  Producers: dsee_wino43_input_f16p, dsee_wino43_dout_f16p, dsee_modulate_bwd_reduce_wino_f16p,
 * dsee_wino43_weights[_table] (split = 4).  fp32 everywhere else (activations, statistics, master weights, Adam). */
int dsee_gemm_f16p_pre(const void* A1, const void* B1, void* C16, long M, int N, int K, long rows_per_group, int b_rows,
                       const float* amax_a, float a_bound, const float* amax_b, float* cscale, hipStream_t stream);
int dsee_gemm_f16p_tn_pqpre(const void* P1, const void* Q1, float* C, int groups, long T, int rows_p, int rows_q, int ldc,
                            int splits, const float* amax_dy, float p_bound, const float* amax_x, float q_bound,
                            hipStream_t stream);
/* (the saved modulation factor `scale` / `out_scale` of the two f16p entry points below is fp16 [N][H][W][C] as well: it is only
 * ever read by the backward pass, dsee_modulate_bwd_apply_amax takes it with scale_f16 = 1) */
int dsee_modulate_bwd_reduce_wino_f16p(const float* dh, const float* h, const float* x, const float* scale,
                                       const float* mean, const float* invstd, void* dM1, int rows, float* sums, int N,
                                       int H, int W, int C, float slope, float* workspace, const float* amax_g, float bound,
                                       const uint32_t* sign_mask,
        hipStream_t stream);
/* round 6: dsee_spade_fused_fwd on the one-wave-per-SIMD kernel of csrc/spade_fused_w4.hip (4 waves x four 16 x 16 blocks, every
 * MFMA followed by its share of the fragment reads, LDS-DMA requests and the fold / Y-update VALU work; 256 AGPR-resident output
 * accumulators per lane): same arguments, bit-identical results.  Two-term fp16x2 operands, K = 128 or 160. */
int dsee_spade_fused_fwd_w4(const void* V2, const void* U2, const float* amax_cat, float v_bound, const float* amax_u,
                            const float* bias_packed, const float* x, const float* mean, const float* invstd, float* out_h,
                            float* out_scale, int N, int H, int W, int C, int rows, int K, int groups, float add_one,
                            float slope, float* amax_h, float* amax_xhat, uint32_t* sign_mask, hipStream_t stream);
int dsee_spade_fused_fwd_f16p(const void* V1, const void* U1, const float* amax_cat, float v_bound, const float* amax_u,
                              const float* bias_packed, const float* x, const float* mean, const float* invstd, float* out_h,
                              float* out_scale, int N, int H, int W, int C, int rows, int K, int groups, float add_one,
                              float slope, float* amax_h, float* amax_xhat, uint32_t* sign_mask,
                              hipStream_t stream);
int dsee_wino43_input_f16p(const float* x, void* V1, int N, int H, int W, int C, const float* amax_x, float bound,
                           hipStream_t stream);
int dsee_wino43_dout_f16p(const float* dy, void* dM1, int N, int H, int W, int C, const float* amax_dy, float bound,
                          float* workspace, float* dbias, float* dnoise0, uint64_t seed0, uint64_t offset0,
                          float* dnoise1, uint64_t seed1, uint64_t offset1, hipStream_t stream);
/* output / adjoint-input transforms of the scaled-fp16 products (mscale / dvscale = the GEMM's *cscale) that also emit the
 * BatchNorm statistics rows / max |dx| their fp32 counterparts emit */
int dsee_wino43_output_stats_f16(const void* M16, const float* bias, const float* residual, int residual_ld, float* y, int N,
                                 int H, int W, int C, int act, float slope, const float* noise_w, uint64_t noise_seed,
                                 uint64_t noise_offset, const float* res_noise_w, uint64_t res_noise_seed,
                                 uint64_t res_noise_offset, const float* mscale, float* stats_part, hipStream_t stream);
int dsee_wino43_input_adjoint_amax_f16(const void* dV16, float* dx, int N, int H, int W, int C, const float* dvscale,
                                       float* amax_dx, hipStream_t stream);
/* Evaluation metrics on the device: per image PSNR, SSIM and RMSE of `fake` against `real`, both fp32 NHWC
 * [N][H][W][Cs] in [-1, 1] (channels 0..2 used).  Replaces MetricsEvaluator.collect_samples' per-sample CPU loop
 * (evaluator/evaluation.py:88-137: util/util.py:72-103 tensor2im quantisation, evaluator/calculate_PSNR_SSIM.py:71-79
 * calculate_psnr, :81-122 calculate_ssim -- 11x11 Gaussian window, sigma 1.5, valid region, float64 -- and the RMSE of
 * evaluation.py:107-110).  out [N][3] doubles = {psnr (inf for identical images), ssim, rmse}; workspace from
 * dsee_psnr_ssim_workspace; deterministic (no atomics).  H, W > 10. */
size_t dsee_psnr_ssim_workspace(int N, int H, int W);
int dsee_psnr_ssim(const float* fake, const float* real, int N, int H, int W, int Cs, double* workspace,
                   size_t workspace_bytes, double* out, hipStream_t stream);
/* Test hook (no reference counterpart): fills the LDS of every CU with NaN bit patterns, so that a pipelined kernel
 * launched next shows a read of a not-yet-landed LDS stage as NaN instead of as stale but plausible data.  sink: one float. */
int dsee_selftest_lds_poison(float* sink, hipStream_t stream);
int dsee_conv2d_fwd_grouped(const dsee_conv_geom* g, const float* in, const float* w_packed, long group_stride,
                            float* out, hipStream_t stream);
/* Weight gradient of the same convs in the Winograd domain (backward of architecture.py:98,122):
 *   dM = dsee_wino43_dout(dy) = A dY A^T  [36][T][Cout_s];  V = dsee_wino43_input(x);
 *   dsee_wino43_wgrad: dU[xi] = dM[xi]^T V[xi] (one split-K MFMA launch over 36 groups), dw = G^T dU G  (OIHW).
 * T % 32 == 0; workspace from dsee_wino43_wgrad_workspace. */
/* SEAN / SPADE gamma-beta GEMM (normalization.py:107-120,167-213,258-286) in the Winograd domain:
 *   U  = dsee_wino43_weights_table(w2a, table)   [36][N][rows][ca+32]  per-image weights (style tables)
 *   M  = dsee_conv2d_fwd_grouped(V = wino43_input(cat), U)   with 36*N groups of T/N tiles
 *   h, scale = dsee_wino43_output_modulate(M, ...)   output transform + BN-normalise + modulate + LeakyReLU
 * backward: dsee_wino43_wgrad_table (dw2a, dtable) and the plain Winograd data gradient with a ReLU-mask epilogue. */
int dsee_wino43_output_modulate(const float* M, const float* bias_packed, const float* x, const float* mean,
                                const float* invstd, float* out_h, float* out_scale, int N, int H, int W, int C,
                                int rows, float add_one, float slope, const float* mscale, hipStream_t stream);
int dsee_wino43_weights_table(const float* w2a, const float* table, float* U, int N, int rows, int ca, int split,
                              const float* amax_w, hipStream_t stream);
/* The fused form (round 3; deepsee_amd/csrc/spade_fused.hip) of the same forward -- normalization.py:107-120 (SPADE),
 * :167-213 (SEAN), :258-286 (PureSEAN above max_fm_size) + the LeakyReLU of architecture.py:92,114 -- in which the
 * Winograd-domain product M never reaches HBM: one workgroup walks all 36 transform positions of 64 tiles x 32 channels,
 * folds every position's MFMA result into the 4x4 output tiles in registers and normalises / modulates / activates in
 * its epilogue.
 *   V2 = dsee_wino43_input_f16x2(cat, amax_cat, v_bound)  [K/16][36*T][2][16] fp16: the split transform of the embedding,
 *        scaled by a power of two known before the transform runs (|B^T d B| <= 100 max|d|; v_bound >= 100)
 *   U2 = dsee_wino43_weights[_table](..., split = 2, amax_u)   [36*groups][K/16][rows][2][16] fp16
 *   h [, scale] = dsee_spade_fused_fwd(...)   K = 128 (SPADE / capped) or 160 (SEAN: 128 + 32 one-hot), rows = 2 C,
 *        C % 32 == 0, (H/4)*(W/4) % 64 == 0, groups = N (per-image style tables) or 1; out_scale may be NULL.
 *        amax_h (optional, 64-line form, zeroed by the caller): receives max |h| -- the bound the convolution that consumes h
 *        needs BEFORE its input transform runs (dsee_wino43_input_f16x2 + dsee_gemm_f16x2_pre); amax_xhat (optional, same
 *        form): receives max |xhat|, which with max |dh| bounds the gamma/beta gradient of the backward pass
 *        (dsee_modulate_bwd_reduce_wino_f16x2). */
#define DSEE_WINO_V_BOUND 100.0f   /* |B^T d B| <= 100 max|d| for the F(4x4,3x3) input transform */
int dsee_wino43_input_f16x2(const float* x, void* V2, int N, int H, int W, int C, const float* amax_x, float bound,
                            hipStream_t stream);
/* sign_mask (round 4; optional, here and in dsee_modulate_bwd_reduce_wino_f16x2 / _f16p / dsee_modulate_bwd_apply_amax): the
 * LeakyReLU branch of h as bits -- [C/32][N*H*W] words, bit 8 * (c % 4) + (c % 32) / 4 = (h[pixel][c] > 0): the order the
 * forward kernel's lanes vote in, the layout its blocks (64 tiles x 32 channels) write whole lines of -- written by the fused
 * forward; the two backward passes then read 1/32 of the bytes of h (which is all they ever needed of it: `h` may be NULL there). */
int dsee_spade_fused_fwd(const void* V2, const void* U2, const float* amax_cat, float v_bound, const float* amax_u,
                         const float* bias_packed, const float* x, const float* mean, const float* invstd, float* out_h,
                         float* out_scale, int N, int H, int W, int C, int rows, int K, int groups, float add_one,
                         float slope, float* amax_h, float* amax_xhat, uint32_t* sign_mask,
                         hipStream_t stream);
size_t dsee_wino43_wgrad_table_workspace(long T, int N, int ca, int rows);
int dsee_wino43_wgrad_table(const float* V, const float* dM, float* workspace, size_t workspace_bytes, float* dw2a,
                            float* dtable, long T, int N, int ca, int rows, int L, int split, const float* amax_v,
                            const float* amax_dm, hipStream_t stream);
int dsee_wino43_dout(const float* dy, float* dM, int N, int H, int W, int C, float* amax, hipStream_t stream);
/* The same transform with the channel sums that read the same dY riding along: dbias[c] = sum_px dY (bias gradient of the
 * convolution, architecture.py:98,122) and dnoise_k[c] = sum_px dY * eps_k, the gradients of up to two NoiseInjection
 * weights (architecture.py:111-112 noise_middle, :127,133-134 noise_skip; normalization.py:289-304) with eps_k the Philox
 * stream (seed_k, offset_k) dsee_wino43_output drew in the forward pass.  Replaces the separate dsee_channel_dot /
 * dsee_channel_dot_rng passes over dY.  Any of the three outputs may be NULL (not all); 256 % (C/4) == 0. */
/* ... and written PRE-SPLIT (round 3): dM2 [C/16][36*T][2][16] fp16 with the power-of-two scale of bound x *amax_dy, bound
 * >= DSEE_WINO_DM_BOUND, amax_dy = max |dY| written by dY's producer (dsee_modulate_bwd_apply, dsee_sumpool).
 * ROW FACTORS (round 4): EVERY dM this library writes (dsee_wino43_dout[_sums / _split_t / _f16x2 / _f16p],
 * dsee_modulate_bwd_reduce_wino[_f16x2 / _f16p]) holds f_i f_j (A dY A^T)[i][j] with f = (1, 1/4, 1/4, 1/16, 1/16, 1): the
 * absolute row sums of A are (1, 4, 4, 15, 15, 1), so the 36 positions would otherwise span gains 1 ... 225 under the ONE
 * power-of-two scale a split operand has, and the low-gain positions -- the corners, which carry the outer taps of a 3x3 weight
 * gradient -- would lose 7.8 bits; with the factors |dM'| <= max |dY| at every position (bound 1).  The factors are exact powers
 * of two and are undone, in fp32, by the consumers of the GEMM results: dsee_wino43_wgrad[_table] in their G^T dU G stage,
 * dsee_wino43_input_adjoint* as they load dV = dM' U^T.
 * Consumers: dsee_gemm_f16x2_pre (adjoint data gradient) and dsee_wino43_wgrad(split = 6).  The three channel
 * sums of dsee_wino43_dout_sums are optional (workspace: dsee_wino43_dout_f16x2_workspace() bytes when any is requested). */
#define DSEE_WINO_DM_BOUND 1.0f   /* |f_i f_j (A dY A^T)[i][j]| <= max|dY| */
size_t dsee_wino43_dout_f16x2_workspace(void);
int dsee_wino43_dout_f16x2(const float* dy, void* dM2, int N, int H, int W, int C, const float* amax_dy, float bound,
                           float* workspace, float* dbias, float* dnoise0, uint64_t seed0, uint64_t offset0,
                           float* dnoise1, uint64_t seed1, uint64_t offset1, hipStream_t stream);
size_t dsee_wino43_dout_sums_workspace(int C);
int dsee_wino43_dout_sums(const float* dy, float* dM, int N, int H, int W, int C, float* amax, float* workspace,
                          float* dbias, float* dnoise0, uint64_t seed0, uint64_t offset0, float* dnoise1, uint64_t seed1,
                          uint64_t offset1, hipStream_t stream);
/* Data gradient from the SAME dM (adjoint form: no second transform of dy): dV = dM x U^T with
 * U^T = dsee_wino43_weights(w, transpose_flip = 2) [36][rows(Cin)][Cout], then dx = sum over tiles of the overlapping
 * 6x6 patches B dV B^T (gather form).  mask != NULL: dx = mask > 0 ? dx : 0 (ReLU backward, like DSEE_ACT_MASK). */
int dsee_wino43_input_adjoint(const float* dV, const float* mask, int mask_ld, float* dx, int N, int H, int W, int C,
                              const float* dvscale, float* amax_dx, hipStream_t stream);   /* amax_dx: optional, 64-line form */
size_t dsee_wino43_wgrad_workspace(long T, int Cin_stored, int Cout_stored);
int dsee_wino43_wgrad(const float* V, const float* dM, float* workspace, size_t workspace_bytes, float* dw_oihw,
                      long T, int Cin_stored, int Cout_stored, int Cout, int Cin, int split, const float* amax_v,
                      const float* amax_dm, hipStream_t stream);
/* split = 6: as split = 5 and dM is the pre-split dM2 of dsee_wino43_dout_f16x2 (bound DSEE_WINO_DM_BOUND), amax_dm = max |dY|
 * (dsee_gemm_f16x2_tn_pqpre).
 * split = 5: V is the PRE-SPLIT fp16x2 transform dsee_wino43_input_f16x2 wrote for the forward pass with bound
 * DSEE_WINO_V_BOUND (cast to const float*), amax_v = max |x| of the layer input (dsee_gemm_f16x2_tn_qpre).
 * split = 3: as split = 2 with two-term fp16 splits (dsee_gemm_f16x2_tn_f32; amax_v / amax_dm = max |V|, max |dM|).
 * split = 2: V / dM are the plain fp32 transforms (dsee_wino43_input / dsee_wino43_dout), transposed and split inside
 * dsee_gemm_bf16x3_tn_f32 (Cout_stored % 256 == 0, Cin_stored == 160 or % 128 == 0).
 * split = 1: V / dM are the transposed bf16x3 operands [36][T/16][C][3][16 tiles] written by the two producers below
 * and the reduction over tiles runs on the bf16 matrix cores (dsee_gemm_bf16x3_tn, fp32-accurate). */
int dsee_wino43_input_split_t(const float* x, void* V3t, int N, int H, int W, int C, hipStream_t stream);
int dsee_wino43_dout_split_t(const float* dy, void* dM3t, int N, int H, int W, int C, hipStream_t stream);
/* fp32 operands [groups*T][rows] (plain transform outputs), transposed + split inside the kernel */
int dsee_gemm_bf16x3_tn_f32(const float* P, const float* Q, float* C, int groups, long T, int rows_p, int rows_q, int ldc,
                            int splits, hipStream_t stream);
int dsee_gemm_bf16x3_tn(const void* P3t, const void* Q3t, float* C, int groups, long T, int rows_p, int rows_q, int ldc,
                        int splits, hipStream_t stream);

/* 3x3 / stride 1 / pad 1 convolution with <= 4 output channels (the generator's to-RGB layer + tanh, sr.py:65,94-95)
 * and its weight gradient, laid out along the input channels on the fp32 VALU (deepsee_amd/csrc/thin.hip): as an
 * implicit GEMM its N dimension would fill 3 of 32 MFMA columns.  out / dout are [N,H,W,4]; C % 256 == 0, W % 64 == 0. */
/* round 3: the same layer as a 1x1 GEMM (z = x . W^T with the [9*Cout][C] weights, row tap*Cout + co; dsee_conv2d_fwd on
 * the fp32 MFMA, x read once at HBM rate) + a 9-point gather with bias and activation, and its adjoint */
int dsee_thin_gather_fwd(const float* z, const float* bias, float* out, int N, int H, int W, int ldz, int Cout, int act,
                         float slope, hipStream_t stream);
int dsee_thin_gather_bwd(const float* dout, const float* out, float* dz, int N, int H, int W, int ldz, int Cout, int act,
                         float slope, hipStream_t stream);
/* round 6: the general gather -- KH x KW taps, stride 1, padding `pad`, out / dout [N,Ho,Wo,4], Ho = H + 2 pad - KH + 1 -- so that
 * the discriminator's last layer (256 -> 1 channels, 4 x 4, padding 2; discriminator.py:78-96, one of 32 MFMA columns as an
 * implicit GEMM) runs as a 16-output 1x1 GEMM + a 16-point gather; its backward is dsee_thin1x1_bwd + this adjoint. */
int dsee_thin_gather_k_fwd(const float* z, const float* bias, float* out, int N, int H, int W, int ldz, int Cout, int KH, int KW,
                           int pad, int act, float slope, hipStream_t stream);
int dsee_thin_gather_k_bwd(const float* dout, const float* out, float* dz, int N, int H, int W, int ldz, int Cout, int KH,
                           int KW, int pad, int act, float slope, hipStream_t stream);
/* Backward of the 27-output 1x1 GEMM the to-RGB layer runs as (y [M][ldz] = x [M][C] . w^T, w [K][C], K <= 32; sr.py:65,94):
 * dx [M][C] = dz w (NULL: skipped), dw [K][C] = dz^T x (NULL: skipped).  Laid out along the C input channels (a thread owns 4
 * channels, dz of a pixel is block-uniform): exact fp32 FMAs at HBM speed where the implicit-GEMM kernels fill 27 of 128 tile
 * columns.  workspace: dsee_thin1x1_bwd_workspace(C, K) bytes (weight gradient only).  C % 4 == 0, K <= ldz <= 32, ldz % 4 == 0.
 * in_lrelu != 0: x = LeakyReLU(pre) is the output of a convolution whose epilogue applied the activation (the last resblock in
 * front of conv_img, sr.py:94) and which leaves the activation's BACKWARD to this call: dx is the gradient w.r.t. `pre`
 * (dx *= x > 0 ? 1 : slope) and amax_dx (optional, 64-line form) receives max |dx| -- in place of that layer's own pass over
 * (dx, x) -> g, 2 GB of traffic at 256^2. */
size_t dsee_thin1x1_bwd_workspace(int C, int K);
int dsee_thin1x1_bwd(const float* dz, int ldz, const float* w, const float* x, float* dx, float* dw, long M, int C, int K,
                     float* workspace, int in_lrelu, float slope, float* amax_dx, hipStream_t stream);
int dsee_conv3x3_thin_fwd(const float* x, const float* w_oihw, const float* bias, float* out, int N, int H, int W, int C,
                          int Cout, int act, float slope, hipStream_t stream);
size_t dsee_conv3x3_thin_wgrad_workspace(int C);
int dsee_conv3x3_thin_wgrad(const float* x, const float* dout, float* workspace, float* dw_oihw, int N, int H, int W,
                            int C, int Cout, int Cin, hipStream_t stream);

/* Fused SPADE / SEAN / PureSEAN normalisation (normalization.py:107-120, 167-213, 258-286) + the
 * LeakyReLU of architecture.py:92,114:  the implicit GEMM produces (gamma-ish, beta-ish) for 32-channel
 * groups side by side (row order of the packed weight: for 64-channel block b, wave w, half h, lane c:
 * row = b*128 + w*64 + h*32 + c  <->  channel b*64 + w*32 + c, h = 0 gamma / 1 beta) and the epilogue writes
 *   scale = acc_gamma + bias_gamma + add_one ;  h = lrelu(((x-mean)*invstd) * scale + acc_beta + bias_beta)
 * so gamma/beta never reach HBM.  `scale` is saved for the backward pass.
 *
 * SEAN style half as a table: the style map is constant per region, so conv(style_map, W_s) is a conv over the one-hot
 * label map with per-image weights T[n][tap][row][r] = sum_s W_s[row][s][tap] * style[n][r][s] (SURVEY B-7).  When
 * style_table != NULL the last 32 input channels of `in` must hold the one-hot label (19 classes padded to 32,
 * dsee_label_onehot) and their 9 K-slabs read T instead of w_packed (which then covers the first shared_cin channels):
 * +288 K instead of +1152 K for the style half.  Requires korder 1 and Ho*Wo % 128 == 0. */
int dsee_conv2d_modulate_fwd(const dsee_conv_geom* g, const float* in, const float* w_packed,
                             const float* style_table, int shared_cin, const float* bias_packed, const float* x,
                             const float* mean, const float* invstd, float* out_h, float* out_scale, int C,
                             float add_one, float slope, hipStream_t stream);

/* dW[co][ci][kh][kw] = sum_m dout[m][co] * in[src(m,tap)][ci]  (conv_backward weight part); split-K over
 * pixels into `workspace` slabs, reduced in fixed order (deterministic).  Only input channels
 * [Cin_first, Cin_first + Cin_real) are produced (Cin_first > 0 needs korder 1 and a multiple of 32). */
size_t dsee_conv2d_wgrad_workspace(const dsee_conv_geom* g);
int dsee_conv2d_wgrad(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                      size_t workspace_bytes, float* dw_oihw, int Cout_real, int Cin_first, int Cin_real,
                      hipStream_t stream);
/* The same with both operands (in, dout) scaled by powers of two from their device maxima and split into two fp16 terms
 * inside the kernel (LDS transpose reads deliver the pixel-major fragments): 3 fp16 MFMA products per multiply-add. */
int dsee_conv2d_wgrad_f16x2(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                            size_t workspace_bytes, float* dw_oihw, int Cout_real, int Cin_first, int Cin_real,
                            const float* amax_in, const float* amax_dout, hipStream_t stream);

/* wgrad of the SEAN modulate GEMM with a per-image style table (see dsee_conv2d_modulate_fwd): one split-K launch with
 * image-aligned splits; shared columns -> dw_oihw [rows][Cin_shared][KH][KW] (may be NULL), one-hot columns per image ->
 * dtable [N][taps][rows][32]. */
size_t dsee_conv2d_wgrad_table_workspace(const dsee_conv_geom* g);
int dsee_conv2d_wgrad_table(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                            size_t workspace_bytes, float* dw_oihw, int Cin_shared, float* dtable, int L,
                            hipStream_t stream);

/* ------------------------------------------------------------------ normalisation statistics / backward
 * Replaces F.batch_norm(training) of the single-device SynchronizedBatchNorm2d branch
 * (sync_batchnorm/batchnorm.py:65-68; biased var + eps, running stats momentum with unbiased var) with
 * groups = 1, and nn.InstanceNorm2d(affine=False) (normalization.py:47-48) with groups = N.
 * mean/invstd are [groups][C].  workspace: dsee_norm_workspace() bytes. */
size_t dsee_norm_workspace(int N, int HW, int C, int groups);
int dsee_norm_stats(const float* x, int N, int HW, int C, int groups, float eps, float momentum, float* mean,
                    float* invstd, float* running_mean, float* running_var, float* workspace, hipStream_t stream);
/* The same in two calls, for several BatchNorms over the SAME tensor (norm_0 and norm_s of a SPADE resblock both normalise
 * the block input, architecture.py:98,127): one pass over x into `workspace` (dsee_norm_workspace bytes, caller keeps it),
 * then one finalize per norm layer (own running statistics). */
/* BatchNorm statistics from a producer's epilogue (sync_batchnorm/batchnorm.py:65-68 without a pass over x): the kernels
 * that write a SPADE/SEAN norm's input -- dsee_upsample_noise_rng_fwd_stats (architecture.py:98,127 input of norm_0),
 * dsee_wino43_output_stats (conv_0 + noise_middle -> norm_1) -- also write rows (count, mean, M2) x C, one per workgroup,
 * stats_part [dsee_stats_part_rows(items)][3][C] with items = the producer's work items (output float4s resp. tile x channel
 * quads); dsee_norm_stats_finalize_parts folds them in row order (Chan), writes mean / invstd and updates the running
 * statistics exactly as dsee_norm_stats does. */
int dsee_stats_part_rows(long items);
int dsee_norm_stats_finalize_parts(const float* part, int rows, int C, float eps, float momentum, float* mean,
                                   float* invstd, float* running_mean, float* running_var, hipStream_t stream);
int dsee_norm_stats_partial(const float* x, int N, int HW, int C, int groups, float* workspace, hipStream_t stream);
int dsee_norm_stats_finalize(const float* workspace, int N, int HW, int C, int groups, float eps, float momentum,
                             float* mean, float* invstd, float* running_mean, float* running_var, hipStream_t stream);
/* SyncBN over RCCL (option `sync_bn`; reference: the DataParallel branch of SynchronizedBatchNorm2d,
 * sync_batchnorm/batchnorm.py:70-145, which sends (sum, ssum) through Python queue pipes to a master GPU and broadcasts
 * mean / inv_std back).  Here every rank reduces its shard to local[2][C] = (mean, M2) (dsee_norm_stats_local), the
 * rows are all-gathered (2*C floats per rank), and dsee_norm_stats_merge folds them in rank order with Chan's update:
 * statistics of the GLOBAL batch, bit-identical on every rank.  clamp != 0: inv_std = max(var, eps)^-1/2 as
 * batchnorm.py:145; clamp == 0: (var + eps)^-1/2 as F.batch_norm.  Running statistics: momentum update with the
 * unbiased global variance (batchnorm.py:134-143). */
int dsee_norm_stats_local(const float* x, int N, int HW, int C, float* local, float* workspace, hipStream_t stream);
int dsee_norm_stats_merge(const float* gathered, int world, long count_per_rank, int C, float eps, float momentum,
                          int clamp, float* mean, float* invstd, float* running_mean, float* running_var,
                          hipStream_t stream);
/* eval-mode BN: mean = running_mean, invstd = 1/sqrt(running_var + eps) (sr_model.py:85-91 inference) */
int dsee_norm_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                         float* invstd, hipStream_t stream);
/* y = act((x - mean) * invstd): InstanceNorm + LeakyReLU (discriminator.py:88-93, encoder.py:83-99) / + tanh
 * (encoder.py:24-27) */
int dsee_norm_act_fwd(const float* x, const float* mean, const float* invstd, float* y, int N, int HW, int C,
                      int groups, int act, float slope, hipStream_t stream);
int dsee_norm_act_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                      float* dx, int N, int HW, int C, int groups, int act, float slope, float* workspace,
                      hipStream_t stream);
/* round 6: the same two with the operand bound of their consumer written in the same pass: amax_y / amax_dx (optional; 2048
 * floats, zeroed by the caller) receive max |y| / max |dx| -- the discriminator's and encoders' convolutions that read them
 * (discriminator.py:78-96, encoder.py:83-99) then need no dsee_absmax pass */
int dsee_norm_act_fwd_amax(const float* x, const float* mean, const float* invstd, float* y, int N, int HW, int C,
                           int groups, int act, float slope, float* amax_y, hipStream_t stream);
int dsee_norm_act_bwd_amax(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                           float* dx, int N, int HW, int C, int groups, int act, float slope, float* workspace,
                           float* amax_dx, hipStream_t stream);
/* backward of dsee_conv2d_modulate_fwd w.r.t. x and (gamma,beta): see SURVEY.md Appendix E.
 * dgb [M][dgb_ld] is written in the packed gamma/beta column order and is the `dout` for the wgrad/dgrad of the
 * gamma/beta convolution; col_sums [2][C] = (sum g*xhat, sum g) are its bias gradients; dx gets `add` added. */
int dsee_modulate_bwd(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                      const float* invstd, const float* add, float* dx, float* dgb, int dgb_ld, float* col_sums, int N,
                      int HW, int C, float slope, float* workspace, hipStream_t stream);

/* The two halves of dsee_modulate_bwd, so that SyncBN can all-reduce the per-channel sums of the BN backward over the
 * global batch in between: reduce writes dgb and sums[4][C] = (sum d, sum d*xhat, sum g*xhat, sum g) of THIS rank's
 * pixels (d = g*scale; rows 2..3 are the bias gradients = col_sums); apply reads sums[0..1] and inv_count =
 * 1 / (number of pixels behind them). */
int dsee_modulate_bwd_reduce(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                             const float* invstd, float* dgb, int dgb_ld, float* sums, int N, int HW, int C, float slope,
                             float* workspace, hipStream_t stream);
/* reduce half with the gamma/beta gradient written directly in the Winograd domain: dM [36][T][rows = 2C] =
 * A (g*xhat | g) A^T in the packed column order, i.e. dsee_wino43_dout(dgb) without dgb ever existing (it is the
 * operand of dsee_wino43_wgrad[_table] and, through dsee_wino43_input_adjoint, of the embedding's data gradient). */
size_t dsee_modulate_bwd_wino_workspace(int N, int H, int W, int C);
int dsee_modulate_bwd_reduce_wino(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                                  const float* invstd, float* dM, int rows, float* sums, int N, int H, int W, int C,
                                  float slope, float* workspace, float* amax, hipStream_t stream);
/* dsee_modulate_bwd_reduce_wino with dM written PRE-SPLIT (round 3): dM2 [rows/16][36*T][2][16] fp16, scale = power of two of
 * bound x *amax_g with *amax_g >= max |dh| * max(1, max |xhat|) (dsee_amax_product of the maxima written by
 * dsee_wino43_input_adjoint_amax and dsee_spade_fused_fwd), bound >= DSEE_WINO_DM_BOUND.  Consumers: dsee_wino43_wgrad[_table]
 * (split = 6) and dsee_gemm_f16x2_pre (adjoint data gradient of the embedding). */
int dsee_modulate_bwd_reduce_wino_f16x2(const float* dh, const float* h, const float* x, const float* scale,
                                        const float* mean, const float* invstd, void* dM2, int rows, float* sums, int N,
                                        int H, int W, int C, float slope, float* workspace, const float* amax_g, float bound,
                                        const uint32_t* sign_mask,
        hipStream_t stream);
int dsee_amax_product(const float* a, const float* b, float floor_b, float* out, hipStream_t stream);
int dsee_wino43_input_adjoint_amax(const float* dV, float* dx, int N, int H, int W, int C, float* amax_dx, hipStream_t stream);
int dsee_modulate_bwd_apply(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                            const float* invstd, const float* sums, const float* add, float* dx, int N, int HW, int C,
                            float inv_count, float slope, hipStream_t stream);
/* ... also writing max |dx| (64-line form): the operand bound of dsee_wino43_dout_f16x2 for the convolution in front of the norm;
 * scale_f16 != 0: `scale` is the fp16 tensor dsee_spade_fused_fwd_f16p wrote (16-bit storage mode) */
int dsee_modulate_bwd_apply_amax(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                                 const float* invstd, const float* sums, const float* add, float* dx, int N, int HW, int C,
                                 float inv_count, float slope, float* amax_dx, int scale_f16, const uint32_t* sign_mask,
        hipStream_t stream);

/* ------------------------------------------------------------------ label-map kernels (uint8 [N][H][W])
 * mlp_shared = ReLU(conv3x3(one-hot)) (normalization.py:98-101) as a 9-tap gather-sum of weight columns. */
int dsee_onehot_conv3x3_pack(const float* w_oihw, float* table, int Co, int L, hipStream_t stream);
/* onehot_coff >= 0: also writes the 32 one-hot label channels of every pixel at that column offset (= dsee_label_onehot);
 * amax (optional, zeroed by the caller): receives max(amax_floor, max |out|) in the 64-line form of dsee_absmax -- the
 * operand bound of the consumer's fp16 scale without a pass over the embedding. */
int dsee_onehot_conv3x3_fwd(const uint8_t* lab, const float* table, const float* bias, float* out, int N, int H, int W,
                            int shift, int L, int Co, int out_ld, int coff, int relu, int onehot_coff, float* amax,
                            float amax_floor, hipStream_t stream);
size_t dsee_onehot_conv3x3_wgrad_workspace(int N, int H, int W, int shift, int L);
int dsee_onehot_conv3x3_wgrad(const uint8_t* lab, const float* dact, int dact_ld, const float* act, int act_ld, int N,
                              int H, int W, int shift, int L, float* dw_oihw, float* dbias, float* workspace,
                              hipStream_t stream);
/* out[m][coff + r] = (label(m) == r), r in [0, 32): the one-hot input channels of the SEAN style table path */
int dsee_label_onehot(const uint8_t* lab, float* out, int N, int H, int W, int shift, int out_ld, int coff,
                      hipStream_t stream);
/* out[m][coff..] = scale * table[n][label(m)][:]   — SEAN style_map (normalization.py:179-185); backward of
 * extract_style_matrix */
int dsee_label_gather(const uint8_t* lab, const float* table, float* out, int N, int H, int W, int shift, int L, int Cs,
                      int out_ld, int coff, float scale, hipStream_t stream);
/* table[n][r][:] = scale * sum_{m in image n, label(m)=r} in[m][coff..]  — extract_style_matrix
 * (encoder.py:36-49, scale = 1/(H*W)); backward of the style_map gather */
size_t dsee_label_segsum_workspace(int N, int H, int W, int shift, int L, int Cs);
int dsee_label_segsum(const uint8_t* lab, const float* in, int ld, int coff, float* table, int N, int H, int W,
                      int shift, int L, int Cs, float scale, float* workspace, hipStream_t stream);

/* ------------------------------------------------------------------ streaming element-wise kernels */
/* y = nearest_up(x, 2^ups) + noise_w[c] * eps   (nn.Upsample sr.py:57 + NoiseInjection normalization.py:299-304) */
int dsee_upsample_noise_fwd(const float* x, const float* eps, const float* noise_w, float* y, int N, int H, int W, int C,
                            int ups, hipStream_t stream);
/* The same with eps generated in registers from the Philox stream (seed, offset) -- bit-identical to
 * dsee_rng_fill(eps, n, seed, offset, 1) followed by the tensor form, without the 4*M*C-byte tensor -- and the
 * gradient of the noise weights against the regenerated eps (normalization.py:303-304). */
int dsee_upsample_noise_rng_fwd(const float* x, const float* noise_w, float* y, int N, int H, int W, int C, int ups,
                                uint64_t seed, uint64_t offset, hipStream_t stream);
int dsee_upsample_noise_rng_fwd_stats(const float* x, const float* noise_w, float* y, int N, int H, int W, int C, int ups,
                                      uint64_t seed, uint64_t offset, float* stats_part, hipStream_t stream);
int dsee_channel_dot_rng(const float* a, float* out, long M, int C, float* workspace, uint64_t seed, uint64_t offset,
                         hipStream_t stream);
int dsee_sumpool(const float* dy, float* dx, int N, int H, int W, int C, int ups, hipStream_t stream);
int dsee_sumpool_amax(const float* dy, float* dx, int N, int H, int W, int C, int ups, float* amax_dx, hipStream_t stream);
/* UpNoise backward in one pass over dy (both stand-alone passes, dsee_sumpool_amax and dsee_channel_dot_rng, read it in full):
 * dx = sum-pool, *amax_dx = max |dx| (optional), dnoise_w[c] = sum dy * eps(seed, offset).  256 % (C/4) == 0. */
size_t dsee_sumpool_dot_rng_workspace(int N, int H, int W, int C, int ups);
int dsee_sumpool_dot_rng(const float* dy, float* dx, int N, int H, int W, int C, int ups, float* amax_dx, float* dnoise_w,
                         float* workspace, uint64_t seed, uint64_t offset, hipStream_t stream);
size_t dsee_channel_dot_workspace(long M, int C);
int dsee_channel_dot(const float* a, const float* b, float* out, long M, int C, float* workspace, hipStream_t stream);
int dsee_act_fwd(const float* x, float* y, long n, int act, float slope, hipStream_t stream);
int dsee_act_bwd(const float* dy, const float* y, float* dx, long n, int act, float slope, hipStream_t stream);
int dsee_act_bwd_amax(const float* dy, const float* y, float* dx, long n, int act, float slope, float* amax_dx,
                      hipStream_t stream);   /* + max |dx| (64-line form) for dsee_wino43_dout_f16x2 */
int dsee_axpby(const float* a, float alpha, const float* b, float beta, float* y, long n, hipStream_t stream);
/* F.avg_pool2d(3, stride 2, pad 1, count_include_pad=False)  (discriminator.py:46-49) */
int dsee_avgpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t stream);
int dsee_avgpool3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, hipStream_t stream);
/* VGG19 MaxPool2d(2,2) (architecture.py:151-181) */
int dsee_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t stream);
int dsee_maxpool2_bwd(const float* dy, const float* x, float* dx, int N, int H, int W, int C, hipStream_t stream);
/* input preparation (base_manager.py:28-66, data/preprocessor.py:17-41) */
int dsee_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cs, hipStream_t stream);
int dsee_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int Cs, hipStream_t stream);
int dsee_label_to_u8(const float* label, uint8_t* out, long n, hipStream_t stream);
int dsee_bicubic_down(const float* x, float* y, int N, int H, int W, int S, int cs_in, int cs_out, hipStream_t stream);
/* Device input pipeline (replaces the PIL -> float CPU tensors of data/base_dataset.py:87-116,171-201):
 * uint8 HWC images -> NHWC RGB0 fp32 with ToTensor + Normalize((.5,.5,.5),(.5,.5,.5)) and the per-sample horizontal
 * flip (flip[n] != 0; NULL = none); uint8 label maps flipped the same way with 255 ('unknown') -> unknown_to. */
int dsee_image_u8_to_nhwc(const uint8_t* img, const uint8_t* flip, float* out, int N, int H, int W, int cs_out,
                          hipStream_t stream);
int dsee_label_u8_prepare(const uint8_t* lab, const uint8_t* flip, uint8_t* out, int N, int H, int W, int unknown_to,
                          hipStream_t stream);
/* cat([input_semantics, image], dim=1) of sr_model.py:655-668 in NHWC, and the image part of its gradient */
int dsee_build_d_input(const uint8_t* lab, const float* img, float* out, long pixels, int L, int Cs, int img_cs,
                       hipStream_t stream);
int dsee_build_d_input_amax(const uint8_t* lab, const float* img, float* out, long pixels, int L, int Cs, int img_cs,
                            float* amax_out, hipStream_t stream);     /* + max |out| folded into amax_out (optional) */
int dsee_extract_image_grad(const float* din, float* dimg, long pixels, int L, int Cs, int img_cs, hipStream_t stream);
/* Philox4x32-10 fill: N(0,1) (normal != 0) or U[0,1) — replaces tensor.normal_() / torch.rand_like on device */
int dsee_rng_fill(float* out, long n, uint64_t seed, uint64_t offset, int normal, hipStream_t stream);
/* Device-side epoch added to the offset of EVERY Philox stream drawn by the library (dsee_rng_fill,
 * dsee_upsample_noise_rng_fwd, dsee_channel_dot_rng, the noise forms of dsee_wino43_output): epoch_dev points to one
 * uint64 in device memory (NULL: epoch 0).  The (seed, offset) arguments travel by value and are frozen in a captured
 * hipGraph; advancing *epoch_dev between replays (a device-side add, itself part of the graph) gives every replayed
 * training step fresh NoiseInjection draws (normalization.py:299-304 draws a new tensor per forward). */
int dsee_rng_set_epoch(const uint64_t* epoch_dev);

/* ------------------------------------------------------------------ losses (loss.py:68-79,114-119; sr_model.py:529-539)
 * mode 0: L1(a,b)  1: -x (hinge, generator)  2: -min(x-1,0) (D, real)  3: -min(-x-1,0) (D, fake).
 * *loss_out += weight*mean; grad = weight * d mean / d a. */
size_t dsee_loss_workspace(void);
int dsee_loss_fwd_bwd(int mode, const float* a, const float* b, float* grad, long rows, int ld, int valid_c,
                      float weight, float* loss_out, float* workspace, hipStream_t stream);
/* backward alone for an arbitrary upstream gradient (device scalar, NULL = 1): grad = *upstream * weight * d mean / d a
 * (the train step backpropagates sum(losses).mean(), trainer_manager.py:36-37, but a caller may re-weight a term) */
int dsee_loss_bwd(int mode, const float* a, const float* b, float* grad, long rows, int ld, int valid_c, float weight,
                  const float* upstream, hipStream_t stream);

/* ------------------------------------------------------------------ SPADE / SEAN parameter packing
 * normalization.py:208-213 (SEAN), :119 (SPADE), :286 (PureSEAN): scale / offset are linear in the gamma / beta / style
 * convolutions, so the layer's GEMM reads ONE blended weight set in the packed [32 gamma rows | 32 beta rows] order.
 * These kernels replace the ~30 (forward) + ~45 (backward) ATen launches that sigmoid / blend / cat / index_select / pad
 * cost per norm layer and pass (deepsee_amd/csrc/sean_pack.hip).  rows = dsee_sean_pack_rows(C). */
int dsee_sean_pack_rows(int C);
int dsee_sean_pack_fwd(const float* w_gamma, const float* w_beta, const float* ws_gamma, const float* ws_beta,
                       const float* b_gamma, const float* b_beta, const float* bs_gamma, const float* bs_beta,
                       const float* alpha_gamma, const float* alpha_beta, int mode, int C, int K, int S, float* w2a,
                       float* wst, float* b2, float* amax_w2a, hipStream_t stream);
size_t dsee_sean_pack_bwd_workspace(void);
int dsee_sean_pack_bwd(const float* w_gamma, const float* w_beta, const float* ws_gamma, const float* ws_beta,
                       const float* b_gamma, const float* b_beta, const float* bs_gamma, const float* bs_beta,
                       const float* alpha_gamma, const float* alpha_beta, int mode, int C, int K, int S,
                       const float* dw2a, const float* dwst, const float* db2, float* dw_gamma, float* dw_beta,
                       float* dws_gamma, float* dws_beta, float* db_gamma, float* db_beta, float* dbs_gamma,
                       float* dbs_beta, float* dalpha, float* workspace, hipStream_t stream);
/* per-image style tables (normalization.py:182-185 as a table, SURVEY B-7): [N*L][9*rows] GEMM result <-> [N][9][rows][32] */
int dsee_style_table_layout(const float* t, float* table, int N, int L, int rows, float* amax, hipStream_t stream);
int dsee_style_table_layout_bwd(const float* dtable, float* dt, int N, int L, int rows, hipStream_t stream);

/* ------------------------------------------------------------------ spectral norm + Adam */
int dsee_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* sigma, float* w_sn, int R, int K,
                           int power_iter, float eps, float* scratch, hipStream_t stream);
int dsee_spectral_norm_bwd(const float* dw, const float* w_sn, const float* u, const float* v, const float* sigma,
                           float* dw_orig, int R, int K, float* scratch, hipStream_t stream);

/* all spectral-normalised layers of a network at once (5 launches instead of 6 + 2 per layer) */
typedef struct dsee_sn_layer {
  const float* w_orig; /* [R][K] */
  float* u;            /* [R], updated in place when power_iter */
  float* v;            /* [K] */
  int64_t out_off;     /* first element of this layer's w_sn in the flat output */
  int64_t saved_off;   /* u at saved_off, v at saved_off + R in the `saved` buffer */
  int32_t R, K;
  int32_t scratch_off; /* K + R floats of scratch */
  int32_t pad_;
} dsee_sn_layer;
int dsee_spectral_norm_group_fwd(const dsee_sn_layer* layers, int nlayers, const int* work_k, int n_k, const int* work_r,
                                 int n_r, const int* work_e, int n_e, int power_iter, float eps, float* scratch,
                                 float* sigma, float* out, float* saved, float* amax, hipStream_t stream);

typedef struct dsee_adam_tensor {
  int64_t offset;      /* first element in the flat buffers */
  int64_t numel;
  int32_t first_block; /* blocks [first_block, first_block + ceil(numel/1024)) belong to this tensor */
  int32_t step;        /* updates already applied (torch state['step']) */
  float lr;
  int32_t active;      /* 0: the reference's p.grad is None -> skipped */
} dsee_adam_tensor;
int dsee_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   const dsee_adam_tensor* tensors, const int* block_tensor, int nblocks, float beta1, float beta2,
                   float eps, float grad_scale, float clip, hipStream_t stream);
/* grad_flat <- the per-parameter gradient tensors of a backward pass (grad_ptrs[t] = device address of parameter t's
 * contiguous fp32 gradient, 0 = none: zeros), active[t] <- grad_ptrs[t] != 0.  One launch in place of the per-parameter
 * AccumulateGrad additions into persistent .grad views (torch's counterpart of optimizer.zero_grad + backward,
 * trainer_manager.py:33,37). */
int dsee_grad_gather(const int64_t* grad_ptrs, const dsee_adam_tensor* tensors, const int* block_tensor, int nblocks,
                     float* grad_flat, int* active, hipStream_t stream);
/* the same on blocks [first_block, first_block + nblocks) only: one launch per all-reduced gradient chunk, so the
 * update of chunk k overlaps the RCCL all-reduce of chunk k+1 (replaces the reduce-to-GPU0 + optimizer.step() of
 * torch.nn.DataParallel, sync_batchnorm/replicate.py:50-94, trainer_manager.py:37-42) */
int dsee_adam_step_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                         const dsee_adam_tensor* tensors, const int* block_tensor, int first_block, int nblocks,
                         float beta1, float beta2, float eps, float grad_scale, float clip, hipStream_t stream);

/* ---- RCCL behind the ABI: the collectives of the data-parallel path for a host without torch.distributed ----------------
 * The reference's only parallelism is torch.nn.DataParallel with SyncBN callbacks (managers/base_manager.py:15-23,
 * sync_batchnorm/replicate.py:50-94: per-iteration parameter broadcast + gradient reduce to GPU 0; batchnorm.py:105-145 +
 * comm.py:46-133: the master's ReduceAddCoalesced / Broadcast of BN sums through Python queues).  Here: one process per GPU,
 * parameters stay resident, and a step exchanges (1) the flat gradient -- dsee_comm_allreduce_sum per chunk, each followed
 * by that chunk's dsee_adam_step_range with grad_scale = 1/world --, (2) once, the start state -- dsee_comm_broadcast --,
 * and only with SyncBN on (3) [2][C] statistics rows -- dsee_comm_allgather feeding dsee_norm_stats_merge, and
 * dsee_comm_allreduce_sum of the BN backward's two sums.
 *
 * Bootstrap: rank 0 calls dsee_comm_unique_id and ships the DSEE_COMM_ID_BYTES to the other ranks over any channel the host
 * has (a file, a socket, its own RPC); every rank then calls dsee_comm_init with ITS HIP device current.  RCCL is resolved
 * with dlopen at the first dsee_comm_* call (DSEE_RCCL_LIB, else librccl.so.1): the library has no link-time dependency on
 * it.  Collectives are enqueued on `stream` (capturable into a hipGraph) and are in place. */
#define DSEE_COMM_ID_BYTES 128
int dsee_comm_unique_id(void* id_out);
int dsee_comm_init(void** comm_out, const void* id, int world, int rank);
int dsee_comm_world(const void* comm);
int dsee_comm_rank(const void* comm);
int dsee_comm_allreduce_sum(void* comm, float* buf, long n, hipStream_t stream);
int dsee_comm_broadcast(void* comm, void* buf, long nbytes, int root, hipStream_t stream);
/* recv [world][nbytes_per_rank] in rank order (a fixed order: dsee_norm_stats_merge then folds bit-identically on every rank) */
int dsee_comm_allgather(void* comm, const void* send, void* recv, long nbytes_per_rank, hipStream_t stream);
int dsee_comm_destroy(void* comm);

/* ---- coarse entry points: whole-block ops for a host without the Python orchestration (deepsee_amd/csrc/coarse.cpp) --------
 * dsee_sean_norm_fwd = ONE SPADE / SEAN normalisation + LeakyReLU of a SPADEResnetBlock, forward (normalization.py:107-120
 * SPADE when table == NULL, :167-213 SEAN with the style half as per-image tables; architecture.py:92,114), i.e. what
 * deepsee_amd/ops.py::SeanNormTable.forward enqueues on the fused fp32 path, bit-identical to it:
 *   actv = ReLU(mlp_shared(one-hot(labels)))           dsee_onehot_conv3x3_pack / _fwd (+ the 32 one-hot channels with a table)
 *   mean, invstd (training: batch statistics, running statistics updated; else from the running statistics)
 *   V2 = split F(4x4,3x3) transform of [actv | one-hot], U2 = transform of the packed gamma|beta weights (+ per-image tables)
 *   h = LeakyReLU((x - mean) invstd (gamma + add_one) + beta)     dsee_spade_fused_fwd: one kernel, M never reaches HBM
 * labels uint8 [N][lab_h][lab_w], read at stride 2^shift (lab_h >> shift == H); w_shared [128][label_nc][3][3], b_shared [128];
 * w2a [2C][128][3][3] and bias_packed [2C] in the packed gamma|beta row order of dsee_sean_pack_fwd; table [N][9][2C][32] or
 * NULL; x, out_h [N][H][W][C]; out_scale (saved modulation factor), sign_mask ([C/32][N*H*W] words), amax_h (2048 floats,
 * zeroed by the caller: receives max |h|) are optional; mean / invstd [C] are outputs.  C % 64 == 0, (H/4)*(W/4) % 64 == 0.
 * The workspace (dsee_sean_norm_fwd_workspace bytes, caller-owned, 256-byte aligned) holds the embedding, the statistics
 * partials, V2, U2 and three operand maxima; nothing is allocated and the stream is never synchronised. */
size_t dsee_sean_norm_fwd_workspace(int N, int H, int W, int C, int label_nc, int has_table);
int dsee_sean_norm_fwd(const uint8_t* labels, int lab_h, int lab_w, int shift, int label_nc, const float* w_shared,
                       const float* b_shared, const float* w2a, const float* table, const float* bias_packed, const float* x,
                       float* running_mean, float* running_var, int training, float eps, float momentum, float add_one,
                       float slope, float* out_h, float* out_scale, uint32_t* sign_mask, float* mean, float* invstd,
                       float* amax_h, int N, int H, int W, int C, void* workspace, size_t workspace_bytes,
                       hipStream_t stream);

/* dsee_spade_resblock_fwd = ONE SPADEResnetBlock.forward (architecture.py:75-147 with fin == fout: identity shortcut; without
 * NoiseInjection, i.e. inference or add_noise off):
 *   out = act(x + conv_1(lrelu(norm_1(conv_0(lrelu(norm_0(x)))))))
 * two dsee_sean_norm_fwd and two Winograd F(4x4,3x3) convolutions on pre-split fp16x2 operands (dsee_wino43_weights split = 2,
 * dsee_wino43_input_f16x2 scaled from the max |h| the fused norm kernel wrote, dsee_gemm_f16x2_pre, dsee_wino43_output with bias /
 * shortcut / activation).  w_conv_* are the convolutions' EFFECTIVE weights [C][C][3][3] (after spectral normalisation:
 * dsee_spectral_norm_fwd), b_conv_* [C] or NULL; a dsee_norm_layer is a HOST struct of device pointers, laid out as
 * dsee_sean_norm_fwd's arguments.  C % 128 == 0, N (H/4) (W/4) % 256 == 0 (DSEE_EUNSUPPORTED otherwise); out_act = DSEE_ACT_*.
 * The workspace (dsee_spade_resblock_fwd_workspace bytes) holds h, conv_0's output, the Winograd-domain operands and product
 * (7.5 GB at N = 8, 256 x 256, C = 512). */
typedef struct dsee_norm_layer {
  const float* w_shared;     /* [128][label_nc][3][3] */
  const float* b_shared;     /* [128] */
  const float* w2a;          /* [2C][128][3][3], packed gamma|beta rows */
  const float* table;        /* [N][9][2C][32] per-image style table, or NULL (SPADE) */
  const float* bias_packed;  /* [2C] */
  float* running_mean;       /* [C] */
  float* running_var;        /* [C] */
  float add_one;             /* 1: SPADE / SEAN (x_hat (1 + gamma) + beta); 0: PureSEAN */
} dsee_norm_layer;
size_t dsee_spade_resblock_fwd_workspace(int N, int H, int W, int C, int label_nc, int has_table);
int dsee_spade_resblock_fwd(const dsee_norm_layer* norm_0, const float* w_conv_0, const float* b_conv_0,
                            const dsee_norm_layer* norm_1, const float* w_conv_1, const float* b_conv_1,
                            const uint8_t* labels, int lab_h, int lab_w, int shift, int label_nc, const float* x, float* out,
                            int out_act, int training, float eps, float momentum, float slope, int N, int H, int W, int C,
                            void* workspace, size_t workspace_bytes, hipStream_t stream);

/* ---- round 6: the TRAINING pair of the hot block (SURVEY 7 "whole resblock fwd/bwd"; architecture.py:75-147 as configs[1] trains
 * it: norm_0 -> LeakyReLU -> conv_0 -> noise_middle -> norm_1 -> LeakyReLU -> conv_1, + the shortcut x + noise_skip(x); x is what
 * noise_in / the upsample in front of the block left -- dsee_upsample_noise_rng_fwd).
 *   dsee_spade_resblock_train_fwd  runs the forward and keeps what the backward needs in `saved` (dsee_spade_resblock_saved_bytes:
 *       per norm the embedding, the modulation factor, the LeakyReLU sign mask, statistics, the split transform of the embedding
 *       and four operand maxima; conv_0's output; the split transforms of both convolutions' inputs -- 0.64 GB at N = 8, 64 x 64,
 *       C = 512).  NoiseInjection draws are Philox streams (seed, offset) regenerated in registers by the output transforms
 *       (dsee_rng_set_epoch applies); noise == NULL or a NULL weight: no injection.  norm_1's batch statistics come from the rows
 *       conv_0's output transform writes (dsee_wino43_output_stats): no pass over conv_0's output.
 *   dsee_spade_resblock_bwd  runs the whole backward pass from `saved`: dout [N,H,W,C] = dL/d(out) with its maximum amax_dout
 *       (2048-float slot: dsee_absmax, or the amax_dx of the block behind), -> dx and max |dx| (amax_dx, 2048 floats), and every
 *       parameter gradient whose pointer in `grads` is not NULL: the convolutions' EFFECTIVE weights (apply
 *       dsee_spectral_norm_bwd for weight_orig) and biases, the two NoiseInjection weights, per norm mlp_shared (dw_shared,
 *       db_shared: give both or neither for SPADE), the packed gamma|beta weights dw2a [2C][128][3][3], the style table dtable
 *       [N][9][2C][32] (SEAN), and dgamma_beta_sums [2][C] = the per-channel sums of the gamma / beta gradients (the bias
 *       gradient in channel order; dsee_sean_pack_bwd maps it to the packed rows).
 * Both sequence the launches deepsee_amd/ops.py makes on the pre-split fp32 path (same kernels, same operands, same order): the
 * results are bit-identical to the Python module's (tests/test_gpu_ops.py::test_coarse_resblock_training_pair).  Shapes the pre-split
 * kernels do not tile are refused (DSEE_EUNSUPPORTED): C a power of two >= 256, N (H/4) (W/4) % 256 == 0, (H/4) (W/4) % 64 == 0,
 * (36 T / 256) (C / 256) >= 512.  Nothing is allocated, the stream is never synchronised. */
typedef struct dsee_block_noise {
  const float* w_middle;  /* [C] noise_middle.weight (architecture.py:111-112) or NULL */
  uint64_t seed_middle, offset_middle;
  const float* w_skip;    /* [C] noise_skip.weight (architecture.py:133-134) or NULL */
  uint64_t seed_skip, offset_skip;
} dsee_block_noise;
typedef struct dsee_norm_grads {
  float* dw_shared;        /* [128][label_nc][3][3] */
  float* db_shared;        /* [128] */
  float* dw2a;             /* [2C][128][3][3], packed rows */
  float* dtable;           /* [N][9][2C][32] (SEAN) */
  float* dgamma_beta_sums; /* [2][C] */
} dsee_norm_grads;
typedef struct dsee_block_grads {
  dsee_norm_grads norm_0, norm_1;
  float* dw_conv_0;        /* [C][C][3][3] */
  float* db_conv_0;        /* [C] */
  float* dw_conv_1;
  float* db_conv_1;
  float* dw_noise_middle;  /* [C] */
  float* dw_noise_skip;    /* [C] */
} dsee_block_grads;
size_t dsee_spade_resblock_saved_bytes(int N, int H, int W, int C, int label_nc, int has_table);
size_t dsee_spade_resblock_train_fwd_workspace(int N, int H, int W, int C, int label_nc, int has_table);
int dsee_spade_resblock_train_fwd(const dsee_norm_layer* norm_0, const float* w_conv_0, const float* b_conv_0,
                                  const dsee_norm_layer* norm_1, const float* w_conv_1, const float* b_conv_1,
                                  const dsee_block_noise* noise, const uint8_t* labels, int lab_h, int lab_w, int shift,
                                  int label_nc, const float* x, float* out, float eps, float momentum, float slope, int N, int H,
                                  int W, int C, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                  hipStream_t stream);
size_t dsee_spade_resblock_bwd_workspace(int N, int H, int W, int C, int label_nc, int has_table, int lab_h, int lab_w, int shift);
int dsee_spade_resblock_bwd(const dsee_norm_layer* norm_0, const float* w_conv_0, const dsee_norm_layer* norm_1,
                            const float* w_conv_1, const dsee_block_noise* noise, const uint8_t* labels, int lab_h, int lab_w,
                            int shift, int label_nc, const float* x, const float* dout, const float* amax_dout,
                            const dsee_block_grads* grads, float* dx, float* amax_dx, float slope, int N, int H, int W, int C,
                            const void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPSEE_HIP_H */
