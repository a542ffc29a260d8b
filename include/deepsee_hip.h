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
} dsee_conv_geom;

/* packed weight shape helpers: rows = round_up(Cout,128), row length = round_up(KH*KW*Cin_stored,32) */
int dsee_conv_kpad(int KH, int KW, int Cin_stored);
int dsee_conv_wrows(int Cout);

/* OIHW fp32 -> GEMM-B layouts.  Optional device scalars: value multiplied by (*scale_num / *scale_den)
 * (scale_den = sigma of spectral norm: W = W_orig / sigma, torch.nn.utils.spectral_norm). */
int dsee_pack_weight_fwd(const float* w_oihw, const float* scale_num, const float* scale_den, float* packed, int Cout,
                         int Cin, int KH, int KW, int Cin_stored, hipStream_t stream);
int dsee_pack_weight_dgrad(const float* w_oihw, const float* scale_num, const float* scale_den, float* packed,
                           int Cout, int Cin, int KH, int KW, int Cout_stored, hipStream_t stream);

/* out = act(conv(in, W) + bias + residual).  Replaces nn.Conv2d / F.conv2d (+ the following
 * LeakyReLU/ReLU/tanh and the resblock's `x_s + dx`): architecture.py:98,122,127,146-147; sr.py:65,94-95;
 * discriminator.py:78-96; encoder.py:83-99,142-158; architecture.py:151-181 (VGG19).
 * With the dgrad geometry + dsee_pack_weight_dgrad it is the data gradient of the same convs. */
int dsee_conv2d_fwd(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                    const float* residual, float* out, int act, float slope, hipStream_t stream);

/* Fused SPADE / SEAN / PureSEAN normalisation (normalization.py:107-120, 167-213, 258-286) + the
 * LeakyReLU of architecture.py:92,114:  the implicit GEMM produces (gamma-ish, beta-ish) for 32-channel
 * groups side by side (row order of the packed weight: for 64-channel block b, wave w, half h, lane c:
 * row = b*128 + w*64 + h*32 + c  <->  channel b*64 + w*32 + c, h = 0 gamma / 1 beta) and the epilogue writes
 *   scale = acc_gamma + bias_gamma + add_one ;  h = lrelu(((x-mean)*invstd) * scale + acc_beta + bias_beta)
 * so gamma/beta never reach HBM.  `scale` is saved for the backward pass. */
int dsee_conv2d_modulate_fwd(const dsee_conv_geom* g, const float* in, const float* w_packed,
                             const float* bias_packed, const float* x, const float* mean, const float* invstd,
                             float* out_h, float* out_scale, int C, float add_one, float slope, hipStream_t stream);

/* dW[co][ci][kh][kw] = sum_m dout[m][co] * in[src(m,tap)][ci]  (conv_backward weight part); split-K over
 * pixels into `workspace` slabs, reduced in fixed order (deterministic). */
size_t dsee_conv2d_wgrad_workspace(const dsee_conv_geom* g);
int dsee_conv2d_wgrad(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                      size_t workspace_bytes, float* dw_oihw, int Cout_real, int Cin_real, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPSEE_HIP_H */
