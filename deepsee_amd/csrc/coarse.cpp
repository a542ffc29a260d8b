// Coarse entry points of the C ABI ("whole-block ops a non-Python host can call"): one call = one SPADE / SEAN
// normalisation forward of a SPADEResnetBlock, or one whole block forward (dsee_spade_resblock_fwd, at the end) (normalization.py:107-120 SPADE, :167-213 SEAN with the style half as per-image
// tables, + the LeakyReLU of architecture.py:92,114).  They only sequence the fine-grained entry points of this library on
// the caller's stream, inside a workspace the caller owns -- the same launches deepsee_amd/ops.py::SeanNormTable.forward makes
// for the fused fp32 path, so the results are bit-identical to the autograd path (tests/test_gpu_ops.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/deepsee_hip.h"

void dsee_set_error(const char* fmt, ...);

namespace {

constexpr int kHidden = 128;          // nhidden of mlp_shared (normalization.py:95)
constexpr long kAmaxFloats = 64 * 32;   // one operand maximum in the 64-line form of dsee_absmax

inline size_t up256(size_t b) { return (b + 255) / 256 * 256; }

struct NormLayout {
  size_t tab, cat, stats, v2, u2, amax, total;
};

NormLayout norm_layout(int N, int H, int W, int C, int nc, int has_table) {
  const int ld = kHidden + (has_table ? 32 : 0), rows = 2 * C;
  const long T = (long)N * (H / 4) * (W / 4);
  NormLayout l;
  size_t o = 0;
  l.amax = o;  o += up256(3 * kAmaxFloats * sizeof(float));                 // max |cat|, max |U|, max |xhat| (zeroed per call)
  l.tab = o;   o += up256((size_t)9 * nc * kHidden * sizeof(float));        // mlp_shared as a [tap][label][128] table
  l.cat = o;   o += up256((size_t)N * H * W * ld * sizeof(float));          // [embedding (128) | one-hot label (32)]
  l.stats = o; o += up256(dsee_norm_workspace(N, H * W, C, 1));
  l.v2 = o;    o += up256((size_t)36 * T * ld * 2 * sizeof(uint16_t));      // split transform of cat
  l.u2 = o;    o += up256((size_t)36 * (has_table ? N : 1) * rows * ld * 2 * sizeof(uint16_t));
  l.total = o;
  return l;
}

#define DSEE_TRY(call)          \
  do {                          \
    const int rc__ = (call);    \
    if (rc__ != DSEE_OK) return rc__; \
  } while (0)

}  // namespace

extern "C" {

size_t dsee_sean_norm_fwd_workspace(int N, int H, int W, int C, int label_nc, int has_table) {
  return norm_layout(N, H, W, C, label_nc, has_table).total;
}

int dsee_sean_norm_fwd(const uint8_t* labels, int lab_h, int lab_w, int shift, int label_nc, const float* w_shared,
                       const float* b_shared, const float* w2a, const float* table, const float* bias_packed, const float* x,
                       float* running_mean, float* running_var, int training, float eps, float momentum, float add_one,
                       float slope, float* out_h, float* out_scale, uint32_t* sign_mask, float* mean, float* invstd,
                       float* amax_h, int N, int H, int W, int C, void* workspace, size_t workspace_bytes,
                       hipStream_t stream) {
  if (!(labels && w_shared && b_shared && w2a && bias_packed && x && out_h && mean && invstd && workspace)) {
    dsee_set_error("dsee_sean_norm_fwd: NULL argument");
    return DSEE_EINVAL;
  }
  if (!((lab_h >> shift) == H && (lab_w >> shift) == W)) {
    dsee_set_error("dsee_sean_norm_fwd: the label map (%d x %d >> %d) does not cover the %d x %d feature map", lab_h, lab_w,
                   shift, H, W);
    return DSEE_EINVAL;
  }
  if (C % 64 != 0) {     // (dsee_sean_pack_fwd packs gamma|beta in groups of 64 channels: rows == 2 C only then)
    dsee_set_error("dsee_sean_norm_fwd: C %% 64 == 0 required (C = %d)", C);
    return DSEE_EUNSUPPORTED;
  }
  if (N <= 0 || H % 4 != 0 || W % 4 != 0 || ((H / 4) * (W / 4)) % 64 != 0) {
    dsee_set_error("dsee_sean_norm_fwd: H and W multiples of 4 with (H/4) (W/4) %% 64 == 0 required (the fused kernel owns blocks of "
                   "64 Winograd tiles of one image): N = %d, H = %d, W = %d", N, H, W);
    return DSEE_EUNSUPPORTED;
  }
  if (label_nc < 1 || (table != nullptr && label_nc > 32)) {
    dsee_set_error("dsee_sean_norm_fwd: label_nc = %d (the style-table path stores the one-hot label in 32 channels)", label_nc);
    return DSEE_EUNSUPPORTED;
  }
  if (!training && !(running_mean && running_var)) {
    dsee_set_error("dsee_sean_norm_fwd: evaluation mode needs the running statistics");
    return DSEE_EINVAL;
  }
  const int has_t = table != nullptr;
  const NormLayout l = norm_layout(N, H, W, C, label_nc, has_t);
  if (workspace_bytes < l.total) {
    dsee_set_error("dsee_sean_norm_fwd: workspace of %zu bytes, %zu needed (dsee_sean_norm_fwd_workspace)", workspace_bytes,
                   l.total);
    return DSEE_EINVAL;
  }
  char* const ws = static_cast<char*>(workspace);
  float* const amax_cat = reinterpret_cast<float*>(ws + l.amax);
  float* const amax_u = amax_cat + kAmaxFloats;
  float* const amax_xhat = amax_u + kAmaxFloats;
  float* const tab = reinterpret_cast<float*>(ws + l.tab);
  float* const cat = reinterpret_cast<float*>(ws + l.cat);
  float* const stats = reinterpret_cast<float*>(ws + l.stats);
  void* const v2 = ws + l.v2;
  float* const u2 = reinterpret_cast<float*>(ws + l.u2);
  const int ld = kHidden + (has_t ? 32 : 0), rows = 2 * C;

  if (hipMemsetAsync(amax_cat, 0, 3 * kAmaxFloats * sizeof(float), stream) != hipSuccess) {
    dsee_set_error("dsee_sean_norm_fwd: hipMemsetAsync failed");
    return DSEE_ELAUNCH;
  }
  // embedding: actv = ReLU(mlp_shared(one-hot labels)) as a 9-tap gather-sum of weight columns, the 32 one-hot channels of the
  // style-table path behind it in the same launch; max |cat| rides along (>= 1 with one-hot channels present)
  DSEE_TRY(dsee_onehot_conv3x3_pack(w_shared, tab, kHidden, label_nc, stream));
  DSEE_TRY(dsee_onehot_conv3x3_fwd(labels, tab, b_shared, cat, N, lab_h, lab_w, shift, label_nc, kHidden, ld, 0, 1,
                                   has_t ? kHidden : -1, amax_cat, has_t ? 1.0f : 0.0f, stream));
  // param-free BatchNorm statistics of x (sync_batchnorm/batchnorm.py:51-68 single-device branch; running stats updated)
  if (training)
    DSEE_TRY(dsee_norm_stats(x, N, H * W, C, 1, eps, momentum, mean, invstd, running_mean, running_var, stats, stream));
  else
    DSEE_TRY(dsee_norm_eval_stats(running_mean, running_var, C, eps, mean, invstd, stream));
  // operands of the gamma/beta convolution in the Winograd domain, split into two fp16 terms by their producers
  DSEE_TRY(dsee_wino43_input_f16x2(cat, v2, N, H, W, ld, amax_cat, DSEE_WINO_V_BOUND, stream));
  DSEE_TRY(dsee_absmax(w2a, (long)rows * kHidden * 9, amax_u, stream));
  if (has_t) {
    DSEE_TRY(dsee_absmax(table, (long)N * 9 * rows * 32, amax_u, stream));
    DSEE_TRY(dsee_wino43_weights_table(w2a, table, u2, N, rows, kHidden, 2, amax_u, stream));
  } else {
    DSEE_TRY(dsee_wino43_weights(w2a, u2, rows, kHidden, 0, 2, amax_u, stream));
  }
  // gamma/beta GEMM + output transform + normalise + modulate + LeakyReLU: one kernel, M never reaches HBM
  return dsee_spade_fused_fwd(v2, u2, amax_cat, DSEE_WINO_V_BOUND, amax_u, bias_packed, x, mean, invstd, out_h, out_scale, N,
                              H, W, C, rows, ld, has_t ? N : 1, add_one, slope, amax_h, out_scale ? amax_xhat : nullptr,
                              sign_mask, stream);
}


/* ---- one SPADEResnetBlock.forward (architecture.py:75-147 with fin == fout: identity shortcut, no NoiseInjection: inference,
 * or training with add_noise off):  out = act(x + conv_1(lrelu(norm_1(conv_0(lrelu(norm_0(x)))))))  -- two dsee_sean_norm_fwd and
 * two Winograd convolutions on pre-split operands (the input transform's scale comes from the max |h| the fused norm kernel
 * wrote: no pass over h). */
static int wino_conv3x3(const float* h, const float* amax_h, const float* w, const float* bias, const float* residual,
                        float* y, int act, float slope, int N, int H, int W, int C, float* amax_w, void* v2, void* u2, float* m,
                        hipStream_t stream) {
  const long T = (long)N * (H / 4) * (W / 4);
  if (hipMemsetAsync(amax_w, 0, kAmaxFloats * sizeof(float), stream) != hipSuccess) {
    dsee_set_error("dsee_spade_resblock_fwd: hipMemsetAsync (max |w|) failed");
    return DSEE_ELAUNCH;
  }
  DSEE_TRY(dsee_absmax(w, (long)C * C * 9, amax_w, stream));
  DSEE_TRY(dsee_wino43_weights(w, static_cast<float*>(u2), C, C, 0, 2, amax_w, stream));
  DSEE_TRY(dsee_wino43_input_f16x2(h, v2, N, H, W, C, amax_h, DSEE_WINO_V_BOUND, stream));
  DSEE_TRY(dsee_gemm_f16x2_pre(v2, u2, m, 36 * T, C, C, T, C, amax_h, DSEE_WINO_V_BOUND, amax_w, stream));
  return dsee_wino43_output(m, bias, residual, C, y, N, H, W, C, act, slope, nullptr, 0, 0, nullptr, 0, 0, nullptr, stream);
}

namespace {
struct BlockLayout {
  size_t norm, h, dx, stat, amax, v2, u2, m, total;
};
BlockLayout block_layout(int N, int H, int W, int C, int nc, int has_table) {
  const long T = (long)N * (H / 4) * (W / 4);
  const size_t act = (size_t)N * H * W * C * sizeof(float);
  BlockLayout l;
  size_t o = 0;
  l.norm = o; o += up256(norm_layout(N, H, W, C, nc, has_table).total);
  l.h = o;    o += up256(act);                                       // lrelu(norm(.)) of the layer in flight
  l.dx = o;   o += up256(act);                                       // conv_0's output
  l.stat = o; o += up256((size_t)2 * C * sizeof(float));             // mean | invstd
  l.amax = o; o += up256(2 * kAmaxFloats * sizeof(float));           // max |h|, max |w|
  l.v2 = o;   o += up256((size_t)36 * T * C * 2 * sizeof(uint16_t)); // split transform of h
  l.u2 = o;   o += up256((size_t)36 * C * C * 2 * sizeof(uint16_t)); // split transform of the weights
  l.m = o;    o += up256((size_t)36 * T * C * sizeof(float));        // Winograd-domain product
  l.total = o;
  return l;
}
}  // namespace

size_t dsee_spade_resblock_fwd_workspace(int N, int H, int W, int C, int label_nc, int has_table) {
  return block_layout(N, H, W, C, label_nc, has_table).total;
}

int dsee_spade_resblock_fwd(const dsee_norm_layer* norm_0, const float* w_conv_0, const float* b_conv_0,
                            const dsee_norm_layer* norm_1, const float* w_conv_1, const float* b_conv_1,
                            const uint8_t* labels, int lab_h, int lab_w, int shift, int label_nc, const float* x, float* out,
                            int out_act, int training, float eps, float momentum, float slope, int N, int H, int W, int C,
                            void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!(norm_0 && norm_1 && w_conv_0 && w_conv_1 && labels && x && out && workspace)) {
    dsee_set_error("dsee_spade_resblock_fwd: NULL argument");
    return DSEE_EINVAL;
  }
  const long T = (long)N * (H / 4) * (W / 4);
  if (C % 128 != 0 || H % 4 != 0 || W % 4 != 0 || T % 256 != 0) {
    dsee_set_error("dsee_spade_resblock_fwd: C %% 128 == 0 and N (H/4) (W/4) %% 256 == 0 required (C = %d, %ld tiles): the "
                   "pre-split Winograd GEMM takes whole 256 x 128 tiles", C, T);
    return DSEE_EUNSUPPORTED;
  }
  const int has_t = norm_0->table != nullptr;
  if (has_t != (norm_1->table != nullptr)) {
    dsee_set_error("dsee_spade_resblock_fwd: both norm layers of a block are SPADE or both SEAN");
    return DSEE_EINVAL;
  }
  const BlockLayout l = block_layout(N, H, W, C, label_nc, has_t);
  if (workspace_bytes < l.total) {
    dsee_set_error("dsee_spade_resblock_fwd: workspace of %zu bytes, %zu needed (dsee_spade_resblock_fwd_workspace)",
                   workspace_bytes, l.total);
    return DSEE_EINVAL;
  }
  char* const ws = static_cast<char*>(workspace);
  float* const h = reinterpret_cast<float*>(ws + l.h);
  float* const dx = reinterpret_cast<float*>(ws + l.dx);
  float* const mean = reinterpret_cast<float*>(ws + l.stat);
  float* const invstd = mean + C;
  float* const amax_h = reinterpret_cast<float*>(ws + l.amax);
  float* const amax_w = amax_h + kAmaxFloats;
  float* const m = reinterpret_cast<float*>(ws + l.m);
  const size_t norm_bytes = l.h - l.norm;
  const dsee_norm_layer* norms[2] = {norm_0, norm_1};
  const float* convw[2] = {w_conv_0, w_conv_1};
  const float* convb[2] = {b_conv_0, b_conv_1};
  const float* in = x;
  for (int i = 0; i < 2; ++i) {
    const dsee_norm_layer* nl = norms[i];
    if (hipMemsetAsync(amax_h, 0, kAmaxFloats * sizeof(float), stream) != hipSuccess) {
      dsee_set_error("dsee_spade_resblock_fwd: hipMemsetAsync (max |h|) failed");
      return DSEE_ELAUNCH;
    }
    DSEE_TRY(dsee_sean_norm_fwd(labels, lab_h, lab_w, shift, label_nc, nl->w_shared, nl->b_shared, nl->w2a, nl->table,
                                nl->bias_packed, in, nl->running_mean, nl->running_var, training, eps, momentum, nl->add_one,
                                slope, h, nullptr, nullptr, mean, invstd, amax_h, N, H, W, C, ws + l.norm, norm_bytes, stream));
    DSEE_TRY(wino_conv3x3(h, amax_h, convw[i], convb[i], i == 1 ? x : nullptr, i == 1 ? out : dx, i == 1 ? out_act : DSEE_ACT_NONE,
                          slope, N, H, W, C, amax_w, ws + l.v2, ws + l.u2, m, stream));
    in = dx;
  }
  return DSEE_OK;
}

}  // extern "C"
