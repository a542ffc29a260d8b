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

// =====================================================================================================================
// Round 6 (VERDICT r5 "missing" #3, SURVEY 7 "whole resblock fwd/bwd"): the TRAINING pair of the hot block -- one call runs
// SPADEResnetBlock.forward as configs[1] trains it (architecture.py:75-147: norm_0 -> LeakyReLU -> conv_0 -> noise_middle ->
// norm_1 -> LeakyReLU -> conv_1, + the shortcut x + noise_skip(x); the block's input x is what noise_in / the upsample left),
// keeping what the backward pass needs in a caller-owned `saved` area; one call runs the whole backward pass from it.  Both only
// sequence this library's fine-grained entry points, in the order and with the operands deepsee_amd/ops.py (SeanNormTable,
// Conv2d on the pre-split Winograd path) uses -- so a host without Python gets the same numbers, bit for bit.
namespace {

struct Arena {
  char* base;
  size_t off;
  template <typename T>
  T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += up256(n * sizeof(T));
    return p;
  }
};

struct NormSaved {
  float *cat, *scale, *mean, *invstd, *amax;   // amax: [4][2048] = max |cat|, max |U| (w2a, table), max |xhat|, max |h|
  uint32_t* mask;
  uint16_t* v2cat;
};
struct BlockSaved {
  NormSaved n[2];
  float *dx0, *amax_w;   // conv_0's output (norm_1's input); [2][2048] = max |w_conv_0|, max |w_conv_1|
  uint16_t* v2h[2];      // split transforms of the two convolutions' inputs
};

BlockSaved saved_layout(Arena& a, int N, int H, int W, int C, int has_t) {
  const int ld = kHidden + (has_t ? 32 : 0);
  const size_t px = (size_t)N * H * W, T = (size_t)N * (H / 4) * (W / 4);
  BlockSaved s;
  for (int i = 0; i < 2; ++i) {
    s.n[i].cat = a.take<float>(px * ld);
    s.n[i].scale = a.take<float>(px * C);
    s.n[i].mask = a.take<uint32_t>(px * (C / 32));
    s.n[i].mean = a.take<float>(C);
    s.n[i].invstd = a.take<float>(C);
    s.n[i].amax = a.take<float>(4 * kAmaxFloats);
    s.n[i].v2cat = a.take<uint16_t>(36 * T * ld * 2);
    s.v2h[i] = a.take<uint16_t>(36 * T * C * 2);
  }
  s.dx0 = a.take<float>(px * C);
  s.amax_w = a.take<float>(2 * kAmaxFloats);
  return s;
}

int train_shape_ok(const char* who, int N, int H, int W, int C, int label_nc, int has_t) {
  const long T = (long)N * (H / 4) * (W / 4);
  if (N <= 0 || H % 4 || W % 4 || ((H / 4) * (W / 4)) % 64 || T % 256 || C % 256 || (C & (C - 1)) ||
      (36 * T / 256) * (C / 256) < 512 || 256 % (C / 4) != 0) {
    dsee_set_error("%s: the pre-split training path takes C a power of two >= 256 (<= 1024), N (H/4) (W/4) %% 256 == 0, (H/4) (W/4) "
                   "%% 64 == 0 and at least 512 GEMM tiles (36 T / 256 x C / 256): N = %d, H = %d, W = %d, C = %d", who, N, H, W, C);
    return DSEE_EUNSUPPORTED;
  }
  if (label_nc < 1 || (has_t && label_nc > 32)) {
    dsee_set_error("%s: label_nc = %d", who, label_nc);
    return DSEE_EUNSUPPORTED;
  }
  return DSEE_OK;
}

int zero(float* p, size_t n, hipStream_t st, const char* who) {
  if (hipMemsetAsync(p, 0, n * sizeof(float), st) != hipSuccess) {
    dsee_set_error("%s: hipMemsetAsync failed", who);
    return DSEE_ELAUNCH;
  }
  return DSEE_OK;
}

int gemm_nt(const void* a2, const void* b2, float* c, long M, int Nn, int K, long rpg, int brows, const float* amax_a, float bound,
            const float* amax_b, hipStream_t st) {
  // (the same choice deepsee_amd/ops.py::_gemm_pre makes: the one-wave-per-SIMD kernel where its shape rules hold)
  if (Nn % 256 == 0 && K % 32 == 0) return dsee_gemm_f16x2_pre_w4(a2, b2, c, M, Nn, K, rpg, brows, amax_a, bound, amax_b, st);
  return dsee_gemm_f16x2_pre(a2, b2, c, M, Nn, K, rpg, brows, amax_a, bound, amax_b, st);
}

struct FwdScratch {
  float *tab, *stats, *h, *m, *part;
  uint16_t *u2n, *u2c;
};
FwdScratch fwd_scratch(Arena& a, int N, int H, int W, int C, int nc, int has_t) {
  const int ld = kHidden + (has_t ? 32 : 0), rows = 2 * C;
  const size_t px = (size_t)N * H * W, T = (size_t)N * (H / 4) * (W / 4);
  FwdScratch f;
  f.tab = a.take<float>((size_t)9 * nc * kHidden);
  f.stats = a.take<float>(dsee_norm_workspace(N, H * W, C, 1) / sizeof(float) + 64);
  f.u2n = a.take<uint16_t>((size_t)36 * (has_t ? N : 1) * rows * ld * 2);
  f.h = a.take<float>(px * C);
  f.u2c = a.take<uint16_t>((size_t)36 * C * C * 2);
  f.m = a.take<float>(36 * T * C);
  f.part = a.take<float>((size_t)dsee_stats_part_rows((long)T * (C / 4)) * 3 * C);
  return f;
}

}  // namespace

extern "C" {

size_t dsee_spade_resblock_saved_bytes(int N, int H, int W, int C, int label_nc, int has_table) {
  (void)label_nc;
  Arena a = {nullptr, 0};
  saved_layout(a, N, H, W, C, has_table);
  return a.off;
}

size_t dsee_spade_resblock_train_fwd_workspace(int N, int H, int W, int C, int label_nc, int has_table) {
  Arena a = {nullptr, 0};
  fwd_scratch(a, N, H, W, C, label_nc, has_table);
  return a.off;
}

int dsee_spade_resblock_train_fwd(const dsee_norm_layer* norm_0, const float* w_conv_0, const float* b_conv_0,
                                  const dsee_norm_layer* norm_1, const float* w_conv_1, const float* b_conv_1,
                                  const dsee_block_noise* noise, const uint8_t* labels, int lab_h, int lab_w, int shift,
                                  int label_nc, const float* x, float* out, float eps, float momentum, float slope, int N, int H,
                                  int W, int C, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                  hipStream_t stream) {
  static const char* who = "dsee_spade_resblock_train_fwd";
  if (!(norm_0 && norm_1 && w_conv_0 && w_conv_1 && labels && x && out && saved && workspace)) {
    dsee_set_error("%s: NULL argument", who);
    return DSEE_EINVAL;
  }
  const int has_t = norm_0->table != nullptr;
  if (has_t != (norm_1->table != nullptr)) {
    dsee_set_error("%s: both norm layers of a block are SPADE or both SEAN", who);
    return DSEE_EINVAL;
  }
  if (!((lab_h >> shift) == H && (lab_w >> shift) == W)) {
    dsee_set_error("%s: the label map (%d x %d >> %d) does not cover the %d x %d feature map", who, lab_h, lab_w, shift, H, W);
    return DSEE_EINVAL;
  }
  DSEE_TRY(train_shape_ok(who, N, H, W, C, label_nc, has_t));
  Arena sa = {static_cast<char*>(saved), 0}, wa = {static_cast<char*>(workspace), 0};
  const BlockSaved S = saved_layout(sa, N, H, W, C, has_t);
  const FwdScratch F = fwd_scratch(wa, N, H, W, C, label_nc, has_t);
  if (saved_bytes < sa.off || workspace_bytes < wa.off) {
    dsee_set_error("%s: saved area of %zu bytes (%zu needed), workspace of %zu bytes (%zu needed)", who, saved_bytes, sa.off,
                   workspace_bytes, wa.off);
    return DSEE_EINVAL;
  }
  const int ld = kHidden + (has_t ? 32 : 0), rows = 2 * C;
  const long T = (long)N * (H / 4) * (W / 4);
  const dsee_norm_layer* norms[2] = {norm_0, norm_1};
  const float* convw[2] = {w_conv_0, w_conv_1};
  const float* convb[2] = {b_conv_0, b_conv_1};
  DSEE_TRY(zero(S.amax_w, 2 * kAmaxFloats, stream, who));
  const float* in = x;
  for (int i = 0; i < 2; ++i) {
    const dsee_norm_layer* nl = norms[i];
    const NormSaved& ns = S.n[i];
    float *amax_cat = ns.amax, *amax_u = ns.amax + kAmaxFloats, *amax_xhat = ns.amax + 2 * kAmaxFloats, *amax_h = ns.amax + 3 * kAmaxFloats;
    DSEE_TRY(zero(ns.amax, 4 * kAmaxFloats, stream, who));
    // ---- SeanNormTable.forward (ops.py): embedding (+ one-hot channels), statistics, operand transforms, fused kernel
    DSEE_TRY(dsee_onehot_conv3x3_pack(nl->w_shared, F.tab, kHidden, label_nc, stream));
    DSEE_TRY(dsee_onehot_conv3x3_fwd(labels, F.tab, nl->b_shared, ns.cat, N, lab_h, lab_w, shift, label_nc, kHidden, ld, 0, 1,
                                     has_t ? kHidden : -1, amax_cat, has_t ? 1.0f : 0.0f, stream));
    if (i == 0)      // (x comes from outside: a statistics pass; norm_1's rows were written by conv_0's output transform)
      DSEE_TRY(dsee_norm_stats(in, N, H * W, C, 1, eps, momentum, ns.mean, ns.invstd, nl->running_mean, nl->running_var, F.stats, stream));
    else
      DSEE_TRY(dsee_norm_stats_finalize_parts(F.part, dsee_stats_part_rows(T * (C / 4)), C, eps, momentum, ns.mean, ns.invstd,
                                              nl->running_mean, nl->running_var, stream));
    DSEE_TRY(dsee_wino43_input_f16x2(ns.cat, ns.v2cat, N, H, W, ld, amax_cat, DSEE_WINO_V_BOUND, stream));
    DSEE_TRY(dsee_absmax(nl->w2a, (long)rows * kHidden * 9, amax_u, stream));
    if (has_t) {
      DSEE_TRY(dsee_absmax(nl->table, (long)N * 9 * rows * 32, amax_u, stream));
      DSEE_TRY(dsee_wino43_weights_table(nl->w2a, nl->table, reinterpret_cast<float*>(F.u2n), N, rows, kHidden, 2, amax_u, stream));
    } else {
      DSEE_TRY(dsee_wino43_weights(nl->w2a, reinterpret_cast<float*>(F.u2n), rows, kHidden, 0, 2, amax_u, stream));
    }
    DSEE_TRY(dsee_spade_fused_fwd(ns.v2cat, F.u2n, amax_cat, DSEE_WINO_V_BOUND, amax_u, nl->bias_packed, in, ns.mean, ns.invstd, F.h,
                                  ns.scale, N, H, W, C, rows, ld, has_t ? N : 1, nl->add_one, slope, amax_h, amax_xhat, ns.mask,
                                  stream));
    // ---- Conv2d.forward on the pre-split Winograd path: V2 of h kept for the weight gradient
    float* amax_w = S.amax_w + i * kAmaxFloats;
    DSEE_TRY(dsee_absmax(convw[i], (long)C * C * 9, amax_w, stream));
    DSEE_TRY(dsee_wino43_weights(convw[i], reinterpret_cast<float*>(F.u2c), C, C, 0, 2, amax_w, stream));
    DSEE_TRY(dsee_wino43_input_f16x2(F.h, S.v2h[i], N, H, W, C, amax_h, DSEE_WINO_V_BOUND, stream));
    DSEE_TRY(gemm_nt(S.v2h[i], F.u2c, F.m, 36 * T, C, C, T, C, amax_h, DSEE_WINO_V_BOUND, amax_w, stream));
    if (i == 0) {
      // conv_0: + bias, + noise_middle (architecture.py:111-112) regenerated in registers, BatchNorm rows of the result for norm_1
      DSEE_TRY(dsee_wino43_output_stats(F.m, convb[0], nullptr, C, S.dx0, N, H, W, C, DSEE_ACT_NONE, slope,
                                        noise ? noise->w_middle : nullptr, noise ? noise->seed_middle : 0,
                                        noise ? noise->offset_middle : 0, nullptr, 0, 0, F.part, stream));
      in = S.dx0;
    } else {
      // conv_1: + bias + the shortcut x + w_skip * eps_skip (architecture.py:127,133-134)
      DSEE_TRY(dsee_wino43_output(F.m, convb[1], x, C, out, N, H, W, C, DSEE_ACT_NONE, slope, nullptr, 0, 0,
                                  noise ? noise->w_skip : nullptr, noise ? noise->seed_skip : 0, noise ? noise->offset_skip : 0,
                                  nullptr, stream));
    }
  }
  return DSEE_OK;
}

}  // extern "C"

namespace {
struct BwdScratch {
  uint16_t *dm2c, *utc, *dm2n, *ute;
  float *wsw, *dv, *dh, *dmid, *sums, *wsmod, *wst, *dve, *dactv, *wse, *amax, *dbias, *chws, *g1, *oh;
  size_t wsw_bytes, wst_bytes, wse_bytes;
};
BwdScratch bwd_scratch(Arena& a, int N, int H, int W, int C, int nc, int has_t, int lab_h, int lab_w, int shift) {
  const int ld = kHidden + (has_t ? 32 : 0), rows = 2 * C;
  const size_t px = (size_t)N * H * W, T = (size_t)N * (H / 4) * (W / 4);
  BwdScratch b;
  b.dm2c = a.take<uint16_t>(36 * T * C * 2);
  b.utc = a.take<uint16_t>((size_t)36 * dsee_conv_wrows(C) * C * 2);
  b.wsw_bytes = dsee_wino43_wgrad_workspace((long)T, C, C);
  b.wsw = a.take<float>(b.wsw_bytes / sizeof(float) + 64);
  b.dv = a.take<float>(36 * T * C);
  b.dh = a.take<float>(px * C);      // gradient w.r.t. a norm's output
  b.dmid = a.take<float>(px * C);    // gradient w.r.t. conv_0's output
  b.dm2n = a.take<uint16_t>(36 * T * rows * 2);
  b.sums = a.take<float>(4 * C);
  b.wsmod = a.take<float>(dsee_modulate_bwd_wino_workspace(N, H, W, C) / sizeof(float) + 64);
  b.wst_bytes = has_t ? dsee_wino43_wgrad_table_workspace((long)T, N, kHidden, rows) : dsee_wino43_wgrad_workspace((long)T, ld, rows);
  b.wst = a.take<float>(b.wst_bytes / sizeof(float) + 64);
  b.ute = a.take<uint16_t>((size_t)36 * dsee_conv_wrows(kHidden) * rows * 2);
  b.dve = a.take<float>(36 * T * kHidden);
  b.dactv = a.take<float>(px * kHidden);
  dsee_conv_geom g = {N, H, W, ld, H, W, kHidden, 3, 3, 1, -1, 1, 0, 0, 1};
  // (SPADE-only norms: mlp_shared's weight gradient runs on the MFMA kernel over 32 materialised one-hot channels, ops.py round 6)
  dsee_conv_geom g32 = {N, H, W, 32, H, W, kHidden, 3, 3, 1, -1, 1, 0, 0, 1};
  const size_t w1 = dsee_conv2d_wgrad_workspace(has_t ? &g : &g32);
  b.wse_bytes = w1;
  b.wse = a.take<float>(w1 / sizeof(float) + 64);
  b.g1 = has_t ? nullptr : a.take<float>(px * kHidden);
  b.oh = has_t ? nullptr : a.take<float>(px * 32);
  b.amax = a.take<float>(8 * kAmaxFloats);
  b.dbias = a.take<float>(3 * C + 64);
  const size_t c1 = dsee_channel_dot_workspace((long)px, kHidden), c2 = dsee_wino43_dout_f16x2_workspace();
  b.chws = a.take<float>((c1 > c2 ? c1 : c2) / sizeof(float) + 64);
  return b;
}
}  // namespace

extern "C" {

size_t dsee_spade_resblock_bwd_workspace(int N, int H, int W, int C, int label_nc, int has_table, int lab_h, int lab_w, int shift) {
  Arena a = {nullptr, 0};
  bwd_scratch(a, N, H, W, C, label_nc, has_table, lab_h, lab_w, shift);
  return a.off;
}

int dsee_spade_resblock_bwd(const dsee_norm_layer* norm_0, const float* w_conv_0, const dsee_norm_layer* norm_1,
                            const float* w_conv_1, const dsee_block_noise* noise, const uint8_t* labels, int lab_h, int lab_w,
                            int shift, int label_nc, const float* x, const float* dout, const float* amax_dout,
                            const dsee_block_grads* grads, float* dx, float* amax_dx, float slope, int N, int H, int W, int C,
                            const void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  static const char* who = "dsee_spade_resblock_bwd";
  if (!(norm_0 && norm_1 && w_conv_0 && w_conv_1 && labels && x && dout && amax_dout && grads && dx && amax_dx && saved && workspace)) {
    dsee_set_error("%s: NULL argument", who);
    return DSEE_EINVAL;
  }
  const int has_t = norm_0->table != nullptr;
  DSEE_TRY(train_shape_ok(who, N, H, W, C, label_nc, has_t));
  Arena sa = {const_cast<char*>(static_cast<const char*>(saved)), 0}, wa = {static_cast<char*>(workspace), 0};
  const BlockSaved S = saved_layout(sa, N, H, W, C, has_t);
  const BwdScratch B = bwd_scratch(wa, N, H, W, C, label_nc, has_t, lab_h, lab_w, shift);
  if (saved_bytes < sa.off || workspace_bytes < wa.off) {
    dsee_set_error("%s: saved area of %zu bytes (%zu needed), workspace of %zu bytes (%zu needed)", who, saved_bytes, sa.off,
                   workspace_bytes, wa.off);
    return DSEE_EINVAL;
  }
  const int ld = kHidden + (has_t ? 32 : 0), rows = 2 * C, rows_tc = dsee_conv_wrows(C), rows_te = dsee_conv_wrows(kHidden);
  const long T = (long)N * (H / 4) * (W / 4);
  const size_t px = (size_t)N * H * W;
  const dsee_norm_layer* norms[2] = {norm_0, norm_1};
  const dsee_norm_grads* ngr[2] = {&grads->norm_0, &grads->norm_1};
  const float* convw[2] = {w_conv_0, w_conv_1};
  float* dwc[2] = {grads->dw_conv_0, grads->dw_conv_1};
  float* dbc[2] = {grads->db_conv_0, grads->db_conv_1};
  DSEE_TRY(zero(B.amax, 8 * kAmaxFloats, stream, who));
  float *amax_dh = B.amax, *amax_g = B.amax + kAmaxFloats, *amax_dmid = B.amax + 2 * kAmaxFloats, *amax_da = B.amax + 3 * kAmaxFloats;

  const float* g = dout;              // gradient w.r.t. the output of the convolution in flight
  const float* amax_gc = amax_dout;
  for (int i = 1; i >= 0; --i) {
    const NormSaved& ns = S.n[i];
    const float *amax_cat = ns.amax, *amax_u = ns.amax + kAmaxFloats, *amax_xhat = ns.amax + 2 * kAmaxFloats, *amax_h = ns.amax + 3 * kAmaxFloats;
    const float* amax_w = S.amax_w + i * kAmaxFloats;
    // ---- Conv2d.backward (ops.py::_wino_wgrad with w_for_dx): ONE A dY A^T, pre-split, carrying the bias / NoiseInjection-weight
    //      sums; weight gradient on the kept V2; data gradient in the adjoint form
    float* dnw0 = (i == 0 && noise && noise->w_middle) ? grads->dw_noise_middle : nullptr;   // conv_0: y += w_middle * eps
    float* dnw1 = (i == 1 && noise && noise->w_skip) ? grads->dw_noise_skip : nullptr;       // conv_1: residual + w_skip * eps
    DSEE_TRY(dsee_wino43_dout_f16x2(g, B.dm2c, N, H, W, C, amax_gc, 1.0f, B.chws, dbc[i] ? B.dbias : nullptr, dnw0 ? B.dbias + C : nullptr,
                                    dnw0 ? noise->seed_middle : 0, dnw0 ? noise->offset_middle : 0, dnw1 ? B.dbias + 2 * C : nullptr,
                                    dnw1 ? noise->seed_skip : 0, dnw1 ? noise->offset_skip : 0, stream));
    if ((dbc[i] && hipMemcpyAsync(dbc[i], B.dbias, C * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) ||
        (dnw0 && hipMemcpyAsync(dnw0, B.dbias + C, C * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) ||
        (dnw1 && hipMemcpyAsync(dnw1, B.dbias + 2 * C, C * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)) {
      dsee_set_error("%s: hipMemcpyAsync of the channel sums failed", who);
      return DSEE_ELAUNCH;
    }
    if (dwc[i])
      DSEE_TRY(dsee_wino43_wgrad(reinterpret_cast<const float*>(S.v2h[i]), reinterpret_cast<const float*>(B.dm2c), B.wsw, B.wsw_bytes,
                                 dwc[i], T, C, C, C, C, 6, amax_h, amax_gc, stream));
    DSEE_TRY(dsee_wino43_weights(convw[i], reinterpret_cast<float*>(B.utc), C, C, 2, 2, amax_w, stream));
    DSEE_TRY(gemm_nt(B.dm2c, B.utc, B.dv, 36 * T, C, C, T, rows_tc, amax_gc, 1.0f, amax_w, stream));
    DSEE_TRY(zero(amax_dh, kAmaxFloats, stream, who));
    DSEE_TRY(dsee_wino43_input_adjoint_amax(B.dv, B.dh, N, H, W, C, amax_dh, stream));

    // ---- SeanNormTable.backward: BN + modulate + LeakyReLU backward (reduce pass writes dM = A (g xhat | g) A^T pre-split; the
    //      apply pass adds the shortcut's gradient for norm_0), table / gamma-beta weight gradient, embedding gradient (adjoint),
    //      mlp_shared gradients
    const float* xin = i == 0 ? x : S.dx0;
    float* dxo = i == 0 ? dx : B.dmid;
    float* amax_dxo = i == 0 ? amax_dx : amax_dmid;
    DSEE_TRY(zero(amax_g, kAmaxFloats, stream, who));
    DSEE_TRY(dsee_amax_product(amax_dh, amax_xhat, 1.0f, amax_g, stream));
    DSEE_TRY(dsee_modulate_bwd_reduce_wino_f16x2(B.dh, nullptr, xin, ns.scale, ns.mean, ns.invstd, B.dm2n, rows, B.sums, N, H, W, C,
                                                 slope, B.wsmod, amax_g, 1.0f, ns.mask, stream));
    DSEE_TRY(zero(amax_dxo, kAmaxFloats, stream, who));
    DSEE_TRY(dsee_modulate_bwd_apply_amax(B.dh, nullptr, xin, ns.scale, ns.mean, ns.invstd, B.sums, i == 0 ? dout : nullptr, dxo, N,
                                          H * W, C, 1.0f / (float)px, slope, amax_dxo, 0, ns.mask, stream));
    DSEE_TRY(dsee_wino43_weights(norms[i]->w2a, reinterpret_cast<float*>(B.ute), rows, kHidden, 2, 2, amax_u, stream));
    if (has_t)
      DSEE_TRY(dsee_wino43_wgrad_table(reinterpret_cast<const float*>(ns.v2cat), reinterpret_cast<const float*>(B.dm2n), B.wst,
                                       B.wst_bytes, ngr[i]->dw2a, ngr[i]->dtable, T, N, kHidden, rows, label_nc, 6, amax_cat, amax_g,
                                       stream));
    else
      DSEE_TRY(dsee_wino43_wgrad(reinterpret_cast<const float*>(ns.v2cat), reinterpret_cast<const float*>(B.dm2n), B.wst, B.wst_bytes,
                                 ngr[i]->dw2a, T, ld, rows, rows, kHidden, 6, amax_cat, amax_g, stream));
    DSEE_TRY(dsee_gemm_f16x2_pre(B.dm2n, B.ute, B.dve, 36 * T, kHidden, rows, T, rows_te, amax_g, 1.0f, amax_u, stream));
    DSEE_TRY(zero(amax_da, kAmaxFloats, stream, who));
    if (has_t) {
      // (the ReLU of the embedding rides in the adjoint transform as a mask on `cat`; mlp_shared -- a convolution over the
      // one-hot label -- takes its weight gradient w.r.t. the one-hot channels already sitting in `cat`)
      DSEE_TRY(dsee_wino43_input_adjoint(B.dve, ns.cat, ld, B.dactv, N, H, W, kHidden, nullptr, amax_da, stream));
      dsee_conv_geom gs = {N, H, W, ld, H, W, kHidden, 3, 3, 1, -1, 1, 0, 0, 1};
      const double flops = 2.0 * (double)px * kHidden * 9.0 * ((label_nc + 31) / 32 * 32);
      if (ngr[i]->dw_shared) {
        if (flops >= 1e9)      // (ops.py::wgrad_raw: split fp16x2 operands above KernelPlan.conv_f16x2_min_flop)
          DSEE_TRY(dsee_conv2d_wgrad_f16x2(&gs, ns.cat, B.dactv, B.wse, B.wse_bytes, ngr[i]->dw_shared, kHidden, kHidden, label_nc,
                                           amax_cat, amax_da, stream));
        else
          DSEE_TRY(dsee_conv2d_wgrad(&gs, ns.cat, B.dactv, B.wse, B.wse_bytes, ngr[i]->dw_shared, kHidden, kHidden, label_nc, stream));
      }
      if (ngr[i]->db_shared) DSEE_TRY(dsee_channel_dot(B.dactv, nullptr, ngr[i]->db_shared, (long)px, kHidden, B.chws, stream));
    } else {
      DSEE_TRY(dsee_wino43_input_adjoint_amax(B.dve, B.dactv, N, H, W, kHidden, amax_da, stream));
      if (ngr[i]->dw_shared && ngr[i]->db_shared) {
        // ReLU backward of the embedding (+ max |g|), the one-hot label channels materialised, then the generic weight gradient
        // (fp16x2 operands above 1 GFLOP like ops.py::wgrad_raw) and the bias gradient as a channel sum
        float *amax_g1 = B.amax + 4 * kAmaxFloats, *amax_oh = B.amax + 5 * kAmaxFloats;
        DSEE_TRY(zero(amax_g1, 2 * kAmaxFloats, stream, who));
        DSEE_TRY(dsee_act_bwd_amax(B.dactv, ns.cat, B.g1, (long)(px * kHidden), DSEE_ACT_RELU, slope, amax_g1, stream));
        DSEE_TRY(dsee_label_onehot(labels, B.oh, N, lab_h, lab_w, shift, 32, 0, stream));
        dsee_conv_geom gs = {N, H, W, 32, H, W, kHidden, 3, 3, 1, -1, 1, 0, 0, 1};
        const double flops = 2.0 * (double)px * kHidden * 9.0 * 32.0;
        if (flops >= 1e9) {
          DSEE_TRY(dsee_absmax(B.oh, (long)(px * 32), amax_oh, stream));
          DSEE_TRY(dsee_conv2d_wgrad_f16x2(&gs, B.oh, B.g1, B.wse, B.wse_bytes, ngr[i]->dw_shared, kHidden, 0, label_nc, amax_oh, amax_g1,
                                           stream));
        } else {
          DSEE_TRY(dsee_conv2d_wgrad(&gs, B.oh, B.g1, B.wse, B.wse_bytes, ngr[i]->dw_shared, kHidden, 0, label_nc, stream));
        }
        DSEE_TRY(dsee_channel_dot(B.g1, nullptr, ngr[i]->db_shared, (long)px, kHidden, B.chws, stream));
      }
    }
    if (ngr[i]->dgamma_beta_sums &&
        hipMemcpyAsync(ngr[i]->dgamma_beta_sums, B.sums + 2 * C, 2 * C * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) {
      dsee_set_error("%s: hipMemcpyAsync of the gamma / beta sums failed", who);
      return DSEE_ELAUNCH;
    }
    g = B.dmid;
    amax_gc = amax_dmid;
  }
  return DSEE_OK;
}

}  // extern "C"
