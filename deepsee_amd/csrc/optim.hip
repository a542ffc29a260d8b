// Parameter-space kernels: spectral normalisation and the fused multi-tensor Adam step.
//
//  * spectral norm = torch.nn.utils.spectral_norm, hook form (call sites architecture.py:40-44,
//    normalization.py:29-30): one power iteration per train-mode forward, in place on u/v (eps 1e-12),
//    sigma = u^T W v, W = W_orig / sigma; backward dW_orig = (dW - <dW, W> u v^T) / sigma.
//  * Adam = torch.optim.Adam(betas=(beta1,beta2), eps=1e-8) over flat fp32 buffers (sr_model.py:469-495),
//    with the reference's "parameter without gradient is skipped" semantics kept per tensor
//    (the unused encoder branch) and optional clip_grad_value_ (trainer_manager.py:39-41).
#include "dsee_common.h"

namespace {

__device__ __forceinline__ float block_sum256(float v) {
  __shared__ float red[4];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// t[j] = sum_i W[i][j] * u[i]      (W [R][K] row-major); block = 32 columns x 8 row-lanes, coalesced over j
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                      float* __restrict__ t, int R, int K) {
  __shared__ float sv[8][32];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cl;
  float acc = 0.f;
  if (j < K)
    for (int i = lane; i < R; i += 8) acc += W[(size_t)i * K + j] * u[i];
  sv[lane][cl] = acc;
  __syncthreads();
  if (lane == 0 && j < K) {
    for (int l = 1; l < 8; ++l) acc += sv[l][cl];
    t[j] = acc;
  }
}

// s[i] = sum_j W[i][j] * v[j]; one block per row
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ W, const float* __restrict__ v,
                                                     float* __restrict__ s, int K) {
  const int i = blockIdx.x;
  float acc = 0.f;
  for (int j = threadIdx.x; j < K; j += 256) acc += W[(size_t)i * K + j] * v[j];
  acc = block_sum256(acc);
  if (threadIdx.x == 0) s[i] = acc;
}

// out = in / max(||in||, eps); single block
__global__ __launch_bounds__(256) void sn_normalize_kernel(const float* __restrict__ in, float* __restrict__ out, int n,
                                                           float eps) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += in[i] * in[i];
  acc = block_sum256(acc);
  const float d = fmaxf(sqrtf(acc), eps);
  for (int i = threadIdx.x; i < n; i += 256) out[i] = in[i] / d;
}

// sigma = dot(u, s)   (s = W v)
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, int n) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += a[i] * b[i];
  acc = block_sum256(acc);
  if (threadIdx.x == 0) *out = acc;
}

__global__ __launch_bounds__(256) void scale_by_inv_kernel(const float* __restrict__ w, const float* __restrict__ sigma,
                                                           float* __restrict__ out, long n) {
  const float s = *sigma;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = w[i] / s;
}

__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          long n, float* __restrict__ part) {
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += a[i] * b[i];
  acc = block_sum256(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// dWo[i][j] = (dW[i][j] - <dW,Wsn> * u[i] * v[j]) / sigma
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ dW, const float* __restrict__ u,
                                                     const float* __restrict__ v, const float* __restrict__ sigma,
                                                     const float* __restrict__ part, int parts,
                                                     float* __restrict__ dWo, int R, int K) {
  __shared__ float dot_s;
  if (threadIdx.x == 0) {
    float d = 0.f;
    for (int p = 0; p < parts; ++p) d += part[p];
    dot_s = d;
  }
  __syncthreads();
  const float d = dot_s, s = *sigma;
  const long n = (long)R * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / K), c = (int)(i % K);
    dWo[i] = (dW[i] - d * u[r] * v[c]) / s;
  }
}

}  // namespace

extern "C" {

/* One spectral-norm forward.  W_orig [R][K] (= weight_orig viewed [Cout, Cin*k*k]); u [R], v [K] are updated IN
 * PLACE when power_iter != 0 (train-mode forward, also under no_grad); writes sigma (device scalar) and
 * w_sn = W_orig / sigma.  scratch: R + K floats. */
int dsee_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* sigma, float* w_sn, int R, int K,
                           int power_iter, float eps, float* scratch, hipStream_t st) {
  DSEE_CHECK_ARG(w_orig && u && v && sigma && w_sn && scratch && R > 0 && K > 0);
  float* tK = scratch;
  float* tR = scratch + K;
  if (power_iter) {
    sn_wt_u_kernel<<<dsee_cdiv(K, 32), 256, 0, st>>>(w_orig, u, tK, R, K);
    sn_normalize_kernel<<<1, 256, 0, st>>>(tK, v, K, eps);
    sn_w_v_kernel<<<R, 256, 0, st>>>(w_orig, v, tR, K);
    sn_normalize_kernel<<<1, 256, 0, st>>>(tR, u, R, eps);
  } else {
    sn_w_v_kernel<<<R, 256, 0, st>>>(w_orig, v, tR, K);
  }
  sn_dot_kernel<<<1, 256, 0, st>>>(u, tR, sigma, R);
  const long n = (long)R * K;
  scale_by_inv_kernel<<<(int)min(2048L, (n + 255) / 256), 256, 0, st>>>(w_orig, sigma, w_sn, n);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* dW_orig = (dW - <dW, W_sn> u v^T) / sigma.  scratch: 256 floats. */
int dsee_spectral_norm_bwd(const float* dw, const float* w_sn, const float* u, const float* v, const float* sigma,
                           float* dw_orig, int R, int K, float* scratch, hipStream_t st) {
  DSEE_CHECK_ARG(dw && w_sn && u && v && sigma && dw_orig && scratch);
  const long n = (long)R * K;
  const int parts = (int)min(256L, (n + 255) / 256);
  dot_partial_kernel<<<parts, 256, 0, st>>>(dw, w_sn, n, scratch);
  sn_bwd_kernel<<<(int)min(2048L, (n + 255) / 256), 256, 0, st>>>(dw, u, v, sigma, scratch, parts, dw_orig, R, K);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- spectral norm of ALL layers of a network in five launches
// The per-layer form above costs 6 launches (+ 2 clones of u, v for the backward pass) per layer and forward: 42 layers x
// 2 generator / encoder / discriminator passes per training iteration = ~650 launches of sub-microsecond work.  Here the
// stages run once per network with a device-resident layer table (like the Adam descriptors): every block looks its
// (layer, piece) up in a small work list.  Same arithmetic and the same summation order per layer as the per-layer
// kernels (bit-identical results).
namespace {

__global__ __launch_bounds__(256) void sng_wt_u_kernel(const dsee_sn_layer* __restrict__ layers, const int2* __restrict__ work,
                                                       float* __restrict__ scratch) {
  __shared__ float sv[8][32];
  const int2 wk = work[blockIdx.x];
  const dsee_sn_layer L = layers[wk.x];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int j = wk.y * 32 + cl;
  float acc = 0.f;
  if (j < L.K)
    for (int i = lane; i < L.R; i += 8) acc += L.w_orig[(size_t)i * L.K + j] * L.u[i];
  sv[lane][cl] = acc;
  __syncthreads();
  if (lane == 0 && j < L.K) {
    for (int l = 1; l < 8; ++l) acc += sv[l][cl];
    scratch[L.scratch_off + j] = acc;
  }
}

// one block per layer: v = tK / max(||tK||, eps)
__global__ __launch_bounds__(256) void sng_norm_v_kernel(const dsee_sn_layer* __restrict__ layers,
                                                         const float* __restrict__ scratch, float eps) {
  const dsee_sn_layer L = layers[blockIdx.x];
  const float* in = scratch + L.scratch_off;
  float acc = 0.f;
  for (int i = threadIdx.x; i < L.K; i += 256) acc += in[i] * in[i];
  acc = block_sum256(acc);
  const float d = fmaxf(sqrtf(acc), eps);
  for (int i = threadIdx.x; i < L.K; i += 256) L.v[i] = in[i] / d;
}

__global__ __launch_bounds__(256) void sng_w_v_kernel(const dsee_sn_layer* __restrict__ layers, const int2* __restrict__ work,
                                                      float* __restrict__ scratch) {
  const int2 wk = work[blockIdx.x];
  const dsee_sn_layer L = layers[wk.x];
  const int i = wk.y;
  float acc = 0.f;
  for (int j = threadIdx.x; j < L.K; j += 256) acc += L.w_orig[(size_t)i * L.K + j] * L.v[j];
  acc = block_sum256(acc);
  if (threadIdx.x == 0) scratch[L.scratch_off + L.K + i] = acc;
}

// one block per layer: (power_iter) u = tR / max(||tR||, eps); sigma = <u, tR>; copies of u, v for the backward pass
__global__ __launch_bounds__(256) void sng_norm_u_sigma_kernel(const dsee_sn_layer* __restrict__ layers,
                                                               const float* __restrict__ scratch, float eps,
                                                               int power_iter, float* __restrict__ sigma,
                                                               float* __restrict__ saved) {
  const dsee_sn_layer L = layers[blockIdx.x];
  const float* tR = scratch + L.scratch_off + L.K;
  if (power_iter) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < L.R; i += 256) acc += tR[i] * tR[i];
    acc = block_sum256(acc);
    const float d = fmaxf(sqrtf(acc), eps);
    for (int i = threadIdx.x; i < L.R; i += 256) L.u[i] = tR[i] / d;
    __syncthreads();
  }
  float dot = 0.f;
  for (int i = threadIdx.x; i < L.R; i += 256) dot += L.u[i] * tR[i];
  dot = block_sum256(dot);
  if (threadIdx.x == 0) sigma[blockIdx.x] = dot;
  for (int i = threadIdx.x; i < L.R; i += 256) saved[L.saved_off + i] = L.u[i];
  for (int i = threadIdx.x; i < L.K; i += 256) saved[L.saved_off + L.R + i] = L.v[i];
}

// w_sn = w_orig / sigma for a 4096-element piece of a layer; max |w_sn| of the layer for the fp16 operand scales
__global__ __launch_bounds__(256) void sng_scale_kernel(const dsee_sn_layer* __restrict__ layers, const int2* __restrict__ work,
                                                        const float* __restrict__ sigma, float* __restrict__ out,
                                                        float* __restrict__ amax) {
  const int2 wk = work[blockIdx.x];
  const dsee_sn_layer L = layers[wk.x];
  const float s = sigma[wk.x];
  const long n = (long)L.R * L.K, i0 = (long)wk.y * 4096;
  float vmax = 0.f;
  for (long i = i0 + threadIdx.x; i < n && i < i0 + 4096; i += 256) {
    const float v = L.w_orig[i] / s;
    out[L.out_off + i] = v;
    vmax = fmaxf(vmax, fabsf(v));
  }
  dsee_block_atomic_absmax(amax + (size_t)wk.x * (DSEE_AMAX_LINES * DSEE_AMAX_STRIDE), vmax);
}

}  // namespace

extern "C" {

/* Spectral normalisation of all `nlayers` layers of a network (torch.nn.utils.spectral_norm hook semantics, call sites
 * architecture.py:40-44, normalization.py:29-30; SURVEY B-2) in five launches.  layers: device table; work_k / work_r /
 * work_e: device work lists of (layer, 32-column block) / (layer, row) / (layer, 4096-element piece) with n_k / n_r / n_e
 * entries; scratch: sum(R + K) floats at the layers' scratch_off; sigma [nlayers]; out: flat w_sn at the layers' out_off;
 * saved: copies of the (updated) u, v at saved_off for dsee_spectral_norm_bwd; amax: [nlayers][2048] floats, zeroed by
 * the caller, receives max |w_sn| per layer in the 64-line form of dsee_absmax. */
int dsee_spectral_norm_group_fwd(const dsee_sn_layer* layers, int nlayers, const int* work_k, int n_k, const int* work_r,
                                 int n_r, const int* work_e, int n_e, int power_iter, float eps, float* scratch,
                                 float* sigma, float* out, float* saved, float* amax, hipStream_t st) {
  DSEE_CHECK_ARG(layers && nlayers > 0 && work_k && work_r && work_e && n_k > 0 && n_r > 0 && n_e > 0);
  DSEE_CHECK_ARG(scratch && sigma && out && saved && amax);
  if (power_iter) {
    sng_wt_u_kernel<<<n_k, 256, 0, st>>>(layers, reinterpret_cast<const int2*>(work_k), scratch);
    sng_norm_v_kernel<<<nlayers, 256, 0, st>>>(layers, scratch, eps);
  }
  sng_w_v_kernel<<<n_r, 256, 0, st>>>(layers, reinterpret_cast<const int2*>(work_r), scratch);
  sng_norm_u_sigma_kernel<<<nlayers, 256, 0, st>>>(layers, scratch, eps, power_iter, sigma, saved);
  sng_scale_kernel<<<n_e, 256, 0, st>>>(layers, reinterpret_cast<const int2*>(work_e), sigma, out, amax);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"

namespace {

// One block-range per tensor: blocks [first_block, first_block + nblocks) work on tensor t.
// grad_flat[offset(t) ..] <- the gradient tensor of parameter t (zeros where autograd delivered none), active[t] <- 0 / 1:
// ONE launch instead of one AccumulateGrad add per parameter
__global__ __launch_bounds__(256) void grad_gather_kernel(const int64_t* __restrict__ ptrs,
                                                          const dsee_adam_tensor* __restrict__ tensors,
                                                          const int* __restrict__ block_tensor,
                                                          float* __restrict__ grad_flat, int* __restrict__ active) {
  const int blk = (int)blockIdx.x;
  const int t = block_tensor[blk];
  const dsee_adam_tensor d = tensors[t];
  const float* src = reinterpret_cast<const float*>(ptrs[t]);
  const long b0 = (long)(blk - d.first_block) * 1024;
  if (blk == d.first_block && threadIdx.x == 0) {
    active[t] = src != nullptr;
    grad_flat[t] = src ? 1.f : 0.f;   // header of the flat buffer: the flag travels with the first all-reduced chunk
  }
  for (long i = b0 + threadIdx.x; i < d.numel && i < b0 + 1024; i += 256) grad_flat[d.offset + i] = src ? src[i] : 0.f;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                   const dsee_adam_tensor* __restrict__ tensors,
                                                   const int* __restrict__ block_tensor, int first_block,
                                                   float beta1, float beta2, float eps, float grad_scale, float clip) {
  const int blk = first_block + (int)blockIdx.x;
  const int t = block_tensor[blk];
  const dsee_adam_tensor d = tensors[t];
  if (!d.active) return;
  const int step = d.step + 1;  // d.step is bumped by the caller after the last launch of a step (same value on every block)
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  const float step_size = d.lr / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  const long b0 = (long)(blk - d.first_block) * 1024;
  for (long i = b0 + threadIdx.x; i < d.numel && i < b0 + 1024; i += 256) {
    const long o = d.offset + i;
    float g = grad[o] * grad_scale;
    if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
    const float m = beta1 * exp_avg[o] + (1.f - beta1) * g;
    const float v = beta2 * exp_avg_sq[o] + (1.f - beta2) * g * g;
    exp_avg[o] = m;
    exp_avg_sq[o] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    param[o] -= step_size * (m / denom);
  }
}

}  // namespace

extern "C" {

/* Fused Adam over a flat parameter buffer.  `tensors` / `block_tensor` are DEVICE arrays: descriptor t covers
 * elements [offset, offset+numel) and owns blocks [first_block, first_block + ceil(numel/1024)); block_tensor maps
 * each block to its descriptor.  `step` in the descriptor is the number of updates already applied to
 * that tensor (torch's state['step']); inactive tensors (grad is None in the reference) are skipped entirely.
 * grad_scale = 1/world_size after the RCCL sum all-reduce.
 * dsee_adam_step_range updates blocks [first_block, first_block + nblocks) only: the data-parallel step launches one
 * range per all-reduced gradient chunk so the update of chunk k overlaps the all-reduce of chunk k+1. */
int dsee_adam_step_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                         const dsee_adam_tensor* tensors, const int* block_tensor, int first_block, int nblocks,
                         float beta1, float beta2, float eps, float grad_scale, float clip, hipStream_t st) {
  DSEE_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && tensors && block_tensor && first_block >= 0 && nblocks > 0);
  adam_kernel<<<nblocks, 256, 0, st>>>(param, grad, exp_avg, exp_avg_sq, tensors, block_tensor, first_block, beta1,
                                       beta2, eps, grad_scale, clip);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* Collects the per-parameter gradient tensors autograd produced (grad_ptrs[t]: device address of a contiguous fp32
 * tensor of tensors[t].numel elements, or 0 = "p.grad is None") into the flat gradient buffer the all-reduce and the
 * Adam kernel work on, and writes the per-tensor active flags -- into active[] and, as 0 / 1 floats, into grad_flat[t]
 * (the flat buffers start with a header of >= ntensors floats; tensors[t].offset lies behind it).  Replaces the ~290 AccumulateGrad `grad += new` kernels of
 * a backward pass into persistent .grad views (and the zero fill of the flat buffer) by one launch. */
int dsee_grad_gather(const int64_t* grad_ptrs, const dsee_adam_tensor* tensors, const int* block_tensor, int nblocks,
                     float* grad_flat, int* active, hipStream_t st) {
  DSEE_CHECK_ARG(grad_ptrs && tensors && block_tensor && grad_flat && active && nblocks > 0);
  grad_gather_kernel<<<nblocks, 256, 0, st>>>(grad_ptrs, tensors, block_tensor, grad_flat, active);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   const dsee_adam_tensor* tensors, const int* block_tensor, int nblocks, float beta1, float beta2,
                   float eps, float grad_scale, float clip, hipStream_t st) {
  return dsee_adam_step_range(param, grad, exp_avg, exp_avg_sq, tensors, block_tensor, 0, nblocks, beta1, beta2, eps,
                              grad_scale, clip, st);
}

}  // extern "C"
