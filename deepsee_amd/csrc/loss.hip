// Scalar losses with their gradients produced in the same pass (the loss graph is
// g_loss = sum(losses).mean(), so every d(loss_k)/d(input) is known at forward time):
//   hinge-GAN  loss.py:68-79 (for G: -mean(x); for D: -mean(min(+-x - 1, 0)))
//   L1 feature matching / VGG perceptual terms  sr_model.py:529-539, loss.py:114-119
// Each call ACCUMULATES  weight * mean(...)  into *loss_out and writes  weight * d(mean)/dx  to grad.
#include "dsee_common.h"

namespace {

__device__ __forceinline__ void block_sum_to(float v, float* part) {
  __shared__ float red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// mode 0: l = |a-b|, d = sign(a-b);  mode 1: l = -x;  mode 2: l = -min(x-1,0);  mode 3: l = -min(-x-1,0)
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ grad, long n, int ld, int valid_c,
                                                           int mode, float gscale, float* __restrict__ part) {
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const bool ok = (int)(i % ld) < valid_c;  // padded channels carry no loss
    float l = 0.f, d = 0.f;
    if (ok) {
      const float x = a[i];
      if (mode == 0) {
        const float df = x - b[i];
        l = fabsf(df);
        d = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
      } else if (mode == 1) {
        l = -x;
        d = -1.f;
      } else if (mode == 2) {
        const float m = x - 1.f;
        l = m < 0.f ? -m : 0.f;
        d = m < 0.f ? -1.f : 0.f;
      } else {
        const float m = -x - 1.f;
        l = m < 0.f ? -m : 0.f;
        d = m < 0.f ? 1.f : 0.f;
      }
    }
    acc += l;
    if (grad) grad[i] = d * gscale;
  }
  block_sum_to(acc, part);
}

// grad[i] = upstream * gscale * dl/da  (the backward half on its own: the loss graph may scale a term, e.g. loss
// scaling or gradient accumulation with 1/k, so the upstream gradient is a device scalar read here)
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ grad, long n, int ld, int valid_c, int mode,
                                                        float gscale, const float* __restrict__ upstream) {
  const float gs = gscale * (upstream ? *upstream : 1.f);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float d = 0.f;
    if ((int)(i % ld) < valid_c) {
      const float x = a[i];
      if (mode == 0) {
        const float df = x - b[i];
        d = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
      } else if (mode == 1) {
        d = -1.f;
      } else if (mode == 2) {
        d = x - 1.f < 0.f ? -1.f : 0.f;
      } else {
        d = -x - 1.f < 0.f ? 1.f : 0.f;
      }
    }
    grad[i] = d * gs;
  }
}

__global__ void loss_finalize_kernel(const float* __restrict__ part, int parts, float scale, float* __restrict__ out) {
  // one wave, fixed order (lane l folds parts l, l+64, ...; then the butterfly): deterministic
  float v = 0.f;
  for (int i = threadIdx.x; i < parts; i += 64) v += part[i];
  v = wave_sum(v);
  if (threadIdx.x == 0) *out += v * scale;
}

}  // namespace

extern "C" {

size_t dsee_loss_workspace(void) { return 1024 * sizeof(float); }

/* *loss_out += weight * mean_valid(l(a[,b]));  grad = weight * dl/da / count.
 * a/b/grad are [rows][ld] with the first valid_c columns real (NHWC channel padding). */
int dsee_loss_fwd_bwd(int mode, const float* a, const float* b, float* grad, long rows, int ld, int valid_c,
                      float weight, float* loss_out, float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(a && loss_out && workspace && mode >= 0 && mode <= 3 && (mode != 0 || b) && valid_c <= ld);
  const long n = rows * ld;
  const float cnt = (float)rows * (float)valid_c;
  int parts = (int)min(1024L, (n + 255) / 256);
  loss_partial_kernel<<<parts, 256, 0, st>>>(a, b, grad, n, ld, valid_c, mode, weight / cnt, workspace);
  DSEE_LAUNCH_CHECK();
  loss_finalize_kernel<<<1, 64, 0, st>>>(workspace, parts, weight / cnt, loss_out);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* grad = (*upstream) * weight * d mean_valid(l(a[,b])) / da  -- the backward of dsee_loss_fwd_bwd(grad = NULL) for an
 * arbitrary upstream gradient (device scalar; NULL = 1). */
int dsee_loss_bwd(int mode, const float* a, const float* b, float* grad, long rows, int ld, int valid_c, float weight,
                  const float* upstream, hipStream_t st) {
  DSEE_CHECK_ARG(a && grad && mode >= 0 && mode <= 3 && (mode != 0 || b) && valid_c <= ld);
  const long n = rows * ld;
  const float cnt = (float)rows * (float)valid_c;
  loss_grad_kernel<<<(int)min(4096L, (n + 255) / 256), 256, 0, st>>>(a, b, grad, n, ld, valid_c, mode, weight / cnt,
                                                                    upstream);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
