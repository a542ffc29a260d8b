// PSNR / SSIM / RMSE of generated against ground-truth images on the device (reference:
// evaluator/evaluation.py:88-137 -> evaluator/calculate_PSNR_SSIM.py:71-122 on util/util.py:72-103 `tensor2im` images).
//
// Reference semantics kept to the letter:
//   * both images are first quantised like tensor2im: u = uint8(clip((x + 1) / 2 * 255, 0, 255)) -- fp32 arithmetic in that
//     order, truncation;
//   * PSNR = 20 log10(255 / sqrt(mean((u_f - u_r)^2))) over all H*W*3 values (inf for identical images); the squared
//     differences are summed as integers, i.e. exactly;
//   * SSIM: 11x11 Gaussian window (sigma 1.5, cv2.getGaussianKernel: exp(-(i-5)^2 / (2 sigma^2)) normalised in double),
//     "valid" region only ([5:-5, 5:-5]), C1 = (0.01*255)^2, C2 = (0.03*255)^2, float64, mean of the SSIM map over the valid
//     pixels and the three channels (calculate_ssim's channel loop passes the whole image three times: the mean over
//     positions and channels IS its result);
//   * RMSE = sqrt(mean((x_f - x_r)^2)) on the [-1, 1] tensors (evaluation.py:107-110; fp32 there, float64 sums here).
// Two launches: per-block partial sums into the caller's workspace (blocks = N x 3 channels x 16x16-pixel tiles, no
// atomics), then one block per image adds them in a fixed order: the result does not depend on scheduling.
#include "dsee_common.h"

namespace {

constexpr int TS = 16;            // output tile edge
constexpr int HALO = 5;
constexpr int IN = TS + 2 * HALO; // 26

struct GaussWindow { double w[11]; };

__device__ __forceinline__ float to_u8(float x) {
  // (x + 1) / 2 * 255 with fp32 rounding after every operation (no contraction into an fma), clip, truncate
  float v = __fmul_rn(__fadd_rn(x, 1.0f) * 0.5f, 255.0f);
  v = fminf(fmaxf(v, 0.0f), 255.0f);
  return floorf(v);
}

// partial[block][4] = {sum of SSIM map values, integer sum of squared u8 differences, sum of squared [-1,1] differences,
// unused}; blocks of image n: n * per_image .. + per_image - 1
__global__ __launch_bounds__(256) void psnr_ssim_partial_kernel(const float* __restrict__ fake, const float* __restrict__ real,
                                                               double* __restrict__ partial, int H, int W, int Cs,
                                                               int tiles_x, int tiles_y, GaussWindow g) {
  __shared__ float sf[IN][IN + 1], sr[IN][IN + 1];
  __shared__ double hz[5][IN][TS];   // horizontally filtered x, y, xx, yy, xy
  __shared__ double red[3][4];
  const int tile = blockIdx.x % (tiles_x * tiles_y), c = (blockIdx.x / (tiles_x * tiles_y)) % 3;
  const int n = blockIdx.x / (tiles_x * tiles_y * 3);
  const int ty0 = (tile / tiles_x) * TS, tx0 = (tile % tiles_x) * TS;   // tile origin in VALID coordinates
  const int tid = threadIdx.x;
  const float* pf = fake + (long)n * H * W * Cs + c;
  const float* pr = real + (long)n * H * W * Cs + c;
  // the block's own 16x16 pixels of the full image (for PSNR / RMSE every pixel must be counted exactly once: tile
  // (ty, tx) of the full-image tiling owns pixels [16 ty, 16 ty + 16) x [16 tx, 16 tx + 16); the valid-region tiling has
  // fewer tiles, the launch covers ceil(H/16) x ceil(W/16) tiles and SSIM positions outside the valid region are skipped)
  double se_u8 = 0.0, se_f = 0.0, ss = 0.0;
  {
    const int y = ty0 + tid / TS, x = tx0 + tid % TS;
    if (y < H && x < W) {
      const float a = pf[((long)y * W + x) * Cs], b = pr[((long)y * W + x) * Cs];
      const float d8 = to_u8(a) - to_u8(b);
      se_u8 = (double)(d8 * d8);
      const double df = (double)a - (double)b;
      se_f = df * df;
    }
  }
  // SSIM: valid position (vy, vx) in [0, H-10) x [0, W-10) reads image rows vy .. vy+10
  const int VH = H - 2 * HALO, VW = W - 2 * HALO;
  if (ty0 < VH && tx0 < VW) {
    for (int i = tid; i < IN * IN; i += 256) {
      const int yy = i / IN, xx = i % IN, y = ty0 + yy, x = tx0 + xx;
      float a = 0.f, b = 0.f;
      if (y < H && x < W) {
        a = to_u8(pf[((long)y * W + x) * Cs]);
        b = to_u8(pr[((long)y * W + x) * Cs]);
      }
      sf[yy][xx] = a;
      sr[yy][xx] = b;
    }
    __syncthreads();
    for (int i = tid; i < IN * TS; i += 256) {
      const int yy = i / TS, xo = i % TS;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const double a = sf[yy][xo + k], b = sr[yy][xo + k], w = g.w[k];
        s0 += w * a; s1 += w * b; s2 += w * a * a; s3 += w * b * b; s4 += w * a * b;
      }
      hz[0][yy][xo] = s0; hz[1][yy][xo] = s1; hz[2][yy][xo] = s2; hz[3][yy][xo] = s3; hz[4][yy][xo] = s4;
    }
    __syncthreads();
    const int yo = tid / TS, xo = tid % TS;
    if (ty0 + yo < VH && tx0 + xo < VW) {
      double m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const double w = g.w[k];
        m1 += w * hz[0][yo + k][xo]; m2 += w * hz[1][yo + k][xo];
        e11 += w * hz[2][yo + k][xo]; e22 += w * hz[3][yo + k][xo]; e12 += w * hz[4][yo + k][xo];
      }
      const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
      const double m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
      const double s11 = e11 - m11, s22 = e22 - m22, s12 = e12 - m12;
      ss = ((2 * m12 + C1) * (2 * s12 + C2)) / ((m11 + m22 + C1) * (s11 + s22 + C2));
    }
  }
  // block reduction in a fixed order: wave shuffles, then the 4 wave sums
  double v[3] = {ss, se_u8, se_f};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double x = v[q];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    if ((tid & 63) == 0) red[q][tid >> 6] = x;
  }
  __syncthreads();
  if (tid < 3) partial[(long)blockIdx.x * 4 + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

// out[n] = {psnr, ssim, rmse}
__global__ __launch_bounds__(256) void psnr_ssim_finalize_kernel(const double* __restrict__ partial, double* __restrict__ out,
                                                                int per_image, int H, int W) {
  __shared__ double red[3][256];
  const int n = blockIdx.x, tid = threadIdx.x;
  double s[3] = {0, 0, 0};
  for (int i = tid; i < per_image; i += 256)
#pragma unroll
    for (int q = 0; q < 3; ++q) s[q] += partial[((long)n * per_image + i) * 4 + q];
#pragma unroll
  for (int q = 0; q < 3; ++q) red[q][tid] = s[q];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o)
#pragma unroll
      for (int q = 0; q < 3; ++q) red[q][tid] += red[q][tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const double npix = 3.0 * H * W, nvalid = 3.0 * (H - 2 * HALO) * (W - 2 * HALO);
    const double mse = red[1][0] / npix;
    out[n * 3 + 0] = mse == 0.0 ? (double)INFINITY : 20.0 * log10(255.0 / sqrt(mse));
    out[n * 3 + 1] = red[0][0] / nvalid;
    out[n * 3 + 2] = sqrt(red[2][0] / npix);
  }
}

}  // namespace

extern "C" {

size_t dsee_psnr_ssim_workspace(int N, int H, int W) {
  return (size_t)N * 3 * dsee_cdiv(H, TS) * dsee_cdiv(W, TS) * 4 * sizeof(double);
}

int dsee_psnr_ssim(const float* fake, const float* real, int N, int H, int W, int Cs, double* workspace,
                   size_t workspace_bytes, double* out, hipStream_t st) {
  DSEE_CHECK_ARG(fake && real && workspace && out && N > 0 && Cs >= 3 && H > 2 * HALO && W > 2 * HALO);
  DSEE_CHECK_ARG(workspace_bytes >= dsee_psnr_ssim_workspace(N, H, W));
  GaussWindow g;
  double sum = 0.0;
  for (int i = 0; i < 11; ++i) {
    g.w[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
    sum += g.w[i];
  }
  for (int i = 0; i < 11; ++i) g.w[i] /= sum;
  const int tx = dsee_cdiv(W, TS), ty = dsee_cdiv(H, TS), per_image = 3 * tx * ty;
  psnr_ssim_partial_kernel<<<N * per_image, 256, 0, st>>>(fake, real, workspace, H, W, Cs, tx, ty, g);
  DSEE_LAUNCH_CHECK();
  psnr_ssim_finalize_kernel<<<N, 256, 0, st>>>(workspace, out, per_image, H, W);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
