// Streaming (HBM-bound) element-wise kernels of the hot path, NHWC fp32, 16 B per lane.
//
//  * noise injection + nearest x2 upsample  (normalization.py:289-304, sr.py:57,69,87, architecture.py:76-77,111-112,133-134)
//  * activation backward, residual helpers
//  * avg_pool2d(3,2,1,count_include_pad=False) (discriminator.py:46-49), max_pool2d(2,2) (VGG19)
//  * input preparation: label float->uint8, NCHW<->NHWC(pad 4), bicubic downsample + clamp
//    (data/preprocessor.py:17-41, base_manager.py:28-66), discriminator input cat([seg, image]) (sr_model.py:655-668)
//  * counter-based N(0,1)/U(0,1) generators (replaces tensor.normal_() / torch.rand_like on the device)
#include "dsee_common.h"
#include "dsee_rng.h"

namespace {

inline int egrid(long n) { return (int)min(8192L, (n + 255) / 256); }

// y[n,h,w,c] = x[n,h>>ups,w>>ups,c] + nw[c]*eps[n,h,w,c]
__global__ __launch_bounds__(256) void up_noise_fwd_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                           const float* __restrict__ nw, float* __restrict__ y, int N,
                                                           int H, int W, int C, int ups) {
  const long total4 = (long)N * H * W * C / 4;
  const int C4 = C / 4, h0 = H >> ups, w0 = W >> ups;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const dsee_nhwq p = dsee_split_nhwq((unsigned)i, C4, W, H);
    const int q = p.q;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)p.n * h0 + (p.h >> ups)) * w0 + (p.w >> ups)) * C + q * 4);
    if (eps) v += *reinterpret_cast<const f32x4*>(nw + q * 4) * *reinterpret_cast<const f32x4*>(eps + i * 4);
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

// dx[n,h0,w0,c] = sum over the 2^ups x 2^ups block of dy  (+ add)
__global__ __launch_bounds__(256) void sumpool_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H,
                                                      int W, int C, int ups, float* __restrict__ amax = nullptr) {
  float vmax = 0.f;
  const int h0 = H >> ups, w0 = W >> ups, C4 = C / 4, f = 1 << ups;
  const long total4 = (long)N * h0 * w0 * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const dsee_nhwq p = dsee_split_nhwq((unsigned)i, C4, w0, h0);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < f; ++a)
      for (int b = 0; b < f; ++b)
        v += *reinterpret_cast<const f32x4*>(dy + (((size_t)p.n * H + p.h * f + a) * W + p.w * f + b) * C + p.q * 4);
    *reinterpret_cast<f32x4*>(dx + i * 4) = v;
    vmax = fmaxf(vmax, dsee_absmax4(v));
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);
}

// part[blk][C] = sum_pixels a*b  (b optional -> column sums), fixed order
__global__ __launch_bounds__(256) void chdot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ part, long M, int C, int chunk_px) {
  __shared__ f32x4 red[256];
  const int tpp = C / 4, ppb = 256 / tpp > 0 ? 256 / tpp : 1;
  const int q = threadIdx.x % tpp, s = threadIdx.x / tpp;
  const bool active = threadIdx.x < ppb * tpp;
  const long p0 = (long)blockIdx.x * chunk_px, p1 = min(M, p0 + chunk_px);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (active)
    for (long p = p0 + s; p < p1; p += ppb) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a + p * C + q * 4);
      if (b) v *= *reinterpret_cast<const f32x4*>(b + p * C + q * 4);
      acc += v;
    }
  if (active) red[s * tpp + q] = acc;
  __syncthreads();
  if (active && s == 0) {
    f32x4 v = red[q];
    for (int j = 1; j < ppb; ++j) v += red[j * tpp + q];
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + q * 4) = v;
  }
}

// part[parts][C] -> out[C]: block = 4 adjacent channels (one 16-byte load per row) x 256 part-lanes; lane l adds rows l, l + 256,
// ... in order, the lanes then add pairwise in a fixed binary tree through LDS (bit-reproducible).  C % 4 == 0 (entry points).
__global__ __launch_bounds__(256) void chdot_finalize_kernel(const float* __restrict__ part, int parts, int C,
                                                             float* __restrict__ out) {
  __shared__ f32x4 sv[256];
  const int lane = threadIdx.x, c0 = blockIdx.x * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int p = lane; p < parts; p += 256) v += *reinterpret_cast<const f32x4*>(part + (size_t)p * C + c0);
  for (int s = 128; s >= 1; s >>= 1) {
    if (lane >= s && lane < 2 * s) sv[lane] = v;
    __syncthreads();
    if (lane < s) v += sv[lane + s];
  }
  if (lane == 0) {      // (scalar stores: `out` may be a slice of a flat gradient buffer at any 4-byte offset)
    out[c0] = v[0];
    out[c0 + 1] = v[1];
    out[c0 + 2] = v[2];
    out[c0 + 3] = v[3];
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, long total4, int act, float slope,
                                                      float* __restrict__ amax = nullptr) {
  float vmax = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + i * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = d[k] * dsee_act_grad_from_out(v[k], act, slope);
    *reinterpret_cast<f32x4*>(dx + i * 4) = r;
    vmax = fmaxf(vmax, dsee_absmax4(r));
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long total4,
                                                      int act, float slope) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = dsee_act(v[k], act, slope);
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

// y = a*alpha + b*beta  (b optional)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b,
                                                    float beta, float* __restrict__ y, long total4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(a + i * 4) * alpha;
    if (b) v += *reinterpret_cast<const f32x4*>(b + i * 4) * beta;
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

// avg_pool2d(k=3, s=2, p=1, count_include_pad=False)
__global__ __launch_bounds__(256) void avgpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                             int H, int W, int Ho, int Wo, int C) {
  const int C4 = C / 4;
  const long total4 = (long)N * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho), n = (int)(t / Ho);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    for (int a = -1; a <= 1; ++a)
      for (int b = -1; b <= 1; ++b) {
        const int h = oh * 2 + a, w = ow * 2 + b;
        if (h >= 0 && h < H && w >= 0 && w < W) {
          v += *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + h) * W + w) * C + q * 4);
          ++cnt;
        }
      }
    *reinterpret_cast<f32x4*>(y + i * 4) = v / (float)cnt;
  }
}

__global__ __launch_bounds__(256) void avgpool3s2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                             int H, int W, int Ho, int Wo, int C) {
  const int C4 = C / 4;
  const long total4 = (long)N * H * W * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % C4);
    long t = i / C4;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H), n = (int)(t / H);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    // outputs (oh,ow) whose window [2oh-1, 2oh+1] contains h
    for (int oh = (h) / 2; oh <= (h + 1) / 2; ++oh)
      for (int ow = (w) / 2; ow <= (w + 1) / 2; ++ow) {
        if (oh < 0 || oh >= Ho || ow < 0 || ow >= Wo) continue;
        if (abs(oh * 2 - h) > 1 || abs(ow * 2 - w) > 1) continue;
        const int ch = min(H - 1, oh * 2 + 1) - max(0, oh * 2 - 1) + 1;
        const int cw = min(W - 1, ow * 2 + 1) - max(0, ow * 2 - 1) + 1;
        v += *reinterpret_cast<const f32x4*>(dy + (((size_t)n * Ho + oh) * Wo + ow) * C + q * 4) / (float)(ch * cw);
      }
    *reinterpret_cast<f32x4*>(dx + i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                           int H, int W, int C) {
  const int C4 = C / 4, Ho = H / 2, Wo = W / 2;
  const long total4 = (long)N * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho), n = (int)(t / Ho);
    const float* b = x + (((size_t)n * H + oh * 2) * W + ow * 2) * C + q * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(b);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(b + C);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(b + (size_t)W * C);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(b + (size_t)W * C + C);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaxf(v[k], v1[k]), fmaxf(v2[k], v3[k]));
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

// gradient goes to the FIRST maximum in scan order (torch CPU max_pool2d semantics)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dx, int N, int H, int W, int C) {
  const int C4 = C / 4, Ho = H / 2, Wo = W / 2;
  const long total4 = (long)N * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho), n = (int)(t / Ho);
    const size_t o0 = (((size_t)n * H + oh * 2) * W + ow * 2) * C + q * 4;
    const size_t offs[4] = {o0, o0 + C, o0 + (size_t)W * C, o0 + (size_t)W * C + C};
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(x + offs[j]);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    f32x4 r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int best = 0;
      float bv = v[0][k];
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (v[j][k] > bv) {
          bv = v[j][k];
          best = j;
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j][k] = j == best ? g[k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dx + offs[j]) = r[j];
  }
}

// ---- layout / input preparation
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W,
                                    int Cs) {
  const long total = (long)N * H * W * Cs;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cs);
    long t = i / Cs;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H), n = (int)(t / H);
    y[i] = c < C ? x[(((size_t)n * C + c) * H + h) * W + w] : 0.f;
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W,
                                    int Cs) {
  const long total = (long)N * C * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    long t = i / W;
    const int h = (int)(t % H);
    t /= H;
    const int c = (int)(t % C), n = (int)(t / C);
    y[i] = x[(((size_t)n * H + h) * W + w) * Cs + c];
  }
}

__global__ void label_to_u8_kernel(const float* __restrict__ lab, uint8_t* __restrict__ out, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    out[i] = (uint8_t)(long)lab[i];  // .long() truncation (base_manager.py:35-39)
}

// cat([one-hot(label), image]) in NHWC: channels [0,L) one-hot, [L,L+3) image, rest 0 (sr_model.py:655-668)
__global__ void build_d_input_kernel(const uint8_t* __restrict__ lab, const float* __restrict__ img,
                                     float* __restrict__ out, long pixels, int L, int Cs, int img_cs,
                                     float* __restrict__ amax) {
  const long total = pixels * Cs;
  float vmax = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cs);
    const long p = i / Cs;
    float v = 0.f;
    if (c < L) v = lab[p] == c ? 1.f : 0.f;
    else if (c < L + 3) v = img[p * img_cs + (c - L)];
    out[i] = v;
    vmax = fmaxf(vmax, fabsf(v));
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);   // (block-uniform) max |out|: operand bound of the discriminator's first layer
}

// d(image) from d(D input): dimg[p][k] = din[p][L+k]
__global__ void extract_image_grad_kernel(const float* __restrict__ din, float* __restrict__ dimg, long pixels, int L,
                                          int Cs, int img_cs) {
  const long total = pixels * img_cs;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % img_cs);
    const long p = i / img_cs;
    dimg[i] = c < 3 ? din[p * Cs + L + c] : 0.f;
  }
}

// Device input pipeline (data/base_dataset.py:87-116,171-201 without the host-side float tensors): the loader ships
// uint8 images [N][H][W][3] and uint8 label maps [N][H][W]; ToTensor (v/255), Normalize((.5,.5,.5),(.5,.5,.5)), the
// per-sample horizontal flip and the 255 -> label_nc 'unknown' remap happen here, straight into the NHWC RGB0 fp32 /
// uint8 label layout the kernels consume.
__global__ void image_u8_to_nhwc_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ flip,
                                        float* __restrict__ out, int N, int H, int W, int cs) {
  const long total = (long)N * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long r = i / W;
    const int n = (int)(r / H);
    const int ws = (flip && flip[n]) ? W - 1 - w : w;
    const uint8_t* p = img + (r * W + ws) * 3;
    float* o = out + i * cs;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = (float)p[c] / 255.f;   // transforms.ToTensor
      o[c] = (x - 0.5f) / 0.5f;              // transforms.Normalize
    }
    for (int c = 3; c < cs; ++c) o[c] = 0.f;
  }
}

__global__ void label_u8_prepare_kernel(const uint8_t* __restrict__ lab, const uint8_t* __restrict__ flip,
                                        uint8_t* __restrict__ out, int N, int H, int W, int unknown_to) {
  const long total = (long)N * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long r = i / W;
    const int n = (int)(r / H);
    const int ws = (flip && flip[n]) ? W - 1 - w : w;
    const uint8_t v = lab[r * W + ws];
    out[i] = v == 255 ? (uint8_t)unknown_to : v;   // base_dataset.py:95: 'unknown' is opt.label_nc
  }
}

__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;  // PyTorch bicubic
  float x = t + 1.f;
  w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// F.interpolate(mode='bicubic', align_corners=False) + clamp(-1,1); NHWC in (Cs_in) -> NHWC out (Cs_out), 3 channels
__global__ void bicubic_down_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int S,
                                    int cs_in, int cs_out) {
  const long total = (long)N * S * S * cs_out;
  const float sh = (float)H / S, sw = (float)W / S;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cs_out);
    long t = i / cs_out;
    const int ow = (int)(t % S);
    t /= S;
    const int oh = (int)(t % S), n = (int)(t / S);
    if (c >= 3) {
      y[i] = 0.f;
      continue;
    }
    const float fy = sh * (oh + 0.5f) - 0.5f, fx = sw * (ow + 0.5f) - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_coeffs(fy - iy, wy);
    cubic_coeffs(fx - ix, wx);
    float acc = 0.f;
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), H - 1);
      float row = 0.f;
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(ix - 1 + b, 0), W - 1);
        row += wx[b] * x[(((size_t)n * H + yy) * W + xx) * cs_in + c];
      }
      acc += wy[a] * row;
    }
    y[i] = fminf(fmaxf(acc, -1.f), 1.f);
  }
}

// ---- Philox4x32-10 counter RNG: dsee_rng.h
__global__ __launch_bounds__(256) void rng_fill_kernel(float* __restrict__ out, long total4, uint64_t seed,
                                                       uint64_t offset, int normal, const uint64_t* __restrict__ epoch) {
  if (epoch) offset += *epoch;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4(seed, offset + (uint64_t)i, r);
    f32x4 v;
    if (normal) {
      v = box_muller4(r);
    } else {
      v = (f32x4){(float)(r[0] >> 8), (float)(r[1] >> 8), (float)(r[2] >> 8), (float)(r[3] >> 8)} * 5.9604645e-8f;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

// same as up_noise_fwd_kernel with eps = the Philox N(0,1) stream (seed, offset + element/4) generated in registers:
// identical values to dsee_rng_fill(..., seed, offset, normal = 1) followed by the tensor form, without the tensor
template <bool STATS>
__global__ __launch_bounds__(256) void up_noise_rng_fwd_kernel(const float* __restrict__ x, const float* __restrict__ nw,
                                                               float* __restrict__ y, int N, int H, int W, int C, int ups,
                                                               uint64_t seed, uint64_t offset,
                                                               const uint64_t* __restrict__ epoch,
                                                               float* __restrict__ stats_part) {
  if (epoch) offset += *epoch;
  DseeStatsAcc sa;
  sa.init();
  const long total4 = (long)N * H * W * C / 4;
  const int C4 = C / 4, h0 = H >> ups, w0 = W >> ups;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const dsee_nhwq p = dsee_split_nhwq((unsigned)i, C4, W, H);
    const int q = p.q;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)p.n * h0 + (p.h >> ups)) * w0 + (p.w >> ups)) * C + q * 4);
    v += *reinterpret_cast<const f32x4*>(nw + q * 4) * philox_normal4(seed, offset + (uint64_t)i);
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
    if constexpr (STATS) sa.add(v);
  }
  if constexpr (STATS) sa.flush(stats_part, C);
}

// part[blk][C] = sum_pixels a * eps(seed, offset)   (gradient of the noise weights without the eps tensor)
__global__ __launch_bounds__(256) void chdot_rng_partial_kernel(const float* __restrict__ a, float* __restrict__ part,
                                                                long M, int C, int chunk_px, uint64_t seed,
                                                                uint64_t offset, const uint64_t* __restrict__ epoch) {
  if (epoch) offset += *epoch;
  __shared__ f32x4 red[256];
  const int tpp = C / 4, ppb = 256 / tpp > 0 ? 256 / tpp : 1;
  const int q = threadIdx.x % tpp, s = threadIdx.x / tpp;
  const bool active = threadIdx.x < ppb * tpp;
  const long p0 = (long)blockIdx.x * chunk_px, p1 = min(M, p0 + chunk_px);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (active)
    for (long p = p0 + s; p < p1; p += ppb)
      acc += *reinterpret_cast<const f32x4*>(a + p * C + q * 4) * philox_normal4(seed, offset + (uint64_t)(p * tpp + q));
  if (active) red[s * tpp + q] = acc;
  __syncthreads();
  if (active && s == 0) {
    f32x4 v = red[q];
    for (int j = 1; j < ppb; ++j) v += red[j * tpp + q];
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + q * 4) = v;
  }
}

// UpNoise backward in ONE pass over dy: dx = sum-pool(dy) (+ max |dx|) and the NoiseInjection weight gradient
// part[block][c] = sum dy * eps with eps regenerated from the Philox stream the forward drew (the two stand-alone passes, sumpool
// and chdot_rng_partial, each read the 1.07 GB of dy at 256^2).  256 % (C/4) == 0: a thread keeps its channel quad over the
// grid-stride loop; per-block rows folded in block order by chdot_finalize_kernel.
__global__ __launch_bounds__(256) void sumpool_dot_rng_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H,
                                                              int W, int C, int ups, float* __restrict__ amax,
                                                              float* __restrict__ part, uint64_t seed, uint64_t offset,
                                                              const uint64_t* __restrict__ epoch) {
  if (epoch) offset += *epoch;
  __shared__ f32x4 red[256];
  float vmax = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int h0 = H >> ups, w0 = W >> ups, C4 = C / 4, f = 1 << ups;
  const long total4 = (long)N * h0 * w0 * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const dsee_nhwq p = dsee_split_nhwq((unsigned)i, C4, w0, h0);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < f; ++a)
      for (int b = 0; b < f; ++b) {
        const size_t px = ((size_t)p.n * H + p.h * f + a) * W + p.w * f + b;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + px * C + p.q * 4);
        v += d;
        acc += d * philox_normal4(seed, offset + (uint64_t)(px * C4 + p.q));
      }
    *reinterpret_cast<f32x4*>(dx + i * 4) = v;
    vmax = fmaxf(vmax, dsee_absmax4(v));
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);
  red[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < C4) {      // threads t, t + C4, ... share the channel quad t
    f32x4 s = red[threadIdx.x];
    for (int k = threadIdx.x + C4; k < 256; k += C4) s += red[k];
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + threadIdx.x * 4) = s;
  }
}

}  // namespace

extern "C" {

/* y = nearest_up(x) + noise_w[c] * eps with eps = N(0,1) from the Philox stream (seed, offset): the values
 * dsee_rng_fill(seed, offset, normal) would write, generated in registers (normalization.py:289-304 without the tensor) */
int dsee_upsample_noise_rng_fwd(const float* x, const float* noise_w, float* y, int N, int H, int W, int C, int ups,
                                uint64_t seed, uint64_t offset, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && noise_w && C % 4 == 0);
  DSEE_CHECK_ARG((long)N * H * W * C / 4 < (1L << 32));      // (32-bit item index: dsee_split_nhwq)
  up_noise_rng_fwd_kernel<false><<<egrid((long)N * H * W * C / 4), 256, 0, st>>>(x, noise_w, y, N, H, W, C, ups, seed, offset,
                                                                                 dsee_rng_epoch(), nullptr);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* the same pass, also writing the BatchNorm statistics rows of y (the next layer is the param-free BatchNorm of a SPADE/SEAN
 * norm, architecture.py:98 -> normalization.py:107): stats_part [dsee_stats_part_rows(N*H*W*C/4)][3][C], folded by
 * dsee_norm_stats_finalize_parts.  256 % (C/4) == 0. */
int dsee_upsample_noise_rng_fwd_stats(const float* x, const float* noise_w, float* y, int N, int H, int W, int C, int ups,
                                      uint64_t seed, uint64_t offset, float* stats_part, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && noise_w && stats_part && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0);
  DSEE_CHECK_ARG((long)N * H * W * C / 4 < (1L << 32));      // (32-bit item index: dsee_split_nhwq)
  const long items = (long)N * H * W * C / 4;
  const int grid = (int)min((long)DSEE_STATS_ROWS_MAX, (items + 255) / 256);
  up_noise_rng_fwd_kernel<true><<<grid, 256, 0, st>>>(x, noise_w, y, N, H, W, C, ups, seed, offset, dsee_rng_epoch(),
                                                      stats_part);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* out[c] = sum_m a[m][c] * eps[m][c] with the same generated eps: d(noise weight), normalization.py:303-304 */
int dsee_channel_dot_rng(const float* a, float* out, long M, int C, float* workspace, uint64_t seed, uint64_t offset,
                         hipStream_t st) {
  DSEE_CHECK_ARG(a && out && workspace && C % 4 == 0 && C <= 1024);
  long cp = (M + 1023) / 1024;
  if (cp < 64) cp = 64;
  const int parts = (int)((M + cp - 1) / cp);
  chdot_rng_partial_kernel<<<parts, 256, 0, st>>>(a, workspace, M, C, (int)cp, seed, offset, dsee_rng_epoch());
  DSEE_LAUNCH_CHECK();
  chdot_finalize_kernel<<<C / 4, 256, 0, st>>>(workspace, parts, C, out);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_upsample_noise_fwd(const float* x, const float* eps, const float* noise_w, float* y, int N, int H, int W, int C,
                            int ups, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && C % 4 == 0 && (eps == nullptr || noise_w != nullptr));
  DSEE_CHECK_ARG((long)N * H * W * C / 4 < (1L << 32));      // (32-bit item index: dsee_split_nhwq)
  up_noise_fwd_kernel<<<egrid((long)N * H * W * C / 4), 256, 0, st>>>(x, eps, noise_w, y, N, H, W, C, ups);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_sumpool(const float* dy, float* dx, int N, int H, int W, int C, int ups, hipStream_t st) {
  DSEE_CHECK_ARG(dy && dx && C % 4 == 0 && ups >= 1);
  DSEE_CHECK_ARG((long)N * H * W * C / 4 < (1L << 32));      // (32-bit item index: dsee_split_nhwq)
  sumpool_kernel<<<egrid((long)N * (H >> ups) * (W >> ups) * C / 4), 256, 0, st>>>(dy, dx, N, H, W, C, ups);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* dsee_sumpool that also writes max |dx| (64-line form): dx is the output gradient of the previous block's conv_1 */
int dsee_sumpool_amax(const float* dy, float* dx, int N, int H, int W, int C, int ups, float* amax_dx, hipStream_t st) {
  DSEE_CHECK_ARG(dy && dx && amax_dx && C % 4 == 0 && ups >= 1);
  DSEE_CHECK_ARG((long)N * H * W * C / 4 < (1L << 32));      // (32-bit item index: dsee_split_nhwq)
  sumpool_kernel<<<egrid((long)N * (H >> ups) * (W >> ups) * C / 4), 256, 0, st>>>(dy, dx, N, H, W, C, ups, amax_dx);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* UpNoise backward in one pass (normalization.py:289-304 + nn.Upsample, sr.py:57): dx = sum-pool of dy over 2^ups x 2^ups blocks,
 * *amax_dx = max |dx| (64-line form, optional), dnoise_w[c] = sum dy * eps with eps the Philox stream (seed, offset) of the forward.
 * workspace: dsee_sumpool_dot_rng_workspace(N, H, W, C, ups) bytes.  256 % (C/4) == 0. */
size_t dsee_sumpool_dot_rng_workspace(int N, int H, int W, int C, int ups) {
  return (size_t)egrid((long)N * (H >> ups) * (W >> ups) * C / 4) * C * sizeof(float);
}
int dsee_sumpool_dot_rng(const float* dy, float* dx, int N, int H, int W, int C, int ups, float* amax_dx, float* dnoise_w,
                         float* workspace, uint64_t seed, uint64_t offset, hipStream_t st) {
  DSEE_CHECK_ARG(dy && dx && dnoise_w && workspace && C % 4 == 0 && ups >= 1 && 256 % (C / 4) == 0);
  DSEE_CHECK_ARG((long)N * H * W * C / 4 < (1L << 32));      // (32-bit item index: dsee_split_nhwq)
  const int grid = egrid((long)N * (H >> ups) * (W >> ups) * C / 4);
  sumpool_dot_rng_kernel<<<grid, 256, 0, st>>>(dy, dx, N, H, W, C, ups, amax_dx, workspace, seed, offset, dsee_rng_epoch());
  DSEE_LAUNCH_CHECK();
  chdot_finalize_kernel<<<C / 4, 256, 0, st>>>(workspace, grid, C, dnoise_w);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

size_t dsee_channel_dot_workspace(long M, int C) {
  long cp = (M + 1023) / 1024;
  if (cp < 64) cp = 64;
  return (size_t)((M + cp - 1) / cp) * C * sizeof(float);
}

/* out[c] = sum_m a[m][c] * b[m][c]   (b == NULL: column sums).  d(noise weight) = sum dy*eps
 * (normalization.py:303-304), conv bias gradients. */
int dsee_channel_dot(const float* a, const float* b, float* out, long M, int C, float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(a && out && workspace && C % 4 == 0 && C <= 1024);
  long cp = (M + 1023) / 1024;
  if (cp < 64) cp = 64;
  const int parts = (int)((M + cp - 1) / cp);
  chdot_partial_kernel<<<parts, 256, 0, st>>>(a, b, workspace, M, C, (int)cp);
  DSEE_LAUNCH_CHECK();
  chdot_finalize_kernel<<<C / 4, 256, 0, st>>>(workspace, parts, C, out);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_act_fwd(const float* x, float* y, long n, int act, float slope, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && n % 4 == 0);
  act_fwd_kernel<<<egrid(n / 4), 256, 0, st>>>(x, y, n / 4, act, slope);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_act_bwd(const float* dy, const float* y, float* dx, long n, int act, float slope, hipStream_t st) {
  DSEE_CHECK_ARG(dy && y && dx && n % 4 == 0);
  act_bwd_kernel<<<egrid(n / 4), 256, 0, st>>>(dy, y, dx, n / 4, act, slope);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* dsee_act_bwd that also writes max |dx| (64-line form): dx is the output gradient a Winograd layer transforms pre-split */
int dsee_act_bwd_amax(const float* dy, const float* y, float* dx, long n, int act, float slope, float* amax_dx,
                      hipStream_t st) {
  DSEE_CHECK_ARG(dy && y && dx && amax_dx && n % 4 == 0);
  act_bwd_kernel<<<egrid(n / 4), 256, 0, st>>>(dy, y, dx, n / 4, act, slope, amax_dx);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_axpby(const float* a, float alpha, const float* b, float beta, float* y, long n, hipStream_t st) {
  DSEE_CHECK_ARG(a && y && n % 4 == 0);
  axpby_kernel<<<egrid(n / 4), 256, 0, st>>>(a, alpha, b, beta, y, n / 4);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_avgpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && C % 4 == 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  avgpool3s2_fwd_kernel<<<egrid((long)N * Ho * Wo * C / 4), 256, 0, st>>>(x, y, N, H, W, Ho, Wo, C);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_avgpool3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, hipStream_t st) {
  DSEE_CHECK_ARG(dy && dx && C % 4 == 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  avgpool3s2_bwd_kernel<<<egrid((long)N * H * W * C / 4), 256, 0, st>>>(dy, dx, N, H, W, Ho, Wo, C);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && C % 4 == 0 && H % 2 == 0 && W % 2 == 0);
  maxpool2_fwd_kernel<<<egrid((long)N * (H / 2) * (W / 2) * C / 4), 256, 0, st>>>(x, y, N, H, W, C);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_maxpool2_bwd(const float* dy, const float* x, float* dx, int N, int H, int W, int C, hipStream_t st) {
  DSEE_CHECK_ARG(dy && x && dx && C % 4 == 0 && H % 2 == 0 && W % 2 == 0);
  maxpool2_bwd_kernel<<<egrid((long)N * (H / 2) * (W / 2) * C / 4), 256, 0, st>>>(dy, x, dx, N, H, W, C);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cs, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && Cs >= C);
  nchw_to_nhwc_kernel<<<egrid((long)N * H * W * Cs), 256, 0, st>>>(x, y, N, C, H, W, Cs);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int Cs, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && Cs >= C);
  nhwc_to_nchw_kernel<<<egrid((long)N * C * H * W), 256, 0, st>>>(x, y, N, C, H, W, Cs);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_label_to_u8(const float* label, uint8_t* out, long n, hipStream_t st) {
  DSEE_CHECK_ARG(label && out);
  label_to_u8_kernel<<<egrid(n), 256, 0, st>>>(label, out, n);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_image_u8_to_nhwc(const uint8_t* img, const uint8_t* flip, float* out, int N, int H, int W, int cs_out,
                          hipStream_t st) {
  DSEE_CHECK_ARG(img && out && cs_out >= 3 && cs_out % 4 == 0);
  image_u8_to_nhwc_kernel<<<egrid((long)N * H * W), 256, 0, st>>>(img, flip, out, N, H, W, cs_out);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_label_u8_prepare(const uint8_t* lab, const uint8_t* flip, uint8_t* out, int N, int H, int W, int unknown_to,
                          hipStream_t st) {
  DSEE_CHECK_ARG(lab && out && unknown_to >= 0 && unknown_to < 256);
  label_u8_prepare_kernel<<<egrid((long)N * H * W), 256, 0, st>>>(lab, flip, out, N, H, W, unknown_to);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_build_d_input(const uint8_t* lab, const float* img, float* out, long pixels, int L, int Cs, int img_cs,
                       hipStream_t st) {
  return dsee_build_d_input_amax(lab, img, out, pixels, L, Cs, img_cs, nullptr, st);
}

/* ... that also folds max |out| into amax_out (optional; 64-line layout, zeroed by the caller) */
int dsee_build_d_input_amax(const uint8_t* lab, const float* img, float* out, long pixels, int L, int Cs, int img_cs,
                            float* amax_out, hipStream_t st) {
  DSEE_CHECK_ARG(lab && img && out && Cs >= L + 3 && img_cs >= 3);
  build_d_input_kernel<<<egrid(pixels * Cs), 256, 0, st>>>(lab, img, out, pixels, L, Cs, img_cs, amax_out);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_extract_image_grad(const float* din, float* dimg, long pixels, int L, int Cs, int img_cs, hipStream_t st) {
  DSEE_CHECK_ARG(din && dimg && Cs >= L + 3 && img_cs >= 3);
  extract_image_grad_kernel<<<egrid(pixels * img_cs), 256, 0, st>>>(din, dimg, pixels, L, Cs, img_cs);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_bicubic_down(const float* x, float* y, int N, int H, int W, int S, int cs_in, int cs_out, hipStream_t st) {
  DSEE_CHECK_ARG(x && y && cs_in >= 3 && cs_out >= 3);
  bicubic_down_kernel<<<egrid((long)N * S * S * cs_out), 256, 0, st>>>(x, y, N, H, W, S, cs_in, cs_out);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_rng_fill(float* out, long n, uint64_t seed, uint64_t offset, int normal, hipStream_t st) {
  DSEE_CHECK_ARG(out && n % 4 == 0);
  rng_fill_kernel<<<egrid(n / 4), 256, 0, st>>>(out, n / 4, seed, offset, normal, dsee_rng_epoch());
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
