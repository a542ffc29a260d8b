// 3x3 / stride-1 / pad-1 convolutions with a handful of output channels (the generator's to-RGB layer, sr.py:65,94:
// 512 -> 3 at the output resolution).  As an implicit GEMM the N dimension is 3 of a 32-wide MFMA tile (11 TF/s
// forward, 3.7 TF/s weight gradient measured); here the work is laid out along the INPUT channels instead:
//   lane = 4 consecutive input channels (16-byte loads, a wave covers 256 channels of one pixel in one request),
//   a wave walks along an image row keeping the 3x3 window of its channels in registers (3 new loads per pixel),
//   forward : the 9 x 4 x Cout weights of the lane stay in registers, 108 FMAs per pixel, then a wave reduction
//             over channels; lane l keeps the result of pixel l of the 64-pixel segment, waves (channel blocks) are
//             combined through LDS once per segment, bias + activation fused;
//   wgrad   : the 9 x 4 x Cout accumulators of the lane stay in registers (no cross-lane traffic at all), dout of the
//             pixel is wave-uniform; per-block partial sums, then a deterministic reduce into OIHW.
// fp32 VALU arithmetic (exact fp32 FMAs); Cout <= 4, C % 256 == 0, W % 64 == 0.
#include "dsee_common.h"

namespace {

constexpr int TCO = 4;  // output channels handled (stored stride of out / dout is 4)

__device__ __forceinline__ f32x4 ldx(const float* __restrict__ x, int n, int y, int xx, int H, int W, int C, int c) {
  const bool ok = y >= 0 && y < H && xx >= 0 && xx < W;
  return ok ? *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + y) * W + xx) * C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// grid: N*H*(W/64) segments; block: (C/256) waves
__global__ __launch_bounds__(256) void thin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int N,
                                                       int H, int W, int C, int Cout, int act, float slope) {
  __shared__ float part[4][TCO][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = wave * 256 + lane * 4;
  const int segs = W / 64;
  const int seg = blockIdx.x % segs, y = (blockIdx.x / segs) % H, n = blockIdx.x / (segs * H);
  // weights of this lane's 4 channels: wr[co][tap][e] = w[co][c+e][tap]
  f32x4 wr[TCO][9];
#pragma unroll
  for (int co = 0; co < TCO; ++co)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) wr[co][t][e] = co < Cout ? w[((size_t)co * C + c + e) * 9 + t] : 0.f;
  const int x0 = seg * 64;
  f32x4 win[3][3];  // win[dy][dx] = x[y-1+dy][px-1+dx]
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    win[dy][1] = ldx(x, n, y - 1 + dy, x0 - 1, H, W, C, c);
    win[dy][2] = ldx(x, n, y - 1 + dy, x0, H, W, C, c);
  }
  float res[TCO] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < 64; ++s) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      win[dy][0] = win[dy][1];
      win[dy][1] = win[dy][2];
      win[dy][2] = ldx(x, n, y - 1 + dy, x0 + s + 1, H, W, C, c);
    }
#pragma unroll
    for (int co = 0; co < TCO; ++co) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a += win[dy][dx] * wr[co][dy * 3 + dx];
      const float v = wave_sum((a[0] + a[1]) + (a[2] + a[3]));
      res[co] = lane == s ? v : res[co];
    }
  }
#pragma unroll
  for (int co = 0; co < TCO; ++co) part[wave][co][lane] = res[co];
  __syncthreads();
  if (wave == 0) {
    f32x4 o;
#pragma unroll
    for (int co = 0; co < TCO; ++co) {
      float v = part[0][co][lane];
      for (int k = 1; k < nw; ++k) v += part[k][co][lane];
      v += (bias && co < Cout) ? bias[co] : 0.f;
      o[co] = co < Cout ? dsee_act(v, act, slope) : 0.f;
    }
    *reinterpret_cast<f32x4*>(out + (((size_t)n * H + y) * W + x0 + lane) * TCO) = o;
  }
}

// grid: (C/256, nstrip); block: one wave... 256 threads = 4 waves, each wave walks its own segments.
// partial [nstrip*4 waves][TCO][C][9]
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                         float* __restrict__ partial, int N, int H, int W, int C,
                                                         int nwalk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int segs = W / 64;
  const long nseg = (long)N * H * segs;
  const int walker = blockIdx.y * 4 + wave;  // 0 .. nwalk-1
  f32x4 acc[TCO][9];
#pragma unroll
  for (int co = 0; co < TCO; ++co)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[co][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long sg = walker; sg < nseg; sg += nwalk) {
    const int seg = (int)(sg % segs), y = (int)((sg / segs) % H), n = (int)(sg / ((long)segs * H));
    const int x0 = seg * 64;
    f32x4 win[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      win[dy][1] = ldx(x, n, y - 1 + dy, x0 - 1, H, W, C, c);
      win[dy][2] = ldx(x, n, y - 1 + dy, x0, H, W, C, c);
    }
    const float* dr = dout + (((size_t)n * H + y) * W + x0) * TCO;
    for (int s = 0; s < 64; ++s) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        win[dy][0] = win[dy][1];
        win[dy][1] = win[dy][2];
        win[dy][2] = ldx(x, n, y - 1 + dy, x0 + s + 1, H, W, C, c);
      }
      const f32x4 d = *reinterpret_cast<const f32x4*>(dr + s * TCO);  // wave-uniform address
#pragma unroll
      for (int co = 0; co < TCO; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[co][t] += win[t / 3][t % 3] * d[co];
    }
  }
  float* p = partial + (size_t)walker * TCO * C * 9;
#pragma unroll
  for (int co = 0; co < TCO; ++co)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) p[((size_t)co * C + c + e) * 9 + t] = acc[co][t][e];
}

__global__ void thin_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nwalk, int C,
                                         int Cout, int Cin) {
  const long total = (long)Cout * Cin * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % 9);
    const long r = i / 9;
    const int ci = (int)(r % Cin), co = (int)(r / Cin);
    float v = 0.f;
    for (int k = 0; k < nwalk; ++k) v += partial[(((size_t)k * TCO + co) * C + ci) * 9 + t];
    dw[i] = v;
  }
}

constexpr int NWALK = 1024;  // row-segment walkers of the weight gradient (4 per block)

// ---- the same layer as a 1x1 GEMM + a 9-point gather (round 3): z[p][tap*Cout + co] = sum_c x[p][c] w[co][c][tap] is a
// plain [pixels x C] x [C x 9 Cout] product (fp32 MFMA implicit-GEMM kernel, reads x ONCE at HBM rate), and
// out[p][co] = act(bias[co] + sum_tap z[p + d(tap)][tap*Cout + co]) with zero padding outside the image.
// (round 6: KH x KW taps, padding P, output Ho x Wo = H + 2 P - KH + 1 -- the discriminator's last layer, 256 -> 1 channels, 4 x 4,
// padding 2 (discriminator.py:78-96), goes the same way: as an implicit GEMM it filled 1 of 32 MFMA columns at 2-4 TFLOP/s)
__global__ __launch_bounds__(256) void thin_gather_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                              float* __restrict__ out, int N, int H, int W, int ldz,
                                                              int Cout, int act, float slope, int KH, int KW, int P, int Ho,
                                                              int Wo) {
  const long total = (long)N * Ho * Wo;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int x = (int)(p % Wo);
    const long r = p / Wo;
    const int y = (int)(r % Ho);
    const long n = r / Ho;
    float acc[TCO] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < KH * KW; ++t) {
      const int yy = y + t / KW - P, xx = x + t % KW - P;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float* zp = z + ((n * H + yy) * W + xx) * ldz + t * Cout;
#pragma unroll
        for (int co = 0; co < TCO; ++co)
          if (co < Cout) acc[co] += zp[co];
      }
    }
    f32x4 o;
#pragma unroll
    for (int co = 0; co < TCO; ++co) o[co] = co < Cout ? dsee_act(acc[co] + (bias ? bias[co] : 0.f), act, slope) : 0.f;
    *reinterpret_cast<f32x4*>(out + p * TCO) = o;
  }
}

// The same gather with the z rows of a 16 x 16 pixel tile and its halo staged in LDS through coalesced 16-byte loads (H, W
// multiples of 16, ldz % 4 == 0, ldz <= 32): a thread of the form above issues 27 four-byte loads at a 112-byte lane stride --
// 178 us for 59 MB at 256^2.
__global__ __launch_bounds__(256) void thin_gather_fwd_tile_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                                   float* __restrict__ out, int N, int H, int W, int ldz,
                                                                   int Cout, int act, float slope) {
  __shared__ __attribute__((aligned(16))) float zs[18 * 18 * 32];
  const int tw = W >> 4, th = H >> 4;
  const int bx = blockIdx.x % tw, by = (blockIdx.x / tw) % th, n = blockIdx.x / (tw * th);
  const int l4 = ldz >> 2;
  for (int i = threadIdx.x; i < 18 * 18 * l4; i += 256) {
    const int px = i / l4, q = i - px * l4;
    const int hy = px / 18, hx = px - hy * 18;
    const int y = by * 16 + hy - 1, x = bx * 16 + hx - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < H && x >= 0 && x < W) v = *reinterpret_cast<const f32x4*>(z + (((size_t)n * H + y) * W + x) * ldz + q * 4);
    *reinterpret_cast<f32x4*>(zs + px * ldz + q * 4) = v;      // (zero rows outside the image: the padding of the convolution)
  }
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  float acc[TCO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float* zp = zs + ((ty + t / 3) * 18 + tx + t % 3) * ldz + t * Cout;
#pragma unroll
    for (int co = 0; co < TCO; ++co)
      if (co < Cout) acc[co] += zp[co];
  }
  f32x4 o;
#pragma unroll
  for (int co = 0; co < TCO; ++co) o[co] = co < Cout ? dsee_act(acc[co] + (bias ? bias[co] : 0.f), act, slope) : 0.f;
  *reinterpret_cast<f32x4*>(out + (((size_t)n * H + by * 16 + ty) * W + bx * 16 + tx) * TCO) = o;
}

// dz[p][tap*Cout + co] = g[p - d(tap)][co], g = dout * act'(out)   (every element of dz is written, padding columns 0);
// p over the H x W input pixels, g over the Ho x Wo outputs
__global__ __launch_bounds__(256) void thin_gather_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                              float* __restrict__ dz, int N, int H, int W, int ldz, int Cout,
                                                              int act, float slope, int KH, int KW, int P, int Ho, int Wo) {
  const long total = (long)N * H * W;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int x = (int)(p % W);
    const long r = p / W;
    const int y = (int)(r % H);
    const long n = r / H;
    float* zp = dz + p * ldz;
    for (int k = KH * KW * Cout; k < ldz; ++k) zp[k] = 0.f;
    for (int t = 0; t < KH * KW; ++t) {
      const int yy = y - (t / KW - P), xx = x - (t % KW - P);
      const bool ok = yy >= 0 && yy < Ho && xx >= 0 && xx < Wo;
      const long q = (n * Ho + yy) * Wo + xx;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        g = *reinterpret_cast<const f32x4*>(dout + q * TCO);
        const f32x4 o = *reinterpret_cast<const f32x4*>(out + q * TCO);
#pragma unroll
        for (int co = 0; co < TCO; ++co) g[co] *= dsee_act_grad_from_out(o[co], act, slope);
      }
#pragma unroll
      for (int co = 0; co < TCO; ++co)
        if (co < Cout) zp[t * Cout + co] = g[co];
    }
  }
}

// ---- backward of the 27-output 1x1 GEMM the to-RGB layer runs as (ops.conv2d, thin path): K <= 32 rows.
// As implicit GEMMs the two gradients have 27 of a 128-wide tile's columns (data gradient: 1.07 GB written at 1.8 TB/s) or 27
// of its rows (weight gradient: 1.07 GB read at 1.8 TB/s).  Laid out along the 512 input channels instead, like the kernels
// above: a thread owns 4 consecutive channels and keeps the K x 4 weights (data gradient) or K x 4 accumulators (weight
// gradient) in registers; dz of a pixel is block-uniform (its rows go through LDS, 128 pixels at a time, and are read back as
// broadcasts), so a pixel costs one 16-byte load or store and K float4 FMAs per thread.  Exact fp32 FMAs.
constexpr int THIN1_CHUNK = 128;      // pixels whose dz rows a block stages in LDS at a time (<= 16 KB at ldz = 32)

// the block's next chunk of dz rows -> LDS (the rows of consecutive pixels are one contiguous range: coalesced 16-byte loads)
__device__ __forceinline__ void thin1_stage(float* __restrict__ zs, const float* __restrict__ dz, int ldz, long m, int np) {
  const int n4 = np * ldz / 4;
  for (int i = threadIdx.x; i < n4; i += 128)
    reinterpret_cast<f32x4*>(zs)[i] = *reinterpret_cast<const f32x4*>(dz + m * ldz + (size_t)i * 4);
}

template <int KMAX>
__global__ __launch_bounds__(128) void thin1x1_dgrad_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ w,
                                                            float* __restrict__ dx, long M, int C, int K, int px_per_block,
                                                            const float* __restrict__ act_x, float slope,
                                                            float* __restrict__ amax) {
  // act_x (optional): the layer input x = LeakyReLU(pre), whose producer left its activation's backward to this kernel:
  // dx *= x > 0 ? 1 : slope (the pass over dx, x and out that producer would run is 2 GB of traffic at 256^2); amax: max |dx|
  float vmax = 0.f;
  __shared__ __attribute__((aligned(16))) float zs[THIN1_CHUNK * 32];
  const int c = (blockIdx.y * 128 + threadIdx.x) * 4;
  const bool on = c < C;
  f32x4 wr[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    wr[k] = (on && k < K) ? *reinterpret_cast<const f32x4*>(w + (size_t)k * C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
  const long m0 = (long)blockIdx.x * px_per_block, m1 = min(M, m0 + px_per_block);
  for (long mc = m0; mc < m1; mc += THIN1_CHUNK) {
    const int np = (int)min((long)THIN1_CHUNK, m1 - mc);
    __syncthreads();
    thin1_stage(zs, dz, ldz, mc, np);
    __syncthreads();
    if (!on) continue;
    auto pixel = [&](int p, const f32x4& xv) {
      const f32x4* __restrict__ zr = reinterpret_cast<const f32x4*>(zs + p * ldz);     // (uniform address: LDS broadcast)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < KMAX / 4; ++k4) {
        if (k4 * 4 >= K) break;          // (uniform; rows are ldz >= K floats, ldz % 4 == 0: the float4 holding row K-1 is in the row)
        const f32x4 z = zr[k4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += z[e] * wr[k4 * 4 + e];
      }
      if (act_x) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] *= xv[e] > 0.f ? 1.f : slope;
      }
      vmax = fmaxf(vmax, dsee_absmax4(acc));
      __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(dx + (mc + p) * C + c));
    };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 4 <= np; p += 4) {        // (the 4 loads of x -- the activation's sign -- are in flight while the FMAs run)
      f32x4 xv[4] = {z4, z4, z4, z4};
      if (act_x) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(act_x + (mc + p + j) * C + c));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) pixel(p + j, xv[j]);
    }
    for (; p < np; ++p) pixel(p, act_x ? *reinterpret_cast<const f32x4*>(act_x + (mc + p) * C + c) : z4);
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);      // (every thread arrives here)
}

// part[block][k][c] = sum over the block's pixels of dz[m][k] x[m][c]
template <int KMAX>
__global__ __launch_bounds__(128) void thin1x1_wgrad_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ x,
                                                            float* __restrict__ part, long M, int C, int K, int px_per_block) {
  __shared__ __attribute__((aligned(16))) float zs[THIN1_CHUNK * 32];
  const int c = (blockIdx.y * 128 + threadIdx.x) * 4;
  const bool on = c < C;
  f32x4 acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long m0 = (long)blockIdx.x * px_per_block, m1 = min(M, m0 + px_per_block);
  for (long mc = m0; mc < m1; mc += THIN1_CHUNK) {
    const int np = (int)min((long)THIN1_CHUNK, m1 - mc);
    __syncthreads();
    thin1_stage(zs, dz, ldz, mc, np);
    __syncthreads();
    if (!on) continue;
    int p = 0;
    for (; p + 4 <= np; p += 4) {      // (4 independent 16-byte loads in flight per thread)
      f32x4 xv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (mc + p + j) * C + c));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4* __restrict__ zr = reinterpret_cast<const f32x4*>(zs + (p + j) * ldz);
#pragma unroll
        for (int k4 = 0; k4 < KMAX / 4; ++k4) {
          if (k4 * 4 >= K) break;
          const f32x4 z = zr[k4];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[k4 * 4 + e] += z[e] * xv[j];
        }
      }
    }
    for (; p < np; ++p) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (mc + p) * C + c);
      const f32x4* __restrict__ zr = reinterpret_cast<const f32x4*>(zs + p * ldz);
#pragma unroll
      for (int k4 = 0; k4 < KMAX / 4; ++k4) {
        if (k4 * 4 >= K) break;
        const f32x4 z = zr[k4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k4 * 4 + e] += z[e] * xv;
      }
    }
  }
  if (on) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * K + k) * C + c) = acc[k];
  }
}

// dw[k][c] = sum_blocks part[block][k][c] in a fixed order: 8 float4 outputs x 32 block-lanes per workgroup (lane j folds blocks
// j, j + 32, ... with 4 loads in flight, then the 32 lanes are folded in order)
__global__ __launch_bounds__(256) void thin1x1_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                   int blocks, long KC4) {
  __shared__ f32x4 red[32][8];
  const int o = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const long i = (long)blockIdx.x * 8 + o;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (i < KC4) {
    int b = lane;
    for (; b + 96 < blocks; b += 128) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(part + ((size_t)b * KC4 + i) * 4);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(part + ((size_t)(b + 32) * KC4 + i) * 4);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(part + ((size_t)(b + 64) * KC4 + i) * 4);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(part + ((size_t)(b + 96) * KC4 + i) * 4);
      v += (a0 + a1) + (a2 + a3);
    }
    for (; b < blocks; b += 32) v += *reinterpret_cast<const f32x4*>(part + ((size_t)b * KC4 + i) * 4);
  }
  red[lane][o] = v;
  __syncthreads();
  if (lane == 0 && i < KC4) {
    for (int j = 1; j < 32; ++j) v += red[j][o];
    *reinterpret_cast<f32x4*>(dw + i * 4) = v;
  }
}

constexpr int THIN1_BLOCKS = 2048;   // pixel blocks of the two kernels (8 per CU; 2048 x K x C partial sums)

}  // namespace

extern "C" {

/* out [N,H,W,4] = act(conv3x3(x [N,H,W,C], w OIHW [Cout][C][3][3]) + bias), Cout <= 4 (unused channels written 0).
 * Replaces the generator's to-RGB convolution + tanh (sr.py:65,94-95).  C % 256 == 0 (<= 1024), W % 64 == 0. */
int dsee_conv3x3_thin_fwd(const float* x, const float* w_oihw, const float* bias, float* out, int N, int H, int W, int C,
                          int Cout, int act, float slope, hipStream_t st) {
  DSEE_CHECK_ARG(x && w_oihw && out && Cout >= 1 && Cout <= TCO && C % 256 == 0 && C <= 1024 && W % 64 == 0);
  thin_fwd_kernel<<<(unsigned)((long)N * H * (W / 64)), C / 4, 0, st>>>(x, w_oihw, bias, out, N, H, W, C, Cout, act,
                                                                       slope);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* to-RGB convolution (sr.py:94-95) as GEMM + gather: out [N,H,W,4] = act(bias + 9-point gather of z [N,H,W,ldz]), z = the
 * 1x1 convolution of x with the [9*Cout][C] weights (row tap*Cout + co); ldz >= 9*Cout.  The backward form writes dz from
 * dout (the gradient w.r.t. the activated output) and out. */
int dsee_thin_gather_fwd(const float* z, const float* bias, float* out, int N, int H, int W, int ldz, int Cout, int act,
                         float slope, hipStream_t st) {
  DSEE_CHECK_ARG(z && out && Cout >= 1 && Cout <= TCO && ldz >= 9 * Cout);
  if (H % 16 == 0 && W % 16 == 0 && ldz % 4 == 0 && ldz <= 32 && (long)N * (H / 16) * (W / 16) < (1L << 31)) {
    thin_gather_fwd_tile_kernel<<<N * (H / 16) * (W / 16), 256, 0, st>>>(z, bias, out, N, H, W, ldz, Cout, act, slope);
    DSEE_LAUNCH_CHECK();
    return DSEE_OK;
  }
  thin_gather_fwd_kernel<<<(int)min(8192L, ((long)N * H * W + 255) / 256), 256, 0, st>>>(z, bias, out, N, H, W, ldz, Cout,
                                                                                         act, slope, 3, 3, 1, H, W);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_thin_gather_bwd(const float* dout, const float* out, float* dz, int N, int H, int W, int ldz, int Cout, int act,
                         float slope, hipStream_t st) {
  DSEE_CHECK_ARG(dout && out && dz && Cout >= 1 && Cout <= TCO && ldz >= 9 * Cout);
  thin_gather_bwd_kernel<<<(int)min(8192L, ((long)N * H * W + 255) / 256), 256, 0, st>>>(dout, out, dz, N, H, W, ldz, Cout,
                                                                                         act, slope, 3, 3, 1, H, W);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* The general form (round 6): KH x KW taps, stride 1, padding `pad`; z [N,H,W,ldz] (ldz >= KH*KW*Cout), out / dout
 * [N,Ho,Wo,4] with Ho = H + 2 pad - KH + 1.  The discriminator's last layer (256 -> 1, 4 x 4, padding 2: discriminator.py:78-96). */
int dsee_thin_gather_k_fwd(const float* z, const float* bias, float* out, int N, int H, int W, int ldz, int Cout, int KH, int KW,
                           int pad, int act, float slope, hipStream_t st) {
  DSEE_CHECK_ARG(z && out && Cout >= 1 && Cout <= TCO && KH >= 1 && KW >= 1 && pad >= 0 && ldz >= KH * KW * Cout);
  const int Ho = H + 2 * pad - KH + 1, Wo = W + 2 * pad - KW + 1;
  DSEE_CHECK_ARG(Ho > 0 && Wo > 0);
  thin_gather_fwd_kernel<<<(int)min(8192L, ((long)N * Ho * Wo + 255) / 256), 256, 0, st>>>(z, bias, out, N, H, W, ldz, Cout,
                                                                                           act, slope, KH, KW, pad, Ho, Wo);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_thin_gather_k_bwd(const float* dout, const float* out, float* dz, int N, int H, int W, int ldz, int Cout, int KH,
                           int KW, int pad, int act, float slope, hipStream_t st) {
  DSEE_CHECK_ARG(dout && out && dz && Cout >= 1 && Cout <= TCO && KH >= 1 && KW >= 1 && pad >= 0 && ldz >= KH * KW * Cout);
  const int Ho = H + 2 * pad - KH + 1, Wo = W + 2 * pad - KW + 1;
  DSEE_CHECK_ARG(Ho > 0 && Wo > 0);
  thin_gather_bwd_kernel<<<(int)min(8192L, ((long)N * H * W + 255) / 256), 256, 0, st>>>(dout, out, dz, N, H, W, ldz, Cout,
                                                                                         act, slope, KH, KW, pad, Ho, Wo);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

size_t dsee_conv3x3_thin_wgrad_workspace(int C) { return (size_t)NWALK * TCO * C * 9 * sizeof(float); }

/* dw OIHW [Cout][Cin][3][3] of the same convolution from dout [N,H,W,4] (pre-activation gradient); Cin <= C. */
int dsee_conv3x3_thin_wgrad(const float* x, const float* dout, float* workspace, float* dw_oihw, int N, int H, int W,
                            int C, int Cout, int Cin, hipStream_t st) {
  DSEE_CHECK_ARG(x && dout && workspace && dw_oihw && Cout >= 1 && Cout <= TCO && C % 256 == 0 && W % 64 == 0);
  DSEE_CHECK_ARG(Cin <= C);
  thin_wgrad_kernel<<<dim3(C / 256, NWALK / 4), 256, 0, st>>>(x, dout, workspace, N, H, W, C, NWALK);
  DSEE_LAUNCH_CHECK();
  const long total = (long)Cout * Cin * 9;
  thin_wgrad_reduce_kernel<<<(int)min(4096L, (total + 255) / 256), 256, 0, st>>>(workspace, dw_oihw, NWALK, C, Cout,
                                                                                 Cin);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* Backward of y [M][ldz] = x [M][C] . w^T with w [K][C], K <= 32 (the 27-output 1x1 GEMM of the to-RGB layer, sr.py:65,94, in
 * ops.conv2d's thin path): dx [M][C] = dz w (NULL: skipped) and dw [K][C] = dz^T x (NULL: skipped).  C % 4 == 0, K <= ldz <= 32,
 * ldz % 4 == 0 (the padding columns of dz must be finite).
 * workspace: dsee_thin1x1_bwd_workspace(C, K) bytes (weight gradient only).
 * in_lrelu != 0: x = LeakyReLU(pre) came out of a producer that leaves its activation's backward to this call -- dx is the
 * gradient w.r.t. `pre` (dx *= x > 0 ? 1 : slope); amax_dx (optional, 64-line form): max |dx|. */
size_t dsee_thin1x1_bwd_workspace(int C, int K) { return (size_t)THIN1_BLOCKS * K * C * sizeof(float); }

int dsee_thin1x1_bwd(const float* dz, int ldz, const float* w, const float* x, float* dx, float* dw, long M, int C, int K,
                     float* workspace, int in_lrelu, float slope, float* amax_dx, hipStream_t st) {
  DSEE_CHECK_ARG(dz && M > 0 && C % 4 == 0 && K > 0 && K <= 32 && ldz >= K && ldz <= 32 && ldz % 4 == 0);
  DSEE_CHECK_ARG((!dx || w) && (!dw || (x && workspace)) && (!in_lrelu || x));
  const int ppb = (int)((M + THIN1_BLOCKS - 1) / THIN1_BLOCKS), blocks = (int)((M + ppb - 1) / ppb);
  const dim3 grid(blocks, dsee_cdiv(C, 512));
  if (dx) {
    const float* ax = in_lrelu ? x : nullptr;
    if (K <= 28) thin1x1_dgrad_kernel<28><<<grid, 128, 0, st>>>(dz, ldz, w, dx, M, C, K, ppb, ax, slope, amax_dx);
    else thin1x1_dgrad_kernel<32><<<grid, 128, 0, st>>>(dz, ldz, w, dx, M, C, K, ppb, ax, slope, amax_dx);
    DSEE_LAUNCH_CHECK();
  }
  if (dw) {
    if (K <= 28) thin1x1_wgrad_kernel<28><<<grid, 128, 0, st>>>(dz, ldz, x, workspace, M, C, K, ppb);
    else thin1x1_wgrad_kernel<32><<<grid, 128, 0, st>>>(dz, ldz, x, workspace, M, C, K, ppb);
    DSEE_LAUNCH_CHECK();
    const long kc4 = (long)K * C / 4;
    thin1x1_wgrad_reduce_kernel<<<dsee_cdiv(kc4, 8), 256, 0, st>>>(workspace, dw, blocks, kc4);
    DSEE_LAUNCH_CHECK();
  }
  return DSEE_OK;
}

}  // extern "C"
