// Per-channel statistics and the backward halves of the normalisation layers (NHWC fp32, HBM-bound).
//
//  * sync-free BatchNorm batch statistics (sync_batchnorm/batchnorm.py:65-68 == F.batch_norm, biased var + eps,
//    running stats momentum .1 with unbiased var) and InstanceNorm2d(affine=False) statistics
//    (normalization.py:47-48) are the same kernel with `groups` = 1 or N.
//  * InstanceNorm + LeakyReLU / tanh apply (discriminator.py:88-93, encoder.py:24-27,83-99).
//  * backward of IN+act and of the fused BN + SPADE/SEAN modulate + LeakyReLU.
//
// Layout of every reduction: a block owns a contiguous pixel range of one group; thread t owns the channel
// quad (t % tpp) of pixel slot (t / tpp), streams float4 loads (16 B/lane, coalesced across the quad
// dimension), combines the slots through LDS in a fixed order and writes one partial row; a finalize kernel
// folds the partial rows in index order, so results are bit-reproducible run to run.
#include "dsee_common.h"

namespace {

struct RedGeom {
  int C, tpp, ppb;      // channels, threads per pixel (C/4), pixels per block iteration
  int groups, P;        // pixels per group
  int chunks, chunk_px; // per group
};

RedGeom make_geom(int N, int HW, int C, int groups) {
  RedGeom g;
  g.C = C;
  g.tpp = C / 4;
  g.ppb = 256 / g.tpp;
  if (g.ppb < 1) g.ppb = 1;
  g.groups = groups;
  g.P = (N / groups) * HW;
  long want = ((long)g.P * groups + 1023) / 1024;   // <= ~1024 chunks in all: the finalize walks them serially per lane
  long cp = want > (long)g.ppb * 4 ? want : (long)g.ppb * 4;
  cp = (cp + g.ppb - 1) / g.ppb * g.ppb;
  g.chunk_px = (int)cp;
  g.chunks = (g.P + g.chunk_px - 1) / g.chunk_px;
  return g;
}

// combine K float4 accumulators over the pixel slots of a block (fixed order), result valid for slot 0
template <int K>
__device__ __forceinline__ void block_combine(f32x4 (&acc)[K], int q, int s, int tpp, int ppb, bool active) {
  __shared__ f32x4 red[K * 256];
  if (active) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[(k * ppb + s) * tpp + q] = acc[k];
  }
  __syncthreads();
  if (active && s == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      f32x4 v = red[(k * ppb) * tpp + q];
      for (int j = 1; j < ppb; ++j) v += red[(k * ppb + j) * tpp + q];
      acc[k] = v;
    }
  }
}

// ---- statistics: shifted sums (shift = first pixel of the chunk) -> (count, mean, M2) partials
__global__ __launch_bounds__(256) void stats_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                            RedGeom g) {
  const int tid = threadIdx.x, q = tid % g.tpp, s = tid / g.tpp;
  const bool active = s < g.ppb && q < g.tpp && tid < g.ppb * g.tpp;
  const int grp = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * g.chunk_px, p1 = min(g.P, p0 + g.chunk_px);
  const float* xb = x + (size_t)grp * g.P * g.C;
  f32x4 acc[2];
  acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 shift = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    shift = *reinterpret_cast<const f32x4*>(xb + (size_t)p0 * g.C + q * 4);
    for (int p = p0 + s; p < p1; p += g.ppb) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)p * g.C + q * 4) - shift;
      acc[0] += v;
      acc[1] += v * v;
    }
  }
  block_combine<2>(acc, q, s, g.tpp, g.ppb, active);
  if (active && s == 0) {
    const float n = (float)(p1 - p0);
    f32x4 mean = shift + acc[0] / n;
    f32x4 m2 = acc[1] - acc[0] * acc[0] / n;
    float* o = part + ((size_t)(grp * g.chunks + chunk) * 2) * g.C + q * 4;
    *reinterpret_cast<f32x4*>(o) = mean;
    *reinterpret_cast<f32x4*>(o + g.C) = m2;
  }
}

// block = 8 channels x 32 chunk-lanes: lane l merges chunks l, l+32, ... (Chan's parallel update), lane 0 then merges
// the 32 partial triples in lane order -> fixed order, bit-reproducible (2048 chunks: 64 serial steps per lane).
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ part,
                                                             float* __restrict__ mean_out,
                                                             float* __restrict__ invstd_out,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var,
                                                             RedGeom g, float eps, float momentum) {
  __shared__ float sn[32][8], sm[32][8], s2[32][8];
  const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + cl;
  const bool ok = i < g.groups * g.C;
  const int grp = ok ? i / g.C : 0, c = ok ? i % g.C : 0;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  if (ok)
    for (int k = lane; k < g.chunks; k += 32) {
      const int p0 = k * g.chunk_px, p1 = min(g.P, p0 + g.chunk_px);
      const float nb = (float)(p1 - p0);
      const float mb = part[((size_t)(grp * g.chunks + k) * 2) * g.C + c];
      const float m2b = part[((size_t)(grp * g.chunks + k) * 2 + 1) * g.C + c];
      const float d = mb - mean, nt = n + nb;
      mean += d * nb / nt;
      m2 += m2b + d * d * n * nb / nt;
      n = nt;
    }
  sn[lane][cl] = n;
  sm[lane][cl] = mean;
  s2[lane][cl] = m2;
  __syncthreads();
  if (lane != 0 || !ok) return;
  for (int l = 1; l < 32; ++l) {
    const float nb = sn[l][cl];
    if (nb == 0.f) continue;
    const float d = sm[l][cl] - mean, nt = n + nb;
    mean += d * nb / nt;
    m2 += s2[l][cl] + d * d * n * nb / nt;
    n = nt;
  }
  const float var = m2 / n;  // biased
  mean_out[i] = mean;
  invstd_out[i] = 1.0f / sqrtf(var + eps);  // precise form (not rsqrtf) for parity with torch
  if (run_mean && g.groups == 1) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
  }
}

// Fold of the (count, mean, M2) rows a producer kernel wrote (DseeStatsAcc::flush): block = 4 adjacent channels (one 16-byte
// load per row and quantity) x 256 row-lanes; lane l merges rows l, l+256, ... in order, then the lanes merge pairwise in a
// fixed binary tree through LDS (fixed order: bit-reproducible; 8 dependent merges instead of a 63-step serial fold).
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float qb) {
  if (nb == 0.f) return;
  const float d = mb - mean, nt = n + nb;
  mean += d * nb / nt;
  m2 += qb + d * d * n * nb / nt;
  n = nt;
}

__global__ __launch_bounds__(256) void stats_finalize_parts_kernel(const float* __restrict__ part, int rows, int C,
                                                                   float* __restrict__ mean_out,
                                                                   float* __restrict__ invstd_out,
                                                                   float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                   float eps, float momentum) {
  __shared__ float4 sn[256], sm[256], s2[256];
  const int lane = threadIdx.x, c0 = blockIdx.x * 4;      // (C % 4 == 0: checked by the entry point)
  float n[4] = {0.f, 0.f, 0.f, 0.f}, mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = lane; k < rows; k += 256) {
    const float* r = part + (size_t)k * 3 * C + c0;
    const float4 nb = *reinterpret_cast<const float4*>(r);
    const float4 mb = *reinterpret_cast<const float4*>(r + C);
    const float4 qb = *reinterpret_cast<const float4*>(r + 2 * C);
    chan_merge(n[0], mean[0], m2[0], nb.x, mb.x, qb.x);
    chan_merge(n[1], mean[1], m2[1], nb.y, mb.y, qb.y);
    chan_merge(n[2], mean[2], m2[2], nb.z, mb.z, qb.z);
    chan_merge(n[3], mean[3], m2[3], nb.w, mb.w, qb.w);
  }
  for (int s = 128; s >= 1; s >>= 1) {
    if (lane >= s && lane < 2 * s) {
      sn[lane] = make_float4(n[0], n[1], n[2], n[3]);
      sm[lane] = make_float4(mean[0], mean[1], mean[2], mean[3]);
      s2[lane] = make_float4(m2[0], m2[1], m2[2], m2[3]);
    }
    __syncthreads();
    if (lane < s) {
      const float4 nb = sn[lane + s], mb = sm[lane + s], qb = s2[lane + s];
      chan_merge(n[0], mean[0], m2[0], nb.x, mb.x, qb.x);
      chan_merge(n[1], mean[1], m2[1], nb.y, mb.y, qb.y);
      chan_merge(n[2], mean[2], m2[2], nb.z, mb.z, qb.z);
      chan_merge(n[3], mean[3], m2[3], nb.w, mb.w, qb.w);
    }
    __syncthreads();
  }
  if (lane != 0) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    const float var = m2[j] / n[j];  // biased
    mean_out[c] = mean[j];
    invstd_out[c] = 1.0f / sqrtf(var + eps);
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean[j];
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n[j] / fmaxf(n[j] - 1.f, 1.f));
    }
  }
}

// SyncBN over RCCL, forward half 1: this rank's shard reduced to ONE (mean, M2) row pair per channel (same fixed-order
// Chan merge as stats_finalize_kernel, no eps / running statistics) -- the 2*C floats that travel.
__global__ __launch_bounds__(256) void stats_local_kernel(const float* __restrict__ part, float* __restrict__ local,
                                                          RedGeom g) {
  __shared__ float sn[32][8], sm[32][8], s2[32][8];
  const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  const bool ok = c < g.C;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  if (ok)
    for (int k = lane; k < g.chunks; k += 32) {
      const int p0 = k * g.chunk_px, p1 = min(g.P, p0 + g.chunk_px);
      const float nb = (float)(p1 - p0);
      const float mb = part[((size_t)k * 2) * g.C + c];
      const float m2b = part[((size_t)k * 2 + 1) * g.C + c];
      const float d = mb - mean, nt = n + nb;
      mean += d * nb / nt;
      m2 += m2b + d * d * n * nb / nt;
      n = nt;
    }
  sn[lane][cl] = n;
  sm[lane][cl] = mean;
  s2[lane][cl] = m2;
  __syncthreads();
  if (lane != 0 || !ok) return;
  for (int l = 1; l < 32; ++l) {
    const float nb = sn[l][cl];
    if (nb == 0.f) continue;
    const float d = sm[l][cl] - mean, nt = n + nb;
    mean += d * nb / nt;
    m2 += s2[l][cl] + d * d * n * nb / nt;
    n = nt;
  }
  local[c] = mean;
  local[g.C + c] = m2;
}

// forward half 2: merge the gathered rows [world][2][C] in rank order (every rank computes the same bits) into the
// statistics of the GLOBAL batch.  clamp != 0: inv_std = max(var, eps)^-1/2 (the reference's DataParallel branch,
// sync_batchnorm/batchnorm.py:128-145); clamp == 0: (var + eps)^-1/2 (F.batch_norm, the single-device branch).
__global__ void stats_merge_kernel(const float* __restrict__ gathered, int world, float count, int C, float eps,
                                   float momentum, int clamp, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, float* __restrict__ run_mean,
                                   float* __restrict__ run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int r = 0; r < world; ++r) {
    const float mb = gathered[((size_t)r * 2) * C + c], m2b = gathered[((size_t)r * 2 + 1) * C + c];
    const float d = mb - mean, nt = n + count;
    mean += d * count / nt;
    m2 += m2b + d * d * n * count / nt;
    n = nt;
  }
  const float var = m2 / n;
  mean_out[c] = mean;
  invstd_out[c] = 1.0f / sqrtf(clamp ? fmaxf(var, eps) : var + eps);
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
  }
}

__global__ void eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                  float* __restrict__ mean_out, float* __restrict__ invstd_out, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    mean_out[c] = run_mean[c];
    invstd_out[c] = 1.0f / sqrtf(run_var[c] + eps);
  }
}

// ---- y = act((x - mean[g][c]) * invstd[g][c])
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float* __restrict__ y,
                                                           long total4, int C, long group_elems, int act, float slope,
                                                           float* __restrict__ amax = nullptr) {
  float vmax = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    const unsigned px = (unsigned)i / ((unsigned)C >> 2);       // (32-bit divisions: dsee_common.h)
    const int c = (int)((unsigned)i - px * ((unsigned)C >> 2)) * 4;
    const int grp = (int)(px / (unsigned)(group_elems / C));
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + e);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)grp * C + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)grp * C + c);
    f32x4 r = (v - mu) * is;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = dsee_act(r[k], act, slope);
    *reinterpret_cast<f32x4*>(y + e) = r;
    vmax = fmaxf(vmax, dsee_absmax4(r));
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);   // (block-uniform) max |y|: operand bound of the direct layer that reads y
}

// ---- backward, pass 1: per-(group,channel) sums.
// MODE 0 (IN + act):   g = dy * act'(y);            S0 = sum g,        S1 = sum g*xhat
// MODE 1 (BN modulate): g = dh * lrelu'(h); d = g*scale; S0 = sum d, S1 = sum d*xhat, S2 = sum g*xhat, S3 = sum g
//   and writes dgb[m][packed(gamma c)] = g*xhat, dgb[m][packed(beta c)] = g  (the "dout" of the gamma/beta conv).
template <int MODE>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              float* __restrict__ dgb, int dgb_ld,
                                                              float* __restrict__ part, RedGeom g, int act,
                                                              float slope) {
  constexpr int K = MODE == 0 ? 2 : 4;
  const int tid = threadIdx.x, q = tid % g.tpp, s = tid / g.tpp;
  const bool active = tid < g.ppb * g.tpp;
  const int grp = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * g.chunk_px, p1 = min(g.P, p0 + g.chunk_px);
  const size_t gb = (size_t)grp * g.P;
  f32x4 acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (active) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (size_t)grp * g.C + q * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + (size_t)grp * g.C + q * 4);
    // packed gamma/beta column of channel c: b*128 + w*64 + h*32 + cc with c = b*64 + w*32 + cc
    const int c0 = q * 4;
    const int pcol = (c0 >> 6) * 128 + ((c0 >> 5) & 1) * 64 + (c0 & 31);
    for (int p = p0 + s; p < p1; p += g.ppb) {
      const size_t o = (gb + p) * g.C + c0;
      const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + o);
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
      const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mu) * is;
      f32x4 gg;
#pragma unroll
      for (int k = 0; k < 4; ++k) gg[k] = dv[k] * dsee_act_grad_from_out(yv[k], act, slope);
      if constexpr (MODE == 0) {
        acc[0] += gg;
        acc[1] += gg * xh;
      } else {
        const f32x4 d = gg * *reinterpret_cast<const f32x4*>(scale + o);
        const f32x4 gx = gg * xh;
        acc[0] += d;
        acc[1] += d * xh;
        acc[2] += gx;
        acc[3] += gg;
        float* row = dgb + (gb + p) * (size_t)dgb_ld + pcol;
        *reinterpret_cast<f32x4*>(row) = gx;
        *reinterpret_cast<f32x4*>(row + 32) = gg;
      }
    }
  }
  block_combine<K>(acc, q, s, g.tpp, g.ppb, active);
  if (active && s == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k)
      *reinterpret_cast<f32x4*>(part + ((size_t)(grp * g.chunks + chunk) * K + k) * g.C + q * 4) = acc[k];
  }
}

// A d : 4 -> 6 (winograd.hip: adjoint of the output transform)
__device__ __forceinline__ void a6n(const f32x4 (&d)[4], f32x4 (&o)[6]) {
  const f32x4 s02 = d[0] + d[2], s13 = d[1] + d[3], t02 = d[0] + 4.f * d[2], t13 = 2.f * d[1] + 8.f * d[3];
  o[0] = d[0];
  o[1] = s02 + s13;
  o[2] = s02 - s13;
  o[3] = t02 + t13;
  o[4] = t02 - t13;
  o[5] = d[3];
}

// 4x4 tile of one channel quad -> its 36 Winograd-domain values A v A^T, stored at dM[xi][t][col..col+3]
__device__ __forceinline__ float store_ata(const f32x4 (&v)[4][4], float* __restrict__ dM, long T, long t, int ld,
                                           int col) {
  float vmax = 0.f;
  f32x4 tmp[6][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 c4[4], o[6];
#pragma unroll
    for (int k = 0; k < 4; ++k) c4[k] = v[k][j];
    a6n(c4, o);
#pragma unroll
    for (int k = 0; k < 6; ++k) tmp[k][j] = o[k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    f32x4 o[6];
    a6n(tmp[k], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const f32x4 of = o[j] * dsee_dm_posf(k * 6 + j);      // (every dM carries the row factors of dsee_common.h)
      *reinterpret_cast<f32x4*>(dM + ((size_t)(k * 6 + j) * T + t) * ld + col) = of;
      vmax = fmaxf(vmax, dsee_absmax4(of));
    }
  }
  return vmax;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4n __attribute__((ext_vector_type(4)));

// the saved modulation factor of element offset o: fp32, or -- 16-bit storage mode -- fp16 (written by dsee_spade_fused_fwd_f16p)
template <bool S16>
__device__ __forceinline__ f32x4 ld_scale4(const float* __restrict__ scale, size_t o) {
  if constexpr (S16) {
    const f16x4n h = *reinterpret_cast<const f16x4n*>(reinterpret_cast<const _Float16*>(scale) + o);
    return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  } else {
    return *reinterpret_cast<const f32x4*>(scale + o);
  }
}

// The same 36 values written into the PRE-SPLIT fp16x2 image dM2 [rows/16][36*T][2][16] (winograd.hip:
// wino43_dout_f16x2_kernel): the four lanes of a 16-column slab (consecutive channel quads, same tile) exchange halves so
// that every lane stores 16 contiguous bytes of the 64-byte row (term 0 | term 1).  `l` = lane, col = packed column.
// (`tmp` = A v, the column half of A v A^T, formed by the caller: the reduce kernel builds it column by column while its loads
// arrive, so that g * xhat itself never has to be held -- 330 -> <= 256 registers, two waves per SIMD instead of one)
__device__ __forceinline__ void store_ata_rows_split(const f32x4 (&tmp)[6][4], unsigned char* __restrict__ dM2, long T, long t,
                                                     int col, float sc, int l) {
  const bool odd = (l & 1) != 0;
  const size_t slab = (size_t)36 * T * 64;
  unsigned char* rowp = dM2 + (size_t)(col >> 4) * slab + (size_t)t * 64 + (odd ? 32 + ((l & 3) - 1) * 8 : (l & 3) * 8);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    f32x4 o[6];
    a6n(tmp[k], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      _Float16 h0[4], h1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = o[j][e] * (sc * dsee_dm_posf(k * 6 + j));      // (row factors: dsee_common.h)
        h0[e] = (_Float16)x;
        h1[e] = (_Float16)(x - (float)h0[e]);
      }
      auto pk = [](_Float16 a, _Float16 b) {
        return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
      };
      const unsigned p0a = pk(h0[0], h0[1]), p0b = pk(h0[2], h0[3]), p1a = pk(h1[0], h1[1]), p1b = pk(h1[2], h1[3]);
      const unsigned sa = odd ? p0a : p1a, sb = odd ? p0b : p1b;
      const unsigned ra = (unsigned)__builtin_amdgcn_mov_dpp((int)sa, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
      const unsigned rb = (unsigned)__builtin_amdgcn_mov_dpp((int)sb, 0xB1, 0xF, 0xF, true);
      const u32x4 wv = odd ? (u32x4){ra, rb, p1a, p1b} : (u32x4){p0a, p0b, ra, rb};
      __builtin_nontemporal_store(wv, reinterpret_cast<u32x4*>(rowp + (size_t)(k * 6 + j) * T * 64));
    }
  }
}

// ... and into the PACKED ONE-TERM image dM1 [rows/32][36*T][32] fp16 of the 16-bit storage mode (one scaled fp16 term per
// element; lane pairs exchange halves across pairs of positions, dsee_common.h)
__device__ __forceinline__ void store_ata_rows_pk(const f32x4 (&tmp)[6][4], unsigned char* __restrict__ dM1, long T, long t,
                                                  int col, float sc, int l) {
  const bool odd = (l & 1) != 0;
  const size_t slab = (size_t)36 * T * 64;
  unsigned char* rowp = dM1 + (size_t)(col >> 5) * slab + (size_t)t * 64 + (((col & 31) >> 2) & ~1) * 8;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    f32x4 o[6];
    a6n(tmp[k], o);
#pragma unroll
    for (int j = 0; j < 6; j += 2)
      dsee_store_pk_pair(rowp + (size_t)(k * 6 + j) * T * 64, (size_t)T * 64, odd, o[j] * (sc * dsee_dm_posf(k * 6 + j)),
                         o[j + 1] * (sc * dsee_dm_posf(k * 6 + j + 1)));      // (row factors: dsee_common.h)
  }
}

// Pass 1 of the BN + modulate + LeakyReLU backward with the gamma/beta gradient written straight in the Winograd
// domain: dM[xi][tile][packed gamma col] = (A (g*xhat) A^T)[xi], [packed beta col] = (A g A^T)[xi] -- the operand of the
// weight / table gradient AND (adjoint form) of the embedding's data gradient; the [M][2C] tensor dgb and the separate
// A . A^T pass over it never exist.  Thread = (4x4 tile, channel quad); per-channel sums as in norm_bwd_reduce_kernel<1>
// (one partial row per block, folded in block order by sums_finalize_kernel).  Needs 256 % (C/4) == 0.
__global__ __launch_bounds__(256) void norm_bwd_reduce_wino_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dM, int rows, float* __restrict__ part, int N, int H, int W, int C, float slope,
    float* __restrict__ amax) {
  __shared__ f32x4 red[4 * 256];
  float vmax = 0.f;
  const int C4 = C / 4, th = H / 4, tw = W / 4;
  const long T = (long)N * th * tw, total = T * C4;
  const int q = threadIdx.x % C4, c0 = q * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), is = *reinterpret_cast<const f32x4*>(invstd + c0);
  const int pcol = (c0 >> 6) * 128 + ((c0 >> 5) & 1) * 64 + (c0 & 31);
  f32x4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit divisions: item counts fit, the ISA has no integer divide -- dsee_common.h)
    const long t = (long)((unsigned)i / (unsigned)C4);  // (i % C4 == q: gridDim.x * 256 is a multiple of C4)
    const unsigned r_ = (unsigned)t / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx = (int)((unsigned)t - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    f32x4 gg[4][4], gx[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t o = (((size_t)n * H + ty * 4 + k) * W + tx * 4 + j) * C + c0;
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + o);
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mu) * is;
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = dv[e] * (yv[e] > 0.f ? 1.f : slope);
        const f32x4 d = g * *reinterpret_cast<const f32x4*>(scale + o);
        gg[k][j] = g;
        gx[k][j] = g * xh;
        acc[0] += d;
        acc[1] += d * xh;
        acc[2] += gx[k][j];
        acc[3] += g;
      }
    vmax = fmaxf(vmax, store_ata(gx, dM, T, t, rows, pcol));
    vmax = fmaxf(vmax, store_ata(gg, dM, T, t, rows, pcol + 32));
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);   // (amax is block-uniform)
  // fold the block's threads that share a channel quad (slots s = tid / C4), fixed order
  const int s = threadIdx.x / C4, ns = 256 / C4;
#pragma unroll
  for (int k = 0; k < 4; ++k) red[(k * ns + s) * C4 + q] = acc[k];
  __syncthreads();
  if (s == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 v = red[(k * ns) * C4 + q];
      for (int j = 1; j < ns; ++j) v += red[(k * ns + j) * C4 + q];
      *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 4 + k) * C + c0) = v;
    }
  }
}

// The pre-split form with its own thread mapping: a wave = 4 consecutive tiles x 64 channels (lane = (tile l >> 4, channel quad
// l & 15)), so that the 16 lanes of a tile read 256 contiguous bytes of every pixel and a store instruction writes, per
// 16-column slab, the 64-byte rows of 4 consecutive tiles = 256 contiguous bytes (the (tile, quad) mapping of the fp32 form
// would scatter 64-byte rows: 2.66 vs 2.2 ms at N = 8, 256^2, C = 512).  A wave keeps its 64 channels for its whole loop
// (gridDim.x * 4 is a multiple of C/64); the per-channel sums are folded over the wave's 4 tile lanes and written as
// part[global wave][4][64], summed per channel group in wave order by norm_bwd_split_sums_kernel.
template <bool PK>
__global__ __launch_bounds__(256, 2) void norm_bwd_reduce_wino_split_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ invstd,
    unsigned char* __restrict__ dM2, float* __restrict__ part, int N, int H, int W, int C, float slope,
    const float* __restrict__ amax, float bound, const unsigned* __restrict__ mask) {
  // mask (optional): the sign bits of y written by the fused forward ([C/32][pixel] words) -- y itself is then not read
  const float sc = dsee_pow2_scale(bound * dsee_amax_read(amax));
  const int ncg = C >> 6, th = H / 4, tw = W / 4;
  const long T = (long)N * th * tw, total = (T >> 2) * ncg * 64;
  const int l = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int c0 = (int)(gw % ncg) * 64 + (l & 15) * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), is = *reinterpret_cast<const f32x4*>(invstd + c0);
  const int pcol = (c0 >> 6) * 128 + ((c0 >> 5) & 1) * 64 + (c0 & 31);
  f32x4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit divisions: item counts fit, the ISA has no integer divide -- dsee_common.h)
    const long t = (long)(((unsigned)(i >> 6) / (unsigned)ncg) * 4 + (l >> 4));
    const unsigned r_ = (unsigned)t / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx_ = (int)((unsigned)t - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    // column j of the tile at a time: g is kept (its own transform follows), g * xhat goes straight into the column half of
    // its transform
    f32x4 gg[4][4], tx[6][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 cx[4], o6[6];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t px = ((size_t)n * H + ty * 4 + k) * W + tx_ * 4 + j, o = px * C + c0;
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + o);
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mu) * is;
        f32x4 g;
        if (mask) {
          // ([C/32][N*H*W] words, bit 8 * (c & 3) + ((c & 31) >> 2): the layout the forward kernel's lanes vote in)
          const unsigned bits = mask[(size_t)(c0 >> 5) * ((size_t)N * H * W) + px] >> ((c0 & 31) >> 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = dv[e] * ((bits >> (8 * e)) & 1u ? 1.f : slope);
        } else {
          const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = dv[e] * (yv[e] > 0.f ? 1.f : slope);
        }
        const f32x4 d = g * ld_scale4<PK>(scale, o);
        gg[k][j] = g;
        cx[k] = g * xh;
        acc[0] += d;
        acc[1] += d * xh;
        acc[2] += cx[k];
        acc[3] += g;
        if (k == 1) __builtin_amdgcn_sched_barrier(0);   // (8 loads per scheduling group: see below)
      }
      a6n(cx, o6);
#pragma unroll
      for (int k = 0; k < 6; ++k) tx[k][j] = o6[k];
      // (keeps the 16 loads of the next column from being hoisted above this column's arithmetic: with all 64 in flight per
      // thread the kernel needs > 256 registers; two waves per SIMD x 16 loads cover the HBM latency as well)
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PK) store_ata_rows_pk(tx, dM2, T, t, pcol, sc, l);
    else store_ata_rows_split(tx, dM2, T, t, pcol, sc, l);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 c4[4], o6[6];
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = gg[k][j];
      a6n(c4, o6);
#pragma unroll
      for (int k = 0; k < 6; ++k) tx[k][j] = o6[k];
    }
    if constexpr (PK) store_ata_rows_pk(tx, dM2, T, t, pcol + 32, sc, l);
    else store_ata_rows_split(tx, dM2, T, t, pcol + 32, sc, l);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[k][e] += __shfl_xor(acc[k][e], 16, 64);
      acc[k][e] += __shfl_xor(acc[k][e], 32, 64);
    }
  if (l < 16) {
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(part + ((size_t)gw * 4 + k) * 64 + l * 4) = acc[k];
  }
}

// sums[k][c] = sum over the waves that own channel group c / 64 (global wave index = group mod C/64) of part[wave][k][c % 64]
__global__ __launch_bounds__(256) void norm_bwd_split_sums_kernel(const float* __restrict__ part, int waves, int C,
                                                                  float* __restrict__ sums) {
  __shared__ float sv[32][8];
  const int k = blockIdx.y, cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl, ncg = C >> 6;
  float v = 0.f;
  if (c < C)
    for (int g = (c >> 6) + lane * ncg; g < waves; g += 32 * ncg) v += part[((size_t)g * 4 + k) * 64 + (c & 63)];
  sv[lane][cl] = v;
  __syncthreads();
  if (lane == 0 && c < C) {
    for (int i = 1; i < 32; ++i) v += sv[i][cl];
    sums[(size_t)k * C + c] = v;
  }
}

__global__ __launch_bounds__(64) void amax_product_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float floor_b, float* __restrict__ out) {
  const float va = dsee_amax_read(a), vb = fmaxf(dsee_amax_read(b), floor_b);
  if (threadIdx.x == 0) out[0] = va * vb;
}

__global__ __launch_bounds__(256) void sums_finalize_kernel(const float* __restrict__ part, float* __restrict__ sums,
                                                            int K, RedGeom g) {
  // sums[k][grp][c] = sum over chunks; 8 outputs x 32 chunk-lanes per block, lanes folded in order
  __shared__ float sv[32][8];
  const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + cl;
  const bool ok = i < K * g.groups * g.C;
  float v = 0.f;
  if (ok) {
    const int c = i % g.C, grp = (i / g.C) % g.groups, k = i / (g.C * g.groups);
    for (int ch = lane; ch < g.chunks; ch += 32) v += part[((size_t)(grp * g.chunks + ch) * K + k) * g.C + c];
  }
  sv[lane][cl] = v;
  __syncthreads();
  if (lane == 0 && ok) {
    for (int l = 1; l < 32; ++l) v += sv[l][cl];
    sums[i] = v;
  }
}

// ---- backward, pass 2: dx = invstd * (d - S0/M - xhat * S1/M) [+ add]
template <int MODE, bool S16 = false>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ sums,
                                                             const float* __restrict__ add, float* __restrict__ dx,
                                                             long total4, int C, long group_elems, int groups,
                                                             float inv_count, int act, float slope,
                                                             float* __restrict__ amax = nullptr,
                                                             const unsigned* __restrict__ mask = nullptr) {
  // mask (MODE 1, LeakyReLU): sign bits of y ([C/32][pixel] words); y is then not read
  float vmax = 0.f;
  // (32-bit index arithmetic: item counts fit 32 bits -- the hosts check -- and a 64-bit division is ~150 instructions here)
  const unsigned C4 = (unsigned)C >> 2, gpx = (unsigned)(group_elems / C), npx = (unsigned)(total4 / C4);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    const unsigned px = (unsigned)i / C4;
    const int c = (int)((unsigned)i - px * C4) * 4;
    const int grp = groups == 1 ? 0 : (int)(px / gpx);
    const size_t sc = (size_t)grp * C + c;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + sc);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + sc);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + sc);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + (size_t)groups * C + sc);
    const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + e);
    const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + e) - mu) * is;
    f32x4 d;
    if (mask) {
      const unsigned bits = mask[(size_t)(c >> 5) * npx + px] >> ((c & 31) >> 2);
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = dv[k] * ((bits >> (8 * k)) & 1u ? 1.f : slope);
    } else {
      const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = dv[k] * dsee_act_grad_from_out(yv[k], act, slope);
    }
    if constexpr (MODE == 1) d = d * ld_scale4<S16>(scale, (size_t)e);
    f32x4 r = is * (d - s0 * inv_count - xh * (s1 * inv_count));
    if (add) r += *reinterpret_cast<const f32x4*>(add + e);
    *reinterpret_cast<f32x4*>(dx + e) = r;
    vmax = fmaxf(vmax, dsee_absmax4(r));
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);   // (block-uniform) max |dx|: operand bound of the consumer's A dY A^T
}

int grid_for(long total4) { return (int)min(8192L, (total4 + 255) / 256); }

}  // namespace

extern "C" {

size_t dsee_norm_workspace(int N, int HW, int C, int groups) {
  if (C % 4 || C > 1024 || groups < 1 || N % groups) return 0;
  RedGeom g = make_geom(N, HW, C, groups);
  return (size_t)g.groups * g.chunks * 4 * g.C * sizeof(float) + (size_t)4 * g.groups * g.C * sizeof(float);
}

int dsee_norm_stats(const float* x, int N, int HW, int C, int groups, float eps, float momentum, float* mean,
                    float* invstd, float* running_mean, float* running_var, float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(x && mean && invstd && workspace);
  DSEE_CHECK_ARG(C % 4 == 0 && C <= 1024 && groups >= 1 && N % groups == 0);
  RedGeom g = make_geom(N, HW, C, groups);
  stats_partial_kernel<<<dim3(g.chunks, g.groups), 256, 0, st>>>(x, workspace, g);
  DSEE_LAUNCH_CHECK();
  stats_finalize_kernel<<<dsee_cdiv((long)g.groups * C, 8), 256, 0, st>>>(workspace, mean, invstd, running_mean,
                                                                           running_var, g, eps, momentum);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* The two halves of dsee_norm_stats as separate calls: several BatchNorms over the SAME tensor (norm_0 and norm_s of a
 * resblock, architecture.py:98,127) share one pass over x and differ only in the running statistics they update. */
int dsee_norm_stats_partial(const float* x, int N, int HW, int C, int groups, float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(x && workspace && C % 4 == 0 && C <= 1024 && groups >= 1 && N % groups == 0);
  RedGeom g = make_geom(N, HW, C, groups);
  stats_partial_kernel<<<dim3(g.chunks, g.groups), 256, 0, st>>>(x, workspace, g);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_norm_stats_finalize(const float* workspace, int N, int HW, int C, int groups, float eps, float momentum,
                             float* mean, float* invstd, float* running_mean, float* running_var, hipStream_t st) {
  DSEE_CHECK_ARG(workspace && mean && invstd && C % 4 == 0 && C <= 1024 && groups >= 1 && N % groups == 0);
  RedGeom g = make_geom(N, HW, C, groups);
  stats_finalize_kernel<<<dsee_cdiv((long)g.groups * C, 8), 256, 0, st>>>(workspace, mean, invstd, running_mean,
                                                                           running_var, g, eps, momentum);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* Rows (count, mean, M2) x C written by a producer's epilogue (dsee_upsample_noise_rng_fwd_stats, dsee_wino43_output_stats):
 * rows = dsee_stats_part_rows(work items of the producer). */
int dsee_stats_part_rows(long items) { return (int)min((long)DSEE_STATS_ROWS_MAX, (items + 255) / 256); }

int dsee_norm_stats_finalize_parts(const float* part, int rows, int C, float eps, float momentum, float* mean,
                                   float* invstd, float* running_mean, float* running_var, hipStream_t st) {
  DSEE_CHECK_ARG(part && mean && invstd && rows >= 1 && C % 4 == 0 && C <= 1024);
  stats_finalize_parts_kernel<<<dsee_cdiv(C, 4), 256, 0, st>>>(part, rows, C, mean, invstd, running_mean, running_var, eps,
                                                               momentum);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* SyncBN-over-RCCL (option; reference: the DataParallel branch of SynchronizedBatchNorm2d,
 * sync_batchnorm/batchnorm.py:70-145).  dsee_norm_stats_local reduces this rank's shard to local[2][C] = (mean, M2);
 * the caller all-gathers the rows of all ranks (2*C floats each) and dsee_norm_stats_merge folds them in rank order. */
int dsee_norm_stats_local(const float* x, int N, int HW, int C, float* local, float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(x && local && workspace && C % 4 == 0 && C <= 1024);
  RedGeom g = make_geom(N, HW, C, 1);
  stats_partial_kernel<<<dim3(g.chunks, 1), 256, 0, st>>>(x, workspace, g);
  DSEE_LAUNCH_CHECK();
  stats_local_kernel<<<dsee_cdiv(C, 8), 256, 0, st>>>(workspace, local, g);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_norm_stats_merge(const float* gathered, int world, long count_per_rank, int C, float eps, float momentum,
                          int clamp, float* mean, float* invstd, float* running_mean, float* running_var,
                          hipStream_t st) {
  DSEE_CHECK_ARG(gathered && mean && invstd && world >= 1 && count_per_rank > 0 && C > 0);
  stats_merge_kernel<<<dsee_cdiv(C, 256), 256, 0, st>>>(gathered, world, (float)count_per_rank, C, eps, momentum, clamp,
                                                        mean, invstd, running_mean, running_var);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_norm_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                         float* invstd, hipStream_t st) {
  DSEE_CHECK_ARG(running_mean && running_var && mean && invstd);
  eval_stats_kernel<<<dsee_cdiv(C, 256), 256, 0, st>>>(running_mean, running_var, mean, invstd, C, eps);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_norm_act_fwd(const float* x, const float* mean, const float* invstd, float* y, int N, int HW, int C,
                      int groups, int act, float slope, hipStream_t st) {
  return dsee_norm_act_fwd_amax(x, mean, invstd, y, N, HW, C, groups, act, slope, nullptr, st);
}

/* ... that also writes max |y| into amax_y (optional; 64-line layout, zeroed by the caller): the bound of the next direct layer */
int dsee_norm_act_fwd_amax(const float* x, const float* mean, const float* invstd, float* y, int N, int HW, int C,
                      int groups, int act, float slope, float* amax_y, hipStream_t st) {
  DSEE_CHECK_ARG(x && mean && invstd && y && C % 4 == 0 && N % groups == 0 && (long)N * HW * C / 4 < (1L << 32));
  const long total4 = (long)N * HW * C / 4;
  norm_act_fwd_kernel<<<grid_for(total4), 256, 0, st>>>(x, mean, invstd, y, total4, C, (long)(N / groups) * HW * C, act,
                                                        slope, amax_y);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_norm_act_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                      float* dx, int N, int HW, int C, int groups, int act, float slope, float* workspace,
                      hipStream_t st) {
  return dsee_norm_act_bwd_amax(dy, y, x, mean, invstd, dx, N, HW, C, groups, act, slope, workspace, nullptr, st);
}

/* IN + act backward: dx from (dy, y, x); amax_dx (optional, zeroed by the caller) receives max |dx|. */
int dsee_norm_act_bwd_amax(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                      float* dx, int N, int HW, int C, int groups, int act, float slope, float* workspace,
                      float* amax_dx, hipStream_t st) {
  DSEE_CHECK_ARG(dy && y && x && mean && invstd && dx && workspace);
  DSEE_CHECK_ARG(C % 4 == 0 && C <= 1024 && N % groups == 0);
  RedGeom g = make_geom(N, HW, C, groups);
  float* sums = workspace + (size_t)g.groups * g.chunks * 4 * g.C;
  norm_bwd_reduce_kernel<0><<<dim3(g.chunks, g.groups), 256, 0, st>>>(dy, y, x, nullptr, mean, invstd, nullptr, 0,
                                                                       workspace, g, act, slope);
  DSEE_LAUNCH_CHECK();
  sums_finalize_kernel<<<dsee_cdiv((long)2 * g.groups * C, 8), 256, 0, st>>>(workspace, sums, 2, g);
  DSEE_LAUNCH_CHECK();
  const long total4 = (long)N * HW * C / 4;
  DSEE_CHECK_ARG(total4 < (1L << 32));
  norm_bwd_apply_kernel<0><<<grid_for(total4), 256, 0, st>>>(dy, y, x, nullptr, mean, invstd, sums, nullptr, dx, total4,
                                                             C, (long)(N / groups) * HW * C, g.groups,
                                                             1.0f / (float)g.P, act, slope, amax_dx);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* BN + modulate + LeakyReLU backward:
 *   in : dh, h (saved output), x, scale (saved), mean/invstd [C]
 *   out: dgb [M][dgb_ld] in packed gamma/beta order (g*xhat | g), col_sums [2][C] = (sum g*xhat, sum g),
 *        dx = invstd*(g*scale - mean(g*scale) - xhat*mean(g*scale*xhat)) + add
 * Two halves so that the data-parallel SyncBN option can all-reduce the per-channel sums in between:
 *   dsee_modulate_bwd_reduce -> sums[4][C] = (sum d, sum d*xhat, sum g*xhat, sum g) over THIS rank's pixels, d = g*scale
 *   dsee_modulate_bwd_apply  <- sums[0..1] (local, or summed over the ranks) and inv_count = 1 / (pixels behind them) */
int dsee_modulate_bwd_reduce(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                             const float* invstd, float* dgb, int dgb_ld, float* sums, int N, int HW, int C, float slope,
                             float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(dh && h && x && scale && mean && invstd && dgb && sums && workspace);
  DSEE_CHECK_ARG(C % 4 == 0 && C <= 1024 && dgb_ld >= (C + 63) / 64 * 128);
  RedGeom g = make_geom(N, HW, C, 1);
  norm_bwd_reduce_kernel<1><<<dim3(g.chunks, 1), 256, 0, st>>>(dh, h, x, scale, mean, invstd, dgb, dgb_ld, workspace, g,
                                                                DSEE_ACT_LRELU, slope);
  DSEE_LAUNCH_CHECK();
  sums_finalize_kernel<<<dsee_cdiv((long)4 * C, 8), 256, 0, st>>>(workspace, sums, 4, g);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* The same reduce half with the gamma/beta gradient produced directly in the Winograd domain:
 * dM [36][T][rows] = A (g*xhat | g) A^T in the packed column order (T = N*(H/4)*(W/4), rows = 2C) -- what
 * dsee_wino43_dout(dgb) would give, without dgb.  C % 64 == 0, 256 % (C/4) == 0, H, W % 4 == 0.
 * workspace: dsee_modulate_bwd_wino_workspace(). */
static int wino_reduce_blocks(int N, int H, int W, int C) {
  const long total = (long)N * (H / 4) * (W / 4) * (C / 4);
  long b = (total + 255) / 256;
  return (int)(b < 2048 ? b : 2048);
}

size_t dsee_modulate_bwd_wino_workspace(int N, int H, int W, int C) {
  // fp32 form: one row [4][C] per block; pre-split form: one row [4][64] per wave (4 per block)
  return (size_t)(wino_reduce_blocks(N, H, W, C) + 4) * 4 * (C > 256 ? C : 256) * sizeof(float);   // (+4: grid rounded up to C/64 groups)
}

int dsee_modulate_bwd_reduce_wino(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                                  const float* invstd, float* dM, int rows, float* sums, int N, int H, int W, int C,
                                  float slope, float* workspace, float* amax, hipStream_t st) {
  DSEE_CHECK_ARG(dh && h && x && scale && mean && invstd && dM && sums && workspace);
  DSEE_CHECK_ARG(C % 64 == 0 && C <= 1024 && 256 % (C / 4) == 0 && rows == 2 * C && H % 4 == 0 && W % 4 == 0);
  DSEE_CHECK_ARG((long)N * H * W * C / 64 < (1L << 32));      // (32-bit item index in the kernels: dsee_common.h)
  const int blocks = wino_reduce_blocks(N, H, W, C);
  norm_bwd_reduce_wino_kernel<<<blocks, 256, 0, st>>>(dh, h, x, scale, mean, invstd, dM, rows, workspace, N, H, W, C,
                                                      slope, amax);
  DSEE_LAUNCH_CHECK();
  RedGeom g = make_geom(N, H * W, C, 1);
  g.chunks = blocks;  // one partial row [4][C] per block
  sums_finalize_kernel<<<dsee_cdiv((long)4 * C, 8), 256, 0, st>>>(workspace, sums, 4, g);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"

template <bool PK>
static int reduce_wino_split_launch(const float* dh, const float* h, const float* x, const float* scale,
                                        const float* mean, const float* invstd, void* dM2, int rows, float* sums, int N,
                                        int H, int W, int C, float slope, float* workspace, const float* amax_g, float bound,
                                        const uint32_t* sign_mask, hipStream_t st) {
  DSEE_CHECK_ARG(dh && (h || sign_mask) && x && scale && mean && invstd && dM2 && sums && workspace && amax_g && bound >= DSEE_WINO_DM_BOUND);
  DSEE_CHECK_ARG(C % 64 == 0 && C <= 1024 && 256 % (C / 4) == 0 && rows == 2 * C && H % 4 == 0 && W % 4 == 0);
  DSEE_CHECK_ARG((long)N * H * W * C / 64 < (1L << 32));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(((long)N * (H / 4) * (W / 4)) % 16 == 0);
  // a grid whose wave count is a multiple of the C/64 channel groups (every wave keeps its channels)
  const int ncg = C / 64, m = ncg / (ncg % 4 == 0 ? 4 : (ncg % 2 == 0 ? 2 : 1));
  int blocks = wino_reduce_blocks(N, H, W, C) / m * m;
  if (blocks < m) blocks = m;
  DSEE_CHECK_ARG(blocks <= wino_reduce_blocks(N, H, W, C) || blocks == m);
  norm_bwd_reduce_wino_split_kernel<PK><<<blocks, 256, 0, st>>>(dh, h, x, scale, mean, invstd,
                                                            reinterpret_cast<unsigned char*>(dM2), workspace, N, H, W, C, slope,
                                                            amax_g, bound, sign_mask);
  DSEE_LAUNCH_CHECK();
  norm_bwd_split_sums_kernel<<<dim3(dsee_cdiv(C, 8), 4), 256, 0, st>>>(workspace, blocks * 4, C, sums);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}


extern "C" {

/* ... with dM written PRE-SPLIT: dM2 [rows/16][36*T][2][16] fp16 scaled by the power of two of bound x *amax_g, where
 * *amax_g >= max |dh| * max(1, max |xhat|) is known before the kernel runs (dsee_amax_product of the maxima the producers of dh
 * and of the forward pass wrote) and bound >= DSEE_WINO_DM_BOUND.  T % 16 == 0. */
int dsee_modulate_bwd_reduce_wino_f16x2(const float* dh, const float* h, const float* x, const float* scale,
                                        const float* mean, const float* invstd, void* dM2, int rows, float* sums, int N,
                                        int H, int W, int C, float slope, float* workspace, const float* amax_g, float bound,
                                        const uint32_t* sign_mask, hipStream_t st) {
  return reduce_wino_split_launch<false>(dh, h, x, scale, mean, invstd, dM2, rows, sums, N, H, W, C, slope, workspace, amax_g,
                                         bound, sign_mask, st);
}

/* 16-bit storage mode: the same pass writing dM as the PACKED ONE-TERM operand dM1 [rows/32][36*T][32] fp16 (consumers:
 * dsee_wino43_wgrad[_table] split = 7, dsee_gemm_f16p_pre) */
int dsee_modulate_bwd_reduce_wino_f16p(const float* dh, const float* h, const float* x, const float* scale,
                                       const float* mean, const float* invstd, void* dM1, int rows, float* sums, int N,
                                       int H, int W, int C, float slope, float* workspace, const float* amax_g, float bound,
                                       const uint32_t* sign_mask, hipStream_t st) {
  return reduce_wino_split_launch<true>(dh, h, x, scale, mean, invstd, dM1, rows, sums, N, H, W, C, slope, workspace, amax_g,
                                        bound, sign_mask, st);
}

/* out (64-line slot, zeroed by the caller) <- max(a) * max(floor_b, max(b)): the operand bound of a product of two tensors */
int dsee_amax_product(const float* a, const float* b, float floor_b, float* out, hipStream_t st) {
  DSEE_CHECK_ARG(a && b && out);
  amax_product_kernel<<<1, 64, 0, st>>>(a, b, floor_b, out);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_modulate_bwd_apply(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                            const float* invstd, const float* sums, const float* add, float* dx, int N, int HW, int C,
                            float inv_count, float slope, hipStream_t st) {
  DSEE_CHECK_ARG(dh && h && x && scale && mean && invstd && sums && dx && C % 4 == 0 && inv_count > 0.f);
  const long total4 = (long)N * HW * C / 4;
  DSEE_CHECK_ARG(total4 < (1L << 32));
  norm_bwd_apply_kernel<1><<<grid_for(total4), 256, 0, st>>>(dh, h, x, scale, mean, invstd, sums, add, dx, total4, C,
                                                             (long)N * HW * C, 1, inv_count, DSEE_ACT_LRELU, slope);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* the same, also writing max |dx| (64-line form, zeroed by the caller): dx is the output gradient of the convolution in front
 * of this norm, whose A dY A^T transform is then written pre-split with the scale known in advance (dsee_wino43_dout_f16x2) */
int dsee_modulate_bwd_apply_amax(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                                 const float* invstd, const float* sums, const float* add, float* dx, int N, int HW, int C,
                                 float inv_count, float slope, float* amax_dx, int scale_f16, const uint32_t* sign_mask,
                                 hipStream_t st) {
  DSEE_CHECK_ARG(dh && (h || sign_mask) && x && scale && mean && invstd && sums && dx && amax_dx && C % 4 == 0 && inv_count > 0.f);
  DSEE_CHECK_ARG(!sign_mask || C % 32 == 0);
  const long total4 = (long)N * HW * C / 4;
  DSEE_CHECK_ARG(total4 < (1L << 32));
  if (scale_f16)
    norm_bwd_apply_kernel<1, true><<<grid_for(total4), 256, 0, st>>>(dh, h, x, scale, mean, invstd, sums, add, dx, total4, C,
                                                                     (long)N * HW * C, 1, inv_count, DSEE_ACT_LRELU, slope,
                                                                     amax_dx, sign_mask);
  else
    norm_bwd_apply_kernel<1><<<grid_for(total4), 256, 0, st>>>(dh, h, x, scale, mean, invstd, sums, add, dx, total4, C,
                                                               (long)N * HW * C, 1, inv_count, DSEE_ACT_LRELU, slope, amax_dx,
                                                               sign_mask);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_modulate_bwd(const float* dh, const float* h, const float* x, const float* scale, const float* mean,
                      const float* invstd, const float* add, float* dx, float* dgb, int dgb_ld, float* col_sums, int N,
                      int HW, int C, float slope, float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(col_sums && workspace && C % 4 == 0 && C <= 1024);
  RedGeom g = make_geom(N, HW, C, 1);
  float* sums = workspace + (size_t)g.chunks * 4 * g.C;
  int rc = dsee_modulate_bwd_reduce(dh, h, x, scale, mean, invstd, dgb, dgb_ld, sums, N, HW, C, slope, workspace, st);
  if (rc != DSEE_OK) return rc;
  (void)hipMemcpyAsync(col_sums, sums + 2 * C, (size_t)2 * C * sizeof(float), hipMemcpyDeviceToDevice, st);
  return dsee_modulate_bwd_apply(dh, h, x, scale, mean, invstd, sums, add, dx, N, HW, C, 1.0f / (float)g.P, slope, st);
}

}  // extern "C"
