// Internal helpers shared by every translation unit of libdeepsee_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/deepsee_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void dsee_set_error(const char* fmt, ...);
const uint64_t* dsee_rng_epoch();   // device pointer registered with dsee_rng_set_epoch (NULL: epoch 0), capi_core.cpp

#define DSEE_CHECK_ARG(cond)                                                        \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      dsee_set_error("%s:%d: argument check failed: %s", __FILE__, __LINE__, #cond); \
      return DSEE_EINVAL;                                                           \
    }                                                                               \
  } while (0)

#define DSEE_LAUNCH_CHECK()                                                            \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      dsee_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return DSEE_ELAUNCH;                                                             \
    }                                                                                  \
  } while (0)

static inline int dsee_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float dsee_act(float v, int act, float slope) {
  if (act == DSEE_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == DSEE_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DSEE_ACT_TANH) return tanhf(v);
  return v;  // DSEE_ACT_NONE, DSEE_ACT_MASK (handled by the conv epilogue)
}

// d(act)/d(pre) expressed through the saved OUTPUT y (valid for lrelu/relu/tanh).
__device__ __forceinline__ float dsee_act_grad_from_out(float y, int act, float slope) {
  if (act == DSEE_ACT_LRELU) return y > 0.f ? 1.f : slope;
  if (act == DSEE_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == DSEE_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- fp16x2 operand scaling (gemm_bf16x3.hip): power-of-two scale that maps max |x| = amax into [2^13, 2^14)
// (amax = 0 / inf / nan: 1).  Used identically by the producers that pre-split an operand and by the GEMM kernels.
__device__ __forceinline__ float dsee_pow2_scale(float amax) {
  const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu);
  if (e == 0 || e == 255) return 1.f;
  int e2 = 267 - e;  // 127 + 14 - (e - 127) - 1
  e2 = e2 < 1 ? 1 : (e2 > 254 ? 254 : e2);
  return __builtin_bit_cast(float, (unsigned)e2 << 23);
}

// Operand maxima for the fp16 scales.  Accesses of many CUs to ONE address serialise in the L2 (~10 ns each: 65 536
// per-wave atomics cost 0.65 ms per launch), so a maximum lives in DSEE_AMAX_LINES separate cache lines
// (amax[i * DSEE_AMAX_STRIDE], i < 64; the true maximum is the max over them): a block reduces through LDS, then ONE
// thread looks at its line (an L2-served load; a stale value only costs a redundant atomic, never a missed update) and
// issues the atomic max only if it would raise it.  Max is order independent: deterministic.
constexpr int DSEE_AMAX_LINES = 64, DSEE_AMAX_STRIDE = 32;   // 64 lines of 128 bytes = 2048 floats per operand

// every thread of the (<= 1024-thread) block must call this, outside divergent control flow
__device__ __forceinline__ void dsee_block_atomic_absmax(float* amax, float v) {
  __shared__ float dsee_amax_red[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) dsee_amax_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 1; i < nw; ++i) v = fmaxf(v, dsee_amax_red[i]);
    unsigned* line = reinterpret_cast<unsigned*>(amax + (blockIdx.x & (DSEE_AMAX_LINES - 1)) * DSEE_AMAX_STRIDE);
    const unsigned bits = __builtin_bit_cast(unsigned, v);
    const unsigned cur = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bits > cur) atomicMax(line, bits);
  }
}

// per-wave form (no block barrier: callable from an epilogue whose waves finish at different times); all 64 lanes call it
__device__ __forceinline__ void dsee_wave_atomic_absmax(float* amax, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) {
    const unsigned slot = (blockIdx.x + blockIdx.y * 7u) * 4u + (threadIdx.x >> 6);
    unsigned* line = reinterpret_cast<unsigned*>(amax + (slot & (DSEE_AMAX_LINES - 1)) * DSEE_AMAX_STRIDE);
    const unsigned bits = __builtin_bit_cast(unsigned, v);
    const unsigned cur = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bits > cur) atomicMax(line, bits);
  }
}

// BatchNorm statistics in a producer's epilogue (sync_batchnorm/batchnorm.py:65-68): a kernel whose threads keep
// one channel quad for their whole grid-stride loop (gridDim.x * 256 a multiple of C/4) accumulates shifted sums of what it
// stores, and the block writes ONE row (count, mean, M2) x C to part[blockIdx.x][3][C]; dsee_norm_stats_finalize_parts
// folds the rows with Chan's update in block order.  Deterministic: fixed order inside the thread, the block and the fold.
constexpr int DSEE_STATS_ROWS_MAX = 1024;
struct DseeStatsAcc {
  f32x4 shift, a0, a1;
  float n;
  __device__ __forceinline__ void init() {
    shift = a0 = a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    n = 0.f;
  }
  __device__ __forceinline__ void add(const f32x4& v) {
    if (n == 0.f) shift = v;
    const f32x4 d = v - shift;
    a0 += d;
    a1 += d * d;
    n += 1.f;
  }
  // every thread of the 256-thread block must call this (C/4 <= 256, 256 % (C/4) == 0)
  __device__ __forceinline__ void flush(float* __restrict__ part, int C) {
    __shared__ f32x4 st_mean[256], st_m2[256];
    __shared__ float st_n[256];
    f32x4 mean = shift, m2 = {0.f, 0.f, 0.f, 0.f};
    if (n > 0.f) {
      mean = shift + a0 / n;
      m2 = a1 - a0 * a0 / n;
    }
    st_mean[threadIdx.x] = mean;
    st_m2[threadIdx.x] = m2;
    st_n[threadIdx.x] = n;
    __syncthreads();
    const int C4 = C / 4, per = 256 / C4;
    if ((int)threadIdx.x < C4) {
      float nn = n;
      for (int k = 1; k < per; ++k) {
        const int j = k * C4 + threadIdx.x;
        const float nb = st_n[j];
        if (nb > 0.f) {
          const f32x4 d = st_mean[j] - mean;
          const float nt = nn + nb;
          mean += d * (nb / nt);
          m2 += st_m2[j] + d * d * (nn * nb / nt);
          nn = nt;
        }
      }
      float* o = part + (size_t)blockIdx.x * 3 * C + threadIdx.x * 4;
      *reinterpret_cast<f32x4*>(o) = (f32x4){nn, nn, nn, nn};
      *reinterpret_cast<f32x4*>(o + C) = mean;
      *reinterpret_cast<f32x4*>(o + 2 * C) = m2;
    }
  }
};

// the maximum (all 64 lanes of the calling wave must be active; every lane gets the value)
__device__ __forceinline__ float dsee_amax_read(const float* amax) {
  float v = amax[(threadIdx.x & 63) * DSEE_AMAX_STRIDE];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float dsee_absmax4(const f32x4& v) {
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// ---- packed one-term fp16 image of the 16-bit storage mode (gemm_bf16x3.hip): a lane holds 4 channels = 8 bytes per transform
// position; the lanes of a pair (adjacent channel quads of one 32-channel slab) exchange halves across a PAIR of positions -- the
// even lane writes 16 bytes of position xi0 (its own quad + the partner's), the odd lane 16 bytes of position xi0 + 1 -- so that
// every store is 16 bytes and a (tile, slab, position) row is written whole.
typedef unsigned dsee_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned dsee_pk2h(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
}
// o0 / o1 = this lane's 4 channels at positions xi0 / xi0 + 1 (already scaled); row_xi0 = the pair's 16-byte chunk in the row
// of position xi0, pos_bytes = distance between the rows of consecutive positions
__device__ __forceinline__ void dsee_store_pk_pair(unsigned char* row_xi0, size_t pos_bytes, bool odd, const f32x4& o0,
                                                   const f32x4& o1) {
  const unsigned a0 = dsee_pk2h(o0[0], o0[1]), a1 = dsee_pk2h(o0[2], o0[3]), b0 = dsee_pk2h(o1[0], o1[1]), b1 = dsee_pk2h(o1[2], o1[3]);
  const unsigned s0 = odd ? a0 : b0, s1 = odd ? a1 : b1;      // what the partner stores: the even lane's xi0 + 1, the odd lane's xi0
  const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
  const dsee_u32x4 wv = odd ? (dsee_u32x4){r0, r1, b0, b1} : (dsee_u32x4){a0, a1, r0, r1};
  __builtin_nontemporal_store(wv, reinterpret_cast<dsee_u32x4*>(row_xi0 + (odd ? pos_bytes : 0)));

}

// Row factors of the PRE-SPLIT dM = A dY A^T images (round 4).  The rows of A have absolute sums (1, 4, 4, 15, 15, 1), so the 36
// positions of dM range over gains 1 ... 225 and one power-of-two scale per tensor leaves the low-gain positions (the corners,
// which carry most of a 3x3 weight gradient's outer taps) 7.8 bits short.  The producers therefore write
// dM'[i][j] = f_i f_j dM[i][j] with f = (1, 1/4, 1/4, 1/16, 1/16, 1) -- exact power-of-two factors, |dM'| <= max |dY| at every
// position (DSEE_WINO_DM_BOUND = 1) -- and the two consumers of the GEMM results undo them in fp32, again exactly: the weight
// gradient's G^T dU G stage and the data gradient's adjoint input transform multiply position (i, j) by r_i r_j, r = 1 / f.
__host__ __device__ constexpr float dsee_dm_rowf(int i) { return (i == 0 || i == 5) ? 1.f : (i < 3 ? 0.25f : 0.0625f); }
__host__ __device__ constexpr float dsee_dm_rowr(int i) { return (i == 0 || i == 5) ? 1.f : (i < 3 ? 4.f : 16.f); }
__host__ __device__ constexpr float dsee_dm_posf(int xi) { return dsee_dm_rowf(xi / 6) * dsee_dm_rowf(xi % 6); }
__host__ __device__ constexpr float dsee_dm_posr(int xi) { return dsee_dm_rowr(xi / 6) * dsee_dm_rowr(xi % 6); }

// Index arithmetic of the NHWC walkers.  The ISA has no integer divide: a 64-bit division by a run-time value compiles to ~150
// instructions with branches, and three of them per float4 item (n, h, w, channel quad) left up_noise / sumpool instruction-bound
// at ~2 TB/s.  Item counts fit 32 bits (the hosts check), and a 32-bit unsigned division is ~20 instructions.
struct dsee_nhwq {
  int n, h, w, q;
};
__device__ __forceinline__ dsee_nhwq dsee_split_nhwq(unsigned i, unsigned C4, unsigned W, unsigned H) {
  dsee_nhwq r;
  const unsigned t = i / C4, u = t / W, n = u / H;
  r.q = (int)(i - t * C4);
  r.w = (int)(t - u * W);
  r.h = (int)(u - n * H);
  r.n = (int)n;
  return r;
}
