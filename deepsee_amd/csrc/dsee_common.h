// Internal helpers shared by every translation unit of libdeepsee_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/deepsee_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void dsee_set_error(const char* fmt, ...);

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

// *amax = max(*amax, max over the wave of v) for v >= 0 (order independent -> deterministic).  Atomics on one address
// serialise in the L2 (~12 ns each): the wave first looks at the current maximum (an L2-served load; a stale value only
// costs a redundant atomic, never a missed update) and issues the atomic only if it would raise it -- after the first few
// waves almost none do.
__device__ __forceinline__ void dsee_wave_atomic_absmax(float* amax, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) {
    const unsigned bits = __builtin_bit_cast(unsigned, v);
    const unsigned cur = __hip_atomic_load(reinterpret_cast<unsigned*>(amax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bits > cur) atomicMax(reinterpret_cast<unsigned*>(amax), bits);
  }
}
__device__ __forceinline__ float dsee_absmax4(const f32x4& v) {
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
