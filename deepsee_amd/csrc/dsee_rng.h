// Philox4x32-10 counter RNG + Box-Muller shared by the kernels that regenerate NoiseInjection's eps in registers
// (elementwise.hip: dsee_rng_fill / dsee_upsample_noise_rng_fwd / dsee_channel_dot_rng; winograd.hip: the output transform
// with the noise_middle injection fused).  Element i (float4 granularity, NHWC linear order) of a stream (seed, offset)
// is philox_normal4(seed, offset + i): every consumer sees the values dsee_rng_fill would have written.
#pragma once
#include "dsee_common.h"

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
  c[0] = h1 ^ c[1] ^ k0;
  c[1] = l1;
  c[2] = h0 ^ c[3] ^ k1;
  c[3] = l0;
}

__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = c[k];
}

// Box-Muller on two pairs of 32-bit draws
__device__ __forceinline__ f32x4 box_muller4(const uint32_t (&r)[4]) {
  const float u0 = ((float)r[0] + 1.f) * 2.3283064e-10f, u1 = (float)r[1] * 2.3283064e-10f;
  const float u2 = ((float)r[2] + 1.f) * 2.3283064e-10f, u3 = (float)r[3] * 2.3283064e-10f;
  const float a = sqrtf(-2.f * __logf(u0)), b = sqrtf(-2.f * __logf(u2));
  float s0, c0, s1, c1;
  __sincosf(6.2831853f * u1, &s0, &c0);
  __sincosf(6.2831853f * u3, &s1, &c1);
  return (f32x4){a * c0, a * s0, b * c1, b * s1};
}

__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t ctr) {
  uint32_t r[4];
  philox4(seed, ctr, r);
  return box_muller4(r);
}

