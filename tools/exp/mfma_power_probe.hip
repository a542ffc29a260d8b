// What the socket power cap leaves of the dense 16-bit matrix peak: MFMAs only, operands resident in registers (random bit patterns
// loaded once from HBM, or zeros), no LDS, no memory traffic in the loop.  One wave per SIMD, 16 independent 32x32x16 accumulators.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_power_probe tools/exp/mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool BF>
__global__ __launch_bounds__(256, 1) void k(const u32x4* src, float* out, int iters) {
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(blockIdx.x * 256 + threadIdx.x) * 8 + i];
    b[i] = src[(blockIdx.x * 256 + threadIdx.x) * 8 + 4 + i];
  }
  f32x16 acc[16];
  for (int t = 0; t < 16; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if constexpr (BF)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t & 3]), __builtin_bit_cast(bf16x8, b[t >> 2]), acc[t], 0, 0, 0);
      else
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t & 3]), __builtin_bit_cast(f16x8, b[t >> 2]), acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 16; ++t) s += acc[t][threadIdx.x & 15];
  if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
  const bool bf = argc > 1 && !strcmp(argv[1], "bf16");
  const bool zero = argc > 2 && !strcmp(argv[2], "zeros");
  const double secs = argc > 3 ? atof(argv[3]) : 3.0;
  const int blocks = 256, iters = 4000;
  size_t n = (size_t)blocks * 256 * 8 * 4;
  unsigned* h = (unsigned*)malloc(n * 4);
  srand(1);
  for (size_t i = 0; i < n; ++i) {
    // two 16-bit floats with random sign / mantissa and exponents within a few binades of 1 (no inf / nan)
    unsigned lo = (rand() & 0x83FF) | ((13 + rand() % 5) << 10), hi = (rand() & 0x83FF) | ((13 + rand() % 5) << 10);
    if (bf) { lo = (rand() & 0x807F) | ((125 + rand() % 5) << 7); hi = (rand() & 0x807F) | ((125 + rand() % 5) << 7); }
    h[i] = zero ? 0u : (lo | (hi << 16));
  }
  unsigned* d; float* o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) { if (bf) k<true><<<blocks, 256>>>((u32x4*)d, o, iters); else k<false><<<blocks, 256>>>((u32x4*)d, o, iters); }
  hipDeviceSynchronize();
  double total_ms = 0; long launches = 0;
  while (total_ms < secs * 1e3) {
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) { if (bf) k<true><<<blocks, 256>>>((u32x4*)d, o, iters); else k<false><<<blocks, 256>>>((u32x4*)d, o, iters); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); total_ms += ms; launches += 20;
    if (launches % 200 == 0) { float last = ms / 20; (void)last; }
  }
  // the last batch's rate = the steady state under the power cap
  hipEventRecord(e0);
  for (int w = 0; w < 20; ++w) { if (bf) k<true><<<blocks, 256>>>((u32x4*)d, o, iters); else k<false><<<blocks, 256>>>((u32x4*)d, o, iters); }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 20.0 * blocks * 4 * (double)iters * 16 * 2.0 * 32 * 32 * 16;
  printf("%s %s: steady state %.1f TFLOP/s dense (%.3f of 2516.6), average over %.1f s %.1f TFLOP/s\n", bf ? "bf16" : "fp16",
         zero ? "zeros" : "random", flop / ms / 1e9, flop / ms / 1e9 / 2516.6, total_ms / 1e3,
         (double)launches / 20 * flop / total_ms / 1e9);
  return 0;
}
