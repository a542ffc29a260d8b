#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* o, const float* a, const float* b) {
  f32x4 y = {0.f, 0.f, 0.f, 0.f};
  asm volatile("" : "+a"(y));
  float av = a[threadIdx.x], bv = b[threadIdx.x];
  asm volatile("s_nop 4\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(y) : "v"(av), "v"(bv));
  asm volatile("s_nop 7");
  o[threadIdx.x * 4 + 0] = y[0]; o[threadIdx.x * 4 + 1] = y[1]; o[threadIdx.x * 4 + 2] = y[2]; o[threadIdx.x * 4 + 3] = y[3];
}
int main() {
  float *o, *a, *b; hipMalloc(&o, 1024); hipMalloc(&a, 256); hipMalloc(&b, 256);
  float ha[64], hb[64], ho[256];
  for (int i = 0; i < 64; ++i) { ha[i] = 1.f + (i & 3); hb[i] = 100.f + i; }
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(o, a, b); hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int n = 0; n < 4; ++n) { float want = (1.f + n) * (100.f + l); if (ho[l * 4 + n] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, n, ho[l*4+n], want); ++bad; } }
  printf("mfma 4x4x1 layout check: %d mismatches (D[lane][reg i] = A[lane&~3 | i] * B[lane])\n", bad);
  return bad != 0;
}
