// Experiment: semantics of ds_read_b64_tr_b16 on gfx950 (which lane receives which LDS element).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(int mode, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int addr;  // in shorts
  if (mode == 0) addr = l * 4;                       // each lane its own 8 bytes, consecutive
  else if (mode == 1) addr = (l & 15) * 64 + (l >> 4) * 4;   // 16 rows of 64 shorts (128 B stride), lane-group picks 4-col chunk
  else addr = (l & 15) * 48 + (l >> 4) * 4;          // row stride 96 B
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(lds + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 3; ++mode) {
    k<<<1, 64>>>(mode, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); printf("\n"); }
  }
  return 0;
}
