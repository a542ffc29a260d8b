// Experiment: semantics of ds_read_b64_tr_b16 on gfx950 (which lane receives which LDS element).
// Build: hipcc --offload-arch=gfx950 -O2 tr16.hip -o tr16 ; run on the GPU box.
// Observed (round 1): within each 16-lane group the 16 lanes' 8-byte pieces form a 4 x 16 matrix -- row e (0..3) is
// supplied by lanes 4e..4e+3 (each lane's address points at 4 consecutive bf16 = columns 4q..4q+3 of that row; the rows
// may be anywhere in LDS) -- and lane i receives column i: out[i][e] = piece[lane 4e + (i>>2)][i & 3].
// => an MFMA operand whose reduction index is the SLOW axis of the LDS image ([k][m], m contiguous) can be fetched with
//    two tr reads per 8 k's (a transposed A/B path for gemm_bf16x3.hip; not used yet).
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
