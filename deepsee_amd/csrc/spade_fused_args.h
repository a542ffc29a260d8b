// Shared between spade_fused.hip (8- / 16-wave forms) and spade_fused_w4.hip (one wave per SIMD): the kernel argument block and
// the compile-time loop helpers.  Internal to libdeepsee_hip.so.
#pragma once
#include <type_traits>
#include <utility>

#include "dsee_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct FusedArgs {
  const unsigned char* V2;   // [K/16][36*T][2][16] fp16: the split Winograd transform of cat (dsee_wino43_input_f16x2)
  const unsigned char* U2;   // [36*G][K/16][rows][2][16] fp16 (dsee_wino43_weights[_table], split = 2)
  const float* amax_v;       // device maximum the V scale was derived from (times v_bound)
  const float* amax_u;
  const float* bias;         // packed [rows]
  const float* x;
  const float* mean;
  const float* invstd;
  float* out;
  float* amax_h;   // optional: max |out| (64-line form) -- the operand bound of the convolution that consumes h
  float* amax_xhat;   // optional: max |xhat| -- with max |dh| the bound of the backward pass's gamma/beta gradient
  float* scale;              // may be NULL
  int cshift;                // log2(4 C) when a sign mask is written (C a power of two then)
  unsigned* mask;            // optional: sign bits of h, [C/32][pixel] words, bit 8 (c & 3) + ((c & 31) >> 2): what the backward
                             // pass needs of h (the LeakyReLU branch) in 1/32 of the bytes
  long T;                    // tiles of the whole batch
  long v_slab_bytes, u_slab_bytes, u_group_bytes;
  unsigned v_bytes, u_bytes; // sizes of the two operand tensors (< 4 GB)
  int tpi, tw;               // tiles per image / per tile row
  int H, W, C, rows;
  int G;                     // weight groups per position: images (per-image tables) or 1
  float v_bound, add_one, slope;
  float* stamps;             // measurement builds (DSEE_FUSED_ABL & 32): per-wave cycle totals
};

template <int I>
using ic = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(ic<Is>{}), ...);
}
// f(ic<0>{}), ..., f(ic<N-1>{}): loop indices that stay compile-time constants through generic lambdas
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const unsigned char*)(((unsigned long long)hi << 32) | lo);
}


}  // namespace
