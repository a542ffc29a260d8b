// Winograd F(4x4, 3x3) transforms for the 3x3 / stride-1 / pad-1 convolutions (Lavin & Gray 2016, interpolation
// points 0, +-1, +-2, inf).  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  per 4x4 output tile and (cin, cout) pair: the
// element-wise product summed over cin is 36 independent GEMMs  M[xi][tile][cout] = V[xi][tile][cin] * U[xi][cin][cout]
// (2.25 M*Cin*Cout MACs instead of 9), run on the fp32 MFMA implicit-GEMM kernel as a grouped 1x1 convolution
// (dsee_conv2d_fwd_grouped).  The two transforms here are streaming kernels (16 B per lane along C):
//   input  : x [N][H][W][C]            -> V [36][T][C],  T = N*(H/4)*(W/4) tiles, zero padding at the image border
//   output : M [36][T][Cout] (+bias, +residual, activation) -> y [N][H][W][Cout]
//   weights: w OIHW [Cout][Cin][3][3]  -> U [36][rows(Cout)][Cin]   (forward)   or, with `transpose_flip`,
//                                         U'[36][rows(Cin)][Cout] of the 180-degree-rotated kernel (data gradient)
// Replaces the same ATen convolutions as conv_mfma.hip (architecture.py:98,122 and their backward).
#include "dsee_common.h"
#include "dsee_rng.h"

namespace {

// bf16x3 split of fp32 values (gemm_bf16x3.hip): x = x0 + x1 + x2 exactly, each term a bf16
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3(float x, unsigned short (&h)[3]) {
  const __bf16 b0 = (__bf16)x;
  const float r1 = x - (float)b0;
  const __bf16 b1 = (__bf16)r1;
  const __bf16 b2 = (__bf16)(r1 - (float)b1);
  h[0] = __builtin_bit_cast(unsigned short, b0);
  h[1] = __builtin_bit_cast(unsigned short, b1);
  h[2] = __builtin_bit_cast(unsigned short, b2);
}
// element (row, k) term p of a slab-major split matrix [K/16][rows][3][16]
__device__ __forceinline__ size_t split_index(size_t row, int k, size_t rows, int p) {
  return (((size_t)(k >> 4) * rows + row) * 3 + p) * 16 + (k & 15);
}
__device__ __forceinline__ void store_split4(unsigned short* base, size_t row, int k, size_t rows, const f32x4& v) {
  unsigned short h[4][3];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3(v[e], h[e]);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const u16x4 w = {h[0][p], h[1][p], h[2][p], h[3][p]};
    *reinterpret_cast<u16x4*>(base + split_index(row, k, rows, p)) = w;
  }
}

// split 4 values into 3 terms, each packed as two dwords of 2 bf16
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4_packed(const f32x4& v, unsigned (&pk)[3][2]) {
  unsigned short h[4][3];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3(v[e], h[e]);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    pk[p][0] = (unsigned)h[0][p] | ((unsigned)h[1][p] << 16);
    pk[p][1] = (unsigned)h[2][p] | ((unsigned)h[3][p] << 16);
  }
}
// Two outputs (va -> row_a, vb -> row_b) of a lane pair (even / odd lane = adjacent 8-byte pieces of one 16-byte chunk):
// the even lane hands its vb piece to the odd lane and receives the odd lane's va piece (one DPP swap per dword),
// so every lane issues ONE 16-byte store per term instead of two 8-byte stores.  `idx8(row)` -> element index of the
// pair's 16-byte chunk for term 0 (terms are 16 elements apart).
__device__ __forceinline__ void store_split_pair(unsigned short* base, const f32x4& va, const f32x4& vb, bool odd,
                                                 size_t ia, size_t ib) {
  unsigned pa[3][2], pb[3][2];
  split4_packed(va, pa);
  split4_packed(vb, pb);
  const size_t at = odd ? ib : ia;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const unsigned s0 = odd ? pa[p][0] : pb[p][0], s1 = odd ? pa[p][1] : pb[p][1];
    const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
    const u32x4 w = odd ? (u32x4){r0, r1, pb[p][0], pb[p][1]} : (u32x4){pa[p][0], pa[p][1], r0, r1};
    *reinterpret_cast<u32x4*>(base + at + p * 16) = w;
  }
}

// 4 consecutive channels of a Winograd-domain GEMM output: fp32, or scaled fp16 in the half-precision compute mode
// (BASELINE configs[2]; the transforms are linear, so the power-of-two scale is undone once on their result)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <typename TM>
__device__ __forceinline__ f32x4 ldm4(const TM* p) {
  if constexpr (sizeof(TM) == 4) {
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
}

// The same element through a buffer load: the 36 transform planes of M / dV are `plane` elements apart, so a thread needs
// ONE 32-bit offset (tile, column) for all of them and the plane base is scalar (buffer resource built per plane by the
// scalar unit) -- instead of a 64-bit address pair per load, which cost the output transforms ~70 VGPRs and two VALU
// operations per load.  plane * sizeof(TM) < 4 GB (checked by the launchers).
// cache policy of the M / dV plane reads: 2 = nt (streamed once, no reuse: the fused SPADE transform gains 7 %)
#ifndef DSEE_M_AUX
#define DSEE_M_AUX 2
#endif
// activation-sized results leave with non-temporal stores (each is consumed by another kernel after > L2 of other
// traffic; the fused SPADE transform gains 3.5 %)
#ifndef DSEE_NT_STORE
#define DSEE_NT_STORE 1
#endif
#ifndef DSEE_NT_STORE2
#define DSEE_NT_STORE2 1   // output 0.62 -> 0.58 ms, input 0.68 -> 0.64 ms at N = 8, 256^2, C = 512
#endif
__device__ __forceinline__ void st4(float* p, const f32x4& v) {
  if constexpr (DSEE_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void st4b(float* p, const f32x4& v) {
  if constexpr (DSEE_NT_STORE2) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
template <typename TM>
__device__ __forceinline__ f32x4 ldm4b(const TM* M, int xi, size_t plane, unsigned voff) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(M + (size_t)xi * plane), 0, 0xFFFFFFFE, 0x00020000);
  if constexpr (sizeof(TM) == 4) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, DSEE_M_AUX));
  } else {
    const f16x4 h = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, DSEE_M_AUX));
    return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
}

// one column / row of B^T d : 6 -> 6
__device__ __forceinline__ void bt6(const f32x4 (&d)[6], f32x4 (&o)[6]) {
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
  o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
  o[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
  o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// A^T m : 6 -> 4
__device__ __forceinline__ void at4(const f32x4 (&m)[6], f32x4 (&o)[4]) {
  const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + m[5];
}

// Wave-wide linear store of one 16-row x 96-byte block of a split matrix (rows contiguous in memory): every lane drops
// its 4 values x 3 terms (8 bytes per term) into a wave-private 1536-byte LDS image of the block at
// row*96 + term*32 + quad*8 and the wave then writes the image with fully contiguous 16-byte-per-lane stores
// (64 + 32 lanes) -- 1 KB runs per store instruction instead of 32-byte pieces at a 96-byte stride.
__device__ __forceinline__ void store_block_linear(unsigned* sb, unsigned short* gblock, int l, int row, int quad,
                                                   const f32x4& v) {
  unsigned pk[3][2];
  split4_packed(v, pk);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const uint2 w = {pk[p][0], pk[p][1]};
    *reinterpret_cast<uint2*>(sb + row * 24 + p * 8 + quad * 2) = w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const u32x4 a = *reinterpret_cast<const u32x4*>(sb + l * 4);
  u32x4 b = {0u, 0u, 0u, 0u};
  if (l < 32) b = *reinterpret_cast<const u32x4*>(sb + 256 + l * 4);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  *reinterpret_cast<u32x4*>(gblock + l * 8) = a;
  if (l < 32) *reinterpret_cast<u32x4*>(gblock + 512 + l * 8) = b;
}

// Output modes of the transform kernels: fp32 rows | bf16x3 slab-major with k = channel (forward / data-gradient GEMM
// operands) | bf16x3 slab-major with k = tile (weight-gradient operands, [xi][T/16][C][3][16 tiles])
constexpr int OUT_F32 = 0, OUT_SPLIT = 1, OUT_SPLIT_T = 2;

// split modes: a wave = 16 consecutive tiles x one 16-channel slab (T % 16 == 0, C % 16 == 0), lane = (tile l>>2,
// channel quad l&3); its three term stores then fill 1536 contiguous bytes of either split layout.  Waves are
// ordered strip-major: 256 consecutive tile groups (4096 tiles, an L2-sized piece of x) for slab 0, the same tiles
// for slab 1, ... so that every (xi, slab) output stream is written in long sequential runs while x is read from
// HBM once.
__device__ __forceinline__ void wave_tile_quad(long i, int C4, long T, int& q, long& t, long& tg, int& kb) {
  // (32-bit: item counts fit -- the hosts check -- and the ISA has no integer divide, dsee_common.h)
  const unsigned C16 = (unsigned)C4 >> 2;
  const unsigned G16 = (unsigned)(T >> 4);
  const unsigned S = G16 % 256 == 0 ? 256 : G16;
  const unsigned w = (unsigned)(i >> 6);
  const int l = (int)(i & 63);
  const unsigned strip = w / (C16 * S), rem = w - strip * (C16 * S), kbu = rem / S;
  kb = (int)kbu;
  tg = (long)(strip * S + (rem - kbu * S));
  q = kb * 4 + (l & 3);
  t = tg * 16 + (l >> 2);
}

// OUT_SPLIT_T: transpose the wave's 16 tiles x 16 channels through LDS so that lane (channel l>>2, tile quad l&3)
// holds 4 consecutive tiles of one channel, then split and store 8 bytes per term
__device__ __forceinline__ void emit_split_t(float* tb, unsigned* lb, unsigned short* base, int xi, long T, int C, int l,
                                             long tblk, int ch0, const f32x4& o) {
#pragma unroll
  for (int e = 0; e < 4; ++e) tb[(4 * (l & 3) + e) * 20 + (l >> 2)] = o[e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (l >> 2) * 20 + 4 * (l & 3));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // rows = the 16 channels ch0.. of tile block tblk (contiguous 96-byte rows): one linear 1536-byte block
  store_block_linear(lb, base + (((size_t)xi * (T >> 4) + tblk) * C + ch0) * 48, l, l >> 2, l & 3, v);
}

template <int OUT>
__global__ __launch_bounds__(256) void wino43_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N,
                                                           int H, int W, int C, float* __restrict__ amax) {
  float vmax = 0.f;  // OUT_F32 only: max |V| for the fp16x2 GEMM's operand scale
  __shared__ __attribute__((aligned(16))) float tbuf[OUT == OUT_SPLIT_T ? 4 : 1][16 * 20];
  __shared__ __attribute__((aligned(16))) unsigned lbuf[OUT != OUT_F32 ? 4 : 1][384];
  const int C4 = C / 4, th = H / 4, tw = W / 4;
  const long T = (long)N * th * tw, total = T * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q;
    long t;
    long tg = 0;
    int kb = 0;
    if constexpr (OUT == OUT_SPLIT) {
      // k = channel layout: a wave = 8 consecutive tiles x 32 channels (two slabs).  Loads are whole 128-byte lines
      // (8 quads of a pixel), stores two 768-byte runs (8 rows x 96 B of each slab) through the LDS image.
      // Strip-major wave order as below (C % 32 == 0, T % 8 == 0).
      const int C32 = C4 >> 3;
      const long G8 = T >> 3;
      const long S = G8 % 512 == 0 ? 512 : G8;
      const long w = i >> 6;
      const int l = (int)(i & 63);
      const long strip = w / (C32 * S), rem = w - strip * (C32 * S);
      kb = (int)(rem / S);            // 32-channel block
      tg = strip * S + rem % S;       // group of 8 tiles
      q = kb * 8 + (l & 7);
      t = tg * 8 + (l >> 3);
    } else if constexpr (OUT == OUT_SPLIT_T) {
      wave_tile_quad(i, C4, T, q, t, tg, kb);
    } else {
      t = (long)((unsigned)i / (unsigned)C4);
      q = (int)((unsigned)i - (unsigned)t * (unsigned)C4);
    }
    // (32-bit divisions: T < 2^31, and a 64-bit division is ~150 instructions on this ISA -- dsee_common.h)
    const unsigned tu_ = (unsigned)t, r_ = tu_ / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx = (int)(tu_ - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    f32x4 tmp[6][6];  // tmp[row][col] = (B^T d)[row][col]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int xx = tx * 4 - 1 + j;
      f32x4 col[6], o[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int yy = ty * 4 - 1 + k;
        const bool ok = xx >= 0 && xx < W && yy >= 0 && yy < H;
        col[k] = ok ? *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + yy) * W + xx) * C + q * 4)
                    : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      bt6(col, o);
#pragma unroll
      for (int k = 0; k < 6; ++k) tmp[k][j] = o[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      f32x4 o[6];
      bt6(tmp[k], o);  // (B^T d) B : same combination along the row
      if constexpr (OUT == OUT_SPLIT) {
        const int l = (int)(i & 63);
        unsigned* sb = lbuf[threadIdx.x >> 6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          unsigned pk[3][2];
          split4_packed(o[j], pk);
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const uint2 wv = {pk[p][0], pk[p][1]};
            *reinterpret_cast<uint2*>(sb + ((l & 7) >> 2) * 192 + (l >> 3) * 24 + p * 8 + (l & 3) * 2) = wv;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const u32x4 c0 = *reinterpret_cast<const u32x4*>(sb + l * 4);
          u32x4 c1 = {0u, 0u, 0u, 0u};
          if (l < 32) c1 = *reinterpret_cast<const u32x4*>(sb + 256 + l * 4);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // chunk l of the image: run l / 48 (slab 2*kb + run), 16-byte piece l % 48; second pass: chunks 64 + l
          unsigned short* g0 = reinterpret_cast<unsigned short*>(V) +
                               split_index((size_t)(k * 6 + j) * T + tg * 8, kb * 32, (size_t)36 * T, 0);
          unsigned short* g1 = reinterpret_cast<unsigned short*>(V) +
                               split_index((size_t)(k * 6 + j) * T + tg * 8, kb * 32 + 16, (size_t)36 * T, 0);
          *reinterpret_cast<u32x4*>((l < 48 ? g0 + l * 8 : g1 + (l - 48) * 8)) = c0;
          if (l < 32) *reinterpret_cast<u32x4*>(g1 + (l + 16) * 8) = c1;
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if constexpr (OUT == OUT_SPLIT_T)
          emit_split_t(tbuf[threadIdx.x >> 6], lbuf[threadIdx.x >> 6], reinterpret_cast<unsigned short*>(V), k * 6 + j, T, C, (int)(i & 63), tg, kb * 16, o[j]);
        else {
          st4b(V + ((size_t)(k * 6 + j) * T + t) * C + q * 4, o[j]);
          vmax = fmaxf(vmax, dsee_absmax4(o[j]));
        }
      }
    }
  }
  if constexpr (OUT == OUT_F32) {
    if (amax) dsee_block_atomic_absmax(amax, vmax);   // (amax is block-uniform)
  }
}

// The same transform written as the pre-split fp16x2 A operand of the fused SPADE kernel (spade_fused.hip):
// V2 [C/16][36*T][2][16] fp16 -- per 16-channel slab and row (position, tile) 32 bytes of term 0 then 32 bytes of term 1 --
// scaled by dsee_pow2_scale(bound * max|x|).  |B^T d B| <= 100 max|d| (absolute row sums of B^T: 10), so the scale is known
// BEFORE the transform runs: the producer of x supplies max|x| and no second pass over V is needed.  A wave = 16
// consecutive tiles x one slab, lane = (tile l >> 2, channel quad l & 3); lane pairs swap halves so that every lane stores
// 16 contiguous bytes and a wave instruction fills 1 KB (16 rows x 64 B) of the image.
// WIDE (C % 64 == 0): a wave = 4 consecutive tiles x four slabs (64 channels), lane = (tile l >> 4, slab (l >> 2) & 3, quad
// l & 3): the 16 lanes of a tile read 256 contiguous bytes of every pixel (64-byte runs cost the narrow form 20 % of its
// bandwidth) and a store instruction still writes whole 64-byte rows, 256 bytes contiguous per slab.
// PK (16-bit storage mode, opt.precision = "fp16"): the same transform written as the PACKED ONE-TERM operand
// V1 [C/32][36*T][32] fp16 -- the 64-byte row of the image holds 32 channels of one scaled fp16 term instead of 2 terms x 16
// channels (gemm_bf16x3.hip, "packed one-term"): half the bytes per element, same row size, same consumers' DMA streams.
// A lane holds 4 channels = 8 bytes per transform position; the lanes of a pair (adjacent channel quads) exchange halves
// across PAIRS of positions -- the even lane writes 16 bytes of position 2j (its own quad + the partner's), the odd lane 16
// bytes of position 2j + 1 -- so every store is 16 bytes and a (tile, 32-channel slab, position) row is written whole.
// Lane mappings: WIDE (C % 64 == 0) as above, 4 tiles x 64 channels per wave; else (C % 32 == 0) 8 tiles x 32 channels.
template <bool WIDE, bool PK>
__device__ __forceinline__ void f16_lane_map(long w, int l, int nkb, int& kb, long& t) {
  if constexpr (WIDE) {
    kb = (int)(w % nkb) * 4 + ((l >> 2) & 3);
    t = (w / nkb) * 4 + (l >> 4);
  } else if constexpr (PK) {
    kb = (int)(w % nkb) * 2 + ((l >> 2) & 1);
    t = (w / nkb) * 8 + (l >> 3);
  } else {
    kb = (int)(w % nkb);
    t = (w / nkb) * 16 + (l >> 2);
  }
}
template <bool WIDE, bool PK>
__device__ __forceinline__ int f16_nkb(int C) { return WIDE ? C >> 6 : (PK ? C >> 5 : C >> 4); }
template <bool WIDE, bool PK>
__device__ __forceinline__ long f16_waves(long T, int nkb) { return (WIDE ? T >> 2 : (PK ? T >> 3 : T >> 4)) * nkb; }
template <bool WIDE, bool PK = false>
__global__ __launch_bounds__(256) void wino43_input_f16x2_kernel(const float* __restrict__ x, unsigned char* __restrict__ V2,
                                                                 int N, int H, int W, int C,
                                                                 const float* __restrict__ amax, float bound) {
  const float sc = dsee_pow2_scale(bound * dsee_amax_read(amax));
  const int nkb = f16_nkb<WIDE, PK>(C), th = H / 4, tw = W / 4;
  const long T = (long)N * th * tw, total = f16_waves<WIDE, PK>(T, nkb) * 64;
  const size_t slab = (size_t)36 * T * 64;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long w = i >> 6;
    const int l = (int)(i & 63);
    int kb;
    long t;
    f16_lane_map<WIDE, PK>(w, l, nkb, kb, t);
    const int q = kb * 4 + (l & 3);
    // (32-bit divisions: T < 2^31, and a 64-bit division is ~150 instructions on this ISA -- dsee_common.h)
    const unsigned tu_ = (unsigned)t, r_ = tu_ / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx = (int)(tu_ - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    f32x4 tmp[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int xx = tx * 4 - 1 + j;
      f32x4 col[6], o[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int yy = ty * 4 - 1 + k;
        const bool ok = xx >= 0 && xx < W && yy >= 0 && yy < H;
        col[k] = ok ? *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + yy) * W + xx) * C + q * 4)
                    : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      bt6(col, o);
#pragma unroll
      for (int k = 0; k < 6; ++k) tmp[k][j] = o[k];
    }
    const bool odd = (l & 1) != 0;
    if constexpr (PK) {
      unsigned char* rowp = V2 + (size_t)(q >> 3) * slab + (size_t)t * 64 + ((q & 7) & ~1) * 8;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        f32x4 o[6];
        bt6(tmp[k], o);
#pragma unroll
        for (int j = 0; j < 6; j += 2)
          dsee_store_pk_pair(rowp + (size_t)(k * 6 + j) * T * 64, (size_t)T * 64, odd, o[j] * sc, o[j + 1] * sc);
      }
      continue;
    }
    unsigned char* rowp = V2 + (size_t)kb * slab + (size_t)t * 64 + (odd ? 32 + ((l & 3) - 1) * 8 : (l & 3) * 8);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      f32x4 o[6];
      bt6(tmp[k], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        _Float16 h0[4], h1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = o[j][e] * sc;
          h0[e] = (_Float16)v;
          h1[e] = (_Float16)(v - (float)h0[e]);
        }
        auto pk = [](_Float16 a, _Float16 b) {
          return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
        };
        const unsigned p0a = pk(h0[0], h0[1]), p0b = pk(h0[2], h0[3]), p1a = pk(h1[0], h1[1]), p1b = pk(h1[2], h1[3]);
        // the even lane hands its term-1 half to the odd lane and receives the odd lane's term-0 half
        const unsigned sa = odd ? p0a : p1a, sb = odd ? p0b : p1b;
        const unsigned ra = (unsigned)__builtin_amdgcn_mov_dpp((int)sa, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        const unsigned rb = (unsigned)__builtin_amdgcn_mov_dpp((int)sb, 0xB1, 0xF, 0xF, true);
        const u32x4 wv = odd ? (u32x4){ra, rb, p1a, p1b} : (u32x4){p0a, p0b, ra, rb};
        __builtin_nontemporal_store(wv, reinterpret_cast<u32x4*>(rowp + (size_t)(k * 6 + j) * T * 64));   // read next by another kernel
      }
    }
  }
}

// A d : 4 -> 6   (adjoint of at4)
__device__ __forceinline__ void a6(const f32x4 (&d)[4], f32x4 (&o)[6]) {
  const f32x4 s02 = d[0] + d[2], s13 = d[1] + d[3], t02 = d[0] + 4.f * d[2], t13 = 2.f * d[1] + 8.f * d[3];
  o[0] = d[0];
  o[1] = s02 + s13;
  o[2] = s02 - s13;
  o[3] = t02 + t13;
  o[4] = t02 - t13;
  o[5] = d[3];
}

// dM[xi][t][c] = (A dY A^T)[xi] per 4x4 tile of the output gradient (adjoint of the output transform)
// dM = A dY A^T written as the pre-split fp16x2 operand (same image as wino43_input_f16x2_kernel: dM2 [C/16][36*T][2][16] fp16,
// scaled by dsee_pow2_scale(bound * max|dY|) and by the row factors f_i f_j of dsee_common.h -- the absolute row sums of A are
// 1,4,4,15,15,1, those of diag(f) A are <= 1, so |f_i f_j (A dY A^T)[i][j]| <= max|dY| at every position and the scale is fixed by
// the maximum the producer of dY wrote).  Both consumers take it as it is: the adjoint data-gradient
// GEMM as its A operand (dsee_gemm_f16x2_pre), the weight-gradient TN GEMM as its P operand through LDS transpose reads.
// SUMS: bias / noise-weight gradients as in wino43_dout_kernel; a wave keeps its 16-channel slab for the whole loop
// (gridDim.x * 4 is a multiple of C/16), reduces over its 16 tile lanes and writes part[global wave][3][16].
struct DoutSums;
template <bool SUMS, bool WIDE, bool PK = false>
__global__ __launch_bounds__(256) void wino43_dout_f16x2_kernel(const float* __restrict__ dy, unsigned char* __restrict__ dM2,
                                                                int N, int H, int W, int C, const float* __restrict__ amax,
                                                                float bound, float* __restrict__ part, int want_bias,
                                                                int want_n0, int want_n1, uint64_t seed0, uint64_t off0,
                                                                uint64_t seed1, uint64_t off1,
                                                                const uint64_t* __restrict__ epoch) {
  const float sc = dsee_pow2_scale(bound * dsee_amax_read(amax));
  if constexpr (SUMS) {
    if (epoch) {
      off0 += *epoch;
      off1 += *epoch;
    }
  }
  const int nkb = f16_nkb<WIDE, PK>(C), C4 = C / 4, th = H / 4, tw = W / 4;   // channel groups of a wave (64 / 32 / 16 channels)
  const long T = (long)N * th * tw, total = f16_waves<WIDE, PK>(T, nkb) * 64;
  const size_t slab = (size_t)36 * T * 64;
  f32x4 sb = {0.f, 0.f, 0.f, 0.f}, s0 = sb, s1 = sb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long w = i >> 6;
    const int l = (int)(i & 63);
    int kb;
    long t;
    f16_lane_map<WIDE, PK>(w, l, nkb, kb, t);
    const int q = kb * 4 + (l & 3);
    // (32-bit divisions: T < 2^31, and a 64-bit division is ~150 instructions on this ISA -- dsee_common.h)
    const unsigned tu_ = (unsigned)t, r_ = tu_ / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx = (int)(tu_ - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    f32x4 tmp[6][4];  // tmp[row][col] = (A dY)[row][col]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 col[4], o[6];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t px = ((size_t)n * H + ty * 4 + k) * W + tx * 4 + j;
        col[k] = *reinterpret_cast<const f32x4*>(dy + px * C + q * 4);
        if constexpr (SUMS) {
          sb += col[k];
          if (want_n0) s0 += col[k] * philox_normal4(seed0, off0 + (uint64_t)(px * C4 + q));
          if (want_n1) s1 += col[k] * philox_normal4(seed1, off1 + (uint64_t)(px * C4 + q));
        }
      }
      a6(col, o);
#pragma unroll
      for (int k = 0; k < 6; ++k) tmp[k][j] = o[k];
    }
    const bool odd = (l & 1) != 0;
    if constexpr (PK) {
      unsigned char* rowp = dM2 + (size_t)(q >> 3) * slab + (size_t)t * 64 + ((q & 7) & ~1) * 8;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        f32x4 o[6];
        a6(tmp[k], o);
#pragma unroll
        for (int j = 0; j < 6; j += 2)
          dsee_store_pk_pair(rowp + (size_t)(k * 6 + j) * T * 64, (size_t)T * 64, odd, o[j] * (sc * dsee_dm_posf(k * 6 + j)),
                             o[j + 1] * (sc * dsee_dm_posf(k * 6 + j + 1)));      // (row factors: dsee_common.h)
      }
      continue;
    }
    unsigned char* rowp = dM2 + (size_t)kb * slab + (size_t)t * 64 + (odd ? 32 + ((l & 3) - 1) * 8 : (l & 3) * 8);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      f32x4 o[6];
      a6(tmp[k], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        _Float16 h0[4], h1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = o[j][e] * (sc * dsee_dm_posf(k * 6 + j));      // (row factors: dsee_common.h)
          h0[e] = (_Float16)v;
          h1[e] = (_Float16)(v - (float)h0[e]);
        }
        auto pk = [](_Float16 a, _Float16 b) {
          return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
        };
        const unsigned p0a = pk(h0[0], h0[1]), p0b = pk(h0[2], h0[3]), p1a = pk(h1[0], h1[1]), p1b = pk(h1[2], h1[3]);
        const unsigned sa = odd ? p0a : p1a, sbb = odd ? p0b : p1b;
        const unsigned ra = (unsigned)__builtin_amdgcn_mov_dpp((int)sa, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        const unsigned rb = (unsigned)__builtin_amdgcn_mov_dpp((int)sbb, 0xB1, 0xF, 0xF, true);
        const u32x4 wv = odd ? (u32x4){ra, rb, p1a, p1b} : (u32x4){p0a, p0b, ra, rb};
        __builtin_nontemporal_store(wv, reinterpret_cast<u32x4*>(rowp + (size_t)(k * 6 + j) * T * 64));   // read next by another kernel
      }
    }
  }
  if constexpr (SUMS) {
    // fold the tile lanes of the wave (WIDE: lane bits 4, 5; else bits 2..5); the low QL lanes then hold the sums of the
    // wave's channel quads; part[global wave][3][CW channels]
    constexpr int QL = WIDE ? 16 : (PK ? 8 : 4), CW = QL * 4;
#pragma unroll
    for (int o = QL; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sb[e] += __shfl_xor(sb[e], o, 64);
        s0[e] += __shfl_xor(s0[e], o, 64);
        s1[e] += __shfl_xor(s1[e], o, 64);
      }
    const int l = threadIdx.x & 63;
    if (l < QL) {
      const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
      float* o = part + (size_t)gw * 3 * CW + l * 4;
      if (want_bias) *reinterpret_cast<f32x4*>(o) = sb;
      if (want_n0) *reinterpret_cast<f32x4*>(o + CW) = s0;
      if (want_n1) *reinterpret_cast<f32x4*>(o + 2 * CW) = s1;
    }
  }
}

// out_w[c] = sum over the waves that own channel group c / cw (global wave index = group mod C/cw) of part[wave][w][c % cw]
__global__ __launch_bounds__(256) void dout_f16x2_sums_finalize_kernel(const float* __restrict__ part, int waves, int C,
                                                                       int cw, float* __restrict__ o0,
                                                                       float* __restrict__ o1, float* __restrict__ o2) {
  __shared__ float sv[32][8];
  const int w = blockIdx.y;
  float* out = w == 0 ? o0 : (w == 1 ? o1 : o2);
  if (!out) return;   // (block-uniform)
  const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl, nkb = C / cw;
  float v = 0.f;
  if (c < C)
    for (int g = c / cw + lane * nkb; g < waves; g += 32 * nkb) v += part[(size_t)g * 3 * cw + w * cw + c % cw];
  sv[lane][cl] = v;
  __syncthreads();
  if (lane == 0 && c < C) {
    for (int l = 1; l < 32; ++l) v += sv[l][cl];
    out[c] = v;
  }
}

// SUMS: the passes over dY that accompany this one ride along -- the bias gradient sum_px dY and the gradients of up to two
// NoiseInjection weights sum_px dY * eps (eps regenerated from its Philox stream, as wino43_output<true> drew it) are
// accumulated per thread (gridDim.x * 256 is a multiple of C/4: a thread keeps its channel quad for the whole loop), folded
// per block through LDS and written to part[block][3][C]; chdot_finalize sums the blocks in a fixed order.
struct DoutSums {
  float* part;
  int bias, n0, n1;
  uint64_t seed0, off0, seed1, off1;
  const uint64_t* epoch;
};

template <int OUT, bool SUMS = false>
__global__ __launch_bounds__(256) void wino43_dout_kernel(const float* __restrict__ dy, float* __restrict__ dM, int N,
                                                          int H, int W, int C, float* __restrict__ amax, DoutSums sm) {
  float vmax = 0.f;
  f32x4 sb = {0.f, 0.f, 0.f, 0.f}, s0 = sb, s1 = sb;
  if constexpr (SUMS) {
    if (sm.epoch) {
      sm.off0 += *sm.epoch;
      sm.off1 += *sm.epoch;
    }
  }
  __shared__ __attribute__((aligned(16))) float tbuf[OUT == OUT_SPLIT_T ? 4 : 1][16 * 20];
  __shared__ __attribute__((aligned(16))) unsigned lbuf[OUT != OUT_F32 ? 4 : 1][384];
  const int C4 = C / 4, th = H / 4, tw = W / 4;
  const long T = (long)N * th * tw, total = T * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int q, kb = 0;
    long t, tg = 0;
    if constexpr (OUT != OUT_F32) {
      wave_tile_quad(i, C4, T, q, t, tg, kb);
    } else {
      t = (long)((unsigned)i / (unsigned)C4);
      q = (int)((unsigned)i - (unsigned)t * (unsigned)C4);
    }
    // (32-bit divisions: T < 2^31, and a 64-bit division is ~150 instructions on this ISA -- dsee_common.h)
    const unsigned tu_ = (unsigned)t, r_ = tu_ / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx = (int)(tu_ - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    f32x4 tmp[6][4];  // tmp[row][col] = (A dY)[row][col]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 col[4], o[6];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t px = ((size_t)n * H + ty * 4 + k) * W + tx * 4 + j;
        col[k] = *reinterpret_cast<const f32x4*>(dy + px * C + q * 4);
        if constexpr (SUMS) {
          sb += col[k];
          if (sm.n0) s0 += col[k] * philox_normal4(sm.seed0, sm.off0 + (uint64_t)(px * C4 + q));
          if (sm.n1) s1 += col[k] * philox_normal4(sm.seed1, sm.off1 + (uint64_t)(px * C4 + q));
        }
      }
      a6(col, o);
#pragma unroll
      for (int k = 0; k < 6; ++k) tmp[k][j] = o[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      f32x4 o[6];
      a6(tmp[k], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f32x4 of = o[j] * dsee_dm_posf(k * 6 + j);      // (every dM carries the row factors of dsee_common.h)
        if constexpr (OUT == OUT_SPLIT_T)
          emit_split_t(tbuf[threadIdx.x >> 6], lbuf[threadIdx.x >> 6], reinterpret_cast<unsigned short*>(dM), k * 6 + j, T, C, (int)(i & 63), tg, kb * 16, of);
        else {
          st4b(dM + ((size_t)(k * 6 + j) * T + t) * C + q * 4, of);
          vmax = fmaxf(vmax, dsee_absmax4(of));
        }
      }
    }
  }
  if constexpr (OUT == OUT_F32) {
    if (amax) dsee_block_atomic_absmax(amax, vmax);   // (amax is block-uniform)
  }
  if constexpr (SUMS) {
    __shared__ f32x4 red[3][256];
    red[0][threadIdx.x] = sb;
    red[1][threadIdx.x] = s0;
    red[2][threadIdx.x] = s1;
    __syncthreads();
    const int per = 256 / C4 > 0 ? 256 / C4 : 1;   // threads of this block that share a channel quad
    if ((int)threadIdx.x < C4) {
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        if ((w == 0 && !sm.bias) || (w == 1 && !sm.n0) || (w == 2 && !sm.n1)) continue;
        f32x4 v = red[w][threadIdx.x];
        for (int k = 1; k < per; ++k) v += red[w][k * C4 + threadIdx.x];
        *reinterpret_cast<f32x4*>(sm.part + ((size_t)blockIdx.x * 3 + w) * C + threadIdx.x * 4) = v;
      }
    }
  }
}

// out_w[c] = sum_blocks part[block][w][c] in a fixed order (blockIdx.y = w; 8 channels x 32 part-lanes per block)
__global__ __launch_bounds__(256) void dout_sums_finalize_kernel(const float* __restrict__ part, int parts, int C,
                                                                 float* __restrict__ o0, float* __restrict__ o1,
                                                                 float* __restrict__ o2) {
  __shared__ float sv[32][8];
  const int w = blockIdx.y;
  float* out = w == 0 ? o0 : (w == 1 ? o1 : o2);
  if (!out) return;   // (block-uniform)
  const int cl = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  float v = 0.f;
  if (c < C)
    for (int p = lane; p < parts; p += 32) v += part[((size_t)p * 3 + w) * C + c];
  sv[lane][cl] = v;
  __syncthreads();
  if (lane == 0 && c < C) {
    for (int l = 1; l < 32; ++l) v += sv[l][cl];
    out[c] = v;
  }
}

// noise_w != NULL: y += noise_w[c] * eps with eps = element (pixel, quad) of the Philox N(0,1) stream (seed, offset) --
// the NoiseInjection that follows the convolution (architecture.py:111-112 noise_middle) without its own pass over y
template <bool NOISE, typename TM, bool STATS = false>
__global__ __launch_bounds__(256, 2) void wino43_output_kernel(const TM* __restrict__ M, const float* __restrict__ bias,
                                                            const float* __restrict__ res, int res_ld,
                                                            float* __restrict__ y, int N, int H, int W, int C, int act,
                                                            float slope, const float* __restrict__ noise_w,
                                                            uint64_t seed, uint64_t offset,
                                                            const float* __restrict__ res_nw, uint64_t res_seed,
                                                            uint64_t res_offset, const float* __restrict__ mscale,
                                                            const uint64_t* __restrict__ epoch,
                                                            float* __restrict__ stats_part = nullptr) {
  // STATS: shifted sums of what this thread stores (shift = its bias quad: no extra registers for the shift or a counter --
  // the kernel sits at the 256-VGPR edge of two waves per SIMD)
  f32x4 sa0 = {0.f, 0.f, 0.f, 0.f}, sa1 = sa0;
  if constexpr (NOISE) {
    if (epoch) {
      offset += *epoch;
      res_offset += *epoch;
    }
  }
  const float ms = mscale ? *mscale : 1.f;
  const int C4 = C / 4, th = H / 4, tw = W / 4;
  const long T = (long)N * th * tw, total = T * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = (long)((unsigned)i / (unsigned)C4);
    const int q = (int)((unsigned)i - (unsigned)t * (unsigned)C4);
    // (32-bit divisions: T < 2^31, and a 64-bit division is ~150 instructions on this ISA -- dsee_common.h)
    const unsigned tu_ = (unsigned)t, r_ = tu_ / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx = (int)(tu_ - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    f32x4 tmp[4][6];  // tmp[i][col] = (A^T m)[i][col]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x4 col[6], o[4];
#pragma unroll
      for (int k = 0; k < 6; ++k) col[k] = ldm4b(M, k * 6 + j, (size_t)T * C, (unsigned)((t * C + q * 4) * sizeof(TM)));
      at4(col, o);
#pragma unroll
      for (int k = 0; k < 4; ++k) tmp[k][j] = o[k];
    }
    const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 nw = noise_w ? *reinterpret_cast<const f32x4*>(noise_w + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 rnw = res_nw ? *reinterpret_cast<const f32x4*>(res_nw + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 o[4];
      at4(tmp[k], o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t px = ((size_t)n * H + ty * 4 + k) * W + tx * 4 + j, off = px * C + q * 4;
        f32x4 v = o[j] * ms + b;
        if (act == DSEE_ACT_MASK) {  // backward of a ReLU whose output is `res`
          const f32x4 m = *reinterpret_cast<const f32x4*>(res + px * res_ld + q * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
        } else {
          if (res) {
            // the shortcut x_s = noise_skip(x) (architecture.py:133-134,127) regenerated from x: never materialised
            f32x4 rv = *reinterpret_cast<const f32x4*>(res + px * res_ld + q * 4);
            if constexpr (NOISE) {
              if (res_nw) rv += rnw * philox_normal4(res_seed, res_offset + (uint64_t)(px * C4 + q));
            }
            v += rv;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = dsee_act(v[e], act, slope);
        }
        if constexpr (NOISE) {
          if (noise_w) v += nw * philox_normal4(seed, offset + (uint64_t)(px * C4 + q));
        }
        st4b(y + off, v);
        if constexpr (STATS) {
          const f32x4 d = v - b;
          sa0 += d;
          sa1 += d * d;
        }
      }
    }
  }
  if constexpr (STATS) {
    const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    DseeStatsAcc sa;
    sa.shift = bias ? *reinterpret_cast<const f32x4*>(bias + (i0 % C4) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    sa.a0 = sa0;
    sa.a1 = sa1;
    sa.n = i0 < total ? 16.f * (float)((total - i0 + stride - 1) / stride) : 0.f;
    sa.flush(stats_part, C);
  }
}

// 4x4 output tile of one channel quad: y[k][j] = (A^T m A)[k][j], m[xi] read at column `col` of M [36][T][ld]
template <typename TM>
__device__ __forceinline__ void out_tile(const TM* __restrict__ M, long T, long t, int ld, int col, f32x4 (&y)[4][4]) {
  f32x4 tmp[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    f32x4 c6[6], o[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = ldm4b(M, k * 6 + j, (size_t)T * ld, (unsigned)((t * ld + col) * sizeof(TM)));
    at4(c6, o);
#pragma unroll
    for (int k = 0; k < 4; ++k) tmp[k][j] = o[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) at4(tmp[k], y[k]);
}

// Output transform of the gamma/beta GEMM fused with the SPADE/SEAN modulate + LeakyReLU (same arithmetic as the
// EPI_MODULATE epilogue of conv_mfma.hip; packed column of channel c: (c/64)*128 + ((c%64)/32)*64 + c%32, beta +32)
template <typename TM>
__global__ __launch_bounds__(256, 2) void wino43_output_modulate_kernel(
    const TM* __restrict__ M, const float* __restrict__ bias, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ out,
    float* __restrict__ scale, int N, int H, int W, int C, int rows, float add_one, float slope,
    const float* __restrict__ mscale) {
  const float ms = mscale ? *mscale : 1.f;
  const int C4 = C / 4, th = H / 4, tw = W / 4;
  const long T = (long)N * th * tw, total = T * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = (long)((unsigned)i / (unsigned)C4);
    const int q = (int)((unsigned)i - (unsigned)t * (unsigned)C4), c = q * 4;
    // (32-bit divisions: T < 2^31, and a 64-bit division is ~150 instructions on this ISA -- dsee_common.h)
    const unsigned tu_ = (unsigned)t, r_ = tu_ / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx = (int)(tu_ - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    const int pg = (c >> 6) * 128 + ((c & 63) >> 5) * 64 + (c & 31);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bg = bias ? *reinterpret_cast<const f32x4*>(bias + pg) : z4;
    const f32x4 bb = bias ? *reinterpret_cast<const f32x4*>(bias + pg + 32) : z4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    // gamma first: scale = gamma + bias + add_one is stored and folded with x-hat into y = x-hat * scale, then beta.  One
    // 4x4 tile of 16 float4 is live across the second transform instead of two (422 -> < 256 VGPRs: two waves per SIMD,
    // so the load phase of one wave overlaps the store phase of the other).
    f32x4 y[4][4];
    out_tile(M, T, t, rows, pg, y);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t off = (((size_t)n * H + ty * 4 + k) * W + tx * 4 + j) * C + c;
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + off) - mu) * is;
        const f32x4 sc = y[k][j] * ms + bg + add_one;
        if (scale) st4(scale + off, sc);  // saved for the backward pass only
        y[k][j] = xh * sc + bb;
      }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 yb[4][4];
    out_tile(M, T, t, rows, pg + 32, yb);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t off = (((size_t)n * H + ty * 4 + k) * W + tx * 4 + j) * C + c;
        f32x4 v = y[k][j] + yb[k][j] * ms;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        st4(out + off, v);
      }
  }
}

// B v : 6 -> 6   (adjoint of bt6: B = (B^T)^T)
__device__ __forceinline__ void b6(const f32x4 (&v)[6], f32x4 (&o)[6]) {
  o[0] = 4.f * v[0];
  o[1] = 4.f * (v[2] - v[1]) + 2.f * (v[4] - v[3]) + 4.f * v[5];
  o[2] = -5.f * v[0] - 4.f * (v[1] + v[2]) - v[3] - v[4];
  o[3] = v[1] - v[2] + 2.f * (v[3] - v[4]) - 5.f * v[5];
  o[4] = v[0] + v[1] + v[2] + v[3] + v[4];
  o[5] = v[5];
}

// Adjoint of the input transform: the data gradient of a Winograd convolution from the SAME dM = A dY A^T the weight
// gradient uses.  dV[xi][t][ci] = sum_co dM[xi][t][co] U[xi][co][ci] (one GEMM with the forward U transposed) is the
// gradient w.r.t. V = B^T d B, so every tile contributes the 6x6 patch P_t = B dV_t B^T to dx at rows 4ty-1 .. 4ty+4;
// neighbouring patches overlap by two rows / columns.  Gather form: the thread of (tile, channel quad) writes its own
// 4x4 pixels = interior of its own patch + the last patch row / column of the tiles above / left (B's row 5 = e5: only
// dV row / column 5 is needed) + the first patch row / column of the tiles below / right (B's row 0 = 4 e0) + 4 corner
// scalars.  64 instead of 36 loads per thread, the 28 extra ones from rows of the same planes its neighbours just read.
// mask != NULL: dx = mask > 0 ? dx : 0 (ReLU backward of the SPADE embedding, DSEE_ACT_MASK of the conv epilogues).
// dV = dM' U^T with dM' = f_i f_j dM[i][j] (every dM carries the row factors of dsee_common.h): each position is multiplied by
// r_i r_j as it is loaded (exact powers of two).
template <typename TM>
__global__ __launch_bounds__(256) void wino43_input_adjoint_kernel(const TM* __restrict__ dV,
                                                                   const float* __restrict__ mask, int mask_ld,
                                                                   float* __restrict__ dx, int N, int H, int W, int C,
                                                                   const float* __restrict__ mscale,
                                                                   float* __restrict__ amax = nullptr) {
  const float ms = mscale ? *mscale : 1.f;
  float vmax = 0.f;
  const int C4 = C / 4, th = H / 4, tw = W / 4;
  const long T = (long)N * th * tw, total = T * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = (long)((unsigned)i / (unsigned)C4);
    const int q = (int)((unsigned)i - (unsigned)t * (unsigned)C4);
    // (32-bit divisions: T < 2^31, and a 64-bit division is ~150 instructions on this ISA -- dsee_common.h)
    const unsigned tu_ = (unsigned)t, r_ = tu_ / (unsigned)tw, n_ = r_ / (unsigned)th;
    const int tx = (int)(tu_ - r_ * (unsigned)tw), ty = (int)(r_ - n_ * (unsigned)th), n = (int)n_;
    auto at = [&](int xi, long tt) {
      const f32x4 v = ldm4(dV + ((size_t)xi * T + tt) * C + q * 4);
      return v * dsee_dm_posr(xi);
    };
    // own patch: tmp[a][s] = (B dV)[a][s], then P[a][b] = sum_s tmp[a][s] B[b][s]
    f32x4 tmp[6][6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      f32x4 col[6], o[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) col[rr] = at(rr * 6 + s, t);
      b6(col, o);
#pragma unroll
      for (int a = 0; a < 6; ++a) tmp[a][s] = o[a];
    }
    f32x4 y[4][4];
#pragma unroll
    for (int a = 1; a < 5; ++a) {
      f32x4 o[6];
      b6(tmp[a], o);
#pragma unroll
      for (int b = 1; b < 5; ++b) y[a - 1][b - 1] = o[b];
    }
    const bool up = ty > 0, dn = ty + 1 < th, lf = tx > 0, rt = tx + 1 < tw;
    if (up) {  // P_up[5][b] = sum_s dV_up[5][s] B[b][s]
      f32x4 v[6], o[6];
#pragma unroll
      for (int s = 0; s < 6; ++s) v[s] = at(30 + s, t - tw);
      b6(v, o);
#pragma unroll
      for (int b = 1; b < 5; ++b) y[0][b - 1] += o[b];
    }
    if (dn) {  // P_down[0][b] = 4 sum_s dV_down[0][s] B[b][s]
      f32x4 v[6], o[6];
#pragma unroll
      for (int s = 0; s < 6; ++s) v[s] = at(s, t + tw);
      b6(v, o);
#pragma unroll
      for (int b = 1; b < 5; ++b) y[3][b - 1] += 4.f * o[b];
    }
    if (lf) {  // P_left[a][5] = sum_r B[a][r] dV_left[r][5]
      f32x4 v[6], o[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) v[rr] = at(rr * 6 + 5, t - 1);
      b6(v, o);
#pragma unroll
      for (int a = 1; a < 5; ++a) y[a - 1][0] += o[a];
    }
    if (rt) {  // P_right[a][0] = 4 sum_r B[a][r] dV_right[r][0]
      f32x4 v[6], o[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) v[rr] = at(rr * 6, t + 1);
      b6(v, o);
#pragma unroll
      for (int a = 1; a < 5; ++a) y[a - 1][3] += 4.f * o[a];
    }
    if (up && lf) y[0][0] += at(35, t - tw - 1);
    if (up && rt) y[0][3] += 4.f * at(30, t - tw + 1);
    if (dn && lf) y[3][0] += 4.f * at(5, t + tw - 1);
    if (dn && rt) y[3][3] += 16.f * at(0, t + tw + 1);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t px = ((size_t)n * H + ty * 4 + k) * W + tx * 4 + j;
        f32x4 v = y[k][j] * ms;
        if (mask) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(mask + px * mask_ld + q * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
        }
        st4b(dx + px * C + q * 4, v);
        vmax = fmaxf(vmax, dsee_absmax4(v));
      }
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);   // max |dx|: dx is the gradient w.r.t. a norm's output (-> dM2 bound)
}

// G g G^T of one 3x3 filter -> u[36]
__device__ __forceinline__ void ggt(const float (&g)[3][3], float (&u)[36]) {
  float t[6][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float g0 = g[0][b], g1 = g[1][b], g2 = g[2][b];
    t[0][b] = 0.25f * g0;
    t[1][b] = (-1.f / 6.f) * (g0 + g1 + g2);
    t[2][b] = (-1.f / 6.f) * (g0 - g1 + g2);
    t[3][b] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
    t[4][b] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
    t[5][b] = g2;
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const float t0 = t[a][0], t1 = t[a][1], t2 = t[a][2];
    u[a * 6 + 0] = 0.25f * t0;
    u[a * 6 + 1] = (-1.f / 6.f) * (t0 + t1 + t2);
    u[a * 6 + 2] = (-1.f / 6.f) * (t0 - t1 + t2);
    u[a * 6 + 3] = (1.f / 24.f) * t0 + (1.f / 12.f) * t1 + (1.f / 6.f) * t2;
    u[a * 6 + 4] = (1.f / 24.f) * t0 - (1.f / 12.f) * t1 + (1.f / 6.f) * t2;
    u[a * 6 + 5] = t2;
  }
}

// per-image weights of the SEAN gamma/beta GEMM in the Winograd domain:
// U[xi][n][row][k] = G g G^T,  g = w2a[row][k][:][:] (k < ca, shared)  |  table[n][tap][row][k - ca] (one-hot chunk)
// element (row, k) term p of a slab-major fp16x2 matrix [K/16][rows][2][16]
__device__ __forceinline__ size_t split2_index(size_t row, int k, size_t rows, int p) {
  return (((size_t)(k >> 4) * rows + row) * 2 + p) * 16 + (k & 15);
}
__device__ __forceinline__ void split2(float x, _Float16 (&h)[2]) {
  h[0] = (_Float16)x;
  h[1] = (_Float16)(x - (float)h[0]);
}

__global__ void wino43_weight_table_kernel(const float* __restrict__ w2a, const float* __restrict__ table,
                                           float* __restrict__ U, int N, int rows, int ca, int Kpad, int split,
                                           const float* __restrict__ amax) {
  const long per = (long)rows * Kpad, total = (long)N * per;
  const float sc = split >= 2 ? dsee_pow2_scale(dsee_amax_read(amax)) : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kpad);
    const long rr = i / Kpad;
    const int row = (int)(rr % rows), n = (int)(rr / rows);
    float g[3][3], u[36];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float v = 0.f;
      if (k < ca)
        v = w2a[((size_t)row * ca + k) * 9 + tap];
      else if (k < ca + 32)
        v = table[(((size_t)n * 9 + tap) * rows + row) * 32 + (k - ca)];
      g[tap / 3][tap % 3] = v;
    }
    ggt(g, u);
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) {
      if (split == 4) {   // 16-bit storage mode: packed one-term rows [K/32][rows][32] (Kpad % 32 == 0)
        reinterpret_cast<_Float16*>(U)[((size_t)xi * N + n) * per + ((size_t)(k >> 5) * rows + row) * 32 + (k & 31)] =
            (_Float16)(u[xi] * sc);
      } else if (split == 3) {   // half-precision compute mode: [K/16][rows][16], one scaled fp16 term
        reinterpret_cast<_Float16*>(U)[((size_t)xi * N + n) * per + ((size_t)(k >> 4) * rows + row) * 16 + (k & 15)] =
            (_Float16)(u[xi] * sc);
      } else if (split == 2) {
        _Float16 h[2];
        split2(u[xi] * sc, h);
#pragma unroll
        for (int p = 0; p < 2; ++p)
          reinterpret_cast<_Float16*>(U)[((size_t)xi * N + n) * per * 2 + split2_index(row, k, rows, p)] = h[p];
      } else if (split) {
        unsigned short h[3];
        split3(u[xi], h);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          reinterpret_cast<unsigned short*>(U)[((size_t)xi * N + n) * per * 3 + split_index(row, k, rows, p)] = h[p];
      } else {
        U[(size_t)xi * total + i] = u[xi];
      }
    }
  }
}

// U[xi][row][k]: forward  row = co, k = ci, g = w[co][ci][:][:]
//                dgrad    row = ci, k = co, g = rot180(w[co][ci])        (transpose_flip = 1: the data gradient as a
//                                                                         convolution with the rotated kernel)
//                adjoint  row = ci, k = co, g = w[co][ci]                (transpose_flip = 2: the forward U transposed,
//                                                                         for dV = dM x U^T of the adjoint data gradient)
// blockIdx.y = layer of a batch of equally shaped layers (w, U, amax advance by w_stride / u_stride / 2048 floats)
__global__ void wino43_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int rows,
                                     int Kpad, int transpose_flip, int split, const float* __restrict__ amax,
                                     long w_stride, long u_stride) {
  w += (size_t)blockIdx.y * w_stride;
  U += (size_t)blockIdx.y * u_stride;
  if (amax) amax += (size_t)blockIdx.y * (DSEE_AMAX_LINES * DSEE_AMAX_STRIDE);
  const long total = (long)rows * Kpad;
  const float sc = split >= 2 ? dsee_pow2_scale(dsee_amax_read(amax)) : 1.f;
  const int R = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / Kpad), k = (int)(i % Kpad);
    float g[3][3];
    const bool ok = row < R && k < K;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float v = 0.f;
        if (ok) {
          const int co = transpose_flip ? k : row, ci = transpose_flip ? row : k;
          const int kh = transpose_flip == 1 ? 2 - a : a, kw = transpose_flip == 1 ? 2 - b : b;
          v = w[(((size_t)co * Cin + ci) * 3 + kh) * 3 + kw];
        }
        g[a][b] = v;
      }
    // G g : 6x3
    float t[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float g0 = g[0][b], g1 = g[1][b], g2 = g[2][b];
      t[0][b] = 0.25f * g0;
      t[1][b] = (-1.f / 6.f) * (g0 + g1 + g2);
      t[2][b] = (-1.f / 6.f) * (g0 - g1 + g2);
      t[3][b] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
      t[4][b] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
      t[5][b] = g2;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const float t0 = t[a][0], t1 = t[a][1], t2 = t[a][2];
      float u[6];
      u[0] = 0.25f * t0;
      u[1] = (-1.f / 6.f) * (t0 + t1 + t2);
      u[2] = (-1.f / 6.f) * (t0 - t1 + t2);
      u[3] = (1.f / 24.f) * t0 + (1.f / 12.f) * t1 + (1.f / 6.f) * t2;
      u[4] = (1.f / 24.f) * t0 - (1.f / 12.f) * t1 + (1.f / 6.f) * t2;
      u[5] = t2;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        if (split == 4) {
          reinterpret_cast<_Float16*>(U)[(size_t)(a * 6 + b) * total + ((size_t)(k >> 5) * rows + row) * 32 + (k & 31)] =
              (_Float16)(u[b] * sc);
        } else if (split == 3) {
          reinterpret_cast<_Float16*>(U)[(size_t)(a * 6 + b) * total + ((size_t)(k >> 4) * rows + row) * 16 + (k & 15)] =
              (_Float16)(u[b] * sc);
        } else if (split == 2) {
          _Float16 h[2];
          split2(u[b] * sc, h);
#pragma unroll
          for (int p = 0; p < 2; ++p)
            reinterpret_cast<_Float16*>(U)[(size_t)(a * 6 + b) * total * 2 + split2_index(row, k, rows, p)] = h[p];
        } else if (split) {
          unsigned short h[3];
          split3(u[b], h);
#pragma unroll
          for (int p = 0; p < 3; ++p)
            reinterpret_cast<unsigned short*>(U)[(size_t)(a * 6 + b) * total * 3 + split_index(row, k, rows, p)] = h[p];
        } else {
          U[(size_t)(a * 6 + b) * total + i] = u[b];
        }
      }
    }
  }
}

// ---- the same two weight transforms with 8 consecutive k of a row per thread.  The one-k-per-thread kernels above store every
// fp16 term on its own (72 two-byte stores per weight: 2 TB/s of U at best, store-issue-bound); here a thread forms G g G^T for 8
// neighbouring input channels and writes each position's 8 values as ONE 16-byte vector per term (two for fp32).  Same expressions
// in the same order as ggt(): bit-identical U.  split = 0 (fp32), 2 (two-term fp16x2), 3 / 4 (one scaled fp16 term, [K/16] / packed
// [K/32] rows); the 3-term bf16 form (split = 1) stays on the kernels above.
template <typename Store>
__device__ __forceinline__ void ggt8(const float (&g)[8][9], Store&& st) {
  float t[8][6][3];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float g0 = g[j][b], g1 = g[j][3 + b], g2 = g[j][6 + b];
      t[j][0][b] = 0.25f * g0;
      t[j][1][b] = (-1.f / 6.f) * (g0 + g1 + g2);
      t[j][2][b] = (-1.f / 6.f) * (g0 - g1 + g2);
      t[j][3][b] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
      t[j][4][b] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
      t[j][5][b] = g2;
    }
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t0 = t[j][a][0], t1 = t[j][a][1], t2 = t[j][a][2];
        v[j] = b == 0   ? 0.25f * t0
               : b == 1 ? (-1.f / 6.f) * (t0 + t1 + t2)
               : b == 2 ? (-1.f / 6.f) * (t0 - t1 + t2)
               : b == 3 ? (1.f / 24.f) * t0 + (1.f / 12.f) * t1 + (1.f / 6.f) * t2
               : b == 4 ? (1.f / 24.f) * t0 - (1.f / 12.f) * t1 + (1.f / 6.f) * t2
                        : t2;
      }
      st(a * 6 + b, v);
    }
}

// 8 values of one position -> U.  base = first element of the position's [rows][Kpad] block in units of the split's element
// (fp32 / fp16 / 2 x fp16); row, k0 (% 8 == 0) inside it.
__device__ __forceinline__ void store_u8(float* __restrict__ U, int split, size_t block, size_t per, int rows, int Kpad, int row,
                                         int k0, float sc, const float (&v)[8]) {
  if (split == 0) {
    float* o = U + block * per + (size_t)row * Kpad + k0;
    *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    return;
  }
  _Float16 h0[8], h1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * sc;
    h0[j] = (_Float16)x;
    h1[j] = (_Float16)(x - (float)h0[j]);
  }
  auto pk = [](_Float16 a, _Float16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
  };
  const dsee_u32x4 w0 = {pk(h0[0], h0[1]), pk(h0[2], h0[3]), pk(h0[4], h0[5]), pk(h0[6], h0[7])};
  _Float16* Uh = reinterpret_cast<_Float16*>(U);
  if (split == 2) {
    const dsee_u32x4 w1 = {pk(h1[0], h1[1]), pk(h1[2], h1[3]), pk(h1[4], h1[5]), pk(h1[6], h1[7])};
    _Float16* o = Uh + block * per * 2 + (((size_t)(k0 >> 4) * rows + row) * 2) * 16 + (k0 & 15);
    *reinterpret_cast<dsee_u32x4*>(o) = w0;
    *reinterpret_cast<dsee_u32x4*>(o + 16) = w1;
  } else if (split == 3) {
    *reinterpret_cast<dsee_u32x4*>(Uh + block * per + ((size_t)(k0 >> 4) * rows + row) * 16 + (k0 & 15)) = w0;
  } else {
    *reinterpret_cast<dsee_u32x4*>(Uh + block * per + ((size_t)(k0 >> 5) * rows + row) * 32 + (k0 & 31)) = w0;
  }
}

__global__ __launch_bounds__(256) void wino43_weight_table8_kernel(const float* __restrict__ w2a, const float* __restrict__ table,
                                                                   float* __restrict__ U, int N, int rows, int ca, int Kpad,
                                                                   int split, const float* __restrict__ amax) {
  const unsigned K8 = (unsigned)Kpad >> 3;
  const size_t per = (size_t)rows * Kpad;
  const long total = (long)N * rows * K8;
  const float sc = split >= 2 ? dsee_pow2_scale(dsee_amax_read(amax)) : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i, rr = iu / K8, n = rr / (unsigned)rows;
    const int k0 = (int)(iu - rr * K8) * 8, row = (int)(rr - n * (unsigned)rows);
    float g[8][9];
    if (k0 < ca) {          // 8 shared embedding columns: 72 contiguous floats of w2a
      const f32x4* src = reinterpret_cast<const f32x4*>(w2a + ((size_t)row * ca + k0) * 9);
#pragma unroll
      for (int q = 0; q < 18; ++q) {
        const f32x4 v = src[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) g[(q * 4 + e) / 9][(q * 4 + e) % 9] = v[e];
      }
    } else if (k0 < ca + 32) {      // 8 columns of this image's style table: [tap][row][32]
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* src = table + (((size_t)n * 9 + tap) * rows + row) * 32 + (k0 - ca);
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[e][tap] = a[e];
          g[4 + e][tap] = b[e];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) g[j][tap] = 0.f;
    }
    ggt8(g, [&](int xi, const float (&v)[8]) { store_u8(U, split, (size_t)xi * N + n, per, rows, Kpad, row, k0, sc, v); });
  }
}

__global__ __launch_bounds__(256) void wino43_weight8_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin,
                                                             int rows, int Kpad, int transpose_flip, int split,
                                                             const float* __restrict__ amax, long w_stride, long u_stride) {
  w += (size_t)blockIdx.y * w_stride;
  U += (size_t)blockIdx.y * u_stride;
  if (amax) amax += (size_t)blockIdx.y * (DSEE_AMAX_LINES * DSEE_AMAX_STRIDE);
  const unsigned K8 = (unsigned)Kpad >> 3;
  const size_t per = (size_t)rows * Kpad;
  const long total = (long)rows * K8;
  const float sc = split >= 2 ? dsee_pow2_scale(dsee_amax_read(amax)) : 1.f;
  const int R = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i, row = iu / K8;
    const int k0 = (int)(iu - row * K8) * 8;
    float g[8][9];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      const bool ok = (int)row < R && k < K;
      const int co = transpose_flip ? k : (int)row, ci = transpose_flip ? (int)row : k;
      const float* src = w + ((size_t)co * Cin + ci) * 9;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int a = tap / 3, b = tap % 3;
        const int kh = transpose_flip == 1 ? 2 - a : a, kw = transpose_flip == 1 ? 2 - b : b;
        g[j][tap] = ok ? src[kh * 3 + kw] : 0.f;
      }
    }
    ggt8(g, [&](int xi, const float (&v)[8]) { store_u8(U, split, (size_t)xi, per, rows, Kpad, (int)row, k0, sc, v); });
  }
}

// *amax = max(*amax, max |x[0..n)|); 16 bytes per lane, scalar tail
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ amax) {
  float v = 0.f;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    v = fmaxf(v, dsee_absmax4(*reinterpret_cast<const f32x4*>(x + i * 4)));
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) v = fmaxf(v, fabsf(x[i]));
  dsee_block_atomic_absmax(amax, v);
}

inline int wgrid(long n) { return (int)min(16384L, (n + 255) / 256); }
// the transform kernels decompose a 32-bit item index (tile, channel quad)
inline bool wino_items32(int N, int H, int W, int C) { return (long)N * ((H + 3) / 4) * ((W + 3) / 4) * ((C + 3) / 4) < (1L << 32); }

// wino43_weight8_kernel with the weights staged through LDS.  There a lane gathers its 8 x 9 floats with 72 four-byte loads whose
// addresses lie 288 B (forward) or Cin * 36 B (data gradient) apart across the wave: every load instruction touches 64 cache lines
// and the kernel runs at a fifth of the HBM rate its 47 MB deserve.  Here a 128-thread block owns 16 rows x 64 k of one layer,
// copies the weights it needs as contiguous runs (64 ci x 9 floats per co forward, 16 ci x 9 per co transposed) with 16-byte loads
// into a padded LDS tile (row stride = 1 mod 64 banks / 145: conflict-free for the 8 x 9 gather), and every thread then forms the
// same 8 transforms from LDS.  Same expressions, same stores: bit-identical U.  R % 16 == 0, K % 64 == 0, Cin % 4 == 0.
template <bool T>
__global__ __launch_bounds__(128) void wino43_weight8_tiled_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                                   int Cin, int rows, int Kpad, int transpose_flip, int split,
                                                                   const float* __restrict__ amax, long w_stride, long u_stride) {
  constexpr int RUN = T ? 16 * 9 : 64 * 9, NRUN = T ? 64 : 16, LD = RUN + 1;
  __shared__ float tile[NRUN * LD];
  w += (size_t)blockIdx.z * w_stride;
  U += (size_t)blockIdx.z * u_stride;
  if (amax) amax += (size_t)blockIdx.z * (DSEE_AMAX_LINES * DSEE_AMAX_STRIDE);
  const int row0 = blockIdx.y * 16, kb = blockIdx.x * 64;       // rows: co forward, ci transposed; k: the other one
  const size_t per = (size_t)rows * Kpad;
  const float sc = split >= 2 ? dsee_pow2_scale(dsee_amax_read(amax)) : 1.f;
  const int R = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const int tid = threadIdx.x, r = tid >> 3, k0 = kb + (tid & 7) * 8, row = row0 + r;
  if (row0 < R && kb < K) {
    for (int q = tid; q < NRUN * (RUN / 4); q += 128) {
      const int run = q / (RUN / 4), e = (q - run * (RUN / 4)) * 4;
      const int co = T ? kb + run : row0 + run, ci0 = T ? row0 : kb;
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + ((size_t)co * Cin + ci0) * 9 + e);
      float* d = tile + run * LD + e;
      d[0] = v[0];
      d[1] = v[1];
      d[2] = v[2];
      d[3] = v[3];
    }
  }
  __syncthreads();
  float g[8][9];
  const bool ok = row0 < R && kb < K;      // (a tile is all weights or all padding: R % 16 == 0, K % 64 == 0)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float* src = T ? tile + ((tid & 7) * 8 + j) * LD + r * 9 : tile + r * LD + ((tid & 7) * 8 + j) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int a = tap / 3, b = tap % 3;
      const int kh = transpose_flip == 1 ? 2 - a : a, kw = transpose_flip == 1 ? 2 - b : b;
      g[j][tap] = ok ? src[kh * 3 + kw] : 0.f;
    }
  }
  ggt8(g, [&](int xi, const float (&v)[8]) { store_u8(U, split, (size_t)xi, per, rows, Kpad, row, k0, sc, v); });
}

inline bool weight8_tiled_ok(const float* w, int Cout, int Cin, int transpose_flip, int rows, int Kpad, long w_stride) {
  const int R = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  return R % 16 == 0 && K % 64 == 0 && Cin % 4 == 0 && rows % 16 == 0 && Kpad % 64 == 0 && ((uintptr_t)w & 15) == 0 &&
         w_stride % 4 == 0 && rows / 16 < 65536;
}

inline void launch_weight8(const float* w, float* U, int layers, long w_stride, long u_stride, int Cout, int Cin, int rows,
                           int Kpad, int transpose_flip, int split, const float* amax, hipStream_t st) {
  if (weight8_tiled_ok(w, Cout, Cin, transpose_flip, rows, Kpad, w_stride)) {
    const dim3 grid(Kpad / 64, rows / 16, layers);
    if (transpose_flip)
      wino43_weight8_tiled_kernel<true><<<grid, 128, 0, st>>>(w, U, Cout, Cin, rows, Kpad, transpose_flip, split, amax, w_stride,
                                                               u_stride);
    else
      wino43_weight8_tiled_kernel<false><<<grid, 128, 0, st>>>(w, U, Cout, Cin, rows, Kpad, transpose_flip, split, amax, w_stride,
                                                                u_stride);
  } else {
    wino43_weight8_kernel<<<dim3(wgrid((long)rows * Kpad / 8), layers), 256, 0, st>>>(w, U, Cout, Cin, rows, Kpad, transpose_flip,
                                                                                      split, amax, w_stride, u_stride);
  }
}


}  // namespace

extern "C" {

/* *amax = max(*amax, max |x|): operand bound for the fp16x2 scale of a weight tensor (|G g G^T| <= max |g|) */
int dsee_absmax(const float* x, long n, float* amax, hipStream_t st) {
  DSEE_CHECK_ARG(x && amax && n > 0);
  DSEE_CHECK_ARG(((uintptr_t)x & 15) == 0);
  // one atomic per block at most, spread over 64 cache lines: 2048 blocks cost ~32 same-line atomics, and an
  // activation-sized tensor (the direct convolutions' operands) streams at HBM rate instead of from 128 blocks
  absmax_kernel<<<(int)min(2048L, (n / 4 + 1023) / 1024 + 1), 256, 0, st>>>(x, n, amax);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* amax (optional device scalar, zeroed by the caller): receives max |V| (atomic max: order independent) */
int dsee_wino43_input(const float* x, float* V, int N, int H, int W, int C, float* amax, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(x && V && C % 4 == 0 && H % 4 == 0 && W % 4 == 0);
  wino43_input_kernel<OUT_F32><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(x, V, N, H, W, C, amax);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* same transform, output as the pre-split fp16x2 operand of dsee_spade_fused_fwd: V2 [C/16][36*T][2][16] fp16 scaled by
 * dsee_pow2_scale(bound * *amax_x) with amax_x >= max |x| (device maximum, 64-line form) and bound >= 100 */
int dsee_wino43_input_f16x2(const float* x, void* V2, int N, int H, int W, int C, const float* amax_x, float bound,
                            hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(x && V2 && amax_x && C % 16 == 0 && H % 4 == 0 && W % 4 == 0 && bound >= 100.f);
  DSEE_CHECK_ARG(((long)N * (H / 4) * (W / 4)) % 16 == 0);
  if (C % 64 == 0)
    wino43_input_f16x2_kernel<true><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(
        x, reinterpret_cast<unsigned char*>(V2), N, H, W, C, amax_x, bound);
  else
    wino43_input_f16x2_kernel<false><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(
        x, reinterpret_cast<unsigned char*>(V2), N, H, W, C, amax_x, bound);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* 16-bit storage mode: the same transform written as the PACKED ONE-TERM operand V1 [C/32][36*T][32] fp16 (one scaled fp16 term
 * per element, 64-byte rows of 32 channels), scale dsee_pow2_scale(bound * *amax_x).  C % 32 == 0, T % 8 == 0. */
int dsee_wino43_input_f16p(const float* x, void* V1, int N, int H, int W, int C, const float* amax_x, float bound,
                           hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(x && V1 && amax_x && C % 32 == 0 && H % 4 == 0 && W % 4 == 0 && bound >= 100.f);
  DSEE_CHECK_ARG(((long)N * (H / 4) * (W / 4)) % 8 == 0);
  if (C % 64 == 0)
    wino43_input_f16x2_kernel<true, true><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(
        x, reinterpret_cast<unsigned char*>(V1), N, H, W, C, amax_x, bound);
  else
    wino43_input_f16x2_kernel<false, true><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(
        x, reinterpret_cast<unsigned char*>(V1), N, H, W, C, amax_x, bound);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* same transform, output as bf16x3-split rows for dsee_gemm_bf16x3: V3 [C/16][36*T][3][16] bf16 (C % 32 == 0) */
int dsee_wino43_input_split(const float* x, void* V3, int N, int H, int W, int C, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(x && V3 && C % 32 == 0 && H % 4 == 0 && W % 4 == 0 && ((long)N * (H / 4) * (W / 4)) % 8 == 0);
  wino43_input_kernel<OUT_SPLIT><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(
      x, reinterpret_cast<float*>(V3), N, H, W, C, nullptr);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_wino43_dout(const float* dy, float* dM, int N, int H, int W, int C, float* amax, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(dy && dM && C % 4 == 0 && H % 4 == 0 && W % 4 == 0);
  wino43_dout_kernel<OUT_F32><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(dy, dM, N, H, W, C, amax, DoutSums{});
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

constexpr int DOUT_SUMS_GRID = 2048;

size_t dsee_wino43_dout_sums_workspace(int C) { return (size_t)DOUT_SUMS_GRID * 3 * C * sizeof(float); }

/* dsee_wino43_dout + the channel sums that read the same dY (architecture.py:98,111-112,122,127: bias gradient of the
 * convolution, gradients of the NoiseInjection weights behind it): dbias[c] = sum_px dY, dnoise_k[c] = sum_px dY * eps_k with
 * eps_k the Philox N(0,1) stream (seed_k, offset_k) dsee_wino43_output drew in the forward pass.  Any of the three outputs
 * may be NULL.  workspace: dsee_wino43_dout_sums_workspace(C) bytes.  C in {16, 32, ..., 1024} with 256 % (C/4) == 0. */
int dsee_wino43_dout_sums(const float* dy, float* dM, int N, int H, int W, int C, float* amax, float* workspace,
                          float* dbias, float* dnoise0, uint64_t seed0, uint64_t offset0, float* dnoise1, uint64_t seed1,
                          uint64_t offset1, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(dy && dM && workspace && C % 4 == 0 && H % 4 == 0 && W % 4 == 0);
  DSEE_CHECK_ARG(C / 4 <= 256 && 256 % (C / 4) == 0 && (dbias || dnoise0 || dnoise1));
  const int grid = (int)min((long)DOUT_SUMS_GRID, ((long)N * (H / 4) * (W / 4) * (C / 4) + 255) / 256);
  DoutSums sm{workspace, dbias != nullptr, dnoise0 != nullptr, dnoise1 != nullptr, seed0, offset0, seed1, offset1,
              dsee_rng_epoch()};
  wino43_dout_kernel<OUT_F32, true><<<grid, 256, 0, st>>>(dy, dM, N, H, W, C, amax, sm);
  DSEE_LAUNCH_CHECK();
  dout_sums_finalize_kernel<<<dim3(dsee_cdiv(C, 8), 3), 256, 0, st>>>(workspace, grid, C, dbias, dnoise0, dnoise1);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

size_t dsee_wino43_dout_f16x2_workspace(void) { return (size_t)DOUT_SUMS_GRID * 4 * 3 * 64 * sizeof(float); }

/* dM = A dY A^T as the pre-split fp16x2 operand dM2 [C/16][36*T][2][16] (scale dsee_pow2_scale(bound * *amax_dy), bound >=
 * DSEE_WINO_DM_BOUND, amax_dy >= max |dY| written by dY's producer; positions carry the row factors of dsee_common.h) + optionally the channel sums of dsee_wino43_dout_sums.
 * workspace (dsee_wino43_dout_f16x2_workspace bytes) only when a sum is requested.  C % 16 == 0, T % 16 == 0,
 * 2048 % (C/16) == 0. */
}  // extern "C"

namespace {
template <bool PK>
int dout_f16_launch(const float* dy, void* dM2, int N, int H, int W, int C, const float* amax_dy, float bound,
                    float* workspace, float* dbias, float* dnoise0, uint64_t seed0, uint64_t offset0, float* dnoise1,
                    uint64_t seed1, uint64_t offset1, hipStream_t st) {
  DSEE_CHECK_ARG(dy && dM2 && amax_dy && C % (PK ? 32 : 16) == 0 && H % 4 == 0 && W % 4 == 0 && bound >= DSEE_WINO_DM_BOUND);
  const long T = (long)N * (H / 4) * (W / 4);
  DSEE_CHECK_ARG(T % 16 == 0);
  const bool sums = dbias || dnoise0 || dnoise1, wide = C % 64 == 0;
  const int cw = wide ? 64 : (PK ? 32 : 16), nkb = C / cw;
  const long blocks = (T / 16) * (C / 16) / 4 + 1;
  unsigned char* out = reinterpret_cast<unsigned char*>(dM2);
  if (!sums) {
    const int grid = (int)min(16384L, blocks);
    if (wide)
      wino43_dout_f16x2_kernel<false, true, PK><<<grid, 256, 0, st>>>(dy, out, N, H, W, C, amax_dy, bound, nullptr, 0, 0, 0, 0,
                                                                      0, 0, 0, nullptr);
    else
      wino43_dout_f16x2_kernel<false, false, PK><<<grid, 256, 0, st>>>(dy, out, N, H, W, C, amax_dy, bound, nullptr, 0, 0, 0,
                                                                       0, 0, 0, 0, nullptr);
    DSEE_LAUNCH_CHECK();
    return DSEE_OK;
  }
  DSEE_CHECK_ARG(workspace && C / 16 <= 64);
  // a grid whose wave count is a multiple of the channel groups: every wave keeps its channels for its whole loop
  const int m = nkb / (nkb % 4 == 0 ? 4 : (nkb % 2 == 0 ? 2 : 1));
  long grid = min((long)DOUT_SUMS_GRID, blocks) / m * m;
  if (grid < m) grid = m;
  DSEE_CHECK_ARG((grid * 4) % nkb == 0);
  if (wide)
    wino43_dout_f16x2_kernel<true, true, PK><<<(int)grid, 256, 0, st>>>(dy, out, N, H, W, C, amax_dy, bound, workspace,
                                                                        dbias != nullptr, dnoise0 != nullptr,
                                                                        dnoise1 != nullptr, seed0, offset0, seed1, offset1,
                                                                        dsee_rng_epoch());
  else
    wino43_dout_f16x2_kernel<true, false, PK><<<(int)grid, 256, 0, st>>>(dy, out, N, H, W, C, amax_dy, bound, workspace,
                                                                         dbias != nullptr, dnoise0 != nullptr,
                                                                         dnoise1 != nullptr, seed0, offset0, seed1, offset1,
                                                                         dsee_rng_epoch());
  DSEE_LAUNCH_CHECK();
  dout_f16x2_sums_finalize_kernel<<<dim3(dsee_cdiv(C, 8), 3), 256, 0, st>>>(workspace, (int)grid * 4, C, cw, dbias, dnoise0,
                                                                           dnoise1);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}
}  // namespace

extern "C" {

int dsee_wino43_dout_f16x2(const float* dy, void* dM2, int N, int H, int W, int C, const float* amax_dy, float bound,
                           float* workspace, float* dbias, float* dnoise0, uint64_t seed0, uint64_t offset0,
                           float* dnoise1, uint64_t seed1, uint64_t offset1, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  return dout_f16_launch<false>(dy, dM2, N, H, W, C, amax_dy, bound, workspace, dbias, dnoise0, seed0, offset0, dnoise1, seed1,
                                offset1, st);
}

/* 16-bit storage mode: dM = A dY A^T as the PACKED ONE-TERM operand dM1 [C/32][36*T][32] fp16 (+ the same optional channel
 * sums); workspace as dsee_wino43_dout_f16x2.  C % 32 == 0. */
int dsee_wino43_dout_f16p(const float* dy, void* dM1, int N, int H, int W, int C, const float* amax_dy, float bound,
                          float* workspace, float* dbias, float* dnoise0, uint64_t seed0, uint64_t offset0,
                          float* dnoise1, uint64_t seed1, uint64_t offset1, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  return dout_f16_launch<true>(dy, dM1, N, H, W, C, amax_dy, bound, workspace, dbias, dnoise0, seed0, offset0, dnoise1, seed1,
                               offset1, st);
}

/* weight-gradient operands for dsee_gemm_bf16x3_tn: [36][T/16][C][3][16 tiles] bf16 (T % 16 == 0, C % 16 == 0) */
int dsee_wino43_input_split_t(const float* x, void* V3t, int N, int H, int W, int C, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(x && V3t && C % 16 == 0 && H % 4 == 0 && W % 4 == 0 && ((long)N * (H / 4) * (W / 4)) % 16 == 0);
  wino43_input_kernel<OUT_SPLIT_T><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(
      x, reinterpret_cast<float*>(V3t), N, H, W, C, nullptr);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_wino43_dout_split_t(const float* dy, void* dM3t, int N, int H, int W, int C, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(dy && dM3t && C % 16 == 0 && H % 4 == 0 && W % 4 == 0 && ((long)N * (H / 4) * (W / 4)) % 16 == 0);
  wino43_dout_kernel<OUT_SPLIT_T><<<wgrid((long)N * (H / 4) * (W / 4) * (C / 4)), 256, 0, st>>>(
      dy, reinterpret_cast<float*>(dM3t), N, H, W, C, nullptr, DoutSums{});
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* dsee_wino43_output (fp32 M) that also writes the BatchNorm statistics rows of y for the SPADE/SEAN norm that follows
 * (architecture.py:98-113: conv_0 -> noise_middle -> norm_1): stats_part [dsee_stats_part_rows(T * C/4)][3][C].
 * 256 % (C/4) == 0. */
int dsee_wino43_output_stats(const float* M, const float* bias, const float* residual, int residual_ld, float* y, int N,
                             int H, int W, int C, int act, float slope, const float* noise_w, uint64_t noise_seed,
                             uint64_t noise_offset, const float* res_noise_w, uint64_t res_noise_seed,
                             uint64_t res_noise_offset, float* stats_part, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(M && y && stats_part && C % 4 == 0 && H % 4 == 0 && W % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0);
  DSEE_CHECK_ARG(act != DSEE_ACT_MASK);
  DSEE_CHECK_ARG(!residual || (residual_ld >= C && residual_ld % 4 == 0));
  DSEE_CHECK_ARG(!res_noise_w || (residual && residual_ld == C));
  DSEE_CHECK_ARG((long)N * (H / 4) * (W / 4) * C * 4 < 0xFFFFFFF0L);
  const long items = (long)N * (H / 4) * (W / 4) * (C / 4);
  const int grid = (int)min((long)DSEE_STATS_ROWS_MAX, (items + 255) / 256);
  wino43_output_kernel<true, float, true><<<grid, 256, 0, st>>>(M, bias, residual, residual_ld, y, N, H, W, C, act, slope,
                                                                noise_w, noise_seed, noise_offset, res_noise_w,
                                                                res_noise_seed, res_noise_offset, nullptr, dsee_rng_epoch(),
                                                                stats_part);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* dsee_wino43_output_stats on the scaled-fp16 product M16 of the 16-bit storage mode (dsee_gemm_f16p_pre; *mscale undoes its
 * power-of-two scale) */
int dsee_wino43_output_stats_f16(const void* M16, const float* bias, const float* residual, int residual_ld, float* y, int N,
                                 int H, int W, int C, int act, float slope, const float* noise_w, uint64_t noise_seed,
                                 uint64_t noise_offset, const float* res_noise_w, uint64_t res_noise_seed,
                                 uint64_t res_noise_offset, const float* mscale, float* stats_part, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(M16 && y && mscale && stats_part && C % 4 == 0 && H % 4 == 0 && W % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0);
  DSEE_CHECK_ARG(act != DSEE_ACT_MASK);
  DSEE_CHECK_ARG(!residual || (residual_ld >= C && residual_ld % 4 == 0));
  DSEE_CHECK_ARG(!res_noise_w || (residual && residual_ld == C));
  DSEE_CHECK_ARG((long)N * (H / 4) * (W / 4) * C * 2 < 0xFFFFFFF0L);
  const long items = (long)N * (H / 4) * (W / 4) * (C / 4);
  const int grid = (int)min((long)DSEE_STATS_ROWS_MAX, (items + 255) / 256);
  wino43_output_kernel<true, _Float16, true><<<grid, 256, 0, st>>>(reinterpret_cast<const _Float16*>(M16), bias, residual,
                                                                   residual_ld, y, N, H, W, C, act, slope, noise_w, noise_seed,
                                                                   noise_offset, res_noise_w, res_noise_seed, res_noise_offset,
                                                                   mscale, dsee_rng_epoch(), stats_part);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* dx [N][H][W][C] from dV [36][T][C] (see wino43_input_adjoint_kernel); mask [pixels][mask_ld] optional */
int dsee_wino43_input_adjoint(const float* dV, const float* mask, int mask_ld, float* dx, int N, int H, int W, int C,
                              const float* dvscale, float* amax_dx, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(dV && dx && C % 4 == 0 && H % 4 == 0 && W % 4 == 0);
  DSEE_CHECK_ARG(!mask || (mask_ld >= C && mask_ld % 4 == 0));
  const int grid = wgrid((long)N * (H / 4) * (W / 4) * (C / 4));
  const _Float16* dVh = reinterpret_cast<const _Float16*>(dV);
  if (dvscale) wino43_input_adjoint_kernel<<<grid, 256, 0, st>>>(dVh, mask, mask_ld, dx, N, H, W, C, dvscale, amax_dx);
  else wino43_input_adjoint_kernel<<<grid, 256, 0, st>>>(dV, mask, mask_ld, dx, N, H, W, C, nullptr, amax_dx);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* dsee_wino43_input_adjoint (fp32 dV) that also writes max |dx| (64-line form) */
int dsee_wino43_input_adjoint_amax(const float* dV, float* dx, int N, int H, int W, int C, float* amax_dx, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(dV && dx && amax_dx && C % 4 == 0 && H % 4 == 0 && W % 4 == 0);
  const int grid = wgrid((long)N * (H / 4) * (W / 4) * (C / 4));
  wino43_input_adjoint_kernel<<<grid, 256, 0, st>>>(dV, nullptr, 0, dx, N, H, W, C, nullptr, amax_dx);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* ... and from the scaled-fp16 dV16 of the 16-bit storage mode (*dvscale undoes its power-of-two scale) */
int dsee_wino43_input_adjoint_amax_f16(const void* dV16, float* dx, int N, int H, int W, int C, const float* dvscale,
                                       float* amax_dx, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(dV16 && dx && dvscale && amax_dx && C % 4 == 0 && H % 4 == 0 && W % 4 == 0);
  const int grid = wgrid((long)N * (H / 4) * (W / 4) * (C / 4));
  const _Float16* dVh = reinterpret_cast<const _Float16*>(dV16);
  wino43_input_adjoint_kernel<<<grid, 256, 0, st>>>(dVh, nullptr, 0, dx, N, H, W, C, dvscale, amax_dx);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_wino43_output(const float* M, const float* bias, const float* residual, int residual_ld, float* y, int N,
                       int H, int W, int C, int act, float slope, const float* noise_w, uint64_t noise_seed,
                       uint64_t noise_offset, const float* res_noise_w, uint64_t res_noise_seed,
                       uint64_t res_noise_offset, const float* mscale, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(M && y && C % 4 == 0 && H % 4 == 0 && W % 4 == 0);
  DSEE_CHECK_ARG(act != DSEE_ACT_MASK || residual);
  DSEE_CHECK_ARG(!residual || (residual_ld >= C && residual_ld % 4 == 0));
  DSEE_CHECK_ARG(!res_noise_w || (residual && residual_ld == C && act != DSEE_ACT_MASK));
  DSEE_CHECK_ARG((long)N * (H / 4) * (W / 4) * C * 4 < 0xFFFFFFF0L);   // one transform plane is addressed with 32-bit offsets
  const int grid = wgrid((long)N * (H / 4) * (W / 4) * (C / 4));
  const _Float16* Mb = reinterpret_cast<const _Float16*>(M);
  const bool nz = noise_w || res_noise_w;
#define DSEE_OUT(NOISE, PTR)                                                                                       \
  wino43_output_kernel<NOISE><<<grid, 256, 0, st>>>(PTR, bias, residual, residual_ld, y, N, H, W, C, act, slope, \
                                                    noise_w, noise_seed, noise_offset, res_noise_w, res_noise_seed, \
                                                    res_noise_offset, mscale, dsee_rng_epoch())
  if (mscale) {
    if (nz) DSEE_OUT(true, Mb); else DSEE_OUT(false, Mb);
  } else {
    if (nz) DSEE_OUT(true, M); else DSEE_OUT(false, M);
  }
#undef DSEE_OUT
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* U: [36][dsee_conv_wrows(R)][dsee_conv_kpad(1,1,K)] with (R,K) = (Cout,Cin) forward, (Cin,Cout) data gradient */
int dsee_wino43_weights(const float* w_oihw, float* U, int Cout, int Cin, int transpose_flip, int split,
                        const float* amax_w, hipStream_t st) {
  DSEE_CHECK_ARG(w_oihw && U && (split < 2 || amax_w));
  const int R = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const int rows = dsee_conv_wrows(R), Kpad = dsee_conv_kpad(1, 1, (K + 3) / 4 * 4);
  if (split != 1 && Kpad % 8 == 0)
    launch_weight8(w_oihw, U, 1, 0, 0, Cout, Cin, rows, Kpad, transpose_flip, split, amax_w, st);
  else
    wino43_weight_kernel<<<wgrid((long)rows * Kpad), 256, 0, st>>>(w_oihw, U, Cout, Cin, rows, Kpad, transpose_flip,
                                                                   split, amax_w, 0, 0);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* The same for `layers` equally shaped weights in one launch: w_oihw + i * w_stride floats -> U + i * u_stride floats,
 * amax_w + i * 2048 floats (the layout of dsee_spectral_norm_group_fwd's flat output and maxima). */
int dsee_wino43_weights_batch(const float* w_oihw, float* U, int layers, long w_stride, long u_stride, int Cout, int Cin,
                              int transpose_flip, int split, const float* amax_w, hipStream_t st) {
  DSEE_CHECK_ARG(w_oihw && U && layers > 0 && layers < 65536 && (split < 2 || amax_w));
  const int R = transpose_flip ? Cin : Cout, K = transpose_flip ? Cout : Cin;
  const int rows = dsee_conv_wrows(R), Kpad = dsee_conv_kpad(1, 1, (K + 3) / 4 * 4);
  if (split != 1 && Kpad % 8 == 0)
    launch_weight8(w_oihw, U, layers, w_stride, u_stride, Cout, Cin, rows, Kpad, transpose_flip, split, amax_w, st);
  else
    wino43_weight_kernel<<<dim3(wgrid((long)rows * Kpad), layers), 256, 0, st>>>(w_oihw, U, Cout, Cin, rows, Kpad,
                                                                                 transpose_flip, split, amax_w, w_stride,
                                                                                 u_stride);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_wino43_output_modulate(const float* M, const float* bias_packed, const float* x, const float* mean,
                                const float* invstd, float* out_h, float* out_scale, int N, int H, int W, int C,
                                int rows, float add_one, float slope, const float* mscale, hipStream_t st) {
  DSEE_CHECK_ARG(wino_items32(N, H, W, C));      // (32-bit item index in the kernels: dsee_common.h)
  DSEE_CHECK_ARG(M && x && mean && invstd && out_h && C % 64 == 0 && rows == 2 * C);  // out_scale may be NULL
  DSEE_CHECK_ARG(H % 4 == 0 && W % 4 == 0);
  DSEE_CHECK_ARG((long)N * (H / 4) * (W / 4) * rows * 4 < 0xFFFFFFF0L);   // 32-bit offsets within one transform plane
  const int grid = wgrid((long)N * (H / 4) * (W / 4) * (C / 4));
  if (mscale)
    wino43_output_modulate_kernel<<<grid, 256, 0, st>>>(reinterpret_cast<const _Float16*>(M), bias_packed, x, mean, invstd,
                                                        out_h, out_scale, N, H, W, C, rows, add_one, slope, mscale);
  else
    wino43_output_modulate_kernel<<<grid, 256, 0, st>>>(M, bias_packed, x, mean, invstd, out_h, out_scale, N, H, W, C, rows,
                                                        add_one, slope, nullptr);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* U: [36][N][rows][Kpad(ca + 32)], rows % 128 == 0; w2a [rows][ca][3][3] (NULL if ca == 0), table [N][9][rows][32] */
int dsee_wino43_weights_table(const float* w2a, const float* table, float* U, int N, int rows, int ca, int split,
                              const float* amax_w, hipStream_t st) {
  DSEE_CHECK_ARG(table && U && (ca == 0 || w2a) && ca % 32 == 0 && rows % 128 == 0 && (split < 2 || amax_w));
  const int Kpad = dsee_conv_kpad(1, 1, ca + 32);
  if (split != 1 && Kpad % 8 == 0)
    wino43_weight_table8_kernel<<<wgrid((long)N * rows * Kpad / 8), 256, 0, st>>>(w2a, table, U, N, rows, ca, Kpad, split,
                                                                                  amax_w);
  else
    wino43_weight_table_kernel<<<wgrid((long)N * rows * Kpad), 256, 0, st>>>(w2a, table, U, N, rows, ca, Kpad, split,
                                                                             amax_w);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
