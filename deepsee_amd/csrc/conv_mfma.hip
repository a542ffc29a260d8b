// fp32 MFMA implicit-GEMM convolution family for gfx950 (MI355X), NHWC.
//
// Replaces the ATen/cuDNN convolutions the reference dispatches implicitly
// (SURVEY 2.2): resblock convs architecture.py:34-35,98,122; SPADE/SEAN gamma/beta convs
// normalization.py:102-103,153-159; D convs discriminator.py:78-96; E convs encoder.py:83-99,142-158;
// VGG19 convs architecture.py:151-181; stem/to-RGB sr.py:31,56.
//
// One kernel computes   out[m][n] = sum_{tap,c} in[src(m,tap)][c] * W[n][tap][c]   where the
// source-position map  p = o*mul + off + k*kdir  (then  p>>dshift, p>>ups)  covers forward
// convs of any stride/padding, their data-gradients (mul=1, off=+pad, kdir=-1, dshift=log2 stride)
// and convs that read a nearest-x2-upsampled input without materialising it (ups=1).
//
// Tiling (CDNA4, wave64): 256 threads = 4 waves; each wave owns MTxNT 32x32 accumulator tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD, the chip's fp32 matrix peak 157 TF).
// K is walked in 32-float slabs staged global -> VGPR -> LDS (double-buffered, one barrier per
// slab); LDS rows are padded to 36 floats so the ds_read_b128 fragment reads (lane<32: k..k+3,
// lane>=32: k+4..k+7 of an 8-wide k group) are bank-conflict free.
#include <stdlib.h>

#include "dsee_common.h"

namespace {

struct ConvArgs {
  const float* in;
  const float* w;      // packed [wrows][Kpad], k = (kh*KW+kw)*Cin + c
  const float* bias;   // [Cout] or null (packed order for the modulate epilogue)
  const float* res;    // residual [M][res_ld] or null (act == DSEE_ACT_MASK: ReLU mask source instead of addend)
  int res_ld;
  float* out;          // [M][Cout]
  // modulate epilogue (SPADE / SEAN / PureSEAN)
  const float* mx;     // [M][C] tensor being normalised
  const float* mean;   // [C]
  const float* invstd; // [C]
  float* scale_out;    // [M][C] saved for backward
  float add_one;
  int C;               // channels of mx / out in modulate mode
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int KH, KW, Ktot, Kpad;
  int mul, off, kdir, dshift, ups, korder;
  int margin;  // GEO 1: bytes the A buffer base is moved down so that every tap's scalar offset is >= 0
  // SEAN style as a per-image table (GEO 1 only): K slabs kt >= nk_shared read their B rows from
  // wt[n_img][kt - nk_shared][row][32] instead of the shared packed weight (n_img = image of this M tile)
  const float* wt;
  int nk_shared, wt_rows;
  int wstride;  // row length (floats) of the shared packed weight (== Kpad unless a table supplies the K tail)
  long wgroup_stride;  // grouped GEMM: image n of the input uses the weight matrix at w + n*wgroup_stride (0: shared)
  int act;
  float slope;
  int M;
  // fp16x2 operand mode (conv_igemm_kernel<..., F16 = true>): device maxima of the input tensor and of the packed
  // weights in the 64-line layout of dsee_common.h; NULL = exact fp32 MFMA
  const float* amax_a;
  const float* amax_w;
  // optional: receives max |out| (64-line layout, zeroed by the caller) from the plain epilogue -- the operand bound of the next
  // direct layer, which then needs no dsee_absmax pass over this output
  float* amax_out;
};

constexpr int BK = 32;
constexpr int LDK = 36;  // padded LDS row (floats)
constexpr int FLUSH = 8; // K-slabs per partial-accumulator chain (power of two): 256-deep MFMA chains

enum { EPI_PLAIN = 0, EPI_MODULATE = 1 };

// Shared epilogue: `row_of(i, r)` maps accumulator element r of M-subtile i to the output pixel index (or -1).
template <int MT, int NT, int BN, int EPI, typename RowOf>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], int bn, int wn, int lane,
                                              RowOf row_of) {
  if constexpr (EPI == EPI_PLAIN) {
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = bn * BN + wn * NT * 32 + j * 32 + (lane & 31);
      if (col >= a.Cout) continue;
      const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long row = row_of(i, r);
          if (row >= 0) {
            float v = acc[i][j][r] + b;
            if (a.act == DSEE_ACT_MASK) {
              v = a.res[(size_t)row * a.res_ld + col] > 0.f ? v : 0.f;  // backward of a ReLU whose output is `res`
            } else if (a.res) {
              v += a.res[(size_t)row * a.res_ld + col];
            }
            v = dsee_act(v, a.act, a.slope);
            a.out[(size_t)row * a.Cout + col] = v;
            vmax = fmaxf(vmax, fabsf(v));
          }
        }
    }
    if (a.amax_out) dsee_wave_atomic_absmax(a.amax_out, vmax);
  } else {
    static_assert(EPI != EPI_MODULATE || NT == 2, "modulate pairs gamma/beta tiles");
    // tile j=0 holds (scale-ish) gamma, j=1 holds beta of channel c for the same rows.
    const int c = bn * (BN / 2) + wn * 32 + (lane & 31);
    if (c < a.C) {
      const int pcol = bn * BN + wn * 64 + (lane & 31);
      const float bg = a.bias ? a.bias[pcol] : 0.f;
      const float bb = a.bias ? a.bias[pcol + 32] : 0.f;
      const float mu = a.mean[c], is = a.invstd[c];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long row = row_of(i, r);
          if (row >= 0) {
            const size_t o = (size_t)row * a.C + c;
            const float xh = (a.mx[o] - mu) * is;
            const float sc = acc[i][0][r] + bg + a.add_one;
            const float v = xh * sc + (acc[i][1][r] + bb);
            a.scale_out[o] = sc;
            a.out[o] = v > 0.f ? v : v * a.slope;
          }
        }
    }
  }
}

// GEO = 1: source pixel is an affine function of the output pixel (dshift == 0, ups == 0) and Cin % 32 == 0, so the
//          per-slab address work is a handful of selects; GEO = 0: general geometry (strided dgrads, fused upsample,
//          odd channel counts) with a division per slab.  Both are branch-free inside the K loop so that the address
//          arithmetic and the global loads of slab kt+1 interleave with the MFMAs of slab kt (the matrix pipe takes a
//          new MFMA only every 64 cycles per wave; everything else issues in its shadow).
// F16: the two operands are scaled by powers of two from their device-side maxima and split into two fp16 terms when a
//      slab is stored to LDS (row = [32 k of term 0][32 k of term 1][16 B pad], the same 144 bytes as an fp32 row);
//      6 v_mfma_f32_32x32x16_f16 (48 passes) replace the 16 v_mfma_f32_32x32x2_f32 (256 passes) of a 32-k slab per
//      accumulator tile -- the same arithmetic as the Winograd-domain GEMMs (gemm_bf16x3.hip), as accurate as an sgemm.
template <int MT, int NT, int WM, int WN, int EPI, int GEO, bool F16 = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int BM = MT * WM * 32, BN = NT * WN * 32;
  constexpr int NTHR = WM * WN * 64, RPP = NTHR / 8;  // threads, LDS rows filled per pass
  constexpr int A_CH = BM * 8 / NTHR, B_CH = BN * 8 / NTHR;
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  static_assert(A_CH >= 1 && B_CH >= 1, "at least one chunk per thread");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order.  Hardware places workgroup b on XCD b % 8 (speed only, never correctness): give every XCD
  // a contiguous range of logical tiles and walk the N tiles of one M tile back to back, so the (up to 4) blocks that
  // read the same A rows, and the vertically adjacent M tiles that share halo rows, meet in one XCD's 4 MB L2.
  int bm, bn;
  {
    const int nbn = gridDim.y, total = gridDim.x * nbn;
    const int b = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, idx = b >> 3;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bn = l % nbn;
    bm = l / nbn;
  }
  const int chunk = tid & 7, lrow = tid >> 3;
  const int Hl = a.Hi << a.ups, Wl = a.Wi << a.ups;
  const int dmask = (1 << a.dshift) - 1;
  const int ntaps = a.KH * a.KW;

  // per-row state
  int a_n[A_CH], a_oh[A_CH], a_ow[A_CH];
  bool a_ok[A_CH];
  unsigned a_voff[A_CH];  // GEO 1: byte offset of the row's centre pixel (oh*mul, ow*mul) + this thread's 16-B chunk
  unsigned a_mask[A_CH];  // GEO 1: bit t = tap t reads inside the image
#pragma unroll
  for (int j = 0; j < A_CH; ++j) {
    const int m = bm * BM + lrow + RPP * j;
    a_ok[j] = m < a.M;
    const int mm = a_ok[j] ? m : 0;
    const int ow = mm % a.Wo;
    const int t = mm / a.Wo;
    a_ow[j] = ow;
    a_oh[j] = t % a.Ho;
    a_n[j] = t / a.Ho;
    if constexpr (GEO == 1) {
      const int h0 = a_oh[j] * a.mul + a.off, w0 = a_ow[j] * a.mul + a.off;
      a_voff[j] = (unsigned)((((size_t)(a_n[j] * a.Hi + a_oh[j] * a.mul) * a.Wi + a_ow[j] * a.mul) * a.Cin + chunk * 4) * 4);
      unsigned msk = 0;
      for (int t2 = 0; t2 < ntaps; ++t2) {
        const int ph = h0 + (t2 / a.KW) * a.kdir, pw = w0 + (t2 % a.KW) * a.kdir;
        if (a_ok[j] && ph >= 0 && ph < a.Hi && pw >= 0 && pw < a.Wi) msk |= 1u << t2;
      }
      a_mask[j] = msk;
    }
  }
  // GEO 1 walks K chunk-major (korder 1): slab kt = (32-channel chunk kt / ntaps, tap kt % ntaps), so the 9 taps of a
  // channel chunk re-read the same input pixels in consecutive slabs (L1/L2 hits) instead of 16 slabs apart.
  // Its loads are buffer loads: per-row VGPR offset (constant for the whole kernel) + one SCALAR offset per slab
  // (tap shift + channel chunk), out-of-image taps get offset 0xFFFFFFFF >= num_records and come back as zeros from
  // the hardware range check.  Address work per slab: ~3 VALU per row instead of 64-bit pointer arithmetic.
  int g_tap = 0, g_kh = 0, g_kw = 0, g_cc = 0;
  const float* wrow[B_CH];
  unsigned b_voff[B_CH];
#pragma unroll
  for (int j = 0; j < B_CH; ++j) {
    wrow[j] = a.w + (size_t)(bn * BN + lrow + RPP * j) * a.wstride + chunk * 4;
    b_voff[j] = (unsigned)(((size_t)(lrow + RPP * j) * a.wstride + chunk * 4) * 4);
  }
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(a.in) - a.margin), 0, 0xFFFFFFFE, 0x00020000);
  const int n_img = (bm * BM) / (a.Ho * a.Wo);  // used by the per-image table / grouped-GEMM modes only
  __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.w + (size_t)bn * BN * a.wstride + (size_t)n_img * a.wgroup_stride), 0, 0xFFFFFFFE, 0x00020000);
  // per-image table rows are 32 floats long; this block's rows start at bn*BN
  __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((a.wt ? a.wt : a.w) + (size_t)bn * BN * BK), 0, 0xFFFFFFFE, 0x00020000);
  unsigned t_voff[B_CH];
#pragma unroll
  for (int j = 0; j < B_CH; ++j) t_voff[j] = (unsigned)(((lrow + RPP * j) * BK + chunk * 4) * 4);
  const int nk = a.Kpad / BK;

  // Loads are unconditional (masked rows read element 0 of the tensor); the zero fill is applied when the slab
  // is written to LDS, so nothing in the load section depends on a load result.
  f32x4 ra[A_CH], rb[B_CH];
  float ra_keep[A_CH];
  auto load_tile = [&](int kt) {
    if constexpr (GEO == 1) {
      const int s_tap = __builtin_amdgcn_readfirstlane(g_tap);
      const bool kok = __builtin_amdgcn_readfirstlane(g_cc) * BK < a.Cin;
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
          ((a.off + g_kh * a.kdir) * a.Wi + (a.off + g_kw * a.kdir)) * a.Cin * 4 + a.margin + g_cc * (BK * 4));
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        const bool ok = kok && ((a_mask[j] >> s_tap) & 1u);
        ra[j] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? a_voff[j] : 0xFFFFFFFFu, soff, 0));
      }
      if constexpr (EPI == EPI_MODULATE) {
        // per-image style table for the K tail (block-uniform switch; one load instruction either way)
        const bool sty = a.wt != nullptr && kt >= a.nk_shared;
        const unsigned soff_b = (unsigned)__builtin_amdgcn_readfirstlane(
            sty ? ((n_img * (nk - a.nk_shared) + (kt - a.nk_shared)) * a.wt_rows) * (BK * 4) : kt * (BK * 4));
        const __amdgpu_buffer_rsrc_t rs = sty ? rsrc_t : rsrc_b;
#pragma unroll
        for (int j = 0; j < B_CH; ++j)
          rb[j] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, sty ? t_voff[j] : b_voff[j], soff_b, 0));
      } else {
        const unsigned soff_b = (unsigned)__builtin_amdgcn_readfirstlane(kt * (BK * 4));
#pragma unroll
        for (int j = 0; j < B_CH; ++j)
          rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_voff[j], soff_b, 0));
      }
      // advance to the next slab (scalar selects only): next tap, then next channel chunk
      g_tap += 1;
      g_kw += 1;
      const int kwrap = g_kw == a.KW;
      g_kw = kwrap ? 0 : g_kw;
      g_kh += kwrap;
      const int twrap = g_tap == ntaps;
      g_tap = twrap ? 0 : g_tap;
      g_kh = twrap ? 0 : g_kh;
      g_cc += twrap;
      return;
    } else {
      const int k = kt * BK + chunk * 4;
      const int tap = k / a.Cin;
      const int c = k - tap * a.Cin;
      const int kh = tap / a.KW;
      const int kw = tap - kh * a.KW;
      const bool kok = k < a.Ktot;
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        int ph = a_oh[j] * a.mul + a.off + kh * a.kdir;
        int pw = a_ow[j] * a.mul + a.off + kw * a.kdir;
        bool ok = a_ok[j] && kok && ph >= 0 && pw >= 0 && ((ph | pw) & dmask) == 0;
        ph >>= a.dshift;
        pw >>= a.dshift;
        ok = ok && ph < Hl && pw < Wl;
        ph >>= a.ups;
        pw >>= a.ups;
        const size_t o = ok ? ((size_t)(a_n[j] * a.Hi + ph) * a.Wi + pw) * a.Cin + c : 0;
        ra[j] = *reinterpret_cast<const f32x4*>(a.in + o);
        ra_keep[j] = ok ? 1.f : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wrow[j] + (size_t)kt * BK);
  };
  float sc_a = 1.f, sc_w = 1.f;
  if constexpr (F16) {
    sc_a = dsee_pow2_scale(dsee_amax_read(a.amax_a));
    sc_w = dsee_pow2_scale(dsee_amax_read(a.amax_w));
  }
  // 4 consecutive k of one row -> 4 halfs of each term, at byte 8 * chunk of the row's term-0 / term-1 half
  auto store_split = [&](float* row, const f32x4& v, float sc) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 h0, h1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e] * sc;
      h0[e] = (_Float16)x;
      h1[e] = (_Float16)(x - (float)h0[e]);
    }
    unsigned char* p = reinterpret_cast<unsigned char*>(row) + chunk * 8;
    *reinterpret_cast<h4*>(p) = h0;
    *reinterpret_cast<h4*>(p + 64) = h1;
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * BM * LDK;
    float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      // GEO 1: the buffer range check already returned zeros.  GEO 0: select (not multiply: a masked row may have
      // read Inf/NaN from an unrelated element)
      const f32x4 v = (GEO == 1 || ra_keep[j] != 0.f) ? ra[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (F16) store_split(Ab + (lrow + RPP * j) * LDK, v, sc_a);
      else *reinterpret_cast<f32x4*>(Ab + (lrow + RPP * j) * LDK + chunk * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      if constexpr (F16) store_split(Bb + (lrow + RPP * j) * LDK, rb[j], sc_w);
      else *reinterpret_cast<f32x4*>(Bb + (lrow + RPP * j) * LDK + chunk * 4) = rb[j];
    }
  };

  // Two-level accumulation: the MFMA chain runs over at most FLUSH*32 k's into `part`, which is then folded
  // into `acc`.  A single fp32 chain over K = 4608 has ~eps*sqrt(K) relative error; this brings it to
  // ~eps*(sqrt(128)+sqrt(K/128)), the same class as a blocked CPU GEMM (see tests/test_gpu_model.py).
  f32x16 acc[MT][NT], part[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = part[i][j][r] = 0.f;

  if constexpr (F16) {
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    load_tile(min(1, nk - 1));   // staging registers hold slab 1
    // fragment of k-step s (16 k), term p of the 32-row tile at `row0`: 16 bytes at p * 64 + s * 32 + (lane >> 5) * 16
    const int fo = (lane & 31) * LDK * 4 + (lane >> 5) * 16;
    auto frag = [&](const float* base, int row0, int s16, int p) {
      return *reinterpret_cast<const u32x4v*>(reinterpret_cast<const unsigned char*>(base + row0 * LDK) + fo + p * 64 +
                                              s16 * 32);
    };
    for (int kt = 0; kt < nk; ++kt) {
      const float* Ac = As + cur * BM * LDK;
      const float* Bc = Bs + cur * BN * LDK;
#pragma unroll
      for (int s16 = 0; s16 < 2; ++s16) {
        u32x4v af[MT][2], bf[NT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int p = 0; p < 2; ++p) af[i][p] = frag(Ac, wm * MT * 32 + i * 32, s16, p);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int p = 0; p < 2; ++p) bf[j][p] = frag(Bc, wn * NT * 32 + j * 32, s16, p);
#pragma unroll
        for (int q = 0; q < 3; ++q)   // a1*b0, a0*b1, a0*b0 (smallest first)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              part[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                  __builtin_bit_cast(f16x8v, af[i][q == 0 ? 1 : 0]), __builtin_bit_cast(f16x8v, bf[j][q == 1 ? 1 : 0]),
                  part[i][j], 0, 0, 0);
        if (s16 == 0) {
          __builtin_amdgcn_sched_barrier(0);
          store_tile(cur ^ 1);   // slab kt+1 (loaded a whole slab ago) into the buffer every wave left before the last barrier
        }
      }
      __syncthreads();
      load_tile(min(kt + 2, nk - 1));
      if ((kt & (FLUSH - 1)) == FLUSH - 1 || kt + 1 == nk) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[i][j] += part[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
          }
      }
      cur ^= 1;
    }
    const float oscale = 1.f / (sc_a * sc_w);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= oscale;
  } else {
  load_tile(0);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const int a_frag = (wm * MT * 32 + frow) * LDK + fk, b_frag = (wn * NT * 32 + frow) * LDK + fk;
  // Fragment registers are double-buffered so that the K loop is software-pipelined ACROSS the barrier: the last
  // MFMA group of slab kt (operands already in registers) executes while the block synchronises and while the first
  // fragments of slab kt+1 come back from LDS, instead of leaving the matrix pipe empty for barrier + ds_read latency.
  f32x4 af[2][MT], bf[2][NT];
  auto read_frags = [&](int buf, int kk, int set) {
    const float* Ac = As + buf * BM * LDK + a_frag + kk * 8;
    const float* Bc = Bs + buf * BN * LDK + b_frag + kk * 8;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[set][i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDK);
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[set][j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDK);
  };
  auto mma_group = [&](int set) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          part[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][q], bf[set][j][q], part[i][j], 0, 0, 0);
  };
  load_tile(min(1, nk - 1));  // staging registers now hold slab 1
  read_frags(0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    read_frags(cur, 1, 1);
    mma_group(0);
    read_frags(cur, 2, 0);
    mma_group(1);
    read_frags(cur, 3, 1);
    mma_group(0);
    __builtin_amdgcn_sched_barrier(0);  // first use of slab kt+1's loads (issued one whole slab = 4096 MFMA cycles ago)
    store_tile(cur ^ 1);
    __syncthreads();
    read_frags(cur ^ 1, 0, 0);          // next slab's first fragments ...
    // ... and the global loads of slab kt+2 into the staging registers that were just drained: both issue under the
    // last MFMA group of this slab.  Branch-free: past the end the final slab is simply fetched again.
    load_tile(min(kt + 2, nk - 1));
    mma_group(1);
    __builtin_amdgcn_sched_barrier(0);
    if ((kt & (FLUSH - 1)) == FLUSH - 1 || kt + 1 == nk) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] += part[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
        }
    }
    cur ^= 1;
  }

  }

  // ---- epilogue.  C/D map of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const int rbase = bm * BM + wm * MT * 32 + 4 * (lane >> 5);
  conv_epilogue<MT, NT, BN, EPI>(a, acc, bn, wn, lane, [&](int i, int r) {
    const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
    return row < a.M ? (long)row : -1L;
  });
}

// ---------------------------------------------------------------- 3x3 stride-1 "same" conv with an LDS-staged halo
// The hot layers (512->512 resblock convs, gamma/beta GEMM, their data gradients, VGG) are 3x3 / stride 1 / same
// size.  An M tile is an 8x16 pixel patch; for every 32-channel chunk its 10x18 halo is staged in LDS ONCE and the 9
// taps read their A fragments from it at fixed offsets, so the A operand costs 23 KB of L2->LDS traffic per 9 K-slabs
// instead of 144 KB and no per-slab address arithmetic at all; only the weight slab (16 KB) is fetched per tap.
// Measured motivation (tools/bench_conv_one.py ablations): with either operand's global traffic removed the same
// loop runs at 95 % of the fp32 MFMA peak, with both present at 84 %.
// WM = 2: 8x16 patch, 4 waves, 2 blocks/CU (shipped).  WM = 4 (16x16 patch, 8 waves, 1 block/CU, half the weight
// traffic) measured the same 86 % of peak, so the extra instantiation is not built.
constexpr int HALO_W = 18;

template <int EPI, int WM>
__global__ __launch_bounds__(WM * 128, 2) void conv_halo_kernel(ConvArgs a) {
  constexpr int MT = 2, NT = 2, BN = 128, NTHR = WM * 128, B_CH = BN * 8 / NTHR, RPP = NTHR / 8;
  constexpr int PH = WM * 4, HALO_PX = HALO_W * (PH + 2), HALO_CH = (HALO_PX * 8 + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ah = smem;                    // [180][LDK]
  float* Bs = smem + HALO_PX * LDK;    // [2][BN][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // wm in [0, WM)
  int bm, bn;
  {
    const int nbn = gridDim.y, total = gridDim.x * nbn;
    const int b = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, idx = b >> 3;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bn = l % nbn;
    bm = l / nbn;
  }
  const int txn = a.Wo / 16, tyn = a.Ho / PH;
  const int tx = bm % txn, ty = (bm / txn) % tyn, n_img = bm / (txn * tyn);
  const int chunk = tid & 7, lrow = tid >> 3;
  const int ntaps = 9, ncc = a.Cin / 32, nk = ncc * ntaps;

  // halo element e = tid + 256*j -> pixel e>>3, 16-B chunk e&7; fixed byte offset (or the out-of-range sentinel)
  unsigned h_voff[HALO_CH];
  int h_lds[HALO_CH];
#pragma unroll
  for (int j = 0; j < HALO_CH; ++j) {
    const int e = tid + NTHR * j, hp = e >> 3, hc = e & 7;
    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
    const int y = ty * PH + hy - 1, x = tx * 16 + hx - 1;
    const bool ok = e < HALO_PX * 8 && y >= 0 && y < a.Hi && x >= 0 && x < a.Wi;
    h_voff[j] = ok ? (unsigned)((((size_t)(n_img * a.Hi + y) * a.Wi + x) * a.Cin + hc * 4) * 4) : 0xFFFFFFFFu;
    h_lds[j] = e < HALO_PX * 8 ? hp * LDK + hc * 4 : -1;
  }
  unsigned b_voff[B_CH], t_voff[B_CH];
#pragma unroll
  for (int j = 0; j < B_CH; ++j) {
    b_voff[j] = (unsigned)(((size_t)(lrow + RPP * j) * a.wstride + chunk * 4) * 4);
    t_voff[j] = (unsigned)(((lrow + RPP * j) * BK + chunk * 4) * 4);
  }
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0xFFFFFFFE, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.w + (size_t)bn * BN * a.wstride), 0, 0xFFFFFFFE, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((a.wt ? a.wt : a.w) + (size_t)bn * BN * BK), 0, 0xFFFFFFFE, 0x00020000);

  f32x4 hreg[HALO_CH], rb[B_CH];
  auto load_halo = [&](int cc, bool live) {
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(cc * (BK * 4));
#pragma unroll
    for (int j = 0; j < HALO_CH; ++j)
      hreg[j] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, live ? h_voff[j] : 0xFFFFFFFFu, so, 0));
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int j = 0; j < HALO_CH; ++j)
      if (h_lds[j] >= 0) *reinterpret_cast<f32x4*>(Ah + h_lds[j]) = hreg[j];
  };
  auto load_b = [&](int kt) {
    if constexpr (EPI == EPI_MODULATE) {
      const bool sty = a.wt != nullptr && kt >= a.nk_shared;
      const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(
          sty ? ((n_img * (nk - a.nk_shared) + (kt - a.nk_shared)) * a.wt_rows) * (BK * 4) : kt * (BK * 4));
      const __amdgpu_buffer_rsrc_t rs = sty ? rsrc_t : rsrc_b;
#pragma unroll
      for (int j = 0; j < B_CH; ++j)
        rb[j] = __builtin_bit_cast(f32x4,
                                   __builtin_amdgcn_raw_buffer_load_b128(rs, sty ? t_voff[j] : b_voff[j], so, 0));
    } else {
      const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(kt * (BK * 4));
#pragma unroll
      for (int j = 0; j < B_CH; ++j)
        rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_voff[j], so, 0));
    }
  };
  auto store_b = [&](int buf) {
    float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < B_CH; ++j) *reinterpret_cast<f32x4*>(Bb + (lrow + RPP * j) * LDK + chunk * 4) = rb[j];
  };

  f32x16 acc[MT][NT], part[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = part[i][j][r] = 0.f;

  const int frow = lane & 31, fk = (lane >> 5) * 4;
  int a_frag[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_frag[i] = ((wm * 4 + i * 2 + (frow >> 4)) * HALO_W + (frow & 15)) * LDK + fk;
  const int b_frag = (wn * NT * 32 + frow) * LDK + fk;
  // halo offset of tap (kh,kw): rows/cols 1 + off + k*kdir in {0,1,2}
  auto tap_off = [&](int tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    return ((1 + a.off + kh * a.kdir) * HALO_W + (1 + a.off + kw * a.kdir)) * LDK;
  };
  f32x4 af[2][MT], bf[2][NT];
  auto read_frags = [&](int toff, int buf, int kk, int set) {
#pragma unroll
    for (int i = 0; i < MT; ++i) af[set][i] = *reinterpret_cast<const f32x4*>(Ah + a_frag[i] + toff + kk * 8);
    const float* Bc = Bs + buf * BN * LDK + b_frag + kk * 8;
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[set][j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDK);
  };
  auto mma_group = [&](int set) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          part[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][q], bf[set][j][q], part[i][j], 0, 0, 0);
  };
  auto flush = [&]() {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] += part[i][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
      }
  };

  load_halo(0, true);
  load_b(0);
  int cur = 0;
  for (int cc = 0; cc < ncc; ++cc) {
    __syncthreads();  // nobody reads the previous chunk's halo / weight slabs any more
    store_halo();
    store_b(cur);
    __syncthreads();
    load_halo(min(cc + 1, ncc - 1), cc + 1 < ncc);  // next chunk's halo: 9 slabs of MFMAs cover it
    load_b(min(cc * ntaps + 1, nk - 1));
    read_frags(tap_off(0), cur, 0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kt = cc * ntaps + tap;
      const int toff = tap_off(tap);
      read_frags(toff, cur, 1, 1);
      mma_group(0);
      read_frags(toff, cur, 2, 0);
      mma_group(1);
      read_frags(toff, cur, 3, 1);
      mma_group(0);
      if (tap < 8) {
        __builtin_amdgcn_sched_barrier(0);
        store_b(cur ^ 1);
        __syncthreads();
        read_frags(tap_off(tap + 1), cur ^ 1, 0, 0);
        load_b(min(kt + 2, nk - 1));
        mma_group(1);
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
      } else {
        mma_group(1);
      }
      if ((kt & (FLUSH - 1)) == FLUSH - 1) flush();
    }
  }
  flush();

  conv_epilogue<MT, NT, BN, EPI>(a, acc, bn, wn, lane, [&](int i, int r) {
    const int local = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    return (long)(n_img * a.Ho + ty * PH + (local >> 4)) * a.Wo + tx * 16 + (local & 15);
  });
}

// The halo kernel on split fp16 operands (round 6): same 8 x 16 patch, same per-tap weight slabs, but the halo and the
// weight slab are scaled by their tensors' power-of-two factors and written to LDS as two fp16 terms (the 32-channel row keeps its
// 128 bytes: 64 B of leading terms, 64 B of residuals) ONCE per 32-channel chunk -- the implicit-GEMM kernel converts every input
// element once per tap, nine times -- and the taps run as 3 x v_mfma_f32_32x32x16_f16 per fragment pair on fragments read at shifted
// LDS addresses.  NT = 1: a 64-column tile for the 64-channel layers (VGG conv1_2, `architecture.py:151-181`), NT = 2: 128 columns.
template <int NT>
__global__ __launch_bounds__(256, 2) void conv_halo_f16_kernel(ConvArgs a) {
  constexpr int WM = 2, MT = 2, BN = 2 * NT * 32, NTHR = 256, B_CH = BN * 8 / NTHR, RPP = NTHR / 8;
  constexpr int PH = WM * 4, HALO_PX = HALO_W * (PH + 2), HALO_CH = (HALO_PX * 8 + NTHR - 1) / NTHR;
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ah = smem;                    // [180][LDK]: per pixel 32 channels as [hi x 32 | lo x 32] halves
  float* Bs = smem + HALO_PX * LDK;    // [2][BN][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bm, bn;
  {
    const int nbn = gridDim.y, total = gridDim.x * nbn;
    const int b = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, idx = b >> 3;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bn = l % nbn;
    bm = l / nbn;
  }
  const int txn = a.Wo / 16, tyn = a.Ho / PH;
  const int tx = bm % txn, ty = (bm / txn) % tyn, n_img = bm / (txn * tyn);
  const int chunk = tid & 7, lrow = tid >> 3;
  const int ntaps = 9, ncc = a.Cin / 32, nk = ncc * ntaps;

  unsigned h_voff[HALO_CH];
  int h_lds[HALO_CH];
#pragma unroll
  for (int j = 0; j < HALO_CH; ++j) {
    const int e = tid + NTHR * j, hp = e >> 3, hc = e & 7;      // (hc == chunk: NTHR % 8 == 0)
    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
    const int y = ty * PH + hy - 1, x = tx * 16 + hx - 1;
    const bool ok = e < HALO_PX * 8 && y >= 0 && y < a.Hi && x >= 0 && x < a.Wi;
    h_voff[j] = ok ? (unsigned)((((size_t)(n_img * a.Hi + y) * a.Wi + x) * a.Cin + hc * 4) * 4) : 0xFFFFFFFFu;
    h_lds[j] = e < HALO_PX * 8 ? hp * LDK : -1;
  }
  unsigned b_voff[B_CH];
#pragma unroll
  for (int j = 0; j < B_CH; ++j) b_voff[j] = (unsigned)(((size_t)(lrow + RPP * j) * a.wstride + chunk * 4) * 4);
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0xFFFFFFFE, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.w + (size_t)bn * BN * a.wstride), 0, 0xFFFFFFFE, 0x00020000);

  const float sc_a = dsee_pow2_scale(dsee_amax_read(a.amax_a)), sc_w = dsee_pow2_scale(dsee_amax_read(a.amax_w));
  // 4 consecutive channels of one row -> 4 halves of each term, at byte 8 * chunk of the row's term-0 / term-1 half
  auto store_split = [&](float* row, const f32x4& v, float sc) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 h0, h1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e] * sc;
      h0[e] = (_Float16)x;
      h1[e] = (_Float16)(x - (float)h0[e]);
    }
    unsigned char* q = reinterpret_cast<unsigned char*>(row) + chunk * 8;
    *reinterpret_cast<h4*>(q) = h0;
    *reinterpret_cast<h4*>(q + 64) = h1;
  };

  f32x4 hreg[HALO_CH], rb[B_CH];
  auto load_halo = [&](int cc, bool live) {
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(cc * (BK * 4));
#pragma unroll
    for (int j = 0; j < HALO_CH; ++j)
      hreg[j] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, live ? h_voff[j] : 0xFFFFFFFFu, so, 0));
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int j = 0; j < HALO_CH; ++j)
      if (h_lds[j] >= 0) store_split(Ah + h_lds[j], hreg[j], sc_a);
  };
  auto load_b = [&](int kt) {
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(kt * (BK * 4));
#pragma unroll
    for (int j = 0; j < B_CH; ++j)
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_voff[j], so, 0));
  };
  auto store_b = [&](int buf) {
    float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < B_CH; ++j) store_split(Bb + (lrow + RPP * j) * LDK, rb[j], sc_w);
  };

  f32x16 acc[MT][NT], part[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = part[i][j][r] = 0.f;

  // fragment of k-step s16 (16 channels), term p, of the row at float offset `row`: 16 bytes at p * 64 + s16 * 32 + (lane >> 5) * 16
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  int a_frag[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_frag[i] = ((wm * 4 + i * 2 + (frow >> 4)) * HALO_W + (frow & 15)) * LDK + fk;
  const int b_frag = (wn * NT * 32 + frow) * LDK + fk;
  auto tap_off = [&](int tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    return ((1 + a.off + kh * a.kdir) * HALO_W + (1 + a.off + kw * a.kdir)) * LDK;
  };
  u32x4v af[2][MT][2], bf[2][NT][2];
  auto read_frags = [&](int toff, int buf, int s16, int set) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) af[set][i][p] = *reinterpret_cast<const u32x4v*>(Ah + a_frag[i] + toff + s16 * 8 + p * 16);
    const float* Bc = Bs + buf * BN * LDK + b_frag + s16 * 8;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) bf[set][j][p] = *reinterpret_cast<const u32x4v*>(Bc + j * 32 * LDK + p * 16);
  };
  auto mma_group = [&](int set) {
#pragma unroll
    for (int q = 0; q < 3; ++q)   // a1*b0, a0*b1, a0*b0 (smallest first)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          part[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, af[set][i][q == 0 ? 1 : 0]),
                                                              __builtin_bit_cast(f16x8v, bf[set][j][q == 1 ? 1 : 0]),
                                                              part[i][j], 0, 0, 0);
  };
  auto flush = [&]() {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] += part[i][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
      }
  };

  load_halo(0, true);
  load_b(0);
  int cur = 0;
  for (int cc = 0; cc < ncc; ++cc) {
    __syncthreads();  // nobody reads the previous chunk's halo / weight slabs any more
    store_halo();
    store_b(cur);
    __syncthreads();
    load_halo(min(cc + 1, ncc - 1), cc + 1 < ncc);  // next chunk's halo: 9 taps of MFMAs cover it
    load_b(min(cc * ntaps + 1, nk - 1));
    read_frags(tap_off(0), cur, 0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kt = cc * ntaps + tap;
      read_frags(tap_off(tap), cur, 1, 1);
      mma_group(0);
      if (tap < 8) {
        __builtin_amdgcn_sched_barrier(0);
        store_b(cur ^ 1);     // tap + 1's weights (loaded a whole tap ago) into the buffer every wave left before the last barrier
        __syncthreads();
        read_frags(tap_off(tap + 1), cur ^ 1, 0, 0);
        load_b(min(kt + 2, nk - 1));
        mma_group(1);
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
      } else {
        mma_group(1);
      }
      if ((kt & (FLUSH - 1)) == FLUSH - 1) flush();
    }
  }
  flush();
  const float oscale = 1.f / (sc_a * sc_w);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] *= oscale;

  conv_epilogue<MT, NT, BN, EPI_PLAIN>(a, acc, bn, wn, lane, [&](int i, int r) {
    const int local = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    return (long)(n_img * a.Ho + ty * PH + (local >> 4)) * a.Wo + tx * 16 + (local & 15);
  });
}

// 3 x 3 / stride 1 / pad 1 over a 4-channel (RGB0) input with 64 output channels (VGG conv1_1, `architecture.py:151-181`): K = 36.  The
// implicit-GEMM kernel spends a 32-k LDS slab, two barriers and a gather per tap on it (15 TFLOP/s: 0.165 ms for a layer whose
// only real cost is writing 134 MB).  Here nothing goes through LDS: a lane builds its A fragments straight from global memory -- the
// 8 k of a 32x32x16 fragment are two taps x 4 channels = two 16-byte loads --, the 64 x 48 weights live in registers as split
// fragments for the whole kernel, a wave owns 64 consecutive pixels x 64 channels (36 MFMAs) and the plain epilogue stores them.
__global__ __launch_bounds__(256) void conv3x3_c4_f16_kernel(ConvArgs a) {
  typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kg = lane >> 5;
  const float sc_a = dsee_pow2_scale(dsee_amax_read(a.amax_a)), sc_w = dsee_pow2_scale(dsee_amax_read(a.amax_w));
  auto split8 = [&](const f32x4& u, const f32x4& v, float sc, f16x8v& hi, f16x8v& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (e < 4 ? u[e] : v[e - 4]) * sc;
      hi[e] = (_Float16)x;
      lo[e] = (_Float16)(x - (float)hi[e]);
    }
  };
  f16x8v bh[2][3], bl[2][3];     // weights: column block j, k-slab s (k = s * 16 + kg * 8 ..+7 of row j * 32 + l31)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float* wr = a.w + (size_t)(j * 32 + l31) * a.wstride + s * 16 + kg * 8;
      split8(*reinterpret_cast<const f32x4*>(wr), *reinterpret_cast<const f32x4*>(wr + 4), sc_w, bh[j][s], bl[j][s]);
    }
  const float oscale = 1.f / (sc_a * sc_w);
  const long ntile = ((long)a.M + 63) / 64;
  for (long t = (long)blockIdx.x * 4 + wave; t < ntile; t += (long)gridDim.x * 4) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long p = t * 64 + i * 32 + l31;
      const unsigned pu = (unsigned)(p < a.M ? p : 0), r_ = pu / (unsigned)a.Wi, n_ = r_ / (unsigned)a.Hi;
      const int x = (int)(pu - r_ * (unsigned)a.Wi), y = (int)(r_ - n_ * (unsigned)a.Hi);
      const float* img = a.in + (size_t)n_ * a.Hi * a.Wi * 4;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        f32x4 v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int tap = s * 4 + kg * 2 + q, ky = tap / 3, kx = tap - ky * 3;
          const int yy = y + ky - 1, xx = x + kx - 1;
          const bool ok = p < a.M && tap < 9 && yy >= 0 && yy < a.Hi && xx >= 0 && xx < a.Wi;
          v[q] = ok ? *reinterpret_cast<const f32x4*>(img + ((size_t)yy * a.Wi + xx) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        f16x8v ah, al;
        split8(v[0], v[1], sc_a, ah, al);
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // a1*b0, a0*b1, a0*b0 (smallest first)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j][s], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j][s], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j][s], acc[i][j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= oscale;
    conv_epilogue<2, 2, 64, EPI_PLAIN>(a, acc, 0, 0, lane, [&](int i, int r) {
      const long row = t * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      return row < a.M ? row : -1L;
    });
  }
}

// ---------------------------------------------------------------- weight gradient (split-K)
struct WgradArgs {
  const float* dout;  // [M][Cout]
  const float* in;    // [N][Hi][Wi][Cin]
  float* slab;        // [S][rows][Kpad]
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int KH, KW, Ktot, Kpad;
  int mul, off, kdir, dshift, ups;
  int M, msplit, rows;
  int korder, Kuse;  // slab column order; number of slab columns actually computed (<= Ktot)
  int Kstart;        // first slab column computed (multiple of 128)
  int margin;  // WGEO 1: bytes the `in` buffer base is moved down (most negative tap shift)
  // fp16x2 operand mode (F16 = true): device maxima of dout and in (64-line layout of dsee_common.h), NULL = fp32 MFMA
  const float* amax_dout;
  const float* amax_in;
};

constexpr int WLD = 132;
constexpr int WROWB = 576;  // F16: bytes of one pixel row of a 32 x 128 tile: 128 halfs of term 0 | 128 of term 1 | 64 B pad  // padded LDS row for the 32 x 128 wgrad tiles

// WGEO = 1: stride-1 "same" convolution (mul 1, dshift 0, ups 0, Hi == Ho, Wi == Wo): the source pixel of output
//           pixel m under tap (kh,kw) is m + const, so both operands are read with buffer loads whose per-thread
//           VGPR offset is fixed for the whole kernel and whose per-slab offset is ONE scalar; out-of-image taps
//           get the out-of-range offset and come back as zeros.  WGEO = 0: general geometry (two divisions per row).
// F16: both operands are scaled (powers of two from their device maxima) and split into two fp16 terms when a slab is
//      stored to LDS, in the natural [pixel][column] order; the MFMA fragments -- 8 consecutive PIXELS of one column per
//      lane -- come out of ds_read_b64_tr_b16 (LDS transpose read: within a 16-lane group, lanes 4e..4e+3 address row e,
//      lane i receives column i of the 4 x 16 block), two reads per 8 pixels.  6 v_mfma_f32_32x32x16_f16 per 32-pixel
//      slab and accumulator tile instead of 16 v_mfma_f32_32x32x2_f32.
template <int WGEO, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  // D[i = cout][j = k'] = sum over pixels.  A'[px][co] = dout tile, B'[px][k'] = shifted input tile.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][32][WLD]
  float* Bs = smem + 2 * 32 * WLD;  // [2][32][WLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order (speed only): all (k', cout) tiles of one pixel split share dout/in slabs -> same XCD, adjacent
  int kx, cy, z;
  {
    const int ntile = gridDim.x * gridDim.y, total = ntile * gridDim.z;
    const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = b & 7, idx = b >> 3;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int t = l % ntile;
    z = l / ntile;
    kx = t % gridDim.x;
    cy = t / gridDim.x;
  }
  const int chunk = tid & 31, lrow = tid >> 5;  // 8 rows per pass, 4 passes
  const int m0 = z * a.msplit;
  const int m1 = min(a.M, m0 + a.msplit);
  // B' column (k') owned by this thread is fixed for the whole kernel
  // (korder 1: chunk-major k' = (ci/32)*(taps*32) + tap*32 + ci%32, so that "only the first Kuse columns" means
  //  "only the first Kuse/(taps*32) channel chunks" — the one-hot tail of the SEAN input needs no shared gradient)
  const int kq = a.Kstart + kx * 128 + chunk * 4;
  const bool kok = kq < a.Kuse;
  int tap = 0, cch = 0;
  if (kok) {
    if (a.korder == 0) {
      tap = kq / a.Cin;
      cch = kq - tap * a.Cin;
    } else {
      const int span = a.KH * a.KW * 32, cc = kq / span, rem = kq - cc * span;
      tap = rem >> 5;
      cch = cc * 32 + (rem & 31);
    }
  }
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int co = cy * 128 + chunk * 4;
  const bool cok = co < a.Cout;
  const int Hl = a.Hi << a.ups, Wl = a.Wi << a.ups;
  const int dmask = (1 << a.dshift) - 1;

  // WGEO 1 state: per-row (oh, ow) walked incrementally, fixed VGPR offsets
  const int dy = a.off + kh * a.kdir, dx = a.off + kw * a.kdir;
  int r_oh[4], r_ow[4];
  unsigned va[4], vb[4];
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, 0xFFFFFFFE, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(a.in) - a.margin), 0, 0xFFFFFFFE, 0x00020000);
  if constexpr (WGEO == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + lrow + 8 * j;
      r_ow[j] = m % a.Wo;
      r_oh[j] = (m / a.Wo) % a.Ho;
      va[j] = cok ? (unsigned)(((size_t)(lrow + 8 * j) * a.Cout + co) * 4) : 0xFFFFFFFFu;
      vb[j] = kok ? (unsigned)(((long)(lrow + 8 * j + dy * a.Wi + dx) * a.Cin + cch) * 4 + a.margin) : 0xFFFFFFFFu;
    }
  }

  f32x4 ra[4], rb[4];
  float keep_a[4], keep_b[4];
  // `live` = false for the branch-free dummy prefetches past the last slab: every row is masked, nothing is read
  // (the incremental row state has moved past the split by then and must not be used to validate addresses).
  auto load_tile = [&](int kt, bool live) {
    const int mb = m0 + kt * 32;
    if constexpr (WGEO == 1) {
      const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane((int)(((size_t)mb * a.Cout) * 4));
      const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)(((size_t)mb * a.Cin) * 4));
      const int rem = __builtin_amdgcn_readfirstlane(live ? m1 - mb : 0);  // rows of this slab that exist
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool mok = lrow + 8 * j < rem;
        const int ph = r_oh[j] + dy, pw = r_ow[j] + dx;
        const bool ok = mok && ph >= 0 && ph < a.Hi && pw >= 0 && pw < a.Wi;
        ra[j] = __builtin_bit_cast(f32x4,
                                   __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, mok ? va[j] : 0xFFFFFFFFu, sa, 0));
        rb[j] = __builtin_bit_cast(f32x4,
                                   __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, ok ? vb[j] : 0xFFFFFFFFu, sb, 0));
        // advance the row by 32 pixels (Wo >= 32 guaranteed by the dispatcher -> at most one wrap)
        int ow = r_ow[j] + 32;
        const int wrap = ow >= a.Wo;
        ow -= wrap ? a.Wo : 0;
        int oh = r_oh[j] + wrap;
        oh = oh == a.Ho ? 0 : oh;
        r_ow[j] = ow;
        r_oh[j] = oh;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = mb + lrow + 8 * j;
        const bool mok = live && m < m1;
        const int mm = mok ? m : 0;
        const int ow = mm % a.Wo;
        const int t = mm / a.Wo;
        const int oh = t % a.Ho;
        const int n = t / a.Ho;
        int ph = oh * a.mul + a.off + kh * a.kdir;
        int pw = ow * a.mul + a.off + kw * a.kdir;
        bool ok = mok && kok && ph >= 0 && pw >= 0 && ((ph | pw) & dmask) == 0;
        ph >>= a.dshift;
        pw >>= a.dshift;
        ok = ok && ph < Hl && pw < Wl;
        ph >>= a.ups;
        pw >>= a.ups;
        const bool oka = mok && cok;
        ra[j] = *reinterpret_cast<const f32x4*>(a.dout + (oka ? (size_t)m * a.Cout + co : 0));
        rb[j] = *reinterpret_cast<const f32x4*>(a.in + (ok ? ((size_t)(n * a.Hi + ph) * a.Wi + pw) * a.Cin + cch : 0));
        keep_a[j] = oka ? 1.f : 0.f;
        keep_b[j] = ok ? 1.f : 0.f;
      }
    }
  };
  float sc_a = 1.f, sc_b = 1.f;
  if constexpr (F16) {
    sc_a = dsee_pow2_scale(dsee_amax_read(a.amax_dout));
    sc_b = dsee_pow2_scale(dsee_amax_read(a.amax_in));
  }
  unsigned char* A16 = reinterpret_cast<unsigned char*>(smem);   // F16: [2][32][WROWB] per operand
  unsigned char* B16 = A16 + 2 * 32 * WROWB;
  auto store_split = [&](unsigned char* row, const f32x4& v, float sc) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 h0, h1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e] * sc;
      h0[e] = (_Float16)x;
      h1[e] = (_Float16)(x - (float)h0[e]);
    }
    *reinterpret_cast<h4*>(row + chunk * 8) = h0;
    *reinterpret_cast<h4*>(row + 256 + chunk * 8) = h1;
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 va_ = (WGEO == 1 || keep_a[j] != 0.f) ? ra[j] : z4, vb_ = (WGEO == 1 || keep_b[j] != 0.f) ? rb[j] : z4;
      if constexpr (F16) {
        store_split(A16 + (buf * 32 + lrow + 8 * j) * WROWB, va_, sc_a);
        store_split(B16 + (buf * 32 + lrow + 8 * j) * WROWB, vb_, sc_b);
      } else {
        *reinterpret_cast<f32x4*>(As + (buf * 32 + lrow + 8 * j) * WLD + chunk * 4) = va_;
        *reinterpret_cast<f32x4*>(Bs + (buf * 32 + lrow + 8 * j) * WLD + chunk * 4) = vb_;
      }
    }
  };

  f32x16 acc[2][2], part[2][2];  // two-level accumulation over the pixel dimension (see the forward kernel)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = part[i][j][r] = 0.f;

  const int nk = (m1 - m0 + 31) / 32;
  if constexpr (F16) {
    typedef short v4i16 __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    // this lane's part of a transpose read: row (pixel) (lane & 15) >> 2 of the 4-row block, columns 16 * ((lane >> 4) & 1)
    // + 4 * (lane & 3) .. + 3 of the 32-column tile; pixel block 8 * (lane >> 5) (+ 4 for the second read) of a 16-pixel step
    const int t_row = 8 * (lane >> 5) + ((lane & 15) >> 2), t_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto frag = [&](const unsigned char* base, int buf, int s16, int col0, int term) {
      const unsigned char* p = base + (buf * 32 + 16 * s16 + t_row) * WROWB + term * 256 + (col0 + t_col) * 2;
      const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(p));
      const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(p + 4 * WROWB));
      typedef short v8i16 __attribute__((ext_vector_type(8)));
      const v8i16 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(f16x8v, v);
    };
    if (nk > 0) {
      load_tile(0, true);
      store_tile(0);
    }
    __syncthreads();
    if (nk > 0) load_tile(min(1, nk - 1), nk > 1);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
      for (int s16 = 0; s16 < 2; ++s16) {
        f16x8v fa[2][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            fa[i][p] = frag(A16, cur, s16, wm * 64 + i * 32, p);
            fb[i][p] = frag(B16, cur, s16, wn * 64 + i * 32, p);
          }
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              part[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][q == 0 ? 1 : 0], fb[j][q == 1 ? 1 : 0], part[i][j],
                                                                  0, 0, 0);
        if (s16 == 0) {
          __builtin_amdgcn_sched_barrier(0);
          store_tile(cur ^ 1);
        }
      }
      __syncthreads();
      load_tile(min(kt + 2, nk - 1), kt + 2 < nk);
      if ((kt & (FLUSH - 1)) == FLUSH - 1 || kt + 1 == nk) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] += part[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
          }
      }
      cur ^= 1;
    }
    const float oscale = 1.f / (sc_a * sc_b);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= oscale;
  } else {
  const int fcol = lane & 31, fk = lane >> 5;
  // fragments double-buffered in registers: 4 k-steps (16 MFMAs) per group, the next group's 16 ds_read_b32 are in
  // flight while the current group's MFMAs issue, and the last group of a slab runs across the barrier (as in the
  // forward kernel) so the matrix pipe is not left empty for barrier + LDS latency.
  float fa[2][8], fb[2][8];
  auto read_frags = [&](int buf, int g, int set) {
    const float* Ac = As + buf * 32 * WLD + wm * 64 + fcol + (g * 8 + fk) * WLD;
    const float* Bc = Bs + buf * 32 * WLD + wn * 64 + fcol + (g * 8 + fk) * WLD;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fa[set][2 * q] = Ac[q * 2 * WLD];
      fa[set][2 * q + 1] = Ac[q * 2 * WLD + 32];
      fb[set][2 * q] = Bc[q * 2 * WLD];
      fb[set][2 * q + 1] = Bc[q * 2 * WLD + 32];
    }
  };
  auto mma_group = [&](int set) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      part[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][2 * q], fb[set][2 * q], part[0][0], 0, 0, 0);
      part[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][2 * q], fb[set][2 * q + 1], part[0][1], 0, 0, 0);
      part[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][2 * q + 1], fb[set][2 * q], part[1][0], 0, 0, 0);
      part[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][2 * q + 1], fb[set][2 * q + 1], part[1][1], 0, 0, 0);
    }
  };
  if (nk > 0) {
    load_tile(0, true);
    store_tile(0);
  }
  __syncthreads();
  if (nk > 0) {
    load_tile(min(1, nk - 1), nk > 1);
    read_frags(0, 0, 0);
  }
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    read_frags(cur, 1, 1);
    mma_group(0);
    read_frags(cur, 2, 0);
    mma_group(1);
    read_frags(cur, 3, 1);
    mma_group(0);
    __builtin_amdgcn_sched_barrier(0);  // slab kt+1 was requested a full slab ago; its first use stays below
    store_tile(cur ^ 1);
    __syncthreads();
    read_frags(cur ^ 1, 0, 0);                    // next slab's first fragments under this slab's last MFMA group
    load_tile(min(kt + 2, nk - 1), kt + 2 < nk);  // next prefetch right after the barrier (branch-free)
    mma_group(1);
    __builtin_amdgcn_sched_barrier(0);
    if ((kt & (FLUSH - 1)) == FLUSH - 1 || kt + 1 == nk) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] += part[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
        }
    }
    cur ^= 1;
  }
  }
  float* slab = a.slab + (size_t)z * a.rows * a.Kpad;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = a.Kstart + kx * 128 + wn * 64 + j * 32 + (lane & 31);
    if (col >= a.Kuse) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cy * 128 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < a.rows) slab[(size_t)row * a.Kpad + col] = acc[i][j][r];
      }
  }
}

// ---------------------------------------------------------------- pack / unpack
// k index of (tap, ci): korder 0 = tap-major  tap*C + ci ; korder 1 = chunk-major (ci/32)*(taps*32) + tap*32 + ci%32
__device__ __forceinline__ void k_to_tap_ci(int k, int C, int taps, int korder, int& tap, int& ci) {
  if (korder == 0) {
    tap = k / C;
    ci = k % C;
  } else {
    const int cc = k / (taps * 32), rem = k % (taps * 32);
    tap = rem / 32;
    ci = cc * 32 + (rem & 31);
  }
}

__global__ void pack_fwd_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale_num,
                                const float* __restrict__ inv_scale_den, float* __restrict__ p, int Cout, int Cin,
                                int KH, int KW, int Cin_s, int rows, int Kpad, int korder) {
  // p[row][k(tap, ci)] = w[row][ci][kh][kw] * s
  const long total = (long)rows * Kpad;
  float s = 1.f;
  if (inv_scale_den) s = (inv_scale_num ? *inv_scale_num : 1.f) / *inv_scale_den;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / Kpad), k = (int)(i % Kpad);
    int tap, ci;
    k_to_tap_ci(k, Cin_s, KH * KW, korder, tap, ci);
    float v = 0.f;
    if (row < Cout && tap < KH * KW && ci < Cin) v = w[((size_t)row * Cin + ci) * KH * KW + tap] * s;
    p[i] = v;
  }
}

__global__ void pack_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale_num,
                                  const float* __restrict__ inv_scale_den, float* __restrict__ p, int Cout, int Cin,
                                  int KH, int KW, int Cout_s, int rows, int Kpad, int korder) {
  // p[ci][k(tap, co)] = w[co][ci][kh][kw] * s      (rows index ci)
  const long total = (long)rows * Kpad;
  float s = 1.f;
  if (inv_scale_den) s = (inv_scale_num ? *inv_scale_num : 1.f) / *inv_scale_den;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / Kpad), k = (int)(i % Kpad);
    int tap, co;
    k_to_tap_ci(k, Cout_s, KH * KW, korder, tap, co);
    float v = 0.f;
    if (row < Cin && tap < KH * KW && co < Cout) v = w[((size_t)co * Cin + row) * KH * KW + tap] * s;
    p[i] = v;
  }
}

__global__ void wgrad_reduce_unpack_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int rows,
                                           int Kpad, int Cout, int Cin, int KH, int KW, int Cin_s, int korder,
                                           int ci0) {
  // dw[co][ci][kh][kw] = sum_s slab[s][co][k'(tap, ci0 + ci)]   (fixed order => deterministic)
  const long total = (long)Cout * Cin * KH * KW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % (KH * KW));
    const long t = i / (KH * KW);
    const int ci = ci0 + (int)(t % Cin), co = (int)(t / Cin);
    const size_t kcol = korder == 0 ? (size_t)tap * Cin_s + ci
                                    : (size_t)(ci >> 5) * (KH * KW * 32) + (size_t)tap * 32 + (ci & 31);
    const size_t o = (size_t)co * Kpad + kcol;
    float v = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; ++s) v += slab[(size_t)s * rows * Kpad + o];   // (8 partial loads in flight per thread)
    dw[i] = v;
  }
}

// dT[n][tap][row][r] = sum_{j < s} slab[n*s + j][row][k0 + tap*32 + r]   (per-image style-table gradient)
__global__ void wgrad_table_unpack_kernel(const float* __restrict__ slab, float* __restrict__ dt, int N, int sper,
                                          int rows, int Kpad, int k0, int ntaps, int L) {
  const long total = (long)N * ntaps * rows * 32;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i & 31);
    long t = i >> 5;
    const int row = (int)(t % rows);
    t /= rows;
    const int tap = (int)(t % ntaps), n = (int)(t / ntaps);
    float v = 0.f;
    if (r < L)
#pragma unroll 8
      for (int j = 0; j < sper; ++j)
        v += slab[((size_t)(n * sper + j) * rows + row) * Kpad + k0 + tap * 32 + r];
    dt[i] = v;
  }
}


// Winograd F(4x4,3x3) weight gradient, last stage: dU[xi][co][ci] = sum_j slab[xi*sper + j][co][ci], then
// dg = G^T dU G  -> dw OIHW [Cout][Cin][3][3]
// SPER > 0: the split count as a compile-time constant -- all 36 * SPER loads of a thread are then independent and in flight
// together (with a run-time loop they were issued one split at a time: 190 us for 302 MB at 512 x 512, pure latency).
template <int SPER>
__global__ void wino43_wgrad_finalize_kernel(const float* __restrict__ slab, float* __restrict__ dw, int sper_rt,
                                             int rows, int Kpad, int Cout, int Cin) {
  const int sper = SPER > 0 ? SPER : sper_rt;
  const long total = (long)Cout * Cin;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i / Cin), ci = (int)(i % Cin);
    float t[3][6];  // G^T dU
    {
      float u[6][6];
#pragma unroll
      for (int xi = 0; xi < 36; ++xi) {
        float v = 0.f;
        if constexpr (SPER > 0) {
#pragma unroll
          for (int j = 0; j < SPER; ++j) v += slab[((size_t)(xi * SPER + j) * rows + co) * Kpad + ci];
        } else {
          for (int j = 0; j < sper; ++j) v += slab[((size_t)(xi * sper + j) * rows + co) * Kpad + ci];
        }
        u[xi / 6][xi % 6] = v * dsee_dm_posr(xi);   // (dM = f_i f_j (A dY A^T)[i][j], dsee_common.h: undone here, exactly)
      }
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const float s12 = u[1][b] + u[2][b], d12 = u[1][b] - u[2][b], s34 = u[3][b] + u[4][b], d34 = u[3][b] - u[4][b];
        t[0][b] = 0.25f * u[0][b] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
        t[1][b] = -(1.f / 6.f) * d12 + (1.f / 12.f) * d34;
        t[2][b] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + u[5][b];
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float s12 = t[a][1] + t[a][2], d12 = t[a][1] - t[a][2], s34 = t[a][3] + t[a][4], d34 = t[a][3] - t[a][4];
      dw[i * 9 + a * 3 + 0] = 0.25f * t[a][0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
      dw[i * 9 + a * 3 + 1] = -(1.f / 6.f) * d12 + (1.f / 12.f) * d34;
      dw[i * 9 + a * 3 + 2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[a][5];
    }
  }
}

// Same for the SEAN gamma/beta GEMM with per-image groups (slab index ((xi*N + n)*sper + j)): shared columns k < ca are
// summed over images into dw2a [rows][ca][3][3]; one-hot columns per image into dtable [N][9][rows][32].
template <int SPER>
__global__ void wino43_wgrad_table_finalize_kernel(const float* __restrict__ slab, float* __restrict__ dw2a,
                                                   float* __restrict__ dtable, int N, int sper_rt, int rows, int Kpad,
                                                   int ca, int L) {
  const int sper = SPER > 0 ? SPER : sper_rt;
  // work items: first rows * ca shared elements (sum over the N images), then N * rows * 32 per-image table elements --
  // separate index ranges, so that a wave runs one of the two paths (a thread per (row, column) doing both made every
  // wave walk the table path N times with 4/5 of its lanes idle: 141 us per call at any size)
  const long n_shared = dw2a ? (long)rows * ca : 0, total = n_shared + (long)N * rows * 32;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const bool shared = i < n_shared;
    int row, k, n0, n1;
    if (shared) {
      row = (int)(i / ca);
      k = (int)(i % ca);
      n0 = 0;
      n1 = N;
    } else {
      const long e = i - n_shared;
      k = ca + (int)(e % 32);
      row = (int)((e / 32) % rows);
      n0 = (int)(e / (32L * rows));
      n1 = n0 + 1;
    }
    float acc[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = 0.f;
    for (int n = n0; n < n1; ++n)
#pragma unroll
      for (int xi = 0; xi < 36; ++xi) {
        float v = 0.f;
        if constexpr (SPER > 0) {
#pragma unroll
          for (int j = 0; j < SPER; ++j) v += slab[((size_t)((xi * N + n) * SPER + j) * rows + row) * Kpad + k];
        } else {
          for (int j = 0; j < sper; ++j) v += slab[((size_t)((xi * N + n) * sper + j) * rows + row) * Kpad + k];
        }
        acc[xi] += v;
      }
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] *= dsee_dm_posr(xi);   // (dM = f_i f_j (A dY A^T)[i][j], dsee_common.h)
    float t[3][6], dg[9];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const float u0 = acc[b], u1 = acc[6 + b], u2 = acc[12 + b], u3 = acc[18 + b], u4 = acc[24 + b], u5 = acc[30 + b];
      const float s12 = u1 + u2, d12 = u1 - u2, s34 = u3 + u4, d34 = u3 - u4;
      t[0][b] = 0.25f * u0 - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
      t[1][b] = -(1.f / 6.f) * d12 + (1.f / 12.f) * d34;
      t[2][b] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + u5;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float s12 = t[a][1] + t[a][2], d12 = t[a][1] - t[a][2], s34 = t[a][3] + t[a][4], d34 = t[a][3] - t[a][4];
      dg[a * 3 + 0] = 0.25f * t[a][0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
      dg[a * 3 + 1] = -(1.f / 6.f) * d12 + (1.f / 12.f) * d34;
      dg[a * 3 + 2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[a][5];
    }
    if (shared) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) dw2a[((size_t)row * ca + k) * 9 + tap] = dg[tap];
    } else {
      const int r = k - ca;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) dtable[(((size_t)n0 * 9 + tap) * rows + row) * 32 + r] = r < L ? dg[tap] : 0.f;
    }
  }
}

template <int MT, int NT, int WM, int WN, int EPI, int GEO, bool F16>
int launch_conv_geo_t(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = MT * WM * 32, BN = NT * WN * 32;
  const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<MT, NT, WM, WN, EPI, GEO, F16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(dsee_cdiv(a.M, BM), EPI == EPI_MODULATE ? dsee_cdiv(a.C, BN / 2) : dsee_cdiv(a.Cout, BN));
  conv_igemm_kernel<MT, NT, WM, WN, EPI, GEO, F16><<<grid, WM * WN * 64, lds, st>>>(a);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

template <int MT, int NT, int WM, int WN, int EPI, int GEO>
int launch_conv_geo(const ConvArgs& a, hipStream_t st) {
  if constexpr (EPI == EPI_PLAIN) {
    if (a.amax_a && a.amax_w) return launch_conv_geo_t<MT, NT, WM, WN, EPI, GEO, true>(a, st);
  }
  return launch_conv_geo_t<MT, NT, WM, WN, EPI, GEO, false>(a, st);
}

template <int EPI, int WM>
int launch_conv_halo(const ConvArgs& a, hipStream_t st) {
  const size_t lds = (size_t)(HALO_W * (WM * 4 + 2) + 2 * 128) * LDK * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<EPI, WM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(a.M / (WM * 64), EPI == EPI_MODULATE ? dsee_cdiv(a.C, 64) : dsee_cdiv(a.Cout, 128));
  conv_halo_kernel<EPI, WM><<<grid, WM * 128, lds, st>>>(a);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

static bool halo_ok(const ConvArgs& a) {
  return a.korder == 1 && a.KH == 3 && a.KW == 3 && a.mul == 1 && a.Hi == a.Ho && a.Wi == a.Wo &&
         a.Ho % 8 == 0 && a.Wo % 16 == 0 && a.off == -a.kdir && (a.kdir == 1 || a.kdir == -1) &&
         (long)a.N * a.Hi * a.Wi * a.Cin * 4 + 65536 < 0xFFFFFFFEL;
}

static bool c4_f16_ok(const ConvArgs& a) {
  return a.Cin == 4 && a.Cout == 64 && a.KH == 3 && a.KW == 3 && a.mul == 1 && a.off == -1 && a.kdir == 1 && a.dshift == 0 &&
         a.ups == 0 && a.korder == 0 && a.Hi == a.Ho && a.Wi == a.Wo && a.amax_a && a.amax_w && !a.wt && a.wgroup_stride == 0 &&
         a.Kpad >= 48 && a.M >= 16384;
}

int launch_conv3x3_c4_f16(const ConvArgs& a, hipStream_t st) {
  const long ntile = ((long)a.M + 63) / 64;
  conv3x3_c4_f16_kernel<<<(unsigned)min(2048L, (ntile + 3) / 4), 256, 0, st>>>(a);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

template <int NT>
int launch_conv_halo_f16(const ConvArgs& a, hipStream_t st) {
  constexpr int BN = 2 * NT * 32;
  const size_t lds = (size_t)(HALO_W * 10 + 2 * BN) * LDK * sizeof(float);
  dim3 grid(a.M / 128, dsee_cdiv(a.Cout, BN));
  conv_halo_f16_kernel<NT><<<grid, 256, lds, st>>>(a);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

// split-operand halo kernel: 3 x 3 / stride 1 / pad 1 layers with whole 8 x 16 patches, 32-channel chunks and > 32 output columns
static bool halo_f16_ok(const ConvArgs& a) {
  return halo_ok(a) && a.amax_a && a.amax_w && a.Cin % 32 == 0 && a.Cout > 32 && !a.wt && a.wgroup_stride == 0 &&
         a.dshift == 0 && a.ups == 0 && a.M % 128 == 0;
}

template <int MT, int NT, int WM, int WN, int EPI>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  if (MT == 2 && NT == 2 && WM == 2 && WN == 2 && halo_ok(a) && !a.amax_a) return launch_conv_halo<EPI, 2>(a, st);
  ConvArgs b = a;
  bool fast = a.korder == 1;
  if (fast) {
    long mn = 0;  // most negative tap shift, in bytes
    for (int kh = 0; kh < a.KH; kh += (a.KH > 1 ? a.KH - 1 : 1))
      for (int kw = 0; kw < a.KW; kw += (a.KW > 1 ? a.KW - 1 : 1)) {
        const long t = ((long)(a.off + kh * a.kdir) * a.Wi + (a.off + kw * a.kdir)) * a.Cin * 4;
        if (t < mn) mn = t;
      }
    b.margin = (int)(-mn);
    const long bytes = (long)a.N * a.Hi * a.Wi * a.Cin * 4;
    if (bytes + b.margin + 65536 >= 0xFFFFFFFEL || -mn > (1L << 30)) fast = false;  // 32-bit buffer offsets
  }
  if (!fast && a.korder == 1) {
    dsee_set_error("conv: tensor too large for 32-bit buffer addressing with korder=1 (pack with korder=0)");
    return DSEE_EUNSUPPORTED;
  }
  return fast ? launch_conv_geo<MT, NT, WM, WN, EPI, 1>(b, st) : launch_conv_geo<MT, NT, WM, WN, EPI, 0>(b, st);
}

inline bool small_grid(const ConvArgs& a) { return (long)((a.M + 127) / 128) * ((a.Cout + 127) / 128) < 192; }

int fill_geom(ConvArgs& a, const dsee_conv_geom* g) {
  DSEE_CHECK_ARG(g != nullptr);
  DSEE_CHECK_ARG(g->Cin % 4 == 0 && g->Cout % 4 == 0);
  DSEE_CHECK_ARG(g->N > 0 && g->Hi > 0 && g->Wi > 0 && g->Ho > 0 && g->Wo > 0 && g->KH > 0 && g->KW > 0);
  DSEE_CHECK_ARG(g->dshift >= 0 && g->dshift <= 3 && g->ups >= 0 && g->ups <= 3);
  a.N = g->N; a.Hi = g->Hi; a.Wi = g->Wi; a.Cin = g->Cin; a.Ho = g->Ho; a.Wo = g->Wo; a.Cout = g->Cout;
  a.KH = g->KH; a.KW = g->KW; a.Ktot = g->KH * g->KW * g->Cin; a.Kpad = (a.Ktot + 31) / 32 * 32;
  a.mul = g->mul; a.off = g->off; a.kdir = g->kdir; a.dshift = g->dshift; a.ups = g->ups; a.korder = g->korder;
  a.wstride = a.Kpad; a.wt = nullptr; a.nk_shared = 0; a.wt_rows = 0; a.wgroup_stride = 0;
  DSEE_CHECK_ARG(g->korder == 0 || (g->korder == 1 && g->dshift == 0 && g->ups == 0 && g->Cin % 32 == 0 && g->KH * g->KW <= 32));
  long M = (long)g->N * g->Ho * g->Wo;
  DSEE_CHECK_ARG(M < (1L << 31) && (long)g->N * g->Hi * g->Wi * g->Cin < (1L << 40));
  a.M = (int)M;
  return DSEE_OK;
}

}  // namespace

extern "C" {

int dsee_conv_kpad(int KH, int KW, int Cin_stored) { return (KH * KW * Cin_stored + 31) / 32 * 32; }
int dsee_conv_wrows(int Cout) { return (Cout + 127) / 128 * 128; }

int dsee_pack_weight_fwd(const float* w_oihw, const float* scale_num, const float* scale_den, float* packed, int Cout,
                         int Cin, int KH, int KW, int Cin_stored, int korder, hipStream_t st) {
  DSEE_CHECK_ARG(w_oihw && packed && Cin_stored >= Cin && Cin_stored % 4 == 0);
  DSEE_CHECK_ARG(korder == 0 || (korder == 1 && Cin_stored % 32 == 0));
  const int rows = dsee_conv_wrows(Cout), Kpad = dsee_conv_kpad(KH, KW, Cin_stored);
  const long total = (long)rows * Kpad;
  pack_fwd_kernel<<<(int)min(4096L, (total + 255) / 256), 256, 0, st>>>(w_oihw, scale_num, scale_den, packed, Cout, Cin,
                                                                        KH, KW, Cin_stored, rows, Kpad, korder);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_pack_weight_dgrad(const float* w_oihw, const float* scale_num, const float* scale_den, float* packed,
                           int Cout, int Cin, int KH, int KW, int Cout_stored, int korder, hipStream_t st) {
  DSEE_CHECK_ARG(w_oihw && packed && Cout_stored >= Cout && Cout_stored % 4 == 0);
  DSEE_CHECK_ARG(korder == 0 || (korder == 1 && Cout_stored % 32 == 0));
  const int rows = dsee_conv_wrows(Cin), Kpad = dsee_conv_kpad(KH, KW, Cout_stored);
  const long total = (long)rows * Kpad;
  pack_dgrad_kernel<<<(int)min(4096L, (total + 255) / 256), 256, 0, st>>>(w_oihw, scale_num, scale_den, packed, Cout,
                                                                          Cin, KH, KW, Cout_stored, rows, Kpad, korder);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_conv2d_fwd(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                    const float* residual, int residual_ld, float* out, int act, float slope, hipStream_t st) {
  return dsee_conv2d_fwd_amax(g, in, w_packed, bias, residual, residual_ld, out, act, slope, nullptr, st);
}

/* dsee_conv2d_fwd that also writes max |out| into amax_out (optional; 64-line layout, zeroed by the caller): the operand bound
 * the next direct layer's fp16x2 split needs, from this layer's epilogue instead of a dsee_absmax pass over the output. */
int dsee_conv2d_fwd_amax(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                         const float* residual, int residual_ld, float* out, int act, float slope, float* amax_out,
                         hipStream_t st) {
  ConvArgs a = {};
  a.amax_out = amax_out;
  int rc = fill_geom(a, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && w_packed && out);
  DSEE_CHECK_ARG(act != DSEE_ACT_MASK || residual != nullptr);
  a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out; a.act = act; a.slope = slope;
  a.res_ld = residual_ld > 0 ? residual_ld : a.Cout;
  // round 6: a 128 x 128 grid that would leave most of the 256 CUs idle (the discriminator's 17^2 / 33^2 layers: 19-70 workgroups)
  // runs on 64 x 64 tiles instead
  if (a.Cout > 64 && small_grid(a)) return launch_conv<1, 1, 2, 2, EPI_PLAIN>(a, st);
  if (a.Cout > 64) return launch_conv<2, 2, 2, 2, EPI_PLAIN>(a, st);
  if (a.Cout > 32) return launch_conv<2, 2, 4, 1, EPI_PLAIN>(a, st);
  return launch_conv<1, 1, 4, 1, EPI_PLAIN>(a, st);
}

/* dsee_conv2d_fwd with both operands split into two scaled fp16 terms inside the kernel (3 fp16 MFMA products per
 * multiply-add instead of the fp32 MFMA; as accurate as an sgemm).  amax_in / amax_w: device maxima |in|, |w_packed| in
 * the 64-line layout dsee_absmax writes. */
int dsee_conv2d_fwd_f16x2(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                          const float* residual, int residual_ld, float* out, int act, float slope,
                          const float* amax_in, const float* amax_w, hipStream_t st) {
  return dsee_conv2d_fwd_f16x2_amax(g, in, w_packed, bias, residual, residual_ld, out, act, slope, amax_in, amax_w, nullptr, 0, st);
}

int dsee_conv2d_fwd_f16x2_amax(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                               const float* residual, int residual_ld, float* out, int act, float slope,
                               const float* amax_in, const float* amax_w, float* amax_out, int flags, hipStream_t st) {
  ConvArgs a = {};
  a.amax_out = amax_out;
  int rc = fill_geom(a, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && w_packed && out && amax_in && amax_w);
  DSEE_CHECK_ARG(act != DSEE_ACT_MASK || residual != nullptr);
  a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out; a.act = act; a.slope = slope;
  a.res_ld = residual_ld > 0 ? residual_ld : a.Cout;
  a.amax_a = amax_in; a.amax_w = amax_w;
  // (64-column tiles only: the 128-column instantiation needs 256 accumulator + fragment registers and spills at two blocks per
  // CU; a wider layer runs two column blocks per patch, which converts the patch twice -- still 4.5x less than once per tap)
  if (!(flags & DSEE_CONV_NO_HALO) && c4_f16_ok(a)) return launch_conv3x3_c4_f16(a, st);
  if (!(flags & DSEE_CONV_NO_HALO) && halo_f16_ok(a) && (long)(a.M / 128) * dsee_cdiv(a.Cout, 64) >= 256)
    return launch_conv_halo_f16<1>(a, st);
  // round 6: a 128 x 128 grid that would leave most of the 256 CUs idle (the discriminator's 17^2 / 33^2 layers: 19-70 workgroups)
  // runs on 64 x 64 tiles instead
  if (a.Cout > 64 && small_grid(a)) return launch_conv<1, 1, 2, 2, EPI_PLAIN>(a, st);
  if (a.Cout > 64) return launch_conv<2, 2, 2, 2, EPI_PLAIN>(a, st);
  if (a.Cout > 32) return launch_conv<2, 2, 4, 1, EPI_PLAIN>(a, st);
  return launch_conv<1, 1, 4, 1, EPI_PLAIN>(a, st);
}

/* Grouped GEMM on the implicit-GEMM kernel: a 1x1 "convolution" whose image n multiplies the weight matrix
 * w_packed + n * group_stride (floats).  Used for the 36 Winograd-domain GEMMs (image = transform position). */
int dsee_conv2d_fwd_grouped(const dsee_conv_geom* g, const float* in, const float* w_packed, long group_stride,
                            float* out, hipStream_t st) {
  ConvArgs a = {};
  int rc = fill_geom(a, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && w_packed && out && g->KH == 1 && g->KW == 1 && g->korder == 1 && g->mul == 1);
  DSEE_CHECK_ARG((g->Ho * g->Wo) % 128 == 0 && g->Cout > 64);  // a 128-row tile never straddles two groups
  a.in = in; a.w = w_packed; a.out = out; a.act = DSEE_ACT_NONE; a.res_ld = a.Cout;
  a.wgroup_stride = group_stride;
  return launch_conv<2, 2, 2, 2, EPI_PLAIN>(a, st);
}

int dsee_conv2d_modulate_fwd(const dsee_conv_geom* g, const float* in, const float* w_packed,
                             const float* style_table, int shared_cin, const float* bias_packed, const float* x,
                             const float* mean, const float* invstd, float* out_h, float* out_scale, int C,
                             float add_one, float slope, hipStream_t st) {
  ConvArgs a = {};
  int rc = fill_geom(a, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && x && mean && invstd && out_h && out_scale && C > 0);
  DSEE_CHECK_ARG(g->Cout >= (C + 63) / 64 * 128);  // packed gamma/beta rows
  a.in = in; a.w = w_packed; a.bias = bias_packed; a.out = out_h; a.mx = x; a.mean = mean; a.invstd = invstd;
  a.scale_out = out_scale; a.add_one = add_one; a.C = C; a.slope = slope;
  if (style_table) {
    // the last 32 input channels are the one-hot label (19 padded to 32); their weights are per image
    DSEE_CHECK_ARG(g->korder == 1 && shared_cin % 32 == 0 && shared_cin + 32 == g->Cin);
    DSEE_CHECK_ARG((g->Ho * g->Wo) % 128 == 0);  // an M tile must not straddle two images
    DSEE_CHECK_ARG(shared_cin == 0 || w_packed != nullptr);
    a.wt = style_table;
    a.nk_shared = shared_cin / 32 * g->KH * g->KW;
    a.wt_rows = dsee_conv_wrows(g->Cout);
    a.wstride = dsee_conv_kpad(g->KH, g->KW, shared_cin);
    if (!a.w) a.w = style_table;  // never dereferenced when nk_shared == 0
  } else {
    DSEE_CHECK_ARG(w_packed != nullptr);
  }
  return launch_conv<2, 2, 2, 2, EPI_MODULATE>(a, st);
}

static int wgrad_splits(int M, int tiles) {
  int s = 2048 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  int maxs = (M + 1023) / 1024;  // at least 1024 pixels per split
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return s;
}

size_t dsee_conv2d_wgrad_workspace(const dsee_conv_geom* g) {
  if (!g) return 0;
  const int Kpad = dsee_conv_kpad(g->KH, g->KW, g->Cin);  // upper bound for any Cin_real
  const int rows = g->Cout;
  const long M = (long)g->N * g->Ho * g->Wo;
  const int tiles = dsee_cdiv(Kpad, 128) * dsee_cdiv(rows, 128);
  const int S = wgrad_splits((int)M, tiles);
  return (size_t)S * rows * Kpad * sizeof(float);
}

static int wgrad_launch(WgradArgs& a, int S, hipStream_t st) {
  const int tx = dsee_cdiv(a.Kuse - a.Kstart, 128), ty = dsee_cdiv(a.rows, 128);
  const bool f16 = a.amax_dout && a.amax_in;
  const size_t lds = f16 ? (size_t)4 * 32 * WROWB : (size_t)4 * 32 * WLD * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * 32 * WLD * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * 32 * WLD * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<0, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * WROWB);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * WROWB);
    attr_done = true;
  }
  // fast path: stride-1 "same" conv, rows at least 32 wide, everything addressable with 32-bit byte offsets
  const long in_bytes = (long)a.N * a.Hi * a.Wi * a.Cin * 4, out_bytes = (long)a.M * a.Cout * 4;
  long mn = 0;
  for (int kh = 0; kh < a.KH; kh += (a.KH > 1 ? a.KH - 1 : 1))
    for (int kw = 0; kw < a.KW; kw += (a.KW > 1 ? a.KW - 1 : 1)) {
      const long t = ((long)(a.off + kh * a.kdir) * a.Wi + (a.off + kw * a.kdir)) * a.Cin * 4;
      if (t < mn) mn = t;
    }
  a.margin = (int)(-mn);
  const bool fast = a.mul == 1 && a.dshift == 0 && a.ups == 0 && a.Hi == a.Ho && a.Wi == a.Wo && a.Wo >= 32 &&
                    in_bytes + a.margin + 65536 < 0xFFFFFFFEL && out_bytes + 65536 < 0xFFFFFFFEL && -mn < (1L << 30);
  if (f16) {
    if (fast)
      conv_wgrad_kernel<1, true><<<dim3(tx, ty, S), 256, lds, st>>>(a);
    else
      conv_wgrad_kernel<0, true><<<dim3(tx, ty, S), 256, lds, st>>>(a);
  } else if (fast) {
    conv_wgrad_kernel<1><<<dim3(tx, ty, S), 256, lds, st>>>(a);
  } else {
    conv_wgrad_kernel<0><<<dim3(tx, ty, S), 256, lds, st>>>(a);
  }
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

static void wgrad_fill(WgradArgs& a, const ConvArgs& c, const float* in, const float* dout, float* workspace) {
  a.dout = dout; a.in = in; a.slab = workspace;
  a.N = c.N; a.Hi = c.Hi; a.Wi = c.Wi; a.Cin = c.Cin; a.Ho = c.Ho; a.Wo = c.Wo; a.Cout = c.Cout;
  a.KH = c.KH; a.KW = c.KW; a.Ktot = c.Ktot; a.Kpad = c.Kpad;
  a.mul = c.mul; a.off = c.off; a.kdir = c.kdir; a.dshift = c.dshift; a.ups = c.ups;
  a.M = c.M; a.rows = c.Cout;
}

static int conv2d_wgrad_impl(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                             size_t workspace_bytes, float* dw_oihw, int Cout_real, int Cin_first, int Cin_real,
                             const float* amax_in, const float* amax_dout, hipStream_t st) {
  ConvArgs c = {};
  int rc = fill_geom(c, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && dout && workspace && dw_oihw);
  DSEE_CHECK_ARG(Cout_real <= g->Cout && Cin_first >= 0 && Cin_first + Cin_real <= g->Cin);
  DSEE_CHECK_ARG(Cin_first == 0 || (g->korder == 1 && Cin_first % 32 == 0));
  DSEE_CHECK_ARG(workspace_bytes >= dsee_conv2d_wgrad_workspace(g));
  WgradArgs a = {};
  wgrad_fill(a, c, in, dout, workspace);
  a.amax_in = amax_in; a.amax_dout = amax_dout;
  a.korder = (g->korder == 1 && c.Cin % 32 == 0) ? 1 : 0;
  // chunk-major slabs: only the channel chunks [Cin_first/32, ceil((Cin_first+Cin_real)/32)) are computed
  a.Kuse = a.korder == 1 ? (Cin_first + Cin_real + 31) / 32 * 32 * a.KH * a.KW : a.Ktot;
  a.Kstart = a.korder == 1 ? Cin_first / 32 * 32 * a.KH * a.KW / 128 * 128 : 0;
  const int ty = dsee_cdiv(a.rows, 128);
  const int S = wgrad_splits(a.M, dsee_cdiv(a.Kpad, 128) * ty);  // same split count as the workspace query
  a.msplit = ((a.M + S - 1) / S + 31) / 32 * 32;
  rc = wgrad_launch(a, S, st);
  if (rc) return rc;
  const long total = (long)Cout_real * Cin_real * a.KH * a.KW;
  wgrad_reduce_unpack_kernel<<<(int)min(4096L, (total + 255) / 256), 256, 0, st>>>(
      workspace, dw_oihw, S, a.rows, a.Kpad, Cout_real, Cin_real, a.KH, a.KW, a.Cin, a.korder, Cin_first);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_conv2d_wgrad(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                      size_t workspace_bytes, float* dw_oihw, int Cout_real, int Cin_first, int Cin_real,
                      hipStream_t st) {
  return conv2d_wgrad_impl(g, in, dout, workspace, workspace_bytes, dw_oihw, Cout_real, Cin_first, Cin_real, nullptr,
                           nullptr, st);
}

/* dsee_conv2d_wgrad with both operands split into two scaled fp16 terms inside the kernel (see dsee_conv2d_fwd_f16x2);
 * amax_in / amax_dout: device maxima |in|, |dout| in the 64-line layout dsee_absmax writes. */
int dsee_conv2d_wgrad_f16x2(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                            size_t workspace_bytes, float* dw_oihw, int Cout_real, int Cin_first, int Cin_real,
                            const float* amax_in, const float* amax_dout, hipStream_t st) {
  DSEE_CHECK_ARG(amax_in && amax_dout);
  return conv2d_wgrad_impl(g, in, dout, workspace, workspace_bytes, dw_oihw, Cout_real, Cin_first, Cin_real, amax_in,
                           amax_dout, st);
}

// image-aligned split count for the table variant: S = N * sper, each split = P / sper pixels of one image
static int table_sper(const dsee_conv_geom* g) {
  const int P = g->Ho * g->Wo;
  const int tiles = dsee_cdiv(dsee_conv_kpad(g->KH, g->KW, g->Cin), 128) * dsee_cdiv(g->Cout, 128);
  int want = 2048 / (tiles * g->N > 0 ? tiles * g->N : 1);
  int sper = 1;
  while (sper * 2 <= want && P % (sper * 2) == 0 && (P / (sper * 2)) % 32 == 0 && P / (sper * 2) >= 1024) sper *= 2;
  return sper;
}

size_t dsee_conv2d_wgrad_table_workspace(const dsee_conv_geom* g) {
  if (!g) return 0;
  return (size_t)g->N * table_sper(g) * g->Cout * dsee_conv_kpad(g->KH, g->KW, g->Cin) * sizeof(float);
}

/* Weight gradient of the SEAN modulate GEMM whose last 32 input channels are the one-hot label with per-image
 * weights: ONE split-K launch with image-aligned splits; the shared columns (first Cin_shared channels) are summed
 * over every split into dw_oihw [rows][Cin_shared][3][3] (skipped if NULL / Cin_shared == 0), the one-hot columns
 * are summed per image into dtable [N][taps][rows][32]. */
int dsee_conv2d_wgrad_table(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                            size_t workspace_bytes, float* dw_oihw, int Cin_shared, float* dtable, int L,
                            hipStream_t st) {
  ConvArgs c = {};
  int rc = fill_geom(c, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && dout && workspace && dtable && g->korder == 1 && Cin_shared + 32 == g->Cin && L <= 32);
  DSEE_CHECK_ARG((g->Ho * g->Wo) % 32 == 0 && workspace_bytes >= dsee_conv2d_wgrad_table_workspace(g));
  WgradArgs a = {};
  wgrad_fill(a, c, in, dout, workspace);
  a.korder = 1;
  a.Kuse = a.Ktot;
  a.Kstart = 0;
  const int sper = table_sper(g), S = g->N * sper;
  a.msplit = g->Ho * g->Wo / sper;
  rc = wgrad_launch(a, S, st);
  if (rc) return rc;
  const int taps = a.KH * a.KW;
  if (dw_oihw && Cin_shared > 0) {
    const long total = (long)a.rows * Cin_shared * taps;
    wgrad_reduce_unpack_kernel<<<(int)min(4096L, (total + 255) / 256), 256, 0, st>>>(
        workspace, dw_oihw, S, a.rows, a.Kpad, a.rows, Cin_shared, a.KH, a.KW, a.Cin, 1, 0);
    DSEE_LAUNCH_CHECK();
  }
  const long tt = (long)g->N * taps * a.rows * 32;
  wgrad_table_unpack_kernel<<<(int)min(4096L, (tt + 255) / 256), 256, 0, st>>>(
      workspace, dtable, g->N, sper, a.rows, a.Kpad, Cin_shared / 32 * taps * 32, taps, L);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

static void launch_wgrad_finalize(const float* ws, float* dw, int sper, int rows, int Kpad, int Cout, int Cin,
                                  hipStream_t st) {
  const int grid = (int)min(4096L, ((long)Cout * Cin + 255) / 256);
  switch (sper) {
    case 1: wino43_wgrad_finalize_kernel<1><<<grid, 256, 0, st>>>(ws, dw, sper, rows, Kpad, Cout, Cin); break;
    case 2: wino43_wgrad_finalize_kernel<2><<<grid, 256, 0, st>>>(ws, dw, sper, rows, Kpad, Cout, Cin); break;
    case 4: wino43_wgrad_finalize_kernel<4><<<grid, 256, 0, st>>>(ws, dw, sper, rows, Kpad, Cout, Cin); break;
    case 8: wino43_wgrad_finalize_kernel<8><<<grid, 256, 0, st>>>(ws, dw, sper, rows, Kpad, Cout, Cin); break;
    default: wino43_wgrad_finalize_kernel<0><<<grid, 256, 0, st>>>(ws, dw, sper, rows, Kpad, Cout, Cin);
  }
}

static void launch_wgrad_table_finalize(const float* ws, float* dw2a, float* dtable, int N, int sper, int rows, int Kpad,
                                        int ca, int L, hipStream_t st) {
  const long total = (long)rows * ca + (long)N * rows * 32;
  const int grid = (int)min(4096L, (total + 255) / 256);
  switch (sper) {
    case 1: wino43_wgrad_table_finalize_kernel<1><<<grid, 256, 0, st>>>(ws, dw2a, dtable, N, sper, rows, Kpad, ca, L); break;
    case 2: wino43_wgrad_table_finalize_kernel<2><<<grid, 256, 0, st>>>(ws, dw2a, dtable, N, sper, rows, Kpad, ca, L); break;
    case 4: wino43_wgrad_table_finalize_kernel<4><<<grid, 256, 0, st>>>(ws, dw2a, dtable, N, sper, rows, Kpad, ca, L); break;
    default: wino43_wgrad_table_finalize_kernel<0><<<grid, 256, 0, st>>>(ws, dw2a, dtable, N, sper, rows, Kpad, ca, L);
  }
}

// split count per transform position for the Winograd weight gradient (36 * sper splits in all)
static int wino_sper(long T, int Cin_s, int Cout_s, int groups_per_xi = 1) {
  const int tiles = dsee_cdiv(dsee_conv_kpad(1, 1, Cin_s), 128) * dsee_cdiv(Cout_s, 128) * groups_per_xi;
  const int want = 4608 / (36 * tiles) > 1 ? 4608 / (36 * tiles) : 1;
  int sper = 1;
  while (sper * 2 <= want && T % (sper * 2) == 0 && (T / (sper * 2)) % 32 == 0 && T / (sper * 2) >= 512) sper *= 2;
  return sper;
}

size_t dsee_wino43_wgrad_workspace(long T, int Cin_s, int Cout_s) {
  return (size_t)36 * wino_sper(T, Cin_s, Cout_s) * Cout_s * dsee_conv_kpad(1, 1, Cin_s) * sizeof(float);
}

/* Weight gradient of a 3x3 / stride-1 / pad-1 convolution in the Winograd F(4x4,3x3) domain:
 *   dU[xi] = dM[xi]^T V[xi]  (36 reductions over the T tiles, one split-K MFMA launch),  dw = G^T dU G.
 * V  [36][T][Cin_s]  = dsee_wino43_input(x),  dM [36][T][Cout_s] = dsee_wino43_dout(dy);  T % 32 == 0. */
int dsee_wino43_wgrad(const float* V, const float* dM, float* workspace, size_t workspace_bytes, float* dw_oihw, long T,
                      int Cin_s, int Cout_s, int Cout, int Cin, int split, const float* amax_v, const float* amax_dm,
                      hipStream_t st) {
  DSEE_CHECK_ARG(V && dM && workspace && dw_oihw && T % 32 == 0 && Cin_s % 4 == 0 && Cout_s % 4 == 0);
  DSEE_CHECK_ARG(split < 3 || (amax_v && amax_dm));
  DSEE_CHECK_ARG(Cout <= Cout_s && Cin <= Cin_s && 36 * T < (1L << 31));
  DSEE_CHECK_ARG(workspace_bytes >= dsee_wino43_wgrad_workspace(T, Cin_s, Cout_s));
  if (split) {
    // operands are the transposed bf16x3 layouts of dsee_wino43_dout_split_t / dsee_wino43_input_split_t
    DSEE_CHECK_ARG(Cout_s % 128 == 0 && Cin_s % 32 == 0);
    const int sper = wino_sper(T, Cin_s, Cout_s), Kpad = dsee_conv_kpad(1, 1, Cin_s);
    // split == 2: V / dM are the plain fp32 transforms, transposed + split inside the GEMM
    // split == 3: the same with two-term fp16 splits (3 MFMA products), operand scales from max |dM|, max |V|
    // split == 5: V is the pre-split V2 of the forward pass (dsee_wino43_input_f16x2, bound DSEE_WINO_V_BOUND), amax_v = max |x|
    // split == 6: dM is pre-split as well (dsee_wino43_dout_f16x2, bound DSEE_WINO_DM_BOUND), amax_dm = max |dY|
    // split == 7: 16-bit storage mode -- both operands packed one-term (dsee_wino43_dout_f16p / dsee_wino43_input_f16p)
    int rc = split == 7   ? dsee_gemm_f16p_tn_pqpre(dM, V, workspace, 36, T, Cout_s, Cin_s, Kpad, sper, amax_dm,
                                                    DSEE_WINO_DM_BOUND, amax_v, DSEE_WINO_V_BOUND, st)
             : split == 6 ? dsee_gemm_f16x2_tn_pqpre(dM, V, workspace, 36, T, Cout_s, Cin_s, Kpad, sper, amax_dm,
                                                     DSEE_WINO_DM_BOUND, amax_v, DSEE_WINO_V_BOUND, st)
             : split == 5 ? dsee_gemm_f16x2_tn_qpre(dM, V, workspace, 36, T, Cout_s, Cin_s, Kpad, sper, amax_dm, amax_v,
                                                    DSEE_WINO_V_BOUND, st)
             : split == 4 ? dsee_gemm_f16_tn_f32(dM, V, workspace, 36, T, Cout_s, Cin_s, Kpad, sper, amax_dm, amax_v, st)
             : split == 3 ? dsee_gemm_f16x2_tn_f32(dM, V, workspace, 36, T, Cout_s, Cin_s, Kpad, sper, amax_dm, amax_v, st)
             : split == 2 ? dsee_gemm_bf16x3_tn_f32(dM, V, workspace, 36, T, Cout_s, Cin_s, Kpad, sper, st)
                          : dsee_gemm_bf16x3_tn(dM, V, workspace, 36, T, Cout_s, Cin_s, Kpad, sper, st);
    if (rc) return rc;
    launch_wgrad_finalize(workspace, dw_oihw, sper, Cout_s, Kpad, Cout, Cin, st);
    DSEE_LAUNCH_CHECK();
    return DSEE_OK;
  }
  WgradArgs a = {};
  a.dout = dM; a.in = V; a.slab = workspace;
  a.N = 1; a.Hi = a.Ho = (int)(36 * T / 32); a.Wi = a.Wo = 32; a.Cin = Cin_s; a.Cout = Cout_s;
  a.KH = a.KW = 1; a.Ktot = Cin_s; a.Kpad = dsee_conv_kpad(1, 1, Cin_s);
  a.mul = 1; a.off = 0; a.kdir = 1; a.dshift = 0; a.ups = 0;
  a.M = (int)(36 * T); a.rows = Cout_s;
  a.korder = 0; a.Kuse = a.Ktot; a.Kstart = 0;
  const int sper = wino_sper(T, Cin_s, Cout_s);
  a.msplit = (int)(T / sper);
  int rc = wgrad_launch(a, 36 * sper, st);
  if (rc) return rc;
  launch_wgrad_finalize(workspace, dw_oihw, sper, a.rows, a.Kpad, Cout, Cin, st);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

size_t dsee_wino43_wgrad_table_workspace(long T, int N, int ca, int rows) {
  return (size_t)36 * N * wino_sper(T / N, ca + 32, rows, N) * rows * dsee_conv_kpad(1, 1, ca + 32) * sizeof(float);
}

/* Weight gradient of the SEAN gamma/beta GEMM in the Winograd domain (per-image groups, see dsee_wino43_weights_table):
 * V [36][T][ca+32] = dsee_wino43_input(cat), dM [36][T][rows] = dsee_wino43_dout(dgb), tiles image-major (T/N each,
 * multiple of 32).  Writes dw2a [rows][ca][3][3] (NULL / ca == 0: skipped) and dtable [N][9][rows][32]. */
int dsee_wino43_wgrad_table(const float* V, const float* dM, float* workspace, size_t workspace_bytes, float* dw2a,
                            float* dtable, long T, int N, int ca, int rows, int L, int split, const float* amax_v,
                            const float* amax_dm, hipStream_t st) {
  DSEE_CHECK_ARG(V && dM && workspace && dtable && N > 0 && T % N == 0 && (T / N) % 32 == 0 && ca % 32 == 0);
  DSEE_CHECK_ARG(split < 3 || (amax_v && amax_dm));
  DSEE_CHECK_ARG(rows % 4 == 0 && L <= 32 && 36 * T < (1L << 31));
  DSEE_CHECK_ARG(workspace_bytes >= dsee_wino43_wgrad_table_workspace(T, N, ca, rows));
  const int ld = ca + 32;
  if (split) {
    DSEE_CHECK_ARG(rows % 128 == 0);
    const int sper = wino_sper(T / N, ld, rows, N), Kpad = dsee_conv_kpad(1, 1, ld);
    int rc = split == 7   ? dsee_gemm_f16p_tn_pqpre(dM, V, workspace, 36 * N, T / N, rows, ld, Kpad, sper, amax_dm,
                                                    DSEE_WINO_DM_BOUND, amax_v, DSEE_WINO_V_BOUND, st)
             : split == 6 ? dsee_gemm_f16x2_tn_pqpre(dM, V, workspace, 36 * N, T / N, rows, ld, Kpad, sper, amax_dm,
                                                     DSEE_WINO_DM_BOUND, amax_v, DSEE_WINO_V_BOUND, st)
             : split == 5 ? dsee_gemm_f16x2_tn_qpre(dM, V, workspace, 36 * N, T / N, rows, ld, Kpad, sper, amax_dm, amax_v,
                                                    DSEE_WINO_V_BOUND, st)
             : split == 4 ? dsee_gemm_f16_tn_f32(dM, V, workspace, 36 * N, T / N, rows, ld, Kpad, sper, amax_dm, amax_v, st)
             : split == 3 ? dsee_gemm_f16x2_tn_f32(dM, V, workspace, 36 * N, T / N, rows, ld, Kpad, sper, amax_dm, amax_v, st)
             : split == 2 ? dsee_gemm_bf16x3_tn_f32(dM, V, workspace, 36 * N, T / N, rows, ld, Kpad, sper, st)
                          : dsee_gemm_bf16x3_tn(dM, V, workspace, 36 * N, T / N, rows, ld, Kpad, sper, st);
    if (rc) return rc;
    launch_wgrad_table_finalize(workspace, ca > 0 ? dw2a : nullptr, dtable, N, sper, rows, Kpad, ca, L, st);
    DSEE_LAUNCH_CHECK();
    return DSEE_OK;
  }
  WgradArgs a = {};
  a.dout = dM; a.in = V; a.slab = workspace;
  a.N = 1; a.Hi = a.Ho = (int)(36 * T / 32); a.Wi = a.Wo = 32; a.Cin = ld; a.Cout = rows;
  a.KH = a.KW = 1; a.Ktot = ld; a.Kpad = dsee_conv_kpad(1, 1, ld);
  a.mul = 1; a.off = 0; a.kdir = 1; a.dshift = 0; a.ups = 0;
  a.M = (int)(36 * T); a.rows = rows;
  a.korder = 0; a.Kuse = a.Ktot; a.Kstart = 0;
  const int sper = wino_sper(T / N, ld, rows, N);
  a.msplit = (int)(T / N / sper);
  int rc = wgrad_launch(a, 36 * N * sper, st);
  if (rc) return rc;
  launch_wgrad_table_finalize(workspace, ca > 0 ? dw2a : nullptr, dtable, N, sper, rows, a.Kpad, ca, L, st);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
