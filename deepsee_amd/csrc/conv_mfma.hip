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
#include "dsee_common.h"

namespace {

struct ConvArgs {
  const float* in;
  const float* w;      // packed [wrows][Kpad], k = (kh*KW+kw)*Cin + c
  const float* bias;   // [Cout] or null (packed order for the modulate epilogue)
  const float* res;    // residual [M][Cout] or null
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
  int mul, off, kdir, dshift, ups;
  int act;
  float slope;
  int M;
};

constexpr int BK = 32;
constexpr int LDK = 36;  // padded LDS row (floats)
constexpr int FLUSH = 4; // K-slabs per partial-accumulator chain (power of two)

enum { EPI_PLAIN = 0, EPI_MODULATE = 1 };

template <int MT, int NT, int WM, int WN, int EPI>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int BM = MT * WM * 32, BN = NT * WN * 32;
  constexpr int A_CH = BM * 8 / 256, B_CH = (BN * 8 + 255) / 256;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int bm = blockIdx.x, bn = blockIdx.y;
  const int chunk = tid & 7, lrow = tid >> 3;

  int a_n[A_CH], a_oh[A_CH], a_ow[A_CH];
  bool a_ok[A_CH];
#pragma unroll
  for (int j = 0; j < A_CH; ++j) {
    int m = bm * BM + lrow + 32 * j;
    a_ok[j] = m < a.M;
    int mm = a_ok[j] ? m : 0;
    int ow = mm % a.Wo;
    int t = mm / a.Wo;
    a_ow[j] = ow;
    a_oh[j] = t % a.Ho;
    a_n[j] = t / a.Ho;
  }
  const int Hl = a.Hi << a.ups, Wl = a.Wi << a.ups;
  const int dmask = (1 << a.dshift) - 1;

  f32x4 ra[A_CH], rb[B_CH];
  auto load_tile = [&](int kt) {
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
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(a_n[j] * a.Hi + ph) * a.Wi + pw) * a.Cin + c);
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      int row = lrow + 32 * j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < BN) v = *reinterpret_cast<const f32x4*>(a.w + (size_t)(bn * BN + row) * a.Kpad + k);
      rb[j] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * BM * LDK;
    float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) *reinterpret_cast<f32x4*>(Ab + (lrow + 32 * j) * LDK + chunk * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      int row = lrow + 32 * j;
      if (row < BN) *reinterpret_cast<f32x4*>(Bb + row * LDK + chunk * 4) = rb[j];
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

  const int nk = a.Kpad / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* Ac = As + cur * BM * LDK + (wm * MT * 32 + frow) * LDK + fk;
    const float* Bc = Bs + cur * BN * LDK + (wn * NT * 32 + frow) * LDK + fk;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDK + kk * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDK + kk * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            part[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], bf[j][q], part[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
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
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue.  C/D map of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const int rbase = bm * BM + wm * MT * 32 + 4 * (lane >> 5);
  if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = bn * BN + wn * NT * 32 + j * 32 + (lane & 31);
      if (col >= a.Cout) continue;
      const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
          if (row < a.M) {
            float v = acc[i][j][r] + b;
            if (a.res) v += a.res[(size_t)row * a.Cout + col];
            a.out[(size_t)row * a.Cout + col] = dsee_act(v, a.act, a.slope);
          }
        }
    }
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
          const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
          if (row < a.M) {
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

// ---------------------------------------------------------------- weight gradient (split-K)
struct WgradArgs {
  const float* dout;  // [M][Cout]
  const float* in;    // [N][Hi][Wi][Cin]
  float* slab;        // [S][rows][Kpad]
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int KH, KW, Ktot, Kpad;
  int mul, off, kdir, dshift, ups;
  int M, msplit, rows;
};

constexpr int WLD = 132;  // padded LDS row for the 32 x 128 wgrad tiles

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  // D[i = cout][j = k'] = sum over pixels.  A'[px][co] = dout tile, B'[px][k'] = shifted input tile.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][32][WLD]
  float* Bs = smem + 2 * 32 * WLD;  // [2][32][WLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int kx = blockIdx.x, cy = blockIdx.y, z = blockIdx.z;
  const int chunk = tid & 31, lrow = tid >> 5;  // 8 rows per pass, 4 passes
  const int m0 = z * a.msplit;
  const int m1 = min(a.M, m0 + a.msplit);
  // B' column (k') owned by this thread is fixed for the whole kernel
  const int kq = kx * 128 + chunk * 4;
  const bool kok = kq < a.Ktot;
  const int tap = kok ? kq / a.Cin : 0;
  const int cch = kq - tap * a.Cin;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int co = cy * 128 + chunk * 4;
  const bool cok = co < a.Cout;
  const int Hl = a.Hi << a.ups, Wl = a.Wi << a.ups;
  const int dmask = (1 << a.dshift) - 1;

  f32x4 ra[4], rb[4];
  auto load_tile = [&](int mb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mb + lrow + 8 * j;
      const bool mok = m < m1;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (mok && cok) va = *reinterpret_cast<const f32x4*>(a.dout + (size_t)m * a.Cout + co);
      if (mok && kok) {
        const int ow = m % a.Wo;
        const int t = m / a.Wo;
        const int oh = t % a.Ho;
        const int n = t / a.Ho;
        int ph = oh * a.mul + a.off + kh * a.kdir;
        int pw = ow * a.mul + a.off + kw * a.kdir;
        bool ok = ph >= 0 && pw >= 0 && ((ph | pw) & dmask) == 0;
        ph >>= a.dshift;
        pw >>= a.dshift;
        ok = ok && ph < Hl && pw < Wl;
        ph >>= a.ups;
        pw >>= a.ups;
        if (ok) vb = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(n * a.Hi + ph) * a.Wi + pw) * a.Cin + cch);
      }
      ra[j] = va;
      rb[j] = vb;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<f32x4*>(As + (buf * 32 + lrow + 8 * j) * WLD + chunk * 4) = ra[j];
      *reinterpret_cast<f32x4*>(Bs + (buf * 32 + lrow + 8 * j) * WLD + chunk * 4) = rb[j];
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
  if (nk > 0) {
    load_tile(m0);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  const int fcol = lane & 31, fk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(m0 + (kt + 1) * 32);
    const float* Ac = As + cur * 32 * WLD + wm * 64 + fcol;
    const float* Bc = Bs + cur * 32 * WLD + wn * 64 + fcol;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = kk * 2 + fk;
      const float a0 = Ac[k * WLD], a1 = Ac[k * WLD + 32];
      const float b0 = Bc[k * WLD], b1 = Bc[k * WLD + 32];
      part[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, part[0][0], 0, 0, 0);
      part[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, part[0][1], 0, 0, 0);
      part[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, part[1][0], 0, 0, 0);
      part[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, part[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
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
    __syncthreads();
    cur ^= 1;
  }
  float* slab = a.slab + (size_t)z * a.rows * a.Kpad;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = kx * 128 + wn * 64 + j * 32 + (lane & 31);
    if (col >= a.Kpad) continue;
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
__global__ void pack_fwd_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale_num,
                                const float* __restrict__ inv_scale_den, float* __restrict__ p, int Cout, int Cin,
                                int KH, int KW, int Cin_s, int rows, int Kpad) {
  // p[row][(kh*KW+kw)*Cin_s + ci] = w[row][ci][kh][kw] * s
  const long total = (long)rows * Kpad;
  float s = 1.f;
  if (inv_scale_den) s = (inv_scale_num ? *inv_scale_num : 1.f) / *inv_scale_den;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / Kpad), k = (int)(i % Kpad);
    const int tap = k / Cin_s, ci = k % Cin_s;
    float v = 0.f;
    if (row < Cout && tap < KH * KW && ci < Cin) v = w[((size_t)row * Cin + ci) * KH * KW + tap] * s;
    p[i] = v;
  }
}

__global__ void pack_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale_num,
                                  const float* __restrict__ inv_scale_den, float* __restrict__ p, int Cout, int Cin,
                                  int KH, int KW, int Cout_s, int rows, int Kpad) {
  // p[ci][(kh*KW+kw)*Cout_s + co] = w[co][ci][kh][kw] * s      (rows index ci)
  const long total = (long)rows * Kpad;
  float s = 1.f;
  if (inv_scale_den) s = (inv_scale_num ? *inv_scale_num : 1.f) / *inv_scale_den;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / Kpad), k = (int)(i % Kpad);
    const int tap = k / Cout_s, co = k % Cout_s;
    float v = 0.f;
    if (row < Cin && tap < KH * KW && co < Cout) v = w[((size_t)co * Cin + row) * KH * KW + tap] * s;
    p[i] = v;
  }
}

__global__ void wgrad_reduce_unpack_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int rows,
                                           int Kpad, int Cout, int Cin, int KH, int KW, int Cin_s) {
  // dw[co][ci][kh][kw] = sum_s slab[s][co][(kh*KW+kw)*Cin_s + ci]   (fixed order => deterministic)
  const long total = (long)Cout * Cin * KH * KW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % (KH * KW));
    const long t = i / (KH * KW);
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    const size_t o = (size_t)co * Kpad + (size_t)tap * Cin_s + ci;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += slab[(size_t)s * rows * Kpad + o];
    dw[i] = v;
  }
}

template <int MT, int NT, int WM, int WN, int EPI>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = MT * WM * 32, BN = NT * WN * 32;
  const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<MT, NT, WM, WN, EPI>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(dsee_cdiv(a.M, BM), EPI == EPI_MODULATE ? dsee_cdiv(a.C, BN / 2) : dsee_cdiv(a.Cout, BN));
  conv_igemm_kernel<MT, NT, WM, WN, EPI><<<grid, 256, lds, st>>>(a);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int fill_geom(ConvArgs& a, const dsee_conv_geom* g) {
  DSEE_CHECK_ARG(g != nullptr);
  DSEE_CHECK_ARG(g->Cin % 4 == 0 && g->Cout % 4 == 0);
  DSEE_CHECK_ARG(g->N > 0 && g->Hi > 0 && g->Wi > 0 && g->Ho > 0 && g->Wo > 0 && g->KH > 0 && g->KW > 0);
  DSEE_CHECK_ARG(g->dshift >= 0 && g->dshift <= 3 && g->ups >= 0 && g->ups <= 3);
  a.N = g->N; a.Hi = g->Hi; a.Wi = g->Wi; a.Cin = g->Cin; a.Ho = g->Ho; a.Wo = g->Wo; a.Cout = g->Cout;
  a.KH = g->KH; a.KW = g->KW; a.Ktot = g->KH * g->KW * g->Cin; a.Kpad = (a.Ktot + 31) / 32 * 32;
  a.mul = g->mul; a.off = g->off; a.kdir = g->kdir; a.dshift = g->dshift; a.ups = g->ups;
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
                         int Cin, int KH, int KW, int Cin_stored, hipStream_t st) {
  DSEE_CHECK_ARG(w_oihw && packed && Cin_stored >= Cin && Cin_stored % 4 == 0);
  const int rows = dsee_conv_wrows(Cout), Kpad = dsee_conv_kpad(KH, KW, Cin_stored);
  const long total = (long)rows * Kpad;
  pack_fwd_kernel<<<(int)min(4096L, (total + 255) / 256), 256, 0, st>>>(w_oihw, scale_num, scale_den, packed, Cout, Cin,
                                                                        KH, KW, Cin_stored, rows, Kpad);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_pack_weight_dgrad(const float* w_oihw, const float* scale_num, const float* scale_den, float* packed,
                           int Cout, int Cin, int KH, int KW, int Cout_stored, hipStream_t st) {
  DSEE_CHECK_ARG(w_oihw && packed && Cout_stored >= Cout && Cout_stored % 4 == 0);
  const int rows = dsee_conv_wrows(Cin), Kpad = dsee_conv_kpad(KH, KW, Cout_stored);
  const long total = (long)rows * Kpad;
  pack_dgrad_kernel<<<(int)min(4096L, (total + 255) / 256), 256, 0, st>>>(w_oihw, scale_num, scale_den, packed, Cout,
                                                                          Cin, KH, KW, Cout_stored, rows, Kpad);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_conv2d_fwd(const dsee_conv_geom* g, const float* in, const float* w_packed, const float* bias,
                    const float* residual, float* out, int act, float slope, hipStream_t st) {
  ConvArgs a = {};
  int rc = fill_geom(a, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && w_packed && out);
  a.in = in; a.w = w_packed; a.bias = bias; a.res = residual; a.out = out; a.act = act; a.slope = slope;
  if (a.Cout > 64) return launch_conv<2, 2, 2, 2, EPI_PLAIN>(a, st);
  if (a.Cout > 32) return launch_conv<2, 2, 4, 1, EPI_PLAIN>(a, st);
  return launch_conv<1, 1, 4, 1, EPI_PLAIN>(a, st);
}

int dsee_conv2d_modulate_fwd(const dsee_conv_geom* g, const float* in, const float* w_packed,
                             const float* bias_packed, const float* x, const float* mean, const float* invstd,
                             float* out_h, float* out_scale, int C, float add_one, float slope, hipStream_t st) {
  ConvArgs a = {};
  int rc = fill_geom(a, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && w_packed && x && mean && invstd && out_h && out_scale && C > 0);
  DSEE_CHECK_ARG(g->Cout >= (C + 63) / 64 * 128);  // packed gamma/beta rows
  a.in = in; a.w = w_packed; a.bias = bias_packed; a.out = out_h; a.mx = x; a.mean = mean; a.invstd = invstd;
  a.scale_out = out_scale; a.add_one = add_one; a.C = C; a.slope = slope;
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
  const int Kpad = dsee_conv_kpad(g->KH, g->KW, g->Cin);
  const int rows = g->Cout;
  const long M = (long)g->N * g->Ho * g->Wo;
  const int tiles = dsee_cdiv(Kpad, 128) * dsee_cdiv(rows, 128);
  const int S = wgrad_splits((int)M, tiles);
  return (size_t)S * rows * Kpad * sizeof(float);
}

int dsee_conv2d_wgrad(const dsee_conv_geom* g, const float* in, const float* dout, float* workspace,
                      size_t workspace_bytes, float* dw_oihw, int Cout_real, int Cin_real, hipStream_t st) {
  ConvArgs c = {};
  int rc = fill_geom(c, g);
  if (rc) return rc;
  DSEE_CHECK_ARG(in && dout && workspace && dw_oihw);
  DSEE_CHECK_ARG(Cout_real <= g->Cout && Cin_real <= g->Cin);
  DSEE_CHECK_ARG(workspace_bytes >= dsee_conv2d_wgrad_workspace(g));
  WgradArgs a = {};
  a.dout = dout; a.in = in; a.slab = workspace;
  a.N = c.N; a.Hi = c.Hi; a.Wi = c.Wi; a.Cin = c.Cin; a.Ho = c.Ho; a.Wo = c.Wo; a.Cout = c.Cout;
  a.KH = c.KH; a.KW = c.KW; a.Ktot = c.Ktot; a.Kpad = c.Kpad;
  a.mul = c.mul; a.off = c.off; a.kdir = c.kdir; a.dshift = c.dshift; a.ups = c.ups;
  a.M = c.M; a.rows = c.Cout;
  const int tx = dsee_cdiv(a.Kpad, 128), ty = dsee_cdiv(a.rows, 128);
  const int S = wgrad_splits(a.M, tx * ty);
  a.msplit = ((a.M + S - 1) / S + 31) / 32 * 32;
  const size_t lds = (size_t)4 * 32 * WLD * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    attr_done = true;
  }
  conv_wgrad_kernel<<<dim3(tx, ty, S), 256, lds, st>>>(a);
  DSEE_LAUNCH_CHECK();
  const long total = (long)Cout_real * Cin_real * a.KH * a.KW;
  wgrad_reduce_unpack_kernel<<<(int)min(4096L, (total + 255) / 256), 256, 0, st>>>(
      workspace, dw_oihw, S, a.rows, a.Kpad, Cout_real, Cin_real, a.KH, a.KW, a.Cin);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
