// Parameter-space glue of the SPADE / SEAN / PureSEAN normalisation as two kernels per direction instead of ~30 + ~45
// ATen launches per norm layer and pass (sigmoid, blends, cat, index_select, zeros, permute, pad ... and their autograd
// backwards):
//   scale = sigmoid(a_g) gamma_s + (1 - sigmoid(a_g)) gamma + 1,  offset = sigmoid(a_b) beta_s + (1 - sigmoid(a_b)) beta
// (normalization.py:208-213; SPADE :119; PureSEAN :286) is linear in the four convolutions, so the gamma/beta GEMM reads
// ONE weight set: rows packed as [32 gamma rows | the 32 beta rows of the same channels] (conv_mfma.hip EPI_MODULATE,
// spade_fused.hip), blended with the sigmoid weights.  dsee_sean_pack_fwd writes
//   w2a [rows][K][3][3]   weights over the 128-channel embedding       (modes 0 spade, 1 sean, 3 sean above max_fm_size)
//   wst [9*rows][S]       weights over the style vector, (tap, row)-major: the B operand of the style-table GEMM
//                         T[n*19 + r][tap*rows + row] = sum_s style[n][r][s] * wst[tap*rows + row][s]      (modes 1, 2)
//   b2  [rows]            packed biases
// and dsee_sean_pack_bwd maps their gradients back to the ten parameters (the two alpha gradients through a two-stage
// deterministic reduction).  dsee_style_table_layout[_bwd] turns the GEMM result into the [N][9][rows][32] per-image
// table the kernels read (19 regions padded to 32) and back.
#include "dsee_common.h"

namespace {

struct PackArgs {
  const float *wg, *wb, *wsg, *wsb, *bg, *bb, *bsg, *bsb, *ag, *ab;
  int mode, C, K, S, rows;
};

__device__ __forceinline__ float sigm(const float* a) { return a ? 1.f / (1.f + __expf(-*a)) : 0.f; }

// packed row p -> (beta?, channel)
__device__ __forceinline__ void unpack_row(int p, int& is_beta, int& ch) {
  const int b = p >> 7, rem = p & 127, w = rem >> 6, h = (rem >> 5) & 1, cc = rem & 31;
  is_beta = h;
  ch = b * 64 + w * 32 + cc;
}

__global__ __launch_bounds__(256) void sean_pack_fwd_kernel(PackArgs a, float* __restrict__ w2a, float* __restrict__ wst,
                                                            float* __restrict__ b2, float* __restrict__ amax) {
  float vmax = 0.f;   // max |w2a| for the fp16 operand scale of the gamma/beta GEMM
  const float sg = a.mode == 1 || a.mode == 3 ? sigm(a.ag) : 0.f, sb = a.mode == 1 || a.mode == 3 ? sigm(a.ab) : 0.f;
  const long na = w2a ? (long)a.rows * a.K : 0, ns = wst ? (long)a.rows * a.S : 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < na + ns + a.rows;
       i += (long)gridDim.x * blockDim.x) {
    if (i < na) {   // (row, k): 9 taps of the embedding weights
      const int p = (int)(i / a.K), k = (int)(i % a.K);
      int hb, ch;
      unpack_row(p, hb, ch);
      const float s = hb ? sb : sg;
      float v[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) v[t] = 0.f;
      if (ch < a.C) {
        if (a.mode == 2) {          // puresean above max_fm_size is folded by the caller; plain puresean has no w2a
        } else {
          const float* w = (hb ? a.wb : a.wg) + ((size_t)ch * a.K + k) * 9;
          const float c0 = a.mode == 0 ? 1.f : 1.f - s;
#pragma unroll
          for (int t = 0; t < 9; ++t) v[t] = c0 * w[t];
          if (a.mode == 3) {        // both halves read the same 128 channels: (1 - s) W + s W_style
            const float* ws = (hb ? a.wsb : a.wsg) + ((size_t)ch * a.S + k) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] += s * ws[t];
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        w2a[i * 9 + t] = v[t];
        vmax = fmaxf(vmax, fabsf(v[t]));
      }
    } else if (i < na + ns) {   // (row, s): 9 taps of the style weights, written (tap, row)-major
      const long j = i - na;
      const int p = (int)(j / a.S), sidx = (int)(j % a.S);
      int hb, ch;
      unpack_row(p, hb, ch);
      const float c1 = a.mode == 2 ? 1.f : (hb ? sb : sg);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float v = 0.f;
        if (ch < a.C) v = c1 * ((hb ? a.wsb : a.wsg)[((size_t)ch * a.S + sidx) * 9 + t]);
        wst[((size_t)t * a.rows + p) * a.S + sidx] = v;
      }
    } else {
      const int p = (int)(i - na - ns);
      int hb, ch;
      unpack_row(p, hb, ch);
      float v = 0.f;
      if (ch < a.C) {
        const float s = hb ? sb : sg;
        const float* bp = hb ? a.bb : a.bg;
        const float* bs = hb ? a.bsb : a.bsg;
        if (a.mode == 0) v = bp[ch];
        else if (a.mode == 2) v = bs[ch];
        else v = (1.f - s) * bp[ch] + s * bs[ch];
      }
      b2[p] = v;
    }
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);   // (amax is block-uniform)
}

// gradients of the parameters + per-block partial sums of the two alpha gradients
__global__ __launch_bounds__(256) void sean_pack_bwd_kernel(PackArgs a, const float* __restrict__ dw2a,
                                                            const float* __restrict__ dwst, const float* __restrict__ db2,
                                                            float* __restrict__ dwg, float* __restrict__ dwb,
                                                            float* __restrict__ dwsg, float* __restrict__ dwsb,
                                                            float* __restrict__ dbg, float* __restrict__ dbb,
                                                            float* __restrict__ dbsg, float* __restrict__ dbsb,
                                                            float* __restrict__ partial) {
  const bool blend = a.mode == 1 || a.mode == 3;
  const float sg = blend ? sigm(a.ag) : 0.f, sb = blend ? sigm(a.ab) : 0.f;
  const long na = dw2a ? (long)a.C * 2 * a.K : 0, ns = (dwst || a.mode == 3) ? (long)a.C * 2 * a.S : 0;
  float acc_g = 0.f, acc_b = 0.f;   // sum of d(out)/d(sigmoid) terms, gamma / beta half
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < na + ns + 2 * a.C;
       i += (long)gridDim.x * blockDim.x) {
    if (i < na) {   // (half, channel, k) of the embedding weights
      const int k = (int)(i % a.K);
      const long r = i / a.K;
      const int ch = (int)(r % a.C), hb = (int)(r / a.C);
      const int p = (ch >> 6) * 128 + ((ch & 63) >> 5) * 64 + hb * 32 + (ch & 31);
      const float s = hb ? sb : sg;
      const float* g = dw2a + ((size_t)p * a.K + k) * 9;
      const size_t o = ((size_t)ch * a.K + k) * 9;
      const float* w = (hb ? a.wb : a.wg);
      float* d = hb ? dwb : dwg;
      const float c0 = a.mode == 0 ? 1.f : 1.f - s;
      float dot = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (d) d[o + t] = c0 * g[t];
        if (blend && w) dot -= w[o + t] * g[t];
      }
      (hb ? acc_b : acc_g) += dot;
    } else if (i < na + ns) {   // (half, channel, s) of the style weights
      const long j = i - na;
      const int sidx = (int)(j % a.S);
      const long r = j / a.S;
      const int ch = (int)(r % a.C), hb = (int)(r / a.C);
      const int p = (ch >> 6) * 128 + ((ch & 63) >> 5) * 64 + hb * 32 + (ch & 31);
      const float s = hb ? sb : sg;
      const size_t o = ((size_t)ch * a.S + sidx) * 9;
      const float* ws = hb ? a.wsb : a.wsg;
      float* d = hb ? dwsb : dwsg;
      const float c1 = a.mode == 2 ? 1.f : s;
      float dot = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // mode 3: the style weights were folded into w2a (same k index); else their gradient arrives (tap, row)-major
        const float g = a.mode == 3 ? dw2a[((size_t)p * a.K + sidx) * 9 + t] : dwst[((size_t)t * a.rows + p) * a.S + sidx];
        if (d) d[o + t] = c1 * g;
        if (blend) dot += ws[o + t] * g;
      }
      (hb ? acc_b : acc_g) += dot;
    } else {
      const int r = (int)(i - na - ns);
      const int ch = r % a.C, hb = r / a.C;
      const int p = (ch >> 6) * 128 + ((ch & 63) >> 5) * 64 + hb * 32 + (ch & 31);
      const float s = hb ? sb : sg;
      const float g = db2 ? db2[p] : 0.f;
      float* dp = hb ? dbb : dbg;
      float* ds = hb ? dbsb : dbsg;
      if (a.mode == 0) { if (dp) dp[ch] = g; }
      else if (a.mode == 2) { if (ds) ds[ch] = g; }
      else {
        if (dp) dp[ch] = (1.f - s) * g;
        if (ds) ds[ch] = s * g;
        (hb ? acc_b : acc_g) += ((hb ? a.bsb : a.bsg)[ch] - (hb ? a.bb : a.bg)[ch]) * g;
      }
    }
  }
  // block reduction (fixed order: deterministic)
  __shared__ float red[2][256];
  red[0][threadIdx.x] = acc_g;
  red[1][threadIdx.x] = acc_b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0][0];
    partial[2 * blockIdx.x + 1] = red[1][0];
  }
}

__global__ __launch_bounds__(256) void sean_alpha_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                  const float* __restrict__ ag, const float* __restrict__ ab,
                                                                  float* __restrict__ dalpha) {
  // 2 x 128 threads, fixed summation order (deterministic)
  __shared__ float red[256];
  const int which = threadIdx.x >> 7, t = threadIdx.x & 127;
  float s = 0.f;
  for (int i = t; i < nblk; i += 128) s += partial[2 * i + which];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (t < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (t == 0) {
    const float sg = sigm(which ? ab : ag);
    dalpha[which] = red[threadIdx.x] * sg * (1.f - sg);
  }
}

// out[n][tap][row][r < 32] = r < L ? t[n*L + r][tap*rows + row] : 0      (and the adjoint)
__global__ __launch_bounds__(256) void table_layout_kernel(const float* __restrict__ t, float* __restrict__ out, int N,
                                                           int L, int rows, int bwd, float* __restrict__ amax) {
  float vmax = 0.f;
  const long total = (long)N * 9 * rows * 32;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i & 31);
    const long q = i >> 5;   // (n, tap, row)
    const int row = (int)(q % rows);
    const long q2 = q / rows;
    const int tap = (int)(q2 % 9), n = (int)(q2 / 9);
    if (!bwd) {
      const float v = r < L ? t[((size_t)n * L + r) * (9 * rows) + tap * rows + row] : 0.f;
      out[i] = v;
      vmax = fmaxf(vmax, fabsf(v));
    } else if (r < L) {
      out[((size_t)n * L + r) * (9 * rows) + tap * rows + row] = t[i];
    }
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);
}

inline int pgrid(long n) { return (int)min(4096L, (n + 255) / 256); }

}  // namespace

extern "C" {

int dsee_sean_pack_rows(int C) { return (C + 63) / 64 * 128; }

/* Packed, blended weights of one SPADE / SEAN / PureSEAN norm layer (see the head of this file).  amax_w2a (optional, zeroed
 * by the caller, 64-line form of dsee_absmax) receives max |w2a|; dsee_style_table_layout's `amax` max |table|: pointing both
 * at one slot gives the operand bound of dsee_wino43_weights_table without a pass of its own.  mode 0 spade (w2a, b2),
 * 1 sean (w2a, wst, b2; alpha_* = the two learnable scalars BEFORE the sigmoid), 2 puresean (wst, b2), 3 sean above
 * max_fm_size (w2a = (1 - s) W + s W_style, b2; needs K == S).  Unused pointers NULL. */
int dsee_sean_pack_fwd(const float* w_gamma, const float* w_beta, const float* ws_gamma, const float* ws_beta,
                       const float* b_gamma, const float* b_beta, const float* bs_gamma, const float* bs_beta,
                       const float* alpha_gamma, const float* alpha_beta, int mode, int C, int K, int S, float* w2a,
                       float* wst, float* b2, float* amax_w2a, hipStream_t st) {
  DSEE_CHECK_ARG(mode >= 0 && mode <= 3 && C > 0 && b2);
  DSEE_CHECK_ARG(mode == 2 || (w_gamma && w_beta && b_gamma && b_beta && w2a && K > 0));
  DSEE_CHECK_ARG(mode == 0 || (ws_gamma && ws_beta && bs_gamma && bs_beta && S > 0));
  DSEE_CHECK_ARG((mode != 1 && mode != 2) || wst);
  DSEE_CHECK_ARG((mode != 1 && mode != 3) || (alpha_gamma && alpha_beta));
  DSEE_CHECK_ARG(mode != 3 || K == S);
  PackArgs a = {w_gamma, w_beta, ws_gamma, ws_beta, b_gamma, b_beta, bs_gamma, bs_beta, alpha_gamma, alpha_beta,
                mode, C, K, S, dsee_sean_pack_rows(C)};
  float* wo = mode == 2 ? nullptr : w2a;
  float* so = (mode == 1 || mode == 2) ? wst : nullptr;
  const long n = (wo ? (long)a.rows * K : 0) + (so ? (long)a.rows * S : 0) + a.rows;
  sean_pack_fwd_kernel<<<pgrid(n), 256, 0, st>>>(a, wo, so, b2, amax_w2a);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

size_t dsee_sean_pack_bwd_workspace(void) { return (size_t)4096 * 2 * sizeof(float); }

/* Adjoint of dsee_sean_pack_fwd: gradients of the (up to) eight weight / bias tensors and, modes 1 / 3, dalpha [2] =
 * d/d(alpha_gamma), d/d(alpha_beta).  dwst / dw2a / db2 may be NULL (no gradient arrived); outputs may be NULL. */
int dsee_sean_pack_bwd(const float* w_gamma, const float* w_beta, const float* ws_gamma, const float* ws_beta,
                       const float* b_gamma, const float* b_beta, const float* bs_gamma, const float* bs_beta,
                       const float* alpha_gamma, const float* alpha_beta, int mode, int C, int K, int S,
                       const float* dw2a, const float* dwst, const float* db2, float* dw_gamma, float* dw_beta,
                       float* dws_gamma, float* dws_beta, float* db_gamma, float* db_beta, float* dbs_gamma,
                       float* dbs_beta, float* dalpha, float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(mode >= 0 && mode <= 3 && C > 0 && workspace);
  DSEE_CHECK_ARG(mode == 2 || dw2a);
  DSEE_CHECK_ARG((mode != 1 && mode != 2) || dwst);
  PackArgs a = {w_gamma, w_beta, ws_gamma, ws_beta, b_gamma, b_beta, bs_gamma, bs_beta, alpha_gamma, alpha_beta,
                mode, C, K, S, dsee_sean_pack_rows(C)};
  const float* ga = mode == 2 ? nullptr : dw2a;
  const float* gs = (mode == 1 || mode == 2) ? dwst : nullptr;
  const long n = (ga ? (long)C * 2 * K : 0) + ((gs || mode == 3) ? (long)C * 2 * S : 0) + 2 * C;
  const int grid = pgrid(n);
  sean_pack_bwd_kernel<<<grid, 256, 0, st>>>(a, ga, gs, db2, dw_gamma, dw_beta, dws_gamma, dws_beta, db_gamma, db_beta,
                                             dbs_gamma, dbs_beta, workspace);
  if (dalpha && (mode == 1 || mode == 3))
    sean_alpha_finalize_kernel<<<1, 256, 0, st>>>(workspace, grid, alpha_gamma, alpha_beta, dalpha);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

/* table [N][9][rows][32] <- t [N*L][9*rows] (the style-table GEMM's result, L <= 32 regions; columns >= L zero) */
int dsee_style_table_layout(const float* t, float* table, int N, int L, int rows, float* amax, hipStream_t st) {
  DSEE_CHECK_ARG(t && table && N > 0 && L > 0 && L <= 32 && rows > 0);
  table_layout_kernel<<<pgrid((long)N * 9 * rows * 32), 256, 0, st>>>(t, table, N, L, rows, 0, amax);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_style_table_layout_bwd(const float* dtable, float* dt, int N, int L, int rows, hipStream_t st) {
  DSEE_CHECK_ARG(dtable && dt && N > 0 && L > 0 && L <= 32 && rows > 0);
  table_layout_kernel<<<pgrid((long)N * 9 * rows * 32), 256, 0, st>>>(dtable, dt, N, L, rows, 1, nullptr);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
