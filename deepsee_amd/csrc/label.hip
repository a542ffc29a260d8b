// Label-map indexed kernels: everything the reference does with the 19-channel one-hot `input_semantics`
// tensor (data/preprocessor.py:35-41) is restated on a uint8 label map [N][H][W] (76x fewer bytes).
//
//  * onehot_conv3x3: mlp_shared = ReLU(conv3x3(one-hot seg)) (normalization.py:98-101,110-114,174-175) is a
//    9-tap gather-sum of weight columns; its weight gradient is a label-segmented sum.
//  * label_gather:   style_map[b,:,h,w] = style[b, label[b,h,w], :]   (normalization.py:179-185), also the
//    backward of style pooling.
//  * label_segsum:   S[b,r,c] = scale * sum_{hw: label=r} f[b,h,w,c]   (encoder.py:36-49 extract_style_matrix,
//    divides by H*W not by region area), also the backward of label_gather.
//  * nearest resize of the label map is index math: src = dst << shift  (F.interpolate 'nearest', SURVEY B-3).
#include "dsee_common.h"

namespace {

__device__ __forceinline__ int lab_at(const uint8_t* lab, int n, int H, int W, int shift, int h, int w) {
  return lab[((size_t)n * H + ((size_t)h << shift)) * W + ((size_t)w << shift)];
}

// wt: [9][L][Co] (tap-major table), out[m][out_ld] at channel offset coff, Co % 4 == 0
__global__ __launch_bounds__(256) void onehot_conv_fwd_kernel(const uint8_t* __restrict__ lab,
                                                              const float* __restrict__ wt,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int N, int H, int W, int shift, int R, int Rw, int L,
                                                              int Co, int out_ld, int coff, int relu, int onehot_coff,
                                                              float* __restrict__ amax, float amax_floor) {
  const int tpp = Co / 4, ppb = 256 / tpp;
  const int q = threadIdx.x % tpp, s = threadIdx.x / tpp;
  const long M = (long)N * R * Rw;
  float vmax = amax_floor;   // max |out| for the fp16 operand scale of the consumer (the one-hot channels contribute 1)
  if (s < ppb) {
    const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (long m = (long)blockIdx.x * ppb + s; m < M; m += (long)gridDim.x * ppb) {
      // (32-bit divisions: M < 2^32 -- the host checks --, and the ISA has no integer divide: dsee_common.h)
      const unsigned mu_ = (unsigned)m, t_ = mu_ / (unsigned)Rw, n_ = t_ / (unsigned)R;
      const int w = (int)(mu_ - t_ * (unsigned)Rw), h = (int)(t_ - n_ * (unsigned)R), n = (int)n_;
      f32x4 acc = b;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int hh = h + dy, ww = w + dx;
          if (hh >= 0 && hh < R && ww >= 0 && ww < Rw) {
            const int r = lab_at(lab, n, H, W, shift, hh, ww);
            const int tap = (dy + 1) * 3 + (dx + 1);
            acc += *reinterpret_cast<const f32x4*>(wt + ((size_t)tap * L + r) * Co + q * 4);
          }
        }
      if (relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], 0.f);
      }
      *reinterpret_cast<f32x4*>(out + m * out_ld + coff + q * 4) = acc;
      vmax = fmaxf(vmax, dsee_absmax4(acc));
      if (onehot_coff >= 0 && q < 8) {   // the 32 one-hot label channels of the same pixel (dsee_label_onehot)
        const int r = lab_at(lab, n, H, W, shift, h, w);
        f32x4 oh;
#pragma unroll
        for (int k = 0; k < 4; ++k) oh[k] = (q * 4 + k == r) ? 1.f : 0.f;
        *reinterpret_cast<f32x4*>(out + m * out_ld + onehot_coff + q * 4) = oh;
      }
    }
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);   // (amax is block-uniform; every thread arrives here)
}

// Same function with the whole tap table [9][L][Co] resident in LDS (<= 150 KB: L = 19..32 classes x 128 channels): the
// global form is a chain of two dependent loads per tap (label byte -> table row in L2) and ran at 1.1 TB/s of output;
// here one 1024-thread block per CU copies the table once, walks the pixels with stride gridDim.x and fetches the NEXT
// pixel's nine labels while it sums the current pixel's rows out of LDS.
__global__ __launch_bounds__(1024) void onehot_conv_fwd_lds_kernel(const uint8_t* __restrict__ lab,
                                                                   const float* __restrict__ wt,
                                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                                   int N, int H, int W, int shift, int R, int Rw, int L,
                                                                   int Co, int out_ld, int coff, int relu, int onehot_coff,
                                                                   float* __restrict__ amax, float amax_floor) {
  extern __shared__ __attribute__((aligned(16))) float tab[];
  for (int i = threadIdx.x; i < 9 * L * Co / 4; i += 1024)
    reinterpret_cast<f32x4*>(tab)[i] = reinterpret_cast<const f32x4*>(wt)[i];
  __syncthreads();
  const int tpp = Co / 4, ppb = 1024 / tpp;
  const int q = threadIdx.x % tpp, s = threadIdx.x / tpp;
  const long M = (long)N * R * Rw, stride = (long)gridDim.x * ppb;
  float vmax = amax_floor;
  auto labels = [&](long m, int (&r)[9]) {
    // (32-bit divisions: M < 2^32 -- the host checks --, and the ISA has no integer divide: dsee_common.h)
    const unsigned mu_ = (unsigned)m, t_ = mu_ / (unsigned)Rw, n_ = t_ / (unsigned)R;
    const int w = (int)(mu_ - t_ * (unsigned)Rw), h = (int)(t_ - n_ * (unsigned)R), n = (int)n_;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
      const bool in = hh >= 0 && hh < R && ww >= 0 && ww < Rw;
      r[tap] = in ? lab_at(lab, n, H, W, shift, hh, ww) : -1;
    }
  };
  if (s < ppb) {
    const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    long m = (long)blockIdx.x * ppb + s;
    int cur[9], nxt[9];
    if (m < M) labels(m, cur);
    for (; m < M; m += stride) {
      const bool more = m + stride < M;
      if (more) labels(m + stride, nxt);
      f32x4 acc = b;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        if (cur[tap] >= 0) acc += *reinterpret_cast<const f32x4*>(tab + ((size_t)tap * L + cur[tap]) * Co + q * 4);
      if (relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], 0.f);
      }
      __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + m * out_ld + coff + q * 4));
      vmax = fmaxf(vmax, dsee_absmax4(acc));
      if (onehot_coff >= 0 && q < 8) {   // the 32 one-hot label channels of the same pixel (dsee_label_onehot)
        f32x4 oh;
#pragma unroll
        for (int k = 0; k < 4; ++k) oh[k] = (q * 4 + k == cur[4]) ? 1.f : 0.f;
        *reinterpret_cast<f32x4*>(out + m * out_ld + onehot_coff + q * 4) = oh;
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) cur[tap] = nxt[tap];
    }
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);
}

// ... and with 4 consecutive pixels of a row per thread (Rw % 4 == 0): the 32 threads of a pixel all formed the same 9 label
// addresses and bounds checks, ~300 instructions per 16-byte store (1.8 TB/s of output at 256^2, issue-bound); a run of 4 pixels
// shares its 3 x 6 label window (18 loads for 4 pixels instead of 36) and one index decomposition.
__global__ __launch_bounds__(1024) void onehot_conv_fwd_lds4_kernel(const uint8_t* __restrict__ lab,
                                                                    const float* __restrict__ wt,
                                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                                    int N, int H, int W, int shift, int R, int Rw, int L,
                                                                    int Co, int out_ld, int coff, int relu, int onehot_coff,
                                                                    float* __restrict__ amax, float amax_floor) {
  extern __shared__ __attribute__((aligned(16))) float tab[];
  for (int i = threadIdx.x; i < 9 * L * Co / 4; i += 1024)
    reinterpret_cast<f32x4*>(tab)[i] = reinterpret_cast<const f32x4*>(wt)[i];
  __syncthreads();
  const int tpp = Co / 4, gpb = 1024 / tpp;          // pixel runs per block and step
  const int q = threadIdx.x % tpp, s = threadIdx.x / tpp;
  const unsigned Rw4 = (unsigned)Rw >> 2;
  const long G = (long)N * R * Rw4, stride = (long)gridDim.x * gpb;
  float vmax = amax_floor;
  if (s < gpb) {
    const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* const tq = tab + q * 4;
    for (long g = (long)blockIdx.x * gpb + s; g < G; g += stride) {
      const unsigned gu = (unsigned)g, t_ = gu / Rw4, n_ = t_ / (unsigned)R;
      const int w0 = (int)(gu - t_ * Rw4) * 4, h = (int)(t_ - n_ * (unsigned)R), n = (int)n_;
      int win[3][6];      // labels of rows h-1 .. h+1, columns w0-1 .. w0+4 (-1 outside the image)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int hh = h + a - 1;
        const bool rin = hh >= 0 && hh < R;
        const uint8_t* const row = lab + ((size_t)n * H + ((size_t)(rin ? hh : 0) << shift)) * W;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const int ww = w0 + c - 1;
          win[a][c] = (rin && ww >= 0 && ww < Rw) ? (int)row[(size_t)ww << shift] : -1;
        }
      }
      const size_t m0 = ((size_t)n * R + h) * Rw + w0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        f32x4 acc = b;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int r = win[tap / 3][p + tap % 3];
          if (r >= 0) acc += *reinterpret_cast<const f32x4*>(tq + (tap * L + r) * Co);
        }
        if (relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], 0.f);
        }
        __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + (m0 + p) * out_ld + coff + q * 4));
        vmax = fmaxf(vmax, dsee_absmax4(acc));
        if (onehot_coff >= 0 && q < 8) {   // the 32 one-hot label channels of the same pixel (dsee_label_onehot)
          f32x4 oh;
#pragma unroll
          for (int k = 0; k < 4; ++k) oh[k] = (q * 4 + k == win[1][p + 1]) ? 1.f : 0.f;
          *reinterpret_cast<f32x4*>(out + (m0 + p) * out_ld + onehot_coff + q * 4) = oh;
        }
      }
    }
  }
  if (amax) dsee_block_atomic_absmax(amax, vmax);
}

__global__ void onehot_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co, int L) {
  // w [Co][L][3][3] -> wt [9][L][Co]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * L * Co) return;
  const int co = i % Co, r = (i / Co) % L, tap = i / (Co * L);
  wt[i] = w[((size_t)co * L + r) * 9 + tap];
}

// Weight gradient of the one-hot conv.  Block = 128 channels x 8 tap groups (group g owns tap g, group 7 taps 7 and 8, group 0
// also the bias).  A thread keeps the LT label accumulators of its tap(s) and channel in REGISTERS and updates them with a
// compare-select chain (acc[k] += label == k ? g : 0): the LDS form -- one float atomic per pixel, tap and channel -- was
// bound by the LDS atomic unit at 0.45 ms for 0.5 M pixels whatever the number of waves or the load pipelining.  Fixed
// summation order per thread, partial rows [9*L + 1][128] per block as before.
template <int LT>
__global__ __launch_bounds__(1024) void onehot_conv_wgrad_kernel(const uint8_t* __restrict__ lab,
                                                                const float* __restrict__ dact, int dld,
                                                                const float* __restrict__ act, int ald, int N,
                                                                int H, int W, int shift, int R, int Rw, int L,
                                                                int chunk_px, float* __restrict__ part) {
  const int c = threadIdx.x & 127, tg = threadIdx.x >> 7;
  const int rows = 9 * L + 1;
  const long M = (long)N * R * Rw;
  const long m0 = (long)blockIdx.x * chunk_px, m1 = min(M, m0 + chunk_px);
  const int t0 = tg;
  const bool two = tg == 7;        // (wave-uniform: a wave = 64 channels of one tap group)
  float acc0[LT], acc1[LT], accb = 0.f;
#pragma unroll
  for (int k = 0; k < LT; ++k) acc0[k] = acc1[k] = 0.f;
  constexpr int U = 8;   // pixels per trip: their loads are issued together
  for (long mb = m0; mb < m1; mb += U) {
    float gv[U];
    int r0[U], r1[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const long m = mb + j;
      gv[j] = m < m1 ? (act[(size_t)m * ald + c] > 0.f ? dact[(size_t)m * dld + c] : 0.f) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const long m = mb + j;
      // (32-bit divisions: M < 2^32 -- the host checks --, and the ISA has no integer divide: dsee_common.h)
      const unsigned mu_ = (unsigned)m, t_ = mu_ / (unsigned)Rw, n_ = t_ / (unsigned)R;
      const int w = (int)(mu_ - t_ * (unsigned)Rw), h = (int)(t_ - n_ * (unsigned)R), n = (int)n_;
      {
        const int hh = h + t0 / 3 - 1, ww = w + t0 % 3 - 1;
        r0[j] = (m < m1 && hh >= 0 && hh < R && ww >= 0 && ww < Rw) ? lab_at(lab, n, H, W, shift, hh, ww) : -1;
      }
      r1[j] = -1;
      if (two) {
        const int hh = h + 1, ww = w + 1;   // tap 8
        r1[j] = (m < m1 && hh < R && ww < Rw) ? lab_at(lab, n, H, W, shift, hh, ww) : -1;
      }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const float g = gv[j];
      accb += g;
#pragma unroll
      for (int k = 0; k < LT; ++k) acc0[k] += r0[j] == k ? g : 0.f;
      if (two) {
#pragma unroll
        for (int k = 0; k < LT; ++k) acc1[k] += r1[j] == k ? g : 0.f;
      }
    }
  }
  float* o = part + (size_t)blockIdx.x * rows * 128;
#pragma unroll
  for (int k = 0; k < LT; ++k)
    if (k < L) {
      o[(t0 * L + k) * 128 + c] = acc0[k];
      if (two) o[(8 * L + k) * 128 + c] = acc1[k];
    }
  if (tg == 0) o[(9 * L) * 128 + c] = accb;
}

__global__ void onehot_wgrad_finalize_kernel(const float* __restrict__ part, int nparts, int L, float* __restrict__ dw,
                                             float* __restrict__ db) {
  // dw [128][L][3][3], db [128]
  const int rows = 9 * L + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 128) return;
  const int c = i % 128, row = i / 128;
  float v = 0.f;
  for (int p = 0; p < nparts; ++p) v += part[(size_t)p * rows * 128 + i];
  if (row == 9 * L) {
    db[c] = v;
  } else {
    const int tap = row / L, r = row % L;
    dw[((size_t)c * L + r) * 9 + tap] = v;
  }
}

// out[m][coff + r] = (lab(m) == r), r in [0,32)
__global__ __launch_bounds__(256) void label_onehot_kernel(const uint8_t* __restrict__ lab, float* __restrict__ out,
                                                           int N, int H, int W, int shift, int R, int Rw, int ld,
                                                           int coff) {
  const int q = threadIdx.x & 7, s = threadIdx.x >> 3;  // 8 float4 per pixel, 32 pixels per block pass
  const long M = (long)N * R * Rw;
  for (long m = (long)blockIdx.x * 32 + s; m < M; m += (long)gridDim.x * 32) {
    // (32-bit divisions: M < 2^32 -- the host checks --, and the ISA has no integer divide: dsee_common.h)
    const unsigned mu_ = (unsigned)m, t_ = mu_ / (unsigned)Rw, n_ = t_ / (unsigned)R;
    const int w = (int)(mu_ - t_ * (unsigned)Rw), h = (int)(t_ - n_ * (unsigned)R), n = (int)n_;
    const int r = lab_at(lab, n, H, W, shift, h, w);
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (q * 4 + k == r) ? 1.f : 0.f;
    *reinterpret_cast<f32x4*>(out + m * ld + coff + q * 4) = v;
  }
}

// out[m][ld] @coff = scale * table[n][lab(m)][:]
__global__ __launch_bounds__(256) void label_gather_kernel(const uint8_t* __restrict__ lab,
                                                           const float* __restrict__ table, float* __restrict__ out,
                                                           int N, int H, int W, int shift, int R, int Rw, int L, int Cs,
                                                           int ld, int coff, float scale) {
  const int tpp = Cs / 4, ppb = 256 / tpp;
  const int q = threadIdx.x % tpp, s = threadIdx.x / tpp;
  if (s >= ppb) return;
  const long M = (long)N * R * Rw;
  for (long m = (long)blockIdx.x * ppb + s; m < M; m += (long)gridDim.x * ppb) {
    // (32-bit divisions: M < 2^32 -- the host checks --, and the ISA has no integer divide: dsee_common.h)
    const unsigned mu_ = (unsigned)m, t_ = mu_ / (unsigned)Rw, n_ = t_ / (unsigned)R;
    const int w = (int)(mu_ - t_ * (unsigned)Rw), h = (int)(t_ - n_ * (unsigned)R), n = (int)n_;
    const int r = lab_at(lab, n, H, W, shift, h, w);
    f32x4 v = *reinterpret_cast<const f32x4*>(table + ((size_t)n * L + r) * Cs + q * 4) * scale;
    *reinterpret_cast<f32x4*>(out + m * ld + coff + q * 4) = v;
  }
}

// part[n][chunk][L][Cs] = sum_{pixels of chunk with label r} in[m][coff + c]
__global__ __launch_bounds__(256) void label_segsum_kernel(const uint8_t* __restrict__ lab,
                                                           const float* __restrict__ in, int ld, int coff, int H, int W,
                                                           int shift, int R, int Rw, int L, int Cs, int chunk_px,
                                                           int chunks, float* __restrict__ part) {
  extern __shared__ float accs[];  // [slots][L][Cs]
  const int slots = 256 / Cs > 0 ? 256 / Cs : 1;
  const int c = threadIdx.x % Cs, s = threadIdx.x / Cs;
  const int n = blockIdx.y, chunk = blockIdx.x;
  for (int i = threadIdx.x; i < slots * L * Cs; i += 256) accs[i] = 0.f;
  __syncthreads();
  const int P = R * Rw;
  const int p0 = chunk * chunk_px, p1 = min(P, p0 + chunk_px);
  if (s < slots && threadIdx.x < slots * Cs) {
    for (int p = p0 + s; p < p1; p += slots) {
      const int h = p / Rw, w = p % Rw;
      const int r = lab_at(lab, n, H, W, shift, h, w);
      accs[(s * L + r) * Cs + c] += in[((size_t)n * P + p) * ld + coff + c];
    }
  }
  __syncthreads();
  float* o = part + ((size_t)n * chunks + chunk) * L * Cs;
  for (int i = threadIdx.x; i < L * Cs; i += 256) {
    float v = 0.f;
    for (int k = 0; k < slots; ++k) v += accs[k * L * Cs + i];
    o[i] = v;
  }
}

__global__ void segsum_finalize_kernel(const float* __restrict__ part, int chunks, int LC, int N, float scale,
                                       float* __restrict__ table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * LC) return;
  const int n = i / LC, j = i % LC;
  float v = 0.f;
  for (int k = 0; k < chunks; ++k) v += part[((size_t)n * chunks + k) * LC + j];
  table[i] = v * scale;
}

int seg_chunks(int N, int P, int* chunk_px) {
  int want = 1024 / (N > 0 ? N : 1);
  if (want < 1) want = 1;
  int cp = (P + want - 1) / want;
  if (cp < 256) cp = 256;
  *chunk_px = cp;
  return (P + cp - 1) / cp;
}

int wgrad_parts(long M, int* chunk_px) {
  long cp = (M + 511) / 512;
  if (cp < 256) cp = 256;
  *chunk_px = (int)cp;
  return (int)((M + cp - 1) / cp);
}

}  // namespace

extern "C" {

int dsee_onehot_conv3x3_pack(const float* w_oihw, float* table, int Co, int L, hipStream_t st) {
  DSEE_CHECK_ARG(w_oihw && table && Co > 0 && L > 0);
  onehot_pack_kernel<<<dsee_cdiv((long)9 * L * Co, 256), 256, 0, st>>>(w_oihw, table, Co, L);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_onehot_conv3x3_fwd(const uint8_t* lab, const float* table, const float* bias, float* out, int N, int H, int W,
                            int shift, int L, int Co, int out_ld, int coff, int relu, int onehot_coff, float* amax,
                            float amax_floor, hipStream_t st) {
  DSEE_CHECK_ARG(lab && table && out && Co % 4 == 0 && Co <= 1024 && out_ld % 4 == 0 && coff % 4 == 0);
  DSEE_CHECK_ARG(onehot_coff < 0 || (onehot_coff % 4 == 0 && out_ld >= onehot_coff + 32 && Co >= 32 && L <= 32));
  const int R = H >> shift, Rw = W >> shift;
  const long M = (long)N * R * Rw;
  DSEE_CHECK_ARG(M < (1L << 32));      // (32-bit pixel index in the kernels)
  const size_t lds = (size_t)9 * L * Co * sizeof(float);
  if (lds <= 150 * 1024 && 1024 % (Co / 4) == 0 && M >= 4096) {
    static size_t attr_lds = 0;   // (the block also holds 64 bytes of static LDS: ask for what is needed, not for 160 KB)
    if (lds > attr_lds) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&onehot_conv_fwd_lds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        DSEE_CHECK_ARG(!"onehot_conv_fwd: cannot reserve the LDS table");
      }
      attr_lds = lds;
    }
    const int ppb1k = 1024 / (Co / 4);
    if (Rw % 4 == 0) {
      static size_t attr_lds4 = 0;
      if (lds > attr_lds4) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&onehot_conv_fwd_lds4_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          (void)hipGetLastError();
          DSEE_CHECK_ARG(!"onehot_conv_fwd: cannot reserve the LDS table");
        }
        attr_lds4 = lds;
      }
      onehot_conv_fwd_lds4_kernel<<<(int)min(256L, (M / 4 + ppb1k - 1) / ppb1k), 1024, lds, st>>>(
          lab, table, bias, out, N, H, W, shift, R, Rw, L, Co, out_ld, coff, relu, onehot_coff, amax, amax_floor);
      DSEE_LAUNCH_CHECK();
      return DSEE_OK;
    }
    onehot_conv_fwd_lds_kernel<<<(int)min(256L, (M + ppb1k - 1) / ppb1k), 1024, lds, st>>>(
        lab, table, bias, out, N, H, W, shift, R, Rw, L, Co, out_ld, coff, relu, onehot_coff, amax, amax_floor);
    DSEE_LAUNCH_CHECK();
    return DSEE_OK;
  }
  const int ppb = 256 / (Co / 4);
  onehot_conv_fwd_kernel<<<(int)min(4096L, (M + ppb - 1) / ppb), 256, 0, st>>>(lab, table, bias, out, N, H, W, shift, R,
                                                                                Rw, L, Co, out_ld, coff, relu, onehot_coff, amax,
                                                                                amax_floor);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

size_t dsee_onehot_conv3x3_wgrad_workspace(int N, int H, int W, int shift, int L) {
  int cp;
  const long M = (long)N * (H >> shift) * (W >> shift);
  return (size_t)wgrad_parts(M, &cp) * (9 * L + 1) * 128 * sizeof(float);
}

/* dW_sh[:, r, tap] = sum_{p : lab(p+tap)=r} relu'(act[p]) * dact[p]  (SURVEY Appendix E); Co fixed at 128
 * (normalization.py:95 nhidden).  act/dact are [M][ld] with the 128 channels at offset coff. */
int dsee_onehot_conv3x3_wgrad(const uint8_t* lab, const float* dact, int dact_ld, const float* act, int act_ld, int N,
                              int H, int W, int shift, int L, float* dw_oihw, float* dbias, float* workspace,
                              hipStream_t st) {
  DSEE_CHECK_ARG(lab && dact && act && dw_oihw && dbias && workspace && L <= 32);
  const int R = H >> shift, Rw = W >> shift;
  const long M = (long)N * R * Rw;
  int cp;
  const int parts = wgrad_parts(M, &cp);
  if (L <= 20)
    onehot_conv_wgrad_kernel<20><<<parts, 1024, 0, st>>>(lab, dact, dact_ld, act, act_ld, N, H, W, shift, R, Rw, L, cp, workspace);
  else
    onehot_conv_wgrad_kernel<32><<<parts, 1024, 0, st>>>(lab, dact, dact_ld, act, act_ld, N, H, W, shift, R, Rw, L, cp, workspace);
  DSEE_LAUNCH_CHECK();
  onehot_wgrad_finalize_kernel<<<dsee_cdiv((long)(9 * L + 1) * 128, 256), 256, 0, st>>>(workspace, parts, L, dw_oihw,
                                                                                        dbias);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_label_onehot(const uint8_t* lab, float* out, int N, int H, int W, int shift, int out_ld, int coff,
                      hipStream_t st) {
  DSEE_CHECK_ARG(lab && out && out_ld % 4 == 0 && coff % 4 == 0 && out_ld >= coff + 32);
  const int R = H >> shift, Rw = W >> shift;
  const long M = (long)N * R * Rw;
  label_onehot_kernel<<<(int)min(4096L, (M + 31) / 32), 256, 0, st>>>(lab, out, N, H, W, shift, R, Rw, out_ld, coff);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

int dsee_label_gather(const uint8_t* lab, const float* table, float* out, int N, int H, int W, int shift, int L, int Cs,
                      int out_ld, int coff, float scale, hipStream_t st) {
  DSEE_CHECK_ARG(lab && table && out && Cs % 4 == 0 && Cs <= 1024 && out_ld % 4 == 0 && coff % 4 == 0);
  const int R = H >> shift, Rw = W >> shift;
  const long M = (long)N * R * Rw;
  const int ppb = 256 / (Cs / 4);
  label_gather_kernel<<<(int)min(4096L, (M + ppb - 1) / ppb), 256, 0, st>>>(lab, table, out, N, H, W, shift, R, Rw, L,
                                                                             Cs, out_ld, coff, scale);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

size_t dsee_label_segsum_workspace(int N, int H, int W, int shift, int L, int Cs) {
  int cp;
  const int chunks = seg_chunks(N, (H >> shift) * (W >> shift), &cp);
  return (size_t)N * chunks * L * Cs * sizeof(float);
}

int dsee_label_segsum(const uint8_t* lab, const float* in, int ld, int coff, float* table, int N, int H, int W,
                      int shift, int L, int Cs, float scale, float* workspace, hipStream_t st) {
  DSEE_CHECK_ARG(lab && in && table && workspace && Cs <= 256 && L <= 32);
  const int R = H >> shift, Rw = W >> shift;
  int cp;
  const int chunks = seg_chunks(N, R * Rw, &cp);
  const int slots = 256 / Cs > 0 ? 256 / Cs : 1;
  const size_t lds = (size_t)slots * L * Cs * sizeof(float);
  label_segsum_kernel<<<dim3(chunks, N), 256, lds, st>>>(lab, in, ld, coff, H, W, shift, R, Rw, L, Cs, cp, chunks,
                                                         workspace);
  DSEE_LAUNCH_CHECK();
  segsum_finalize_kernel<<<dsee_cdiv((long)N * L * Cs, 256), 256, 0, st>>>(workspace, chunks, L * Cs, N, scale, table);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
