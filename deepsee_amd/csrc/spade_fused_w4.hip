// Round 6: the fused SPADE / SEAN normalisation forward (spade_fused.hip: gamma/beta Winograd GEMM + output transform folded in
// registers + normalise + modulate + LeakyReLU; normalization.py:107-120, 167-213, architecture.py:92,114) with ONE wave per SIMD.
//
// The 8-wave kernel's cycle stamps (profiles/r05_fused_phases.txt) say a wave spends a quarter of a block each in fragment reads,
// LDS-DMA requests, the fold / Y update and its MFMAs, one after the other, and two in-order waves per SIMD overlap at most two of
// those phases: 140 k cycles per block against 35 k of matrix work.  Round 3's first version had one wave per SIMD already and
// was bound by the in-order issue of that wave -- it left the order of its instructions to the compiler.  This kernel takes
// round 6's GEMM recipe (gemm_w4.hip): program order IS issue order (sched_barrier fences), and every MFMA is followed by its
// share of everything else --
//   * 4 waves = 2 channel halves x 2 tile halves of the same 64 tile x 64 row block; a wave owns FOUR 16 x 16 blocks (gamma and
//     beta rows of 16 channels x two groups of 16 tiles): a fragment is shared by twice the MFMAs (40 instead of 60 KB of LDS
//     reads per position and SIMD), Y = 16 outputs x 16 = 256 accumulator registers (the whole AGPR half of the file),
//   * after each of the 12 MFMAs of a 32-k piece: one of the 8 fragment reads of the NEXT piece, or one of the look-ahead LDS-DMA
//     requests, and a slice of the VALU work -- the fold of the PREVIOUS position's product into T (the product lives in a
//     second set of MFMA accumulators, so no MFMA result is ever waited for) and, in the first position of a row, the
//     Y += At[.][r-1] (x) T update of the previous row, as read x n / fma x n / write x n groups so that the in-order wave does not
//     sit out the AGPR round trip of every element,
//   * the same ring memory (two positions of 2 x NP pieces = the whole 160 KB of LDS for K = 160), piece image, swizzle and
//     arithmetic order as the 8-wave kernel -- the results are bit-identical to it -- but a PIECE-granular protocol: one barrier per
//     piece, the slot a piece leaves is re-requested at once, 2 NP - 1 pieces in flight per CU (see "ring protocol" below).
// Two-term fp16x2 operands only (the 16-bit storage mode keeps the 8-wave kernel).
#include <stdlib.h>

#include "spade_fused_args.h"

// measurement builds (tools/exp/build_fw4.sh): 1 no MFMAs, 2 no fragment reads, 4 no fold / Y update, 8 no look-ahead LDS-DMA,
// 16 no epilogue stores
#ifndef DSEE_FW4_ABL
#define DSEE_FW4_ABL 0
#endif

namespace {

template <int NP, bool WSCALE>
__global__ __launch_bounds__(256, 1) void spade_fused_w4_kernel(FusedArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PIECE = 8192, RD = 2, UREG = RD * NP * PIECE, IPP = 4;   // IPP: LDS-DMA instructions per piece and wave
  static_assert(NP == 4 || NP == 5, "K = 128 or 160");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 1, wt = wave & 1;     // channel half (16 channels), tile half (32 tiles)

  // ---- workgroup -> (tile group of 64 tiles, row group of 64 packed rows): the order of the 8-wave kernel (4 x 8 co-running sets)
  const int rgn = a.rows >> 6;
  const long tgn = a.T >> 6, ntile = tgn * rgn;
  long l;
  {
    const long v = blockIdx.x, q = ntile >> 3, r = ntile & 7, xcd = v & 7, idx = v >> 3;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  long tg;
  int rg;
  if ((rgn % 8) == 0 && (tgn % 4) == 0) {
    const long sup = l >> 5;
    const int in = (int)(l & 31), rh = rgn / 8;
    tg = (sup / rh) * 4 + (in / 8);
    rg = (int)(sup % rh) * 8 + (in % 8);
  } else {
    tg = l / rgn;
    rg = (int)(l % rgn);
  }
  const long t0 = tg * 64;                    // first tile (of the batch)
  const int n = (int)(t0 / a.tpi);            // its image
  const int g = a.G > 1 ? n : 0;

  const float sv = dsee_pow2_scale(a.v_bound * dsee_amax_read(a.amax_v));
  const float su = dsee_pow2_scale(dsee_amax_read(a.amax_u));
  const float oscale = 1.f / (sv * su);

  // ---- LDS-DMA: instruction j (0, 1) of wave w fills rows 8 (w + 4 j) .. + 7 of a piece; lane -> (row, slot chunk) as in the
  //      8-wave kernel (a quad of lanes fetches the 64 contiguous bytes of one (row, slab)); rows + 32 = + 2048 bytes in the slab
  const int dr = 8 * wave + (lane >> 3);
  const int dcc = (lane & 7) ^ ((dr >> 1) & 7);         // ((dr + 32) >> 1) & 7 is the same: one chunk assignment for both
  const int dslab = dcc >> 2;
  const unsigned dlo = (unsigned)(dr * 64 + (dcc & 3) * 16);
  // (two offset registers per operand: the instruction's immediate offset would move the LDS destination as well)
  const unsigned voffu[2] = {dlo + (unsigned)dslab * (unsigned)a.u_slab_bytes, dlo + (unsigned)dslab * (unsigned)a.u_slab_bytes + 2048u};
  const unsigned voffv[2] = {dlo + (unsigned)dslab * (unsigned)a.v_slab_bytes, dlo + (unsigned)dslab * (unsigned)a.v_slab_bytes + 2048u};
  const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(a.U2), 0, (int)a.u_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(a.V2), 0, (int)a.v_bytes, 0x00020000);
  const unsigned PU = (unsigned)(a.G * a.u_group_bytes), PV = (unsigned)(a.T * 64);       // per position
  const unsigned QU = (unsigned)(2 * a.u_slab_bytes), QV = (unsigned)(2 * a.v_slab_bytes);   // per piece
  const unsigned base_u = (unsigned)((long)g * a.u_group_bytes + (long)rg * 4096), base_v = (unsigned)(t0 * 64);
  // request i (0 .. 3: U rows, U rows + 32, V rows, V rows + 32) of piece pc of the position in ring slot group par
  auto dma = [&](int par, int pc, unsigned ou, unsigned ov, auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    unsigned char* dst = smem + (i >= 2 ? UREG : 0) + (par * NP + pc) * PIECE + (wave + 4 * (i & 1)) * 1024;
    if constexpr (i < 2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (__attribute__((address_space(3))) void*)dst, 16, voffu[i & 1], ou + pc * QU, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (__attribute__((address_space(3))) void*)dst, 16, voffv[i & 1], ov + pc * QV, 0, 0);
  };

  // ---- fragment addresses (bytes within a piece): row r, chunk 4 (octet / 2) + 2 term + octet % 2 at slot 8r + (chunk ^ f(r)).
  //      A operand: lane -> (row l % 16 of a 16-row block, octet l / 16): gamma rows 16 wq + i, beta rows + 32 (+ 4096 bytes);
  //      B operand: tiles 32 wt + i, second tile block + 16 rows (+ 2048 bytes): neither shift changes the swizzle
  const int fi = lane & 15, oc = lane >> 4;
  auto foff = [&](int r, int t) {
    const int chunk = 4 * (oc >> 1) + 2 * t + (oc & 1);
    return (unsigned)((8 * r + (chunk ^ ((r >> 1) & 7))) * 16);
  };
  const int rga = 16 * wq + fi, rtv = 32 * wt + fi;
  const unsigned au0 = foff(rga, 0), au1 = foff(rga, 1), au0h = au0 + 7 * PIECE, au1h = au1 + 7 * PIECE;
  const unsigned av0 = UREG + foff(rtv, 0), av1 = UREG + foff(rtv, 1), av0h = av0 + 7 * PIECE, av1h = av1 + 7 * PIECE;
  struct Frag {
    u32x4 u[4];      // g0, g1, b0, b1 (term 0 / 1 of the gamma rows, of the beta rows)
    u32x4 v[2][2];   // [tile block][term]
  };
  // read i (0 .. 7) of the fragments of piece pc in ring slot group par
  auto ldf = [&](Frag& f, auto par_c, auto pc_c, auto i_c) __attribute__((always_inline)) {
    constexpr int so = (decltype(par_c)::value * NP + decltype(pc_c)::value) * PIECE, i = decltype(i_c)::value;
    constexpr bool hi = so >= 7 * PIECE;
    constexpr int io = hi ? so - 7 * PIECE : so;
    static_assert(io >= 0 && io + 4096 + PIECE <= 65536, "window");
    if constexpr (DSEE_FW4_ABL & 2) {
      if constexpr (i < 4) f.u[i] = (u32x4){(unsigned)lane, 1u, 2u, 3u}; else f.v[(i - 4) >> 1][i & 1] = (u32x4){4u, (unsigned)lane, 5u, 6u};
    } else if constexpr (i < 4) {
      const unsigned char* b = smem + ((i & 1) ? (hi ? au1h : au1) : (hi ? au0h : au0));
      f.u[i] = *reinterpret_cast<const u32x4*>(b + io + (i >> 1) * 4096);
    } else {
      const unsigned char* b = smem + ((i & 1) ? (hi ? av1h : av1) : (hi ? av0h : av0));
      f.v[(i - 4) >> 1][i & 1] = *reinterpret_cast<const u32x4*>(b + io + ((i - 4) >> 1) * 2048);
    }
  };

  // ---- register state
  float Y[4][4][2][8];      // [output row i][output column j][tile block][gamma 0..3 | beta 4..7]: 256 AGPRs, updated in place
  f32x4 T[4][2][2];         // [j][tile block][gamma | beta]
  f32x4 P[2][2][2];         // [position parity][tile block][gamma | beta]: MFMA accumulators of the running / the previous position
  Frag F[2];
  static_for<4>([&](auto i) {
    static_for<4>([&](auto j) {
      static_for<16>([&](auto e) {
        float& yr = Y[decltype(i)::value][decltype(j)::value][decltype(e)::value >> 3][decltype(e)::value & 7];
        asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(yr));
      });
    });
  });
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < 4; ++b) T[j][b >> 1][b & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 8; ++b) P[b >> 2][(b >> 1) & 1][b & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Y[i][j][.] += cf * T[j][.] for N consecutive elements e0 .. of tile block tb: read x N, fma x N, write x N
  auto y_rmw = [&](auto i_c, auto j_c, auto tb_c, auto e0_c, auto n_c, float cc) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value, tb = decltype(tb_c)::value, e0 = decltype(e0_c)::value,
                  N = decltype(n_c)::value;
    float t[N];
#pragma unroll
    for (int d = 0; d < N; ++d) {
      float& yr = Y[i][j][tb][e0 + d];
      float& tr = t[d];
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(tr) : "a"(yr));
    }
#pragma unroll
    for (int d = 0; d < N; ++d) t[d] = __builtin_fmaf(cc, T[j][tb][(e0 + d) >> 2][(e0 + d) & 3], t[d]);
#pragma unroll
    for (int d = 0; d < N; ++d) {
      float& yr = Y[i][j][tb][e0 + d];
      float& tr = t[d];
      asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(yr) : "v"(tr));
    }
  };

  // fold unit u (0 .. 15 = (j, tile block, gamma | beta)) of the product Q of a position in column CP of its row into T.
  // At[j][CP]: column 0 = (1,0,0,0), 1 = (1,1,1,1), 2 = (1,-1,1,-1), 3 = (1,2,4,8), 4 = (1,-2,4,-8), 5 = (0,0,0,1); the first
  // touch of a T element in a row assigns (T restarts with every row of positions)
  auto fold_unit = [&](auto cp_c, auto u_c, const f32x4 (&Q)[2][2]) __attribute__((always_inline)) {
    constexpr int CP = decltype(cp_c)::value, u = decltype(u_c)::value, j = u >> 2, tb = (u >> 1) & 1, gb = u & 1;
    const f32x4 m = Q[tb][gb];
    if constexpr (CP == 0) {
      if constexpr (j == 0) T[0][tb][gb] = m;
    } else if constexpr (CP == 1) {
      if constexpr (j == 0) T[0][tb][gb] += m; else T[j][tb][gb] = m;
    } else if constexpr (CP == 2) {
      if constexpr (j & 1) T[j][tb][gb] -= m; else T[j][tb][gb] += m;
    } else if constexpr (CP == 3) {
      constexpr float c = j == 0 ? 1.f : (j == 1 ? 2.f : (j == 2 ? 4.f : 8.f));
      T[j][tb][gb] += c * m;
    } else if constexpr (CP == 4) {
      constexpr float c = j == 0 ? 1.f : (j == 1 ? -2.f : (j == 2 ? 4.f : -8.f));
      T[j][tb][gb] += c * m;
    } else {
      if constexpr (j == 3) T[3][tb][gb] += m;
    }
  };

  // the 12 MFMAs of a piece: product q (g1 v0, g0 v1, g0 v0: smallest terms first) x tile block x (gamma, beta) -- four accumulators
  // in rotation.  FIRST: the position's first piece starts its accumulators (C = 0).
  auto mfma = [&](auto m_c, auto first_c, const Frag& f, f32x4 (&Q)[2][2]) __attribute__((always_inline)) {
    constexpr int m = decltype(m_c)::value, q = m >> 2, tb = (m >> 1) & 1, gb = m & 1;
    constexpr bool FIRST = decltype(first_c)::value != 0 && q == 0;
    if constexpr (DSEE_FW4_ABL & 1) {
      if constexpr (FIRST) Q[tb][gb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      Q[tb][gb][m & 3] += __builtin_bit_cast(float, f.u[2 * gb][0] ^ f.v[tb][0][1]);
    } else {
      const f16x8 ua = __builtin_bit_cast(f16x8, f.u[2 * gb + (q == 0 ? 1 : 0)]);
      const f16x8 vb = __builtin_bit_cast(f16x8, f.v[tb][q == 1 ? 1 : 0]);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      Q[tb][gb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua, vb, FIRST ? z : Q[tb][gb], 0, 0, 0);
    }
  };

  // ---- ring protocol (piece-granular): the ring holds 2 NP pieces, piece (pos, k) in slot (pos % 2) NP + k.  At the barrier that
  //      opens piece P every wave has the fragments of P in registers, so slot(P) is free: the request for piece P + 2 NP goes
  //      into it during piece P, while the fragments of P + 1 are read -- 2 NP - 1 pieces (144 KB at K = 160) are requested or in
  //      flight per CU instead of the stage protocol's ~1.5 stages.  The 8-wave kernel is bound by exactly this window: its
  //      look-ahead requests cost 0.8 of its 2.2 ms, no other part more than 0.25 (profiles/r06_fused_w4.txt).
  constexpr int RING = 2 * NP;
  auto request = [&](int pos, auto slot_c, auto i_c) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;      // = (pos % 2) * NP + k
    dma(slot / NP, slot % NP, base_u + pos * PU, base_v + pos * PV, i_c);
  };
  static_for<RING>([&](auto sl) {      // pieces 0 .. 2 NP - 1 = positions 0 and 1
    static_for<IPP>([&](auto i) { request(decltype(sl)::value / NP, sl, i); });
  });
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPP * (RING - 1)) : "memory");      // piece 0 landed (this wave's rows)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_for<8>([&](auto i) { ldf(F[0], ic<0>{}, ic<0>{}, i); });

  // One transform position (column C of its row; ring slot group and product parity C % 2): NP pieces in two stages.  `valu(k, m)`
  // is the VALU slice that goes behind MFMA m of piece k.
  auto position = [&](int pos, auto c_c, auto&& valu) __attribute__((always_inline)) {
    constexpr int Cc = decltype(c_c)::value, PAR = Cc % RD, PP = Cc & 1;
    const int pl = min(pos + 2, 35);      // the position requested while this one computes (clamped: the surplus lands in dead slots)
    static_for<NP>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      constexpr int fcur = (k + (NP & 1) * PP) & 1;      // fragment set: parity of the running piece count
      // piece P + 1 has landed (this wave's rows) once only the requests of pieces P + 2 .. P + 2 NP - 1 are in flight; the
      // fragments of piece P have arrived; the barrier publishes P + 1 and retires slot(P)
      if constexpr (!(DSEE_FW4_ABL & 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPP * (RING - 2)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      static_for<12>([&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        mfma(m_c, ic<k == 0>{}, F[fcur], P[PP]);
        __builtin_amdgcn_sched_barrier(0);
        // one fragment read of the next piece behind each of the first 8 MFMAs, the 4 requests of piece P + 2 NP behind the last 4
        if constexpr (m < 8) {
          if constexpr (k + 1 < NP)
            ldf(F[1 - fcur], ic<PAR>{}, ic<k + 1>{}, m_c);
          else
            ldf(F[1 - fcur], ic<(PAR + 1) % RD>{}, ic<0>{}, m_c);
        } else if constexpr (!(DSEE_FW4_ABL & 8)) {
          request(pl, ic<PAR * NP + k>{}, ic<m - 8>{});
        }
        if constexpr (!(DSEE_FW4_ABL & 4)) valu(k_c, m_c);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // Row r of the positions.  Position (r, C) folds the product of position (r, C - 1) -- (r - 1, 5) for C = 0 -- under its MFMAs;
  // (r, 0) then also runs Y += At[.][r - 1] (x) T for the finished row r - 1.
#pragma unroll 1
  for (int r = 0; r < 6; ++r) {
    const int q = r - 1;   // At[i][q] (row r - 1 of the positions)
    const float cf[4] = {q < 0 ? 0.f : 1.f, q <= 0 ? 0.f : (q == 1 ? 1.f : (q == 2 ? -1.f : (q == 3 ? 2.f : -2.f))),
                         q <= 0 ? 0.f : (q < 3 ? 1.f : 4.f),
                         q <= 0 ? 0.f : (q == 1 ? 1.f : (q == 2 ? -1.f : (q == 3 ? 8.f : -8.f)))};
    position(r * 6, ic<0>{}, [&](auto k_c, auto m_c) __attribute__((always_inline)) {
      constexpr int s = decltype(k_c)::value * 12 + decltype(m_c)::value, NS = NP * 12;
      if (q >= 0) {     // (wave-uniform: nothing to fold or update in front of the very first position)
        // slots 0 .. 3: the last fold of row r - 1 (column 5: only T[3] takes it)
        if constexpr (s < 4) fold_unit(ic<5>{}, ic<12 + s>{}, P[1]);
        // slots 4 ..: the 64 groups (i, j, tile block, half of the 8 elements) of the Y update, output row i by output row
        else {
          constexpr int g0 = (s - 4) * 64 / (NS - 4), g1 = (s - 3) * 64 / (NS - 4);
          static_for<g1 - g0>([&](auto d) {
            constexpr int gi = g0 + decltype(d)::value, i = gi >> 4, j = (gi >> 2) & 3, tb = (gi >> 1) & 1, e0 = (gi & 1) * 4;
            // rows of At with a zero in column q need no update: q = 0 touches output row 0 only
            if (q > 0 || i == 0) y_rmw(ic<i>{}, ic<j>{}, ic<tb>{}, ic<e0>{}, ic<4>{}, cf[i]);
          });
        }
      }
    });
#define DSEE_POS(C)                                                                                               \
  position(r * 6 + C, ic<C>{}, [&](auto k_c, auto m_c) __attribute__((always_inline)) {                            \
    constexpr int s = decltype(k_c)::value * 12 + decltype(m_c)::value;                                            \
    /* the 16 fold units of the previous position, one behind every third MFMA */                                 \
    if constexpr (s % 3 == 0 && s / 3 < 16) fold_unit(ic<C - 1>{}, ic<s / 3>{}, P[(C - 1) & 1]);                   \
  });
    DSEE_POS(1)
    DSEE_POS(2)
    DSEE_POS(3)
    DSEE_POS(4)
    DSEE_POS(5)
#undef DSEE_POS
  }

  // ---- epilogue.  The last fold (position (5, 5), column 5) and the last row's Y update (At[.][5] = (0, 0, 0, 1): output row 3).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the surplus look-ahead requests write LDS too)
  static_for<4>([&](auto u) { fold_unit(ic<5>{}, ic<12 + decltype(u)::value>{}, P[1]); });
  static_for<16>([&](auto gc) {
    constexpr int gi = decltype(gc)::value, j = gi >> 2, tb = (gi >> 1) & 1, e0 = (gi & 1) * 4;
    y_rmw(ic<3>{}, ic<j>{}, ic<tb>{}, ic<e0>{}, ic<4>{}, 1.f);
  });
  // Per pixel row k of the tiles, the block's gamma / beta values go through LDS into pixel-major order G[px][32 ch], B[px][32 ch]
  // (px = 4 * tile + j; 16-byte chunks XOR-swizzled by the tile), then every thread handles (pixel, channel quad) items with
  // 128-byte-line global accesses: 256 pixels x 8 quads = 8 items per thread and pixel row.
  const int chunk_r = tid & 7;                            // read phase: channel quad of the 32-channel group
  const int cq = rg * 32 + chunk_r * 4;
  unsigned boff[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int px = it * 32 + (tid >> 3);
    const int tl = px >> 2, j = px & 3;
    const int tin = (int)(t0 + tl - (long)n * a.tpi);
    const int ty = tin / a.tw, tx = tin - ty * a.tw;
    boff[it] = ((unsigned)((n * a.H + ty * 4) * a.W + tx * 4 + j) * (unsigned)a.C + (unsigned)cq) * 4u;
  }
  const unsigned np = (unsigned)(a.T * 16);               // pixels of the tensor
  const unsigned rowbytes = (unsigned)(a.W * a.C) * 4u;
  const char* const xb = reinterpret_cast<const char*>(a.x);
  char* const ob = reinterpret_cast<char*>(a.out);
  char* const sb = reinterpret_cast<char*>(a.scale);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float* const Gs = reinterpret_cast<float*>(smem);
  float* const Bs = reinterpret_cast<float*>(smem + 32768);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + cq), is = *reinterpret_cast<const f32x4*>(a.invstd + cq);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 bg = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + rg * 64 + chunk_r * 4) : z4;
  const f32x4 bb = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + rg * 64 + 32 + chunk_r * 4) : z4;
  float hmax = 0.f, xmax = 0.f;
  f32x4 xr[8];
  static_for<4>([&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    // this pixel row's x values are requested before the exchange, so that they travel while it runs
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      unsigned off = boff[it];
      asm volatile("" : "+v"(off));
      xr[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xb + (off + k * rowbytes)));
    }
    static_for<2>([&](auto tb_c) {
      constexpr int tb = decltype(tb_c)::value;
      const int tl_w = 32 * wt + 16 * tb + fi;             // this lane's tile (of tile block tb) within the block
      const int chw = ((wq * 4 + oc) ^ (tl_w & 7)) * 4;    // this lane's channel quad 16 wq + 4 oc, swizzled
      static_for<4>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        const int px = tl_w * 4 + j;
        f32x4 gv, bv;
        static_for<4>([&](auto e_c) {
          constexpr int e = decltype(e_c)::value;
          float yg, yb;
          float& rg_ = Y[k][j][tb][e];
          float& rb_ = Y[k][j][tb][4 + e];
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(yg) : "a"(rg_));
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(yb) : "a"(rb_));
          gv[e] = yg;
          bv[e] = yb;
        });
        *reinterpret_cast<f32x4*>(Gs + px * 32 + chw) = gv;
        *reinterpret_cast<f32x4*>(Bs + px * 32 + chw) = bv;
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int px = it * 32 + (tid >> 3);
      const int tl = px >> 2;
      unsigned off = boff[it];
      asm volatile("" : "+v"(off));
      off += k * rowbytes;
      const int ch = (chunk_r ^ (tl & 7)) * 4;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + px * 32 + ch);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + px * 32 + ch);
      const f32x4 xh = (xr[it] - mu) * is;
      const f32x4 sc = gv * oscale + bg + a.add_one;
      if constexpr (WSCALE && !(DSEE_FW4_ABL & 16)) __builtin_nontemporal_store(sc, reinterpret_cast<f32x4*>(sb + off));
      f32x4 v = (xh * sc + bb) + bv * oscale;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
      if constexpr (!(DSEE_FW4_ABL & 16)) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ob + off));
      if constexpr (WSCALE) {
        // the LeakyReLU branch as bits (spade_fused.hip): lane l = 8 * pixel + channel quad, byte `pixel` of ballot e holds element
        // e of the pixel's 8 quads; word of (pixel, 32-channel group): bit 8 * (c & 3) + ((c & 31) >> 2)
        const int pl = (tid >> 3) & 7;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned long long b = __builtin_amdgcn_ballot_w64(v[e] > 0.f);
          const unsigned half = (pl & 4) ? (unsigned)(b >> 32) : (unsigned)b;
          m |= __builtin_amdgcn_ubfe(half, 8 * (pl & 3), 8) << (8 * e);
        }
        if (chunk_r == 0 && a.mask) a.mask[rg * np + (off >> a.cshift)] = m;
      }
      hmax = fmaxf(hmax, dsee_absmax4(v));
      xmax = fmaxf(xmax, dsee_absmax4(xh));
    }
    if constexpr (k < 3) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  });
  if (a.amax_h) {   // one read-before-atomic max per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hmax = fmaxf(hmax, __shfl_xor(hmax, o, 64));
    if (lane == 0) {
      unsigned* line = reinterpret_cast<unsigned*>(a.amax_h + ((blockIdx.x * 4 + wave) & (DSEE_AMAX_LINES - 1)) * DSEE_AMAX_STRIDE);
      const unsigned bits = __builtin_bit_cast(unsigned, hmax);
      if (bits > __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(line, bits);
    }
  }
  if (a.amax_xhat) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
    if (lane == 0) {
      unsigned* line = reinterpret_cast<unsigned*>(a.amax_xhat + ((blockIdx.x * 4 + wave) & (DSEE_AMAX_LINES - 1)) * DSEE_AMAX_STRIDE);
      const unsigned bits = __builtin_bit_cast(unsigned, xmax);
      if (bits > __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(line, bits);
    }
  }
#endif
}

}  // namespace

extern "C" {

/* dsee_spade_fused_fwd on the one-wave-per-SIMD kernel (see the head of this file): same operands, same arguments, bit-identical
 * results.  K = 128 or 160, two-term fp16x2 operands. */
int dsee_spade_fused_fwd_w4(const void* V2, const void* U2, const float* amax_cat, float v_bound, const float* amax_u,
                            const float* bias_packed, const float* x, const float* mean, const float* invstd, float* out_h,
                            float* out_scale, int N, int H, int W, int C, int rows, int K, int groups, float add_one,
                            float slope, float* amax_h, float* amax_xhat, uint32_t* sign_mask, hipStream_t st) {
  DSEE_CHECK_ARG(V2 && U2 && amax_cat && amax_u && x && mean && invstd && out_h);
  DSEE_CHECK_ARG(rows == 2 * C && C % 32 == 0 && H % 4 == 0 && W % 4 == 0 && (K == 128 || K == 160));
  DSEE_CHECK_ARG(groups == 1 || groups == N);
  const int tpi = (H / 4) * (W / 4);
  DSEE_CHECK_ARG(tpi % 64 == 0);
  const long T = (long)N * tpi;
  FusedArgs a;
  a.V2 = (const unsigned char*)V2;
  a.U2 = (const unsigned char*)U2;
  a.amax_v = amax_cat;
  a.amax_u = amax_u;
  a.bias = bias_packed;
  a.x = x;
  a.mean = mean;
  a.invstd = invstd;
  a.out = out_h;
  a.scale = out_scale;
  a.mask = sign_mask;
  a.cshift = 0;
  while ((4 << a.cshift) < 4 * C) ++a.cshift;
  a.cshift += 2;
  DSEE_CHECK_ARG(!sign_mask || (C & (C - 1)) == 0);             // (the mask index is formed by a shift)
  DSEE_CHECK_ARG((long)N * H * W * C < (1L << 30));             // (32-bit byte offsets into x / h / scale)
  a.amax_h = amax_h;
  a.amax_xhat = amax_xhat;
  a.T = T;
  a.v_slab_bytes = 36L * T * 64;
  a.u_slab_bytes = (long)rows * 64;
  a.u_group_bytes = (long)(K / 16) * rows * 64;
  const long vb = a.v_slab_bytes * (K / 16), ub = a.u_group_bytes * 36 * groups;
  DSEE_CHECK_ARG(vb < 0xFFFFFFF0L && ub < 0xFFFFFFF0L);
  a.v_bytes = (unsigned)vb;
  a.u_bytes = (unsigned)ub;
  a.tpi = tpi;
  a.tw = W / 4;
  a.H = H;
  a.W = W;
  a.C = C;
  a.rows = rows;
  a.G = groups;
  a.v_bound = v_bound;
  a.add_one = add_one;
  a.slope = slope;
  a.stamps = nullptr;
  const long ntile = (T / 64) * (rows / 64);
  DSEE_CHECK_ARG(ntile < 0x7FFFFFFF);
  const int np = K / 32;
  const size_t lds = (size_t)2 * np * 16384;
#define DSEE_FW4(NP, WS)                                                                                             \
  do {                                                                                                               \
    static bool attr_done = false;                                                                                   \
    if (!attr_done) {                                                                                                \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spade_fused_w4_kernel<NP, WS>),              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e != hipSuccess) {                                                                                         \
        dsee_set_error("hipFuncSetAttribute(%zu bytes of LDS): %s", lds, hipGetErrorString(e));                     \
        return DSEE_ELAUNCH;                                                                                         \
      }                                                                                                              \
      attr_done = true;                                                                                              \
    }                                                                                                                \
    spade_fused_w4_kernel<NP, WS><<<(int)ntile, 256, lds, st>>>(a);                                                  \
  } while (0)
  if (np == 5) {
    if (out_scale) DSEE_FW4(5, true); else DSEE_FW4(5, false);
  } else {
    if (out_scale) DSEE_FW4(4, true); else DSEE_FW4(4, false);
  }
#undef DSEE_FW4
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // extern "C"
