// The fused SPADE / SEAN normalisation forward: gamma/beta convolution + BN-normalise + modulate + LeakyReLU in ONE kernel
// (normalization.py:107-120 SPADE, :167-213 SEAN, :258-286 PureSEAN above max_fm_size; architecture.py:92,114 LeakyReLU).
//
//   h = lrelu( (x - mean) * invstd * (conv3x3(cat, W_gamma') + b_gamma + add_one) + conv3x3(cat, W_beta') + b_beta )
//
// The 3x3 convolution over the K = 128 / 160 channel embedding `cat` runs as Winograd F(4x4,3x3): 36 small GEMMs
// M[xi][row][tile] = U[xi][row][:] . V[xi][tile][:] (fp16x2-split operands, 3 MFMA products per multiply-add, fp32
// accumulate -- gemm_bf16x3.hip).  Rounds 1-2 wrote M (4.8 GB at N = 8, 256^2, C = 512) to HBM and read it back in a
// separate output-transform kernel.  Here a workgroup owns 64 tiles x 64 packed rows (= 32 channels, gamma and beta),
// walks ALL 36 transform positions for them and folds each position's product into the 4x4 output tile in registers:
//
//   Y = A^T M A  separably:   T[j] += At[j][c] * M[r][c]   (after every position, 4 partial columns)
//                              Y[i][j] += At[i][r] * T[j]   (after every row r of positions)
//
// so M never exists outside the register file.
//
// Work split: 8 waves (2 row halves x 4 tile quarters), two per SIMD, so that the LDS-DMA issue, LDS reads and the VALU
// folds of one wave overlap the partner wave's matrix work (a first version with 4 waves of 32x32 MFMA blocks, one per
// SIMD, was bound by the in-order issue of a single wave: removing the MFMAs changed nothing, removing DMA / folds /
// fragment reads saved 0.8 / 0.5 / 0.4 ms of 3.0).  A wave owns two 16x16 blocks of v_mfma_f32_16x16x32_f16 -- the 16
// gamma rows and the 16 beta rows of the same 16 channels -- for 16 tiles: a lane ends with ONE tile and 4 consecutive
// channels x (gamma, beta).  Y = 16 outputs x 8 = 128 accumulator registers per wave (AGPRs, updated in place once per row
// of positions: the VALU cannot address them), T = 32, MFMA accumulators 2 x 8.
// Operands travel global -> LDS by buffer_load ... lds in 32-k pieces (64 rows x 128 B = 2 terms x 4 k-octets; one
// instruction per wave and piece) into a ring of two positions; a position is two stages (2 + 2|3 pieces), one barrier
// each, the requests of stage s + 3 are issued during stage s.  The two waves of a SIMD run each piece in complementary order (MFMAs
// first / loads and folds first).  The piece image is XOR-swizzled (16-byte chunk cc of row r
// at slot 8r + (cc ^ ((r >> 1) & 7))): every ds_read_b128 fragment read is conflict free.
// The epilogue requests the block's x values first (64 KB in flight per CU), swaps the results through LDS into
// pixel-major order and reads / writes whole 128-byte lines.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "dsee_common.h"

// Measurement builds only (tools/exp/build_fused_abl.sh): bit 1 no MFMAs, 2 no fragment reads, 4 no fold / Y update,
// 8 no look-ahead LDS-DMA, 16 no epilogue, 32 cycle stamps, 64 all look-ahead requests read the same (cache-hot) piece,
// 128 no sign-mask code.  The shipped library is built without the macro.
#ifndef DSEE_FUSED_ABL
#define DSEE_FUSED_ABL 0
#endif

// Schedule variants for same-box A/B (tools/exp/build_fused_var.sh): bit 1 the early wave group requests the next piece's
// fragments before its MFMAs, 2 s_setprio(1) around the MFMAs of a piece, 4 static priority for waves 4-7, 8 the look-ahead
// LDS-DMA requests of a piece issued between its MFMAs instead of back to back.
#ifndef DSEE_FUSED_SCHED
#define DSEE_FUSED_SCHED 0
#endif

// Which form of the kernel a launch takes unless the environment says otherwise (DSEE_FUSED_W16): 0 = 8 waves of 256
// registers, 1 = 16 waves of 128 (round 5, see spade_fused_fwd16_kernel).
// Shape of the set of 32 workgroups that co-run on one XCD: DSEE_FUSED_SET_A tile groups x 32 / A row groups (a set fetches A V
// strips + 32 / A U strips; 4 x 8 and 8 x 4 are the optimum, measurement builds vary it to check the traffic model of DESIGN 3.12)
#ifndef DSEE_FUSED_SET_A
#define DSEE_FUSED_SET_A 4
#endif
#define DSEE_FUSED_SET_B (32 / DSEE_FUSED_SET_A)

#ifndef DSEE_FUSED_W16_DEFAULT
#define DSEE_FUSED_W16_DEFAULT 0
#endif

namespace {

}  // namespace
#include "spade_fused_args.h"
namespace {

// PK (16-bit storage mode, opt.precision = "fp16"): the operands are PACKED ONE-TERM images ([K/32][rows][32] fp16: the same
// 64-byte rows holding 32 k's of one scaled fp16 term, gemm_bf16x3.hip) -- a piece is then 64 k's (its two "term" halves are
// k's 0-31 and 32-63), a position NP = 2 (K = 128) or 3 (K = 160: the last piece half empty, fetched as zeros) pieces, and a
// piece costs 2 MFMA products per block instead of 3.  Ring, fragment reads, fold and epilogue are unchanged.
// RD: depth of the operand ring in transform positions.  The requests of stage s + 2 RD - 1 are issued during stage s, so
// (2 RD - 1) / 2 positions of operands are in flight per CU: 123 KB with RD = 2 at K = 160 in the two-term form (the whole
// 160 KB of LDS is the ring), and what the kernel sustains is that window divided by the loaded L2 / HBM latency -- round 5
// measured it: no schedule change moves the kernel, removing the requests does.  The packed one-term operands are half the
// bytes, so their ring takes RD = 3 (144 KB at K = 160, 2.5 positions = 102 KB in flight instead of 1.5 = 61 KB).
template <int NP, bool WSCALE, bool PK = false, int RD = 2>
__global__ __launch_bounds__(512) void spade_fused_fwd_kernel(FusedArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PIECE = 8192;                  // 32 k's of one operand: 64 rows x (2 terms x 4 octets x 16 B)   [PK: 64 k's]
  constexpr int UREG = RD * NP * PIECE;        // LDS: [RD NP pieces of U][RD NP pieces of V]; piece (par, k) in slot par * NP + k
  static_assert(RD == 2 || RD == 3, "a row of 6 positions is a whole number of ring turns");
  static_assert(2 * UREG <= 160 * 1024 && 2 * UREG >= 65536, "ring fits the LDS and covers the epilogue's 64 KB");
  static_assert(PK ? (NP == 2 || NP == 3) : (NP == 4 || NP == 5), "K = 128 or 160");
  constexpr bool HALF_LAST = PK && NP == 3;    // K = 160 = 2.5 pieces of 64 k's
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * POSB (>= 64 KB for the epilogue)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 2, wt = wave & 3;     // row half (16 channels), tile quarter (16 tiles)

  // ---- workgroup -> (tile group of 64 tiles, row group of 64 packed rows).  Blocks b, b + 8, ... share an XCD (and
  //      its L2): each XCD walks a contiguous range of the list, 32 consecutive entries (one per CU) = 4 tile groups x 8
  //      row groups, so that a co-running set re-reads 4 V strips and 8 U strips from L2 instead of HBM.
  const int rgn = a.rows >> 6;
  const long tgn = a.T >> 6, ntile = tgn * rgn;
  long l;
  {
    const long v = blockIdx.x, q = ntile >> 3, r = ntile & 7, xcd = v & 7, idx = v >> 3;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  long tg;
  int rg;
  if ((rgn % DSEE_FUSED_SET_B) == 0 && (tgn % DSEE_FUSED_SET_A) == 0) {
    const long sup = l >> 5;
    const int in = (int)(l & 31), rh = rgn / DSEE_FUSED_SET_B;
    tg = (sup / rh) * DSEE_FUSED_SET_A + (in / DSEE_FUSED_SET_B);
    rg = (int)(sup % rh) * DSEE_FUSED_SET_B + (in % DSEE_FUSED_SET_B);
  } else {
    tg = l / rgn;
    rg = (int)(l % rgn);
  }
  const long t0 = tg * 64;                    // first tile (of the batch)
  const int n = (int)(t0 / a.tpi);            // its image
  const int g = a.G > 1 ? n : 0;

  // ---- operand scales (exact powers of two), undone once on the 4x4 results
  const float sv = dsee_pow2_scale(a.v_bound * dsee_amax_read(a.amax_v));
  const float su = dsee_pow2_scale(dsee_amax_read(a.amax_u));
  const float oscale = 1.f / (sv * su);

  // ---- LDS-DMA: wave w fills rows 8w .. 8w+7 of every piece (64 slots); lane -> (row 8w + l/8, slot chunk l%8) fetches
  //      chunk cc = (l%8) ^ f(row) = 4 slab + 2 term + octet%2: bytes (cc % 4) * 16 of the row in 16-k slab 2 piece + cc / 4 --
  //      the 4 lanes of a quad fetch the 64 contiguous bytes of ONE (row, slab) in some order, so the texture addresser sees
  //      one 64-byte segment per quad (round 5; with cc = 4 term + octet a quad straddled both slabs: two segments per quad).
  //      PK: chunk cc = 4 half + octet = k's 8 cc .. 8 cc + 7 of the 64-k piece: bytes (cc % 4) * 16 of the row in 32-k slab
  //      2 piece + cc / 4 (the same formulas).
  const int dr = 8 * wave + (lane >> 3);
  const int dcc = (lane & 7) ^ ((dr >> 1) & 7);
  const int dslab = dcc >> 2;                      // which of the piece's two slabs this lane reads
  const unsigned dlo = (unsigned)(dr * 64 + (dcc & 3) * 16);
  const unsigned voffu = dlo + (unsigned)dslab * (unsigned)a.u_slab_bytes;
  const unsigned voffv = dlo + (unsigned)dslab * (unsigned)a.v_slab_bytes;
  // the half-empty last piece: its second slab does not exist -- out-of-range offsets make the buffer loads return zeros
  const unsigned voffu_l = dslab ? 0xFFFFFFF0u : voffu, voffv_l = dslab ? 0xFFFFFFF0u : voffv;
  const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(a.U2), 0, (int)a.u_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(a.V2), 0, (int)a.v_bytes, 0x00020000);
  const unsigned PU = (unsigned)(a.G * a.u_group_bytes), PV = (unsigned)(a.T * 64);       // per position
  const unsigned QU = (unsigned)(2 * a.u_slab_bytes), QV = (unsigned)(2 * a.v_slab_bytes);   // per piece
  const unsigned base_u = (unsigned)((long)g * a.u_group_bytes + (long)rg * 4096), base_v = (unsigned)(t0 * 64);
  // piece pc of position pos (ring slot group par = pos % RD): one U and one V instruction per wave
  auto dma_u = [&](int par, int pc, unsigned opos) {
    unsigned char* dst = smem + (par * NP + pc) * PIECE + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (__attribute__((address_space(3))) void*)dst, 16,
                                             (HALF_LAST && pc == NP - 1) ? voffu_l : voffu, opos + pc * QU, 0, 0);
  };
  auto dma_v = [&](int par, int pc, unsigned opos) {
    unsigned char* dst = smem + UREG + (par * NP + pc) * PIECE + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (__attribute__((address_space(3))) void*)dst, 16,
                                             (HALF_LAST && pc == NP - 1) ? voffv_l : voffv, opos + pc * QV, 0, 0);
  };

  // ---- fragment addresses (bytes within a piece): row r, chunk 4 (octet / 2) + 2 term + octet % 2 [PK: 4 half + octet] at slot
  //      8r + (chunk ^ f(r)).
  //      A operand: lane -> (row l%16 of its 16-row block, octet l/16); gamma block rows 16 wq + i, beta block rows
  //      32 + 16 wq + i of the 64-row group; B operand: tiles 16 wt + i.
  const int fi = lane & 15, oc = lane >> 4;
  auto foff = [&](int r, int t) {
    const int chunk = PK ? 4 * t + oc : 4 * (oc >> 1) + 2 * t + (oc & 1);
    return (unsigned)((8 * r + (chunk ^ ((r >> 1) & 7))) * 16);
  };
  const int rga = 16 * wq + fi, rtv = 16 * wt + fi;
  // DS instructions address VGPR + 16-bit immediate: two windows (slots 0-6, 7-9) per fragment kind cover each 80 / 64 KB region (the
  // beta block's rows are the gamma block's + 32: same swizzle, + 4096 bytes)
  const unsigned au0 = foff(rga, 0), au1 = foff(rga, 1), au0h = au0 + 7 * PIECE, au1h = au1 + 7 * PIECE;
  const unsigned av0 = UREG + foff(rtv, 0), av1 = UREG + foff(rtv, 1), av0h = av0 + 7 * PIECE, av1h = av1 + 7 * PIECE;
  struct Frag {
    u32x4 g0, g1, b0, b1, v0, v1;
  };
  auto ldf = [&](Frag& f, auto par_c, auto pc_c) {
    constexpr int so = (decltype(par_c)::value * NP + decltype(pc_c)::value) * PIECE;   // slot offset within the region
    constexpr bool hi = so >= 7 * PIECE;
    constexpr int io = hi ? so - 7 * PIECE : so;
    static_assert(io >= 0 && io + 4096 + PIECE <= 65536, "window");
    const unsigned char* bu0 = smem + (hi ? au0h : au0);
    const unsigned char* bu1 = smem + (hi ? au1h : au1);
    const unsigned char* bv0 = smem + (hi ? av0h : av0);
    const unsigned char* bv1 = smem + (hi ? av1h : av1);
    f.g0 = *reinterpret_cast<const u32x4*>(bu0 + io);
    f.g1 = *reinterpret_cast<const u32x4*>(bu1 + io);
    f.b0 = *reinterpret_cast<const u32x4*>(bu0 + io + 4096);
    f.b1 = *reinterpret_cast<const u32x4*>(bu1 + io + 4096);
    f.v0 = *reinterpret_cast<const u32x4*>(bv0 + io);
    f.v1 = *reinterpret_cast<const u32x4*>(bv1 + io);
  };
  // the three products of a piece for both blocks, alternating between the two accumulators
  auto mm = [&](const Frag& f, f32x4& pg, f32x4& pb, auto&& h1, auto&& h2) {
    if constexpr (DSEE_FUSED_SCHED & 2) __builtin_amdgcn_s_setprio(1);
    const f16x8 g0 = __builtin_bit_cast(f16x8, f.g0), g1 = __builtin_bit_cast(f16x8, f.g1);
    const f16x8 b0 = __builtin_bit_cast(f16x8, f.b0), b1 = __builtin_bit_cast(f16x8, f.b1);
    const f16x8 v0 = __builtin_bit_cast(f16x8, f.v0), v1 = __builtin_bit_cast(f16x8, f.v1);
    if constexpr (DSEE_FUSED_ABL & 1) {
      pg[0] += (float)(g1[0] + v0[1]) + (float)(g0[2] + v1[3]);
      pb[1] += (float)(b1[0] + v0[1]) + (float)(b0[2] + v1[3]);
    } else if constexpr (PK) {   // one term: the two chunk groups are k's 0-31 and 32-63 of the piece
      pg = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0, v0, pg, 0, 0, 0);
      pb = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, v0, pb, 0, 0, 0);
      h1();
      pg = __builtin_amdgcn_mfma_f32_16x16x32_f16(g1, v1, pg, 0, 0, 0);
      pb = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, v1, pb, 0, 0, 0);
      h2();
    } else {
      pg = __builtin_amdgcn_mfma_f32_16x16x32_f16(g1, v0, pg, 0, 0, 0);
      pb = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, v0, pb, 0, 0, 0);
      h1();
      pg = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0, v1, pg, 0, 0, 0);
      pb = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, v1, pb, 0, 0, 0);
      h2();
      pg = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0, v0, pg, 0, 0, 0);
      pb = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, v0, pb, 0, 0, 0);
    }
    if constexpr (DSEE_FUSED_SCHED & 2) __builtin_amdgcn_s_setprio(0);
  };

  // Everything from here on exists twice, once per wave group (see `late` below): the accumulator-resident state never
  // crosses a control-flow merge (the register allocator spilled all of it at the merge otherwise).
  auto body = [&](auto late_c) {
  // Y lives in the accumulator half of the register file (128 AGPRs per wave; the VALU cannot address them, so an update
  // is v_accvgpr_read -> v_fmac -> v_accvgpr_write in place, once per row of positions); everything the VALU touches per
  // position (T, two pairs of MFMA accumulators, fragments) stays within the 128 architectural VGPRs.  The file is
  // compiled with -mllvm -amdgpu-mfma-vgpr-form so that the MFMA accumulators do not compete for AGPRs.
  float Y[4][4][8];      // [output row i][output column j][gamma 0..3 | beta 4..7]
  f32x4 T[4][2], P[2];
  Frag F[2];
  static_for<4>([&](auto i) {
    static_for<4>([&](auto j) {
      static_for<8>([&](auto e) {
        float& yr = Y[decltype(i)::value][decltype(j)::value][decltype(e)::value];
        asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(yr));
      });
    });
  });
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < 2; ++b) T[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- ring: RD NP pieces (RD positions), piece (pos, k) in slot (pos % RD) * NP + k.  A position is two stages (pieces
  //      [0, NA) and [NA, NP)), one barrier each; the requests of stage s + 2 RD - 1 (RD = 2: s + 3) are issued during stage s
  //      into the slots of stage s - 1 (dead: every wave consumed those fragments before it arrived at the barrier that opens s).
  constexpr int NA = NP == 2 ? 1 : 2;
  auto issue_stage = [&](int pos, int half) {   // prologue only (run-time indices)
    for (int k = half ? NA : 0; k < (half ? NP : NA); ++k) {
      dma_u(pos % RD, k, base_u + pos * PU);
      dma_v(pos % RD, k, base_v + pos * PV);
    }
  };
  for (int st = 0; st < 2 * RD - 1; ++st) issue_stage(st >> 1, st & 1);
  // stage 0 landed (this wave's rows): only the 2 RD - 2 younger stages = RD - 1 whole positions may be in flight
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP * (RD - 1)) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ldf(F[0], ic<0>{}, ic<0>{});
  // (the two waves of a SIMD, w and w + 4, run the same pieces in complementary order: the early one issues its 6 MFMAs
  // first and its LDS reads / DMA requests / VALU folds afterwards, the late one the other way round -- in lock-step both
  // wanted the same pipe at the same time and nothing overlapped)

  // Y[i][j] += cf[i] * T[j] for the (i, j) pairs [LO, HI) of the 16
  auto y_update = [&](auto lo_c, auto hi_c, const float (&cf)[4]) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    static_for<HI - LO>([&](auto d) {
      constexpr int ij = LO + decltype(d)::value, i = ij >> 2, j = ij & 3;
      static_for<8>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        float y;   // in place: the accumulator register is both input and output, so Y never moves between AGPRs
        float& yr = Y[i][j][e];
        const float cc = cf[i], tt = T[j][e >> 2][e & 3];
        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_fmac_f32 %1, %2, %3\n\tv_accvgpr_write_b32 %0, %1"
                     : "+a"(yr), "=&v"(y)
                     : "s"(cc), "v"(tt));
      });
    });
  };

#if DSEE_FUSED_ABL & 32
  unsigned long long tw_ = 0, tm_ = 0, tl_ = 0, t0_ = 0, t1_ = 0, tstart_ = __builtin_readcyclecounter();
  unsigned long long tq0_ = 0, tq1_ = 0, tq2_ = 0, tv_ = 0;   // fragment reads | DMA requests | fold / Y update | vmcnt wait alone
#endif
  // One transform position = NP pieces in two stages.  During piece k the fragments of piece k + 1 (of the next position at
  // the end) are read, a share of the look-ahead stage s + 3 is requested and `fill(k)` runs (the Y update of the previous
  // row of positions).  Between barriers the waves drift freely.  pos is wave-uniform, its ring slot group PAR = pos % RD and
  // its parity PP
  // compile-time constant.
  auto position = [&](auto late_c, int pos, auto par_c, auto pp_c, auto&& fill) {
    constexpr bool LATE = decltype(late_c)::value != 0;
    constexpr int PAR = decltype(par_c)::value, PP = decltype(pp_c)::value;
    P[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    P[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<2>([&](auto half_c) {
      constexpr int half = decltype(half_c)::value;
      // look-ahead stage s + 2 RD - 1 = (pos + RD - 1, second half) | (pos + RD, first half), clamped to the last position
      // (the surplus requests of the last 2 RD - 1 stages re-read valid memory into slots nobody reads any more)
      const int pl = (DSEE_FUSED_ABL & 64) ? 0 : min(pos + RD - 1 + half, 35);   // (64: every request re-reads position 0)
      const unsigned ou = base_u + pl * PU, ov = base_v + pl * PV;
      constexpr int LPAR = half ? PAR : (PAR + RD - 1) % RD;   // ring slot group of the look-ahead position
      constexpr int L0 = half ? 0 : NA, L1 = half ? NA : NP;   // its pieces
      constexpr int C0 = half ? NA : 0, C1 = half ? NP : NA;   // the pieces computed now
#if DSEE_FUSED_ABL & 32
      t0_ = __builtin_readcyclecounter();
#endif
      // stage s + 1 has landed (this wave's rows) when only the requests of stages s + 2 .. s + 2 RD - 2 are still in flight:
      // RD - 1 stages of this half's piece count and RD - 2 of the other half's
      if constexpr (DSEE_FUSED_ABL & 8)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ((RD - 1) * (C1 - C0) + (RD - 2) * (NP - (C1 - C0)))) : "memory");
#if DSEE_FUSED_ABL & 32
      tv_ += __builtin_readcyclecounter() - t0_;
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#if DSEE_FUSED_ABL & 32
      t1_ = __builtin_readcyclecounter();
      tw_ += t1_ - t0_;
#endif
      static_for<C1 - C0>([&](auto p_c) {
        constexpr int p = decltype(p_c)::value, k = C0 + p;
        constexpr int fcur = (k + (NP & 1) * PP) & 1;    // fragment buffer: parity of the running piece count pos * NP + k
        auto loads = [&](auto frags_c, auto rest_c) {
          constexpr bool FRAGS = decltype(frags_c)::value != 0, REST = decltype(rest_c)::value != 0;
#if DSEE_FUSED_ABL & 32
          unsigned long long q0_ = __builtin_readcyclecounter();
          __builtin_amdgcn_sched_barrier(0);
#endif
          if constexpr (FRAGS && !(DSEE_FUSED_ABL & 2)) {
            if constexpr (k + 1 < NP)
              ldf(F[1 - fcur], ic<PAR>{}, ic<k + 1>{});
            else
              ldf(F[1 - fcur], ic<(PAR + 1) % RD>{}, ic<0>{});
          }
#if DSEE_FUSED_ABL & 32
          __builtin_amdgcn_sched_barrier(0);
          { const unsigned long long tt = __builtin_readcyclecounter(); tq0_ += tt - q0_; q0_ = tt; }
          __builtin_amdgcn_sched_barrier(0);
#endif
          if constexpr (REST && !(DSEE_FUSED_ABL & 8) && !(DSEE_FUSED_SCHED & 8)) {
            // the L1 - L0 look-ahead pieces spread over the C1 - C0 computed ones
            constexpr int a0 = L0 + p * (L1 - L0) / (C1 - C0), a1 = L0 + (p + 1) * (L1 - L0) / (C1 - C0);
            static_for<a1 - a0>([&](auto d) {
              dma_u(LPAR, a0 + decltype(d)::value, ou);
              dma_v(LPAR, a0 + decltype(d)::value, ov);
            });
          }
#if DSEE_FUSED_ABL & 32
          __builtin_amdgcn_sched_barrier(0);
          { const unsigned long long tt = __builtin_readcyclecounter(); tq1_ += tt - q0_; q0_ = tt; }
          __builtin_amdgcn_sched_barrier(0);
#endif
          if constexpr (REST && !(DSEE_FUSED_ABL & 4)) fill(ic<k>{});
#if DSEE_FUSED_ABL & 32
          __builtin_amdgcn_sched_barrier(0);
          { const unsigned long long tt = __builtin_readcyclecounter(); tq2_ += tt - q0_; }
#endif
        };
        // DSEE_FUSED_SCHED & 8: this piece's share of the look-ahead requests is issued BETWEEN its MFMAs, half behind the
        // second and half behind the fourth, instead of back to back in the load part: a request that finds the texture
        // addresser's queue full blocks the (in-order) wave, and 2 - 4 requests in a row from four waves at once did
        auto dma_part = [&](auto part_c) {
          if constexpr ((DSEE_FUSED_SCHED & 8) && !(DSEE_FUSED_ABL & 8)) {
            constexpr int part = decltype(part_c)::value;
            constexpr int a0 = L0 + p * (L1 - L0) / (C1 - C0), a1 = L0 + (p + 1) * (L1 - L0) / (C1 - C0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (a1 - a0 == 1) {
              if constexpr (part == 0) dma_u(LPAR, a0, ou); else dma_v(LPAR, a0, ov);
            } else if constexpr (a1 - a0 == 2) {
              dma_u(LPAR, a0 + part, ou);
              dma_v(LPAR, a0 + part, ov);
            } else {
              static_assert(a1 - a0 == 0, "0, 1 or 2 look-ahead pieces per computed piece");
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        auto h1 = [&]() { dma_part(ic<0>{}); };
        auto h2 = [&]() { dma_part(ic<1>{}); };
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LATE) {
        loads(ic<1>{}, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
#if DSEE_FUSED_ABL & 32
        { const unsigned long long tt = __builtin_readcyclecounter(); tl_ += tt - t1_; t1_ = tt; }
#endif
        mm(F[fcur], P[0], P[1], h1, h2);
        __builtin_amdgcn_sched_barrier(0);
#if DSEE_FUSED_ABL & 32
        { const unsigned long long tt = __builtin_readcyclecounter(); tm_ += tt - t1_; t1_ = tt; }
#endif
      } else {
        // (DSEE_FUSED_SCHED & 1: the early group requests the NEXT piece's fragments before its MFMAs -- the other fragment
        //  buffer is free since the previous piece -- so that their LDS latency runs under its own matrix work instead of in
        //  front of the next piece's; the DMA requests and the fold slices stay behind the MFMAs)
        if constexpr (DSEE_FUSED_SCHED & 1) {
          loads(ic<1>{}, ic<0>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        mm(F[fcur], P[0], P[1], h1, h2);
        __builtin_amdgcn_sched_barrier(0);
#if DSEE_FUSED_ABL & 32
        { const unsigned long long tt = __builtin_readcyclecounter(); tm_ += tt - t1_; t1_ = tt; }
#endif
        loads(ic<(DSEE_FUSED_SCHED & 1) ? 0 : 1>{}, ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
#if DSEE_FUSED_ABL & 32
        { const unsigned long long tt = __builtin_readcyclecounter(); tl_ += tt - t1_; t1_ = tt; }
#endif
      }
      });
    });
  };
  // fold of the product of a position (column CP of its row) into T
  auto fold = [&](auto cp_c, const f32x4& qg, const f32x4& qb) {
    constexpr int CP = decltype(cp_c)::value;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x4 m = b ? qb : qg;
      if constexpr (CP == 0) {
        T[0][b] += m;
      } else if constexpr (CP == 1) {
        T[0][b] += m; T[1][b] += m; T[2][b] += m; T[3][b] += m;
      } else if constexpr (CP == 2) {
        T[0][b] += m; T[1][b] -= m; T[2][b] += m; T[3][b] -= m;
      } else if constexpr (CP == 3) {
        T[0][b] += m; T[1][b] += 2.f * m; T[2][b] += 4.f * m; T[3][b] += 8.f * m;
      } else if constexpr (CP == 4) {
        T[0][b] += m; T[1][b] -= 2.f * m; T[2][b] += 4.f * m; T[3][b] -= 8.f * m;
      } else {
        T[3][b] += m;
      }
    }
  };

  // Row r of the positions: every position folds its product into T when its last MFMA has retired (the partner wave of
  // the SIMD covers the bubble); (r, 0) additionally hides Y += At[.][r-1] (x) T of the previous row in slices under its
  // pieces, then T restarts.  For r = 0 the "previous row" has all-zero coefficients: the same code runs, so the loop body
  // has no conditional blocks.
#pragma unroll 1
  for (int r = 0; r < 6; ++r) {
    const int q = r - 1;   // At[i][q]
    const float cf[4] = {q < 0 ? 0.f : 1.f, q <= 0 ? 0.f : (q == 1 ? 1.f : (q == 2 ? -1.f : (q == 3 ? 2.f : -2.f))),
                         q <= 0 ? 0.f : (q < 3 ? 1.f : 4.f),
                         q <= 0 ? 0.f : (q == 1 ? 1.f : (q == 2 ? -1.f : (q == 3 ? 8.f : -8.f)))};
    position(late_c, r * 6, ic<0>{}, ic<0>{}, [&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      if constexpr (k >= 1) {
        // rows of At with a zero in column q need no update: q = -1 (first row of positions) none, q = 0 only i = 0
        constexpr int NY = NP - 1, qq = k - 1, LO = qq * 16 / NY, HI = (qq + 1) * 16 / NY;
        if constexpr (NP == 5) {          // slice = one output row i = qq
          if (q > 0 || (q == 0 && qq == 0)) y_update(ic<LO>{}, ic<HI>{}, cf);
        } else {
          if (q >= 0) y_update(ic<LO>{}, ic<HI>{}, cf);
        }
        if constexpr (k == NP - 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < 2; ++b) T[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    });
    fold(ic<0>{}, P[0], P[1]);
    // (position r * 6 + C: ring slot group C % RD -- a row is a whole number of ring turns -- and parity C % 2)
#define DSEE_POS(C)                                                            \
  position(late_c, r * 6 + C, ic<C % RD>{}, ic<C % 2>{}, [&](auto) {});        \
  fold(ic<C>{}, P[0], P[1]);
    DSEE_POS(1)
    DSEE_POS(2)
    DSEE_POS(3)
    DSEE_POS(4)
    DSEE_POS(5)
#undef DSEE_POS
  }

  // ---- epilogue.  The x values of the whole block tile (4 pixel rows x 4 items per thread) are requested first, so that
  //      64 KB per CU are in flight while the last Y update and the LDS exchange run.
#if DSEE_FUSED_ABL & 32
  const unsigned long long tmain_ = __builtin_readcyclecounter();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the surplus look-ahead requests write LDS too)
  const int chunk_r = tid & 7;                            // read phase: channel quad of the 32-channel group
  const int cq = rg * 32 + chunk_r * 4;
  // BYTE offsets of this lane's 4 items (pixel row k = 0) into x / h / scale, 32 bits (the host checks the tensor is < 4 GB):
  // half the registers of 64-bit element offsets, and a load / store is `global_* v, v_off, s[base]`
  unsigned boff[4];
  f32x4 xr[4][4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int px = it * 64 + (tid >> 3);
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
  if constexpr (!(DSEE_FUSED_ABL & 16)) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int it = 0; it < 4; ++it)
        xr[k][it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xb + (boff[it] + k * rowbytes)));
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const float cf[4] = {0.f, 0.f, 0.f, 1.f};
    y_update(ic<12>{}, ic<16>{}, cf);   // row 5 of the positions: At[.][5] = (0, 0, 0, 1)
  }
  // Per pixel row k of the tiles, the block's gamma / beta values go through LDS into pixel-major order G[px][32 ch],
  // B[px][32 ch] (px = 4 * tile + j; 16-byte chunks XOR-swizzled by the tile), then every thread handles (pixel, channel
  // quad) items with 128-byte-line global accesses.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float* const Gs = reinterpret_cast<float*>(smem);
  float* const Bs = reinterpret_cast<float*>(smem + 32768);
  const int tl_w = 16 * wt + fi;                          // this lane's tile within the block
  const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + cq), is = *reinterpret_cast<const f32x4*>(a.invstd + cq);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 bg = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + rg * 64 + chunk_r * 4) : z4;
  const f32x4 bb = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + rg * 64 + 32 + chunk_r * 4) : z4;
  const int chw = ((wq * 4 + oc) ^ (tl_w & 7)) * 4;       // write phase: this lane's channel quad 16 wq + 4 oc, swizzled
  float hmax = 0.f, xmax = 0.f;
  static_for<(DSEE_FUSED_ABL & 16) ? 0 : 4>([&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    static_for<4>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      const int px = tl_w * 4 + j;
      f32x4 gv, bv;
      static_for<4>([&](auto e_c) {
        constexpr int e = decltype(e_c)::value;
        float yg, yb;
        float& rg_ = Y[k][j][e];
        float& rb_ = Y[k][j][4 + e];
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(yg) : "a"(rg_));
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(yb) : "a"(rb_));
        gv[e] = yg;
        bv[e] = yb;
      });
      *reinterpret_cast<f32x4*>(Gs + px * 32 + chw) = gv;
      *reinterpret_cast<f32x4*>(Bs + px * 32 + chw) = bv;
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int px = it * 64 + (tid >> 3);
      const int tl = px >> 2;
      // (opaque to the optimiser: otherwise the 16 offsets are formed before the main loop and carried through it -- 45 more
      // spilled registers, the kernel 8 % slower)
      unsigned off = boff[it];
      asm volatile("" : "+v"(off));
      off += k * rowbytes;
      const int ch = (chunk_r ^ (tl & 7)) * 4;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + px * 32 + ch);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + px * 32 + ch);
      const f32x4 xh = (xr[k][it] - mu) * is;
      const f32x4 sc = gv * oscale + bg + a.add_one;
      if constexpr (WSCALE && PK) {      // 16-bit storage mode: the saved modulation factor is fp16 as well (8 bytes per lane)
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 sh = {(_Float16)sc[0], (_Float16)sc[1], (_Float16)sc[2], (_Float16)sc[3]};
        __builtin_nontemporal_store(sh, reinterpret_cast<f16x4*>(sb + (off >> 1)));
      } else if constexpr (WSCALE) {
        __builtin_nontemporal_store(sc, reinterpret_cast<f32x4*>(sb + off));
      }
      f32x4 v = (xh * sc + bb) + bv * oscale;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ob + off));
      if constexpr (WSCALE && !(DSEE_FUSED_ABL & 128)) {
        // The LeakyReLU branch as bits for the backward pass (training only, like `scale`).  v_cmp leaves the 64 lanes' answers
        // in an SGPR pair, so no lane exchange is needed: lane l = 8 * pixel + channel quad, byte `pixel` of ballot e holds
        // element e of the pixel's 8 quads.  Word of (pixel, 32-channel group): bit 8 * (c & 3) + ((c & 31) >> 2).
        // (Computed unconditionally -- a branch on a.mask here costs 45 spilled registers -- and stored by one lane of 8.)
        const int pl = (tid >> 3) & 7;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned long long b = __builtin_amdgcn_ballot_w64(v[e] > 0.f);
          const unsigned half = (pl & 4) ? (unsigned)(b >> 32) : (unsigned)b;
          m |= __builtin_amdgcn_ubfe(half, 8 * (pl & 3), 8) << (8 * e);
        }
        // ([C/32][N*H*W] words: the 8 pixels of a wave are 32 contiguous bytes, the 64 of the block's pass 256)
        if (chunk_r == 0 && a.mask) a.mask[rg * np + (off >> a.cshift)] = m;      // (off >> cshift = the pixel: C = 2^(cshift - 2))
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
  if (a.amax_h) {   // one read-before-atomic max per wave (the block's LDS is all dynamic: no static reduction buffer)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hmax = fmaxf(hmax, __shfl_xor(hmax, o, 64));
    if (lane == 0) {
      unsigned* line = reinterpret_cast<unsigned*>(a.amax_h + ((blockIdx.x * 8 + wave) & (DSEE_AMAX_LINES - 1)) * DSEE_AMAX_STRIDE);
      const unsigned bits = __builtin_bit_cast(unsigned, hmax);
      if (bits > __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(line, bits);
    }
  }
  if (a.amax_xhat) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
    if (lane == 0) {
      unsigned* line = reinterpret_cast<unsigned*>(a.amax_xhat + ((blockIdx.x * 8 + wave) & (DSEE_AMAX_LINES - 1)) * DSEE_AMAX_STRIDE);
      const unsigned bits = __builtin_bit_cast(unsigned, xmax);
      if (bits > __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(line, bits);
    }
  }
#if DSEE_FUSED_ABL & 32
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && blockIdx.x < 64) {   // per wave: stage waits | MFMA parts | load parts | epilogue
    const unsigned long long tend_ = __builtin_readcyclecounter();
    float* o = a.stamps + (blockIdx.x * 8 + wave) * 8;
    o[0] = (float)tw_; o[1] = (float)tm_; o[2] = (float)tl_; o[3] = (float)(tend_ - tmain_);
    o[4] = (float)tq0_; o[5] = (float)tq1_; o[6] = (float)tq2_; o[7] = (float)tv_;
  }
#endif
  };
  if constexpr (DSEE_FUSED_SCHED & 4) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // (the younger half of the workgroup loses VALU arbitration by age otherwise)
  }
  if (wave >= 4) body(ic<1>{}); else body(ic<0>{});
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// The 16-wave form (round 5).  Cycle stamps of the 8-wave kernel above (DSEE_FUSED_ABL & 32, tools/exp/fused_phases.py): a wave
// spends about a quarter of a block each in fragment reads (34.8 k cycles: exactly the time the LDS array needs for ALL eight
// waves' reads, 8.6 MB at 256 B/clk), LDS-DMA requests (31.1 k: the texture addresser takes 64 B/clk), fold / Y update (32.0 k)
// and MFMAs (33.5 k), and 3 k waiting for its own requests to land: every phase saturates a pipe the waves share while it
// runs, and with two in-order waves per SIMD at most two phases overlap -- the block takes ~140 k cycles against ~46 k of
// the busiest pipe.  Here the same 64 tile x 64 row block is owned by SIXTEEN waves of 128 registers, four per SIMD: a wave
// owns ONE 16 x 16 block (16 gamma rows or the 16 beta rows of the same channels, 16 tiles): Y = 16 outputs x 4 = 64 AGPRs,
// T = 16 registers, one MFMA accumulator, one set of four fragments; it issues ONE LDS-DMA instruction per piece (waves 0-7 the
// U rows, 8-15 the V rows) instead of two.  Fragment reads per piece go from 48 to 64 KB per CU (a fragment is shared by fewer
// blocks of the same wave); ring, swizzle, stage structure, fold, epilogue arithmetic and output bits are unchanged.
template <int NP, bool WSCALE, bool PK = false, int RD = 2>
__global__ __launch_bounds__(1024) void spade_fused_fwd16_kernel(FusedArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int PIECE = 8192;
  constexpr int UREG = RD * NP * PIECE;
  static_assert(RD == 2 || RD == 3, "a row of 6 positions is a whole number of ring turns");
  static_assert(2 * UREG <= 160 * 1024 && 2 * UREG >= 65536, "ring fits the LDS and covers the epilogue's 64 KB");
  static_assert(PK ? (NP == 2 || NP == 3) : (NP == 4 || NP == 5), "K = 128 or 160");
  constexpr bool HALF_LAST = PK && NP == 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wb = wave >> 2, wt = wave & 3;     // row block (0, 1: gamma of channels 0-15 / 16-31; 2, 3: beta), tile quarter
  const int wq = wb & 1;
  const bool is_beta = wb >= 2;

  const int rgn = a.rows >> 6;
  const long tgn = a.T >> 6, ntile = tgn * rgn;
  long l;
  {
    const long v = blockIdx.x, q = ntile >> 3, r = ntile & 7, xcd = v & 7, idx = v >> 3;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  long tg;
  int rg;
  if ((rgn % DSEE_FUSED_SET_B) == 0 && (tgn % DSEE_FUSED_SET_A) == 0) {
    const long sup = l >> 5;
    const int in = (int)(l & 31), rh = rgn / DSEE_FUSED_SET_B;
    tg = (sup / rh) * DSEE_FUSED_SET_A + (in / DSEE_FUSED_SET_B);
    rg = (int)(sup % rh) * DSEE_FUSED_SET_B + (in % DSEE_FUSED_SET_B);
  } else {
    tg = l / rgn;
    rg = (int)(l % rgn);
  }
  const long t0 = tg * 64;
  const int n = (int)(t0 / a.tpi);
  const int g = a.G > 1 ? n : 0;

  const float sv = dsee_pow2_scale(a.v_bound * dsee_amax_read(a.amax_v));
  const float su = dsee_pow2_scale(dsee_amax_read(a.amax_u));
  const float oscale = 1.f / (sv * su);

  // ---- LDS-DMA: waves 0-7 fill the U pieces, 8-15 the V pieces, rows 8 (w % 8) .. + 7 of every piece (chunk assignment as above)
  const bool dma_v = wave >= 8;
  const int dw = wave & 7;
  const int dr = 8 * dw + (lane >> 3);
  const int dcc = (lane & 7) ^ ((dr >> 1) & 7);
  const int dslab = dcc >> 2;
  const unsigned dlo = (unsigned)(dr * 64 + (dcc & 3) * 16);
  const unsigned slab_bytes = (unsigned)(dma_v ? a.v_slab_bytes : a.u_slab_bytes);
  const unsigned voff = dlo + (unsigned)dslab * slab_bytes;
  const unsigned voff_l = dslab ? 0xFFFFFFF0u : voff;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)uniform_ptr(dma_v ? a.V2 : a.U2), 0, (int)(dma_v ? a.v_bytes : a.u_bytes), 0x00020000);
  const unsigned PP_ = dma_v ? (unsigned)(a.T * 64) : (unsigned)(a.G * a.u_group_bytes);   // per position
  const unsigned QQ_ = 2 * slab_bytes;                                                      // per piece
  const unsigned base_o = dma_v ? (unsigned)(t0 * 64) : (unsigned)((long)g * a.u_group_bytes + (long)rg * 4096);
  unsigned char* const dbase = smem + (dma_v ? UREG : 0) + dw * 1024;
  auto dma = [&](int par, int pc, unsigned opos) {
    unsigned char* dst = dbase + (par * NP + pc) * PIECE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16,
                                             (HALF_LAST && pc == NP - 1) ? voff_l : voff, opos + pc * QQ_, 0, 0);
  };

  // ---- fragment addresses
  const int fi = lane & 15, oc = lane >> 4;
  auto foff = [&](int r, int t) {
    const int chunk = PK ? 4 * t + oc : 4 * (oc >> 1) + 2 * t + (oc & 1);
    return (unsigned)((8 * r + (chunk ^ ((r >> 1) & 7))) * 16);
  };
  const int rga = 16 * wb + fi, rtv = 16 * wt + fi;
  const unsigned au0 = foff(rga, 0), au1 = foff(rga, 1);
  const unsigned av0 = UREG + foff(rtv, 0), av1 = UREG + foff(rtv, 1);
  struct Frag {
    u32x4 a0, a1, v0, v1;
  };
  auto ldf = [&](Frag& f, auto par_c, auto pc_c) {
    constexpr int so = (decltype(par_c)::value * NP + decltype(pc_c)::value) * PIECE;
    constexpr bool hi = so >= 7 * PIECE;
    constexpr int io = hi ? so - 7 * PIECE : so;
    static_assert(io >= 0 && io + PIECE <= 65536, "window");
    // (the second window's base is formed where it is used: one VALU add instead of four more live registers)
    const unsigned w0 = hi ? 7 * PIECE : 0;
    f.a0 = *reinterpret_cast<const u32x4*>(smem + (au0 + w0) + io);
    f.a1 = *reinterpret_cast<const u32x4*>(smem + (au1 + w0) + io);
    f.v0 = *reinterpret_cast<const u32x4*>(smem + (av0 + w0) + io);
    f.v1 = *reinterpret_cast<const u32x4*>(smem + (av1 + w0) + io);
  };
  auto mm = [&](const Frag& f, f32x4& p) {
    const f16x8 a0 = __builtin_bit_cast(f16x8, f.a0), a1 = __builtin_bit_cast(f16x8, f.a1);
    const f16x8 v0 = __builtin_bit_cast(f16x8, f.v0), v1 = __builtin_bit_cast(f16x8, f.v1);
    if constexpr (PK) {
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, v0, p, 0, 0, 0);
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, v1, p, 0, 0, 0);
    } else {
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, v0, p, 0, 0, 0);
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, v1, p, 0, 0, 0);
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, v0, p, 0, 0, 0);
    }
  };

  float Y[4][4][4];      // [output row i][output column j][4 channels]: 64 AGPRs
  f32x4 T[4], P;
  Frag F;
  static_for<4>([&](auto i) {
    static_for<4>([&](auto j) {
      static_for<4>([&](auto e) {
        float& yr = Y[decltype(i)::value][decltype(j)::value][decltype(e)::value];
        asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(yr));
      });
    });
  });
#pragma unroll
  for (int j = 0; j < 4; ++j) T[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int NA = NP == 2 ? 1 : 2;
  auto issue_stage = [&](int pos, int half) {
    for (int k = half ? NA : 0; k < (half ? NP : NA); ++k) dma(pos % RD, k, base_o + pos * PP_);
  };
  for (int st = 0; st < 2 * RD - 1; ++st) issue_stage(st >> 1, st & 1);
  // (stage 0 is waited for by the first position's own vmcnt + barrier)

  auto y_update = [&](auto lo_c, auto hi_c, const float (&cf)[4]) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    static_for<HI - LO>([&](auto d) {
      constexpr int ij = LO + decltype(d)::value, i = ij >> 2, j = ij & 3;
      static_for<4>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        float y;
        float& yr = Y[i][j][e];
        const float cc = cf[i], tt = T[j][e];
        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_fmac_f32 %1, %2, %3\n\tv_accvgpr_write_b32 %0, %1"
                     : "+a"(yr), "=&v"(y)
                     : "s"(cc), "v"(tt));
      });
    });
  };

  auto position = [&](int pos, auto par_c, auto pp_c, auto&& fill) {
    constexpr int PAR = decltype(par_c)::value;
    P = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<2>([&](auto half_c) {
      constexpr int half = decltype(half_c)::value;
      const int pl = min(pos + RD - 1 + half, 35);
      const unsigned oo = base_o + pl * PP_;
      constexpr int LPAR = half ? PAR : (PAR + RD - 1) % RD;
      constexpr int L0 = half ? 0 : NA, L1 = half ? NA : NP;
      constexpr int C0 = half ? NA : 0, C1 = half ? NP : NA;
      // stage s itself has landed (this wave's rows) when only the requests of the 2 RD - 2 younger stages -- RD - 1 whole
      // positions -- are still in flight: its fragments are read after the barrier, none of the next stage's before the next one
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 1) * NP) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      static_for<C1 - C0>([&](auto p_c) {
        constexpr int p = decltype(p_c)::value, k = C0 + p;
        __builtin_amdgcn_sched_barrier(0);
        // this piece's fragments (one set: with four waves per SIMD another wave's MFMAs cover their LDS latency), this piece's
        // share of the look-ahead requests and of the fold work while they travel, then the MFMAs
        ldf(F, ic<PAR>{}, ic<k>{});
        __builtin_amdgcn_sched_barrier(0);
        constexpr int a0 = L0 + p * (L1 - L0) / (C1 - C0), a1 = L0 + (p + 1) * (L1 - L0) / (C1 - C0);
        static_for<a1 - a0>([&](auto d) { dma(LPAR, a0 + decltype(d)::value, oo); });
        fill(ic<k>{});
        __builtin_amdgcn_sched_barrier(0);
        mm(F, P);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };
  auto fold = [&](auto cp_c, const f32x4& m) {
    constexpr int CP = decltype(cp_c)::value;
    if constexpr (CP == 0) {
      T[0] += m;
    } else if constexpr (CP == 1) {
      T[0] += m; T[1] += m; T[2] += m; T[3] += m;
    } else if constexpr (CP == 2) {
      T[0] += m; T[1] -= m; T[2] += m; T[3] -= m;
    } else if constexpr (CP == 3) {
      T[0] += m; T[1] += 2.f * m; T[2] += 4.f * m; T[3] += 8.f * m;
    } else if constexpr (CP == 4) {
      T[0] += m; T[1] -= 2.f * m; T[2] += 4.f * m; T[3] -= 8.f * m;
    } else {
      T[3] += m;
    }
  };

#pragma unroll 1
  for (int r = 0; r < 6; ++r) {
    const int q = r - 1;
    const float cf[4] = {q < 0 ? 0.f : 1.f, q <= 0 ? 0.f : (q == 1 ? 1.f : (q == 2 ? -1.f : (q == 3 ? 2.f : -2.f))),
                         q <= 0 ? 0.f : (q < 3 ? 1.f : 4.f),
                         q <= 0 ? 0.f : (q == 1 ? 1.f : (q == 2 ? -1.f : (q == 3 ? 8.f : -8.f)))};
    position(r * 6, ic<0>{}, ic<0>{}, [&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      if constexpr (k >= 1) {
        constexpr int NY = NP - 1, qq = k - 1, LO = qq * 16 / NY, HI = (qq + 1) * 16 / NY;
        if constexpr (NP == 5) {
          if (q > 0 || (q == 0 && qq == 0)) y_update(ic<LO>{}, ic<HI>{}, cf);
        } else {
          if (q >= 0) y_update(ic<LO>{}, ic<HI>{}, cf);
        }
        if constexpr (k == NP - 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) T[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    });
    fold(ic<0>{}, P);
#define DSEE_POS16(C)                                                  \
  position(r * 6 + C, ic<C % RD>{}, ic<C % 2>{}, [&](auto) {});        \
  fold(ic<C>{}, P);
    DSEE_POS16(1)
    DSEE_POS16(2)
    DSEE_POS16(3)
    DSEE_POS16(4)
    DSEE_POS16(5)
#undef DSEE_POS16
  }

  // ---- epilogue (as above, 1024 threads: 2 items per thread and pixel row)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int chunk_r = tid & 7;
  const int cq = rg * 32 + chunk_r * 4;
  unsigned boff[2];
  f32x4 xr[4][2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int px = it * 128 + (tid >> 3);
    const int tl = px >> 2, j = px & 3;
    const int tin = (int)(t0 + tl - (long)n * a.tpi);
    const int ty = tin / a.tw, tx = tin - ty * a.tw;
    boff[it] = ((unsigned)((n * a.H + ty * 4) * a.W + tx * 4 + j) * (unsigned)a.C + (unsigned)cq) * 4u;
  }
  const unsigned np = (unsigned)(a.T * 16);
  const unsigned rowbytes = (unsigned)(a.W * a.C) * 4u;
  const char* const xb = reinterpret_cast<const char*>(a.x);
  char* const ob = reinterpret_cast<char*>(a.out);
  char* const sb = reinterpret_cast<char*>(a.scale);
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int it = 0; it < 2; ++it)
      xr[k][it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xb + (boff[it] + k * rowbytes)));
  __builtin_amdgcn_sched_barrier(0);
  {
    const float cf[4] = {0.f, 0.f, 0.f, 1.f};
    y_update(ic<12>{}, ic<16>{}, cf);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float* const Gs = reinterpret_cast<float*>(smem);
  float* const Bs = reinterpret_cast<float*>(smem + 32768);
  float* const Ws = is_beta ? Bs : Gs;                    // this wave's half of the exchange
  const int tl_w = 16 * wt + fi;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + cq), is = *reinterpret_cast<const f32x4*>(a.invstd + cq);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 bg = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + rg * 64 + chunk_r * 4) : z4;
  const f32x4 bb = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + rg * 64 + 32 + chunk_r * 4) : z4;
  const int chw = ((wq * 4 + oc) ^ (tl_w & 7)) * 4;
  float hmax = 0.f, xmax = 0.f;
  static_for<4>([&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    static_for<4>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      const int px = tl_w * 4 + j;
      f32x4 yv;
      static_for<4>([&](auto e_c) {
        constexpr int e = decltype(e_c)::value;
        float y;
        float& r_ = Y[k][j][e];
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(r_));
        yv[e] = y;
      });
      *reinterpret_cast<f32x4*>(Ws + px * 32 + chw) = yv;
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int px = it * 128 + (tid >> 3);
      const int tl = px >> 2;
      unsigned off = boff[it];
      asm volatile("" : "+v"(off));
      off += k * rowbytes;
      const int ch = (chunk_r ^ (tl & 7)) * 4;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + px * 32 + ch);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + px * 32 + ch);
      const f32x4 xh = (xr[k][it] - mu) * is;
      const f32x4 sc = gv * oscale + bg + a.add_one;
      if constexpr (WSCALE && PK) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 sh = {(_Float16)sc[0], (_Float16)sc[1], (_Float16)sc[2], (_Float16)sc[3]};
        __builtin_nontemporal_store(sh, reinterpret_cast<f16x4*>(sb + (off >> 1)));
      } else if constexpr (WSCALE) {
        __builtin_nontemporal_store(sc, reinterpret_cast<f32x4*>(sb + off));
      }
      f32x4 v = (xh * sc + bb) + bv * oscale;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ob + off));
      if constexpr (WSCALE) {
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
  if (a.amax_h) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) hmax = fmaxf(hmax, __shfl_xor(hmax, o, 64));
    if (lane == 0) {
      unsigned* line = reinterpret_cast<unsigned*>(a.amax_h + ((blockIdx.x * 16 + wave) & (DSEE_AMAX_LINES - 1)) * DSEE_AMAX_STRIDE);
      const unsigned bits = __builtin_bit_cast(unsigned, hmax);
      if (bits > __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(line, bits);
    }
  }
  if (a.amax_xhat) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
    if (lane == 0) {
      unsigned* line = reinterpret_cast<unsigned*>(a.amax_xhat + ((blockIdx.x * 16 + wave) & (DSEE_AMAX_LINES - 1)) * DSEE_AMAX_STRIDE);
      const unsigned bits = __builtin_bit_cast(unsigned, xmax);
      if (bits > __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(line, bits);
    }
  }
#endif
}

}  // namespace

static float* g_fused_stamps = nullptr;

#if DSEE_FUSED_ABL
/* measurement builds only (tools/exp/build_fused_abl.sh; the shipped library does not export it): device buffer for the cycle
 * stamps of DSEE_FUSED_ABL & 32 */
extern "C" void dsee_fused_set_stamps(float* p) { g_fused_stamps = p; }
#endif

/* The fused SPADE / SEAN normalisation forward (see the head of this file).  V2 = dsee_wino43_input_f16x2(cat, amax_cat,
 * v_bound), U2 = dsee_wino43_weights[_table](..., split = 2, amax_u); groups = images with per-image tables, else 1. */

static int spade_fused_launch(bool packed, const void* V2, const void* U2, const float* amax_cat, float v_bound,
                              const float* amax_u, const float* bias_packed, const float* x, const float* mean,
                              const float* invstd, float* out_h, float* out_scale, int N, int H, int W, int C, int rows, int K,
                              int groups, float add_one, float slope, float* amax_h, float* amax_xhat, unsigned* sign_mask,
                              hipStream_t st) {
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
  const int kslab = packed ? 32 : 16;          // k's per 64-byte-row slab of the operand images
  a.u_group_bytes = (long)(K / kslab) * rows * 64;
  const long vb = a.v_slab_bytes * (K / kslab), ub = a.u_group_bytes * 36 * groups;
  DSEE_CHECK_ARG(vb < 0xFFFFFFF0L && ub < 0xFFFFFFF0L);   // operand tensors are addressed through one buffer resource each
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
  a.stamps = g_fused_stamps;
  const long ntile = (T / 64) * (rows / 64);
  DSEE_CHECK_ARG(ntile < 0x7FFFFFFF);
  const int np = packed ? (K + 63) / 64 : K / 32;
  const int rd = packed ? 3 : 2;               // ring depth in positions (see the kernel): the packed operands are half the bytes
  const size_t lds = (size_t)rd * np * 16384;  // ring of rd positions: rd * NP pieces of U and of V, 8 KB each
  // DSEE_FUSED_W16 = 1 / 0 selects the 16-wave / 8-wave form (default: DSEE_FUSED_W16_DEFAULT)
  const char* const w16_env = getenv("DSEE_FUSED_W16");      // (read per launch: a test can flip it inside one process)
  const int w16 = w16_env ? atoi(w16_env) : DSEE_FUSED_W16_DEFAULT;
#define DSEE_FUSED_K(KERNEL, THREADS, NP, WS, PKD)                                                                   \
  do {                                                                                                               \
    static bool attr_done = false;                                                                                   \
    if (!attr_done) {                                                                                                \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL<NP, WS, PKD, (PKD ? 3 : 2)>),         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e != hipSuccess) {                                                                                         \
        dsee_set_error("hipFuncSetAttribute(%zu bytes of LDS): %s", lds, hipGetErrorString(e));                     \
        return DSEE_ELAUNCH;                                                                                         \
      }                                                                                                              \
      attr_done = true;                                                                                              \
    }                                                                                                                \
    KERNEL<NP, WS, PKD, (PKD ? 3 : 2)><<<(int)ntile, THREADS, lds, st>>>(a);                                         \
  } while (0)
#define DSEE_FUSED(NP, WS, PKD)                                                                                      \
  do {                                                                                                               \
    if (w16)                                                                                                         \
      DSEE_FUSED_K(spade_fused_fwd16_kernel, 1024, NP, WS, PKD);                                                     \
    else                                                                                                             \
      DSEE_FUSED_K(spade_fused_fwd_kernel, 512, NP, WS, PKD);                                                        \
  } while (0)
  if (packed) {
    if (np == 3) {
      if (out_scale) DSEE_FUSED(3, true, true); else DSEE_FUSED(3, false, true);
    } else {
      if (out_scale) DSEE_FUSED(2, true, true); else DSEE_FUSED(2, false, true);
    }
  } else if (np == 5) {
    if (out_scale) DSEE_FUSED(5, true, false); else DSEE_FUSED(5, false, false);
  } else {
    if (out_scale) DSEE_FUSED(4, true, false); else DSEE_FUSED(4, false, false);
  }
#undef DSEE_FUSED
#undef DSEE_FUSED_K
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

extern "C" {

int dsee_spade_fused_fwd(const void* V2, const void* U2, const float* amax_cat, float v_bound, const float* amax_u,
                         const float* bias_packed, const float* x, const float* mean, const float* invstd, float* out_h,
                         float* out_scale, int N, int H, int W, int C, int rows, int K, int groups, float add_one,
                         float slope, float* amax_h, float* amax_xhat, uint32_t* sign_mask, hipStream_t st) {
  return spade_fused_launch(false, V2, U2, amax_cat, v_bound, amax_u, bias_packed, x, mean, invstd, out_h, out_scale, N, H, W,
                            C, rows, K, groups, add_one, slope, amax_h, amax_xhat, sign_mask, st);
}

/* 16-bit storage mode: the same kernel on PACKED ONE-TERM operands -- V1 = dsee_wino43_input_f16p(cat) [K/32][36*T][32] fp16,
 * U1 = dsee_wino43_weights[_table](split = 4) [36*groups][K/32][rows][32] -- one MFMA product per multiply-add. */
int dsee_spade_fused_fwd_f16p(const void* V1, const void* U1, const float* amax_cat, float v_bound, const float* amax_u,
                              const float* bias_packed, const float* x, const float* mean, const float* invstd, float* out_h,
                              float* out_scale, int N, int H, int W, int C, int rows, int K, int groups, float add_one,
                              float slope, float* amax_h, float* amax_xhat, uint32_t* sign_mask, hipStream_t st) {
  return spade_fused_launch(true, V1, U1, amax_cat, v_bound, amax_u, bias_packed, x, mean, invstd, out_h, out_scale, N, H, W,
                            C, rows, K, groups, add_one, slope, amax_h, amax_xhat, sign_mask, st);
}

}  // extern "C"
