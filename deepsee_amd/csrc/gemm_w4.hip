// Round 6: the Winograd-domain NT GEMM on pre-split operands with ONE wave per SIMD.
//
// gemm_bf16x3.hip's 8-wave kernel gives each wave a 128 x 64 tile and lets the two waves of a SIMD alternate between a load
// and an MFMA interval (ping-pong).  Its slab of 16 k's costs 12 KB of LDS fragment reads per 24 MFMAs and two workgroup
// barriers, and measures at 0.44 of the three-product peak: both intervals run ~1 050 cycles against 768 for the MFMAs
// (profiles/r02_gemm_ablation.md, r06_gemm_ablation.md).  This kernel removes work per flop instead of re-ordering it:
//   * 4 waves, each a 128 x 128 tile of the same 256 x 256 block: 16 KB of fragments per 48 MFMAs (8 instead of 12 KB per 24),
//     256 accumulator registers per lane (the whole AGPR half of the unified file: one wave per SIMD),
//   * ONE barrier per slab; the fragments of slab k+1 are read into a second register set and the LDS-DMA requests of slab
//     k+5 are issued BETWEEN the MFMAs of slab k (sched_group_barrier), so the matrix pipe is fed by a single in-order
//     stream with its memory work in the issue shadow,
//   * a ring of FIVE 32 KB stages = the whole 160 KB of LDS: the 64-byte rows are XOR-swizzled instead of padded (16-byte chunk c
//     of row r sits at chunk c ^ ((r >> 2) & 3): each 16-lane group of a ds_read_b128 covers all 64 banks once), requests run
//     four slabs ahead.
// Operands: the 64-byte-row images of dsee_gemm_f16x2_pre / dsee_gemm_f16p_pre ([K/16][rows][2 terms][16] fp16 two-term,
// [K/32][rows][32] fp16 packed one-term), same scales, same products (a1 b0 + a0 b1 + a0 b0 | a0 b0 + a1 b1), same output.
#include "dsee_common.h"

// measurement builds only (tools/exp/build_w4_abl.sh): 1 one MFMA product per slab and tile instead of NP, 2 no fragment reads, 8 no LDS-DMA, 16 no C stores
#ifndef DSEE_W4_ABL
#define DSEE_W4_ABL 0
#endif
// interleave pattern of the memory work between the MFMAs (0: none, the compiler's own order; 1: DMA first, then reads;
// 2: reads first, then DMA)
// start-up stagger of the blocks of an XCD, in 64ths of a tile (0: none)
#ifndef DSEE_W4_STAGGER
#define DSEE_W4_STAGGER 0
#endif
#ifndef DSEE_W4_SCHED
#define DSEE_W4_SCHED 1
#endif

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct W4Args {
  const unsigned char* A;
  const unsigned char* B;
  void* C;
  const float* amax_a;
  const float* amax_b;
  float* cscale;
  unsigned tiles_m, tiles_n;   // 256-row / 256-column tiles
  int nk;                      // 64-byte-row slabs per tile (even)
  int ldc;
  unsigned rows_per_group;     // rows m / rows_per_group select the B matrix
  long a_slab_bytes, b_slab_bytes, b_group_bytes;
  float a_bound;
  int k_real;                  // reduction length in elements (for the fp16 output's bound)
};

__device__ __forceinline__ const unsigned char* w4_uniform_ptr(const unsigned char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const unsigned char*)(((unsigned long long)hi << 32) | lo);
}

template <int V>
struct IC {
  static constexpr int value = V;
};

template <bool PK, bool C16>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(W4Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NST = 5, OPB = 16384, STAGE = 2 * OPB, NP = PK ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned nbn = a.tiles_n, ntile = a.tiles_m * nbn, G = gridDim.x;
  // XCD-aware tile order (block b runs on XCD b % 8): each XCD walks one contiguous eighth of the list, so the 32 blocks that
  // share an L2 work on neighbouring row tiles of the same B matrix
  auto decode = [&](unsigned v, unsigned& bm, unsigned& bn) __attribute__((always_inline)) {
    const unsigned q = ntile >> 3, r = ntile & 7, xcd = v & 7, idx = v >> 3;
    const unsigned l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bm = l / nbn;
    bn = l - bm * nbn;
  };

  const float ama = dsee_amax_read(a.amax_a), amb = dsee_amax_read(a.amax_b);
  float oscale = 1.f / (dsee_pow2_scale(a.a_bound * ama) * dsee_pow2_scale(amb));
  if constexpr (C16) {
    const float sm = 2.f * dsee_pow2_scale((float)a.k_real * a.a_bound * ama * amb);
    oscale *= sm;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.cscale = 1.f / sm;
  }

  // LDS-DMA: instruction j of this wave fills rows 16 (wave + 4 j) .. + 15 of an operand's stage; lane -> row l >> 2, LDS chunk
  // position l & 3, which holds global chunk (l & 3) ^ ((row >> 2) & 3) -- a quad of lanes still covers one whole 64-byte row
  unsigned voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    voff[j] = (unsigned)((16 * (wave + 4 * j) + (lane >> 2)) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
  // fragments: lane -> row (lane & 31) of a 32-row MFMA tile, k-half lane >> 5; term t is chunk 2 t + (lane >> 5), i.e. the two
  // terms of a row sit 32 bytes apart whatever the swizzle
  const unsigned swz = (unsigned)((lane >> 2) & 3), kh = (unsigned)(lane >> 5);
  const unsigned fa0 = (unsigned)((wm * 128 + (lane & 31)) * 64) + ((kh ^ swz) << 4);
  const unsigned fb0 = (unsigned)(OPB + (wn * 128 + (lane & 31)) * 64) + ((kh ^ swz) << 4);

  // ---- the slab stream of this block (runs ahead of the compute stream, across tile boundaries)
  unsigned lt = blockIdx.x;
  int lk = 0, valid = 0;
  const unsigned char *pa = a.A, *pb = a.B;
  auto set_base = [&]() __attribute__((always_inline)) {
    const bool live = lt < ntile;
    unsigned bm, bn;
    decode(live ? lt : (unsigned)blockIdx.x, bm, bn);
    const unsigned group = (bm * 256u) / a.rows_per_group;
    pa = w4_uniform_ptr(a.A + (long)bm * (256 * 64));
    pb = w4_uniform_ptr(a.B + (long)group * a.b_group_bytes + (long)bn * (256 * 64));
    valid = __builtin_amdgcn_readfirstlane(live ? 256 * 64 : 0);   // past the end of the list: zeros land in LDS
  };
  auto issue = [&](int stage_off) __attribute__((always_inline)) {
    if constexpr (!(DSEE_W4_ABL & 8)) {
      __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(pa + lk * a.a_slab_bytes), 0, valid, 0x00020000);
      __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(pb + lk * a.b_slab_bytes), 0, valid, 0x00020000);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        auto* dst = (__attribute__((address_space(3))) void*)(smem + stage_off + (wave + 4 * j) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, voff[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        auto* dst = (__attribute__((address_space(3))) void*)(smem + stage_off + OPB + (wave + 4 * j) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, voff[j], 0, 0, 0);
      }
    }
  };
  auto advance_load = [&]() __attribute__((always_inline)) {
    if (++lk == a.nk) {
      lk = 0;
      lt += G;
      set_base();
    }
  };

  f32x16 acc[4][4];   // defined by the first product of a tile's first slab (MFMA with C = 0), dead after the tile's stores
  u32x4 fA[2][4][2], fB[2][4][2];   // [register set][32-row tile][term / k-half]

  auto read_frags = [&](auto setc, int stage_off) __attribute__((always_inline)) {
    constexpr int S = decltype(setc)::value;
    const unsigned char* sa_ = smem + stage_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if constexpr (DSEE_W4_ABL & 2) {
          fA[S][i][p] = (u32x4){(unsigned)lk, 1u, 2u, (unsigned)lane};
          fB[S][i][p] = (u32x4){(unsigned)lk, 3u, 4u, (unsigned)lane};
        } else {
          fA[S][i][p] = *reinterpret_cast<const u32x4*>(sa_ + ((fa0 ^ (p * 32)) + i * 2048));
          fB[S][i][p] = *reinterpret_cast<const u32x4*>(sa_ + ((fb0 ^ (p * 32)) + i * 2048));
        }
      }
  };

  unsigned ct = blockIdx.x;
  auto store_tile = [&]() __attribute__((always_inline)) {
    unsigned bm, bn;
    decode(ct, bm, bn);
    if constexpr (C16) {
      _Float16* cz = reinterpret_cast<_Float16*>(a.C);
      const bool odd = (lane & 1) != 0;
      int ldc_ = a.ldc;
      asm volatile("" : "+s"(ldc_));   // (opaque: keeps the address arithmetic of the 128 stores inside this block)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long mb = (long)bm * 256 + wm * 128 + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = bn * 256 + wn * 128 + j * 32 + (lane & 31);
          // two columns per store: lanes (n, n + 1) exchange one value per row pair through DPP
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            float y0, y1;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y0) : "a"(acc[i][j][r]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y1) : "a"(acc[i][j][r + 1]));
            const unsigned h0 = __builtin_bit_cast(unsigned short, (_Float16)(y0 * oscale));
            const unsigned h1 = __builtin_bit_cast(unsigned short, (_Float16)(y1 * oscale));
            const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd ? h0 : h1), 0xB1, 0xF, 0xF, true);
            const unsigned w = odd ? (got | (h1 << 16)) : (h0 | (got << 16));
            const long row = mb + ((r + (odd ? 1 : 0)) & 3) + 8 * (r >> 2);
            if constexpr (!(DSEE_W4_ABL & 16)) *reinterpret_cast<unsigned*>(cz + row * ldc_ + (n & ~1)) = w;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // D[m][n]: lane = column, registers = rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  One buffer resource per wave tile, the
      // lane's part of the address in one VGPR, the (tile, row) part in an SGPR: a store costs one SALU op and no VALU
      float* cw = reinterpret_cast<float*>(a.C) + ((long)bm * 256 + wm * 128) * a.ldc + bn * 256 + wn * 128;
      __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)w4_uniform_ptr((const unsigned char*)cw), 0, 0x7FFFFFFF, 0x00020000);
      // (the row stride is made opaque here: otherwise the 256 scalar offsets are loop invariants, get hoisted out of the slab
      // loop and spill)
      int ldc4 = a.ldc * 4;
      asm volatile("" : "+s"(ldc4));
      const unsigned lane_off = (unsigned)((4 * (lane >> 5)) * ldc4 + (lane & 31) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int soff = (i * 32 + (r & 3) + 8 * (r >> 2)) * ldc4 + j * 128;
            float y;   // (explicit read: left to itself the compiler copies whole accumulator tiles into VGPRs inside the slab loop)
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(acc[i][j][r]));
            if constexpr (!(DSEE_W4_ABL & 16))
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y * oscale), rc, lane_off, soff, 0);
          }
          __builtin_amdgcn_sched_barrier(0);   // one accumulator tile at a time: the next slab's fragments stay in registers
        }
    }
  };

#if DSEE_W4_STAGGER
  // Every block walks tiles of the same length, so without this all 256 CUs reach their tile stores in the same microsecond and
  // the 64 MB burst queues behind the fabric (10 k cycles per tile against 4 k at a CU's own store rate).  Blocks of an XCD start
  // up to DSEE_W4_STAGGER/64 of a tile apart (block b runs on XCD b % 8; 32 phases).
  {
    const int phase = (blockIdx.x >> 3) & 31;
    const int units = phase * a.nk * DSEE_W4_STAGGER / 64;     // units of ~1 k cycles (a slab is ~2.1 k cycles)
    for (int u = 0; u < units; ++u) __builtin_amdgcn_s_sleep(16);
  }
#endif
  // ---- prologue: slabs 0 .. 4 requested, slab 0 landed and published, its fragments requested
  set_base();
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    issue(s * STAGE);
    advance_load();
  }
  asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(IC<0>{}, 0);
  int sfill = 0, sread = STAGE;   // stage freed by the slab being computed / stage of the slab after it

#if DSEE_W4_ABL & 32
  unsigned long long st_top = 0, st_vm = 0, st_bar = 0, st_body = 0, st_wvm = 0, st_wbar = 0, st_store = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_bar));
#endif
  // one slab: S = register set of slab k's fragments
  auto step = [&](auto setc, auto firstc) __attribute__((always_inline)) {
    constexpr int S = decltype(setc)::value;
    constexpr bool FIRST = decltype(firstc)::value != 0;   // first slab of a tile: its first product starts the accumulators
    // slab k+1 was requested four slabs ago: this wave's part has landed once only its requests of slabs k+2 .. k+4 remain;
    // the fragments of slab k have arrived; the barrier publishes slab k+1 and retires stage k % 5
#if DSEE_W4_ABL & 32   // cycle stamps: [top of slab -> vmcnt wait done -> barrier released] vs the slab body
    asm volatile("s_memtime %0" : "=s"(st_top));
    if constexpr (!(DSEE_W4_ABL & 8)) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    asm volatile("s_memtime %0" : "=s"(st_vm));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_top), "+s"(st_vm), "+s"(st_bar)::"memory");
    st_body += st_top - st_bar;     // (st_bar: stamp taken right after the previous slab's barrier)
    st_wvm += st_vm - st_top;
    __builtin_amdgcn_s_barrier();
    asm volatile("s_memtime %0" : "=s"(st_bar));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_bar)::"memory");
    st_wbar += st_bar - st_vm;
#else
    if constexpr (!(DSEE_W4_ABL & 8)) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    // Program order IS issue order here: the MFMAs are volatile asm statements with the accumulator pinned in AGPRs ("+a") -- the
    // compiler's own allocation of 256 accumulators + two fragment sets shuttled values between the register halves and spilled
    // -- and every pair of MFMAs is followed by ONE piece of memory work (8 LDS-DMA requests of slab k+5 into the stage slab k
    // occupied, then the 16 fragment reads of slab k+1 into the other register set), fenced by sched_barriers.
    const unsigned char* sr_ = smem + sread;
#if DSEE_W4_ABL & 128   // measurement build: every request reads the first slab of the first tile (cache-hot)
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, valid, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, valid, 0x00020000);
#else
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(pa + lk * a.a_slab_bytes), 0, valid, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(pb + lk * a.b_slab_bytes), 0, valid, 0x00020000);
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 16 * NP; ++m) {
      const int q = m >> 4, j = (m >> 2) & 3, i = m & 3;   // product-major: 16 independent accumulators between two uses of one
      int ia, ib;
      if constexpr (PK) {
        ia = ib = q;              // a0 b0 + a1 b1 (the two k-halves of a 64-byte row)
      } else {
        ia = q == 0 ? 1 : 0;      // a1 b0, a0 b1, a0 b0: smallest terms first
        ib = q == 1 ? 1 : 0;
      }
      if ((DSEE_W4_ABL & 1) && q > 0) {
        // measurement build: one product instead of NP
      } else if (FIRST && q == 0)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc[i][j]) : "v"(fA[S][i][ia]), "v"(fB[S][j][ib]));
      else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fA[S][i][ia]), "v"(fB[S][j][ib]));
      constexpr int PER = PK ? 1 : 2;   // MFMAs per piece of memory work
      if (m % PER == PER - 1 && m / PER < 24) {
        const int slot = m / PER;
        // which piece goes into slot 0 .. 23: t < 8 = LDS-DMA request t, t >= 8 = fragment read t - 8
#if DSEE_W4_SCHED == 2      // reads first
        const int t = slot < 16 ? slot + 8 : slot - 16;
#elif DSEE_W4_SCHED == 3    // two reads, one request, ...
        const int t = slot % 3 == 2 ? slot / 3 : 8 + (slot / 3) * 2 + slot % 3;
#else                       // requests first
        const int t = slot;
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (t < 8) {
          if (!(DSEE_W4_ABL & 8) && !((DSEE_W4_ABL & 64) && t >= 4)) {
            auto* dst = (__attribute__((address_space(3))) void*)(smem + sfill + (t >= 4 ? OPB : 0) + (wave + 4 * (t & 3)) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(t >= 4 ? rb : ra, dst, 16, voff[t & 3], 0, 0, 0);
          }
        } else {
          const int u = t - 8, ti = (u >> 1) & 3, p = u & 1;   // A tile 0 (2 terms), ... A tile 3, B tile 0 ...
          if constexpr (DSEE_W4_ABL & 2) {
            if (u < 8) fA[S ^ 1][ti][p] = (u32x4){(unsigned)lk, 1u, 2u, (unsigned)lane};
            else fB[S ^ 1][ti][p] = (u32x4){(unsigned)lk, 3u, 4u, (unsigned)lane};
          } else {
            if (u < 8) fA[S ^ 1][ti][p] = *reinterpret_cast<const u32x4*>(sr_ + ((fa0 ^ (p * 32)) + ti * 2048));
            else fB[S ^ 1][ti][p] = *reinterpret_cast<const u32x4*>(sr_ + ((fb0 ^ (p * 32)) + ti * 2048));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // nk is even and the request stream runs an odd number of slabs (five) ahead: a tile's last slab is REQUESTED in an even
    // step -- only that half of the unrolled pair carries the request stream's tile-boundary branch
    if constexpr (S == 0) advance_load();
    else ++lk;
    sfill = sfill == (NST - 1) * STAGE ? 0 : sfill + STAGE;
    sread = sread == (NST - 1) * STAGE ? 0 : sread + STAGE;
  };

  while (ct < ntile) {
    step(IC<0>{}, IC<1>{});
    step(IC<1>{}, IC<0>{});
    for (int kk = 2; kk < a.nk; kk += 2) {
      step(IC<0>{}, IC<0>{});
      step(IC<1>{}, IC<0>{});
    }
    // (the last MFMAs' results are read by v_accvgpr_read below: the compiler cannot see that hazard through the asm statements)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#if DSEE_W4_ABL & 32
    unsigned long long s0, s1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0));
    store_tile();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1));
    st_store += s1 - s0;
    st_bar += s1 - s0;   // (keeps the stores out of the next slab's body time)
#else
    store_tile();
#endif
    ct += G;
  }
#if DSEE_W4_ABL & 32
  if (lane == 0 && blockIdx.x < 8) {   // per-wave totals: slab bodies | vmcnt waits | barrier waits | tile stores
    float* o = reinterpret_cast<float*>(a.C) + (blockIdx.x * 4 + wave) * 8;
    o[0] = (float)st_body; o[1] = (float)st_wvm; o[2] = (float)st_wbar; o[3] = (float)st_store;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the block
#endif
}

int w4_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <bool PK, bool C16>
int launch_w4(const W4Args& a, hipStream_t st) {
  constexpr int LDS = 5 * 32768;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<PK, C16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const long ntile = (long)a.tiles_m * a.tiles_n, slots = w4_num_cus();
  gemm_w4_kernel<PK, C16><<<(unsigned)(ntile < slots ? ntile : slots), 256, LDS, st>>>(a);
  DSEE_LAUNCH_CHECK();
  return DSEE_OK;
}

}  // namespace

extern "C" {

/* dsee_gemm_f16x2_pre on the one-wave-per-SIMD kernel (see the top of this file): same operands, same result up to the order of
 * the fp32 accumulation (identical: slabs in order, products smallest first).  N % 256 == 0, (K / 16) even. */
int dsee_gemm_f16x2_pre_w4(const void* A2, const void* B2, float* C, long M, int N, int K, long rows_per_group, int b_rows,
                           const float* amax_a, float a_bound, const float* amax_b, hipStream_t st) {
  DSEE_CHECK_ARG(A2 && B2 && C && amax_a && amax_b && a_bound > 0.f && M > 0 && N > 0 && K > 0 && K % 32 == 0);
  DSEE_CHECK_ARG(rows_per_group % 256 == 0 && M % rows_per_group == 0 && N % 256 == 0 && b_rows >= N);
  DSEE_CHECK_ARG(M < 0x7FFFFFFFL && rows_per_group < 0x7FFFFFFFL && (M / 256) * (N / 256) < 0x7FFFFFFFL);
  W4Args a = {};
  a.A = (const unsigned char*)A2; a.B = (const unsigned char*)B2; a.C = C;
  a.amax_a = amax_a; a.amax_b = amax_b; a.a_bound = a_bound;
  a.tiles_m = (unsigned)(M / 256); a.tiles_n = (unsigned)(N / 256); a.nk = K / 16; a.ldc = N; a.k_real = K;
  a.rows_per_group = (unsigned)rows_per_group;
  a.a_slab_bytes = M * 64; a.b_group_bytes = (long)b_rows * K * 4; a.b_slab_bytes = (long)b_rows * 64;
  return launch_w4<false, false>(a, st);
}

/* dsee_gemm_f16p_pre (16-bit storage mode: packed one-term operands, scaled fp16 product) on the same kernel.  N % 256 == 0,
 * (K / 32) even. */
int dsee_gemm_f16p_pre_w4(const void* A1, const void* B1, void* C16, long M, int N, int K, long rows_per_group, int b_rows,
                          const float* amax_a, float a_bound, const float* amax_b, float* cscale, hipStream_t st) {
  DSEE_CHECK_ARG(A1 && B1 && C16 && amax_a && amax_b && cscale && a_bound > 0.f && M > 0 && N > 0 && K > 0 && K % 64 == 0);
  DSEE_CHECK_ARG(rows_per_group % 256 == 0 && M % rows_per_group == 0 && N % 256 == 0 && b_rows >= N);
  DSEE_CHECK_ARG(M < 0x7FFFFFFFL && rows_per_group < 0x7FFFFFFFL && (M / 256) * (N / 256) < 0x7FFFFFFFL);
  W4Args a = {};
  a.A = (const unsigned char*)A1; a.B = (const unsigned char*)B1; a.C = C16;
  a.amax_a = amax_a; a.amax_b = amax_b; a.a_bound = a_bound; a.cscale = cscale;
  a.tiles_m = (unsigned)(M / 256); a.tiles_n = (unsigned)(N / 256); a.nk = K / 32; a.ldc = N; a.k_real = K;
  a.rows_per_group = (unsigned)rows_per_group;
  a.a_slab_bytes = M * 64; a.b_group_bytes = (long)b_rows * K * 2; a.b_slab_bytes = (long)b_rows * 64;
  return launch_w4<true, true>(a, st);
}

}  // extern "C"
