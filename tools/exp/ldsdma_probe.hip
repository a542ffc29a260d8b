// Experiment (round 5): what does the global -> LDS operand path of one CU actually deliver?
// Both the fused SPADE kernel (80 KB of U/V pieces per transform position) and the 256x256 Winograd-domain GEMM (32 KB per
// 16-k slab) measure ~21-23 B/clk/CU of LDS-DMA inside their loops; this probe times the path alone.
// Build: hipcc --offload-arch=gfx950 -O3 ldsdma_probe.hip -o ldsdma_probe ; run on the GPU box.
//
// One workgroup per CU (160 KB of LDS keeps it alone), W waves; every wave moves 1 KB pieces (64 lanes x 16 B) from a
// source window into its own LDS region, `depth` pieces in flight.  Modes:
//   0  buffer_load_dwordx4 ... lds           (LDS-DMA, the path the kernels use)
//   1  buffer_load_dwordx4 -> VGPR -> ds_write_b128
//   2  buffer_load_dwordx4 -> VGPR only      (no LDS write: the L2 -> CU path alone)
//   3  mode 0 + every wave also reads 3 KB of fragments (ds_read_b128) per piece it moves (the fused kernel's read:fill ratio)
// Source window per workgroup: `win` bytes, either shared by the 32 CUs of an XCD (L2-hot after the first sweep) or private
// and streaming (HBM).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH, int PAT = 0>
__global__ __launch_bounds__(512) void probe(const unsigned char* src, unsigned long long win, unsigned long long stride_blk,
                                             int iters, unsigned* sink, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  // blocks b, b + 8, ... share an XCD: with stride_blk = 0 the 32 CUs of an XCD sweep the same window
  const unsigned char* base = src + (unsigned long long)(stride_blk ? blockIdx.x : (blockIdx.x & 7)) * (stride_blk ? stride_blk : win);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)win, 0x00020000);
  unsigned char* my = smem + wave * (DEPTH * 1024);
  // PAT (mode 0): per-lane source pattern.  0: lane * 16 (1 KB contiguous).  1 / 2: the fused SPADE kernel's piece -- 8 rows x
  // 64 B from each of two slabs half a window apart, lane -> (row l / 8, chunk cc = (l % 8) ^ ((row / 2) % 8)); 1 = the round-3/4
  // assignment cc = 4 term + 2 slab + o (a quad of lanes straddles both slabs), 2 = round 5's cc = 4 slab + 2 term + o (a quad
  // reads the 64 contiguous bytes of one row of one slab, permuted).
  unsigned voff = lane * 16;
  if constexpr (PAT != 0) {
    const int row = lane >> 3, cc = (lane & 7) ^ (((8 * wave + row) >> 1) & 7);
    const int slab = PAT == 1 ? ((cc >> 1) & 1) : (cc >> 2);
    const int within = PAT == 1 ? (2 * (cc >> 2) + (cc & 1)) : (cc & 3);
    voff = (unsigned)(row * 64 + within * 16) + (unsigned)slab * (unsigned)(win >> 1);
  }
  u32x4 acc = {0, 0, 0, 0};
  u32x4 r[DEPTH];
  const unsigned npieces = (unsigned)(win / 1024);   // (PAT: a piece is 512 B in each half of the window)
  unsigned p = wave;   // wave w takes pieces w, w + nw, ...
  if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, d * 1024u, 0);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  // sliding window: DEPTH pieces of this wave in flight at any time
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned so = PAT ? p * 512u : p * 1024u;
      p += nw;
      if (p >= npieces) p -= npieces;
      if constexpr (MODE == 0 || MODE == 3) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");   // slot d's previous piece has landed
        if constexpr (MODE == 3) {
          // 3 KB of fragment reads per piece moved (the fused kernel's read : fill ratio)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + ((wave * DEPTH + d) * 1024 + ((lane * 16 + q * 4096) & (nw * DEPTH * 1024 - 16))));
            acc ^= v;
          }
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(my + d * 1024), 16, voff, so, 0, 0);
      } else if constexpr (MODE == 1) {
        *reinterpret_cast<u32x4*>(my + d * 1024 + lane * 16) = r[d];
        r[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0);
      } else {
        acc ^= r[d];
        r[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (MODE == 0 || MODE == 1 || MODE == 3) acc ^= *reinterpret_cast<u32x4*>(my + lane * 16);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE, int DEPTH, int PAT = 0>
static void run(const char* what, int waves, const unsigned char* src, size_t win, size_t stride, int iters, unsigned* sink,
                unsigned long long* cyc) {
  const int blocks = 256;
  const size_t lds = 160 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, DEPTH, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    probe<MODE, DEPTH, PAT><<<blocks, waves * 64, lds>>>(src, win, stride, iters, sink, cyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mc = 0;
  for (auto v : h) mc += (double)v;
  mc /= blocks;
  const double bytes_cu = (double)iters * DEPTH * waves * 1024.0;
  // s_memtime / readcyclecounter ticks at 100 MHz on gfx9: report bytes per microsecond per CU and chip-wide TB/s
  printf("%-44s waves %d depth %d win %7zu KB %s: %.3f ms  %.1f GB/s per CU  %.2f TB/s chip  (%.0f ticks)\n", what, waves, DEPTH,
         win >> 10, stride ? "private" : "xcd-shared", best, bytes_cu / (best * 1e-3) / 1e9, bytes_cu * blocks / (best * 1e-3) / 1e12, mc);
}

int main() {
  const size_t total = (size_t)3 << 30;
  unsigned char* src;
  unsigned* sink;
  unsigned long long* cyc;
  hipMalloc(&src, total);
  hipMemset(src, 1, total);
  hipMalloc(&sink, 4);
  hipMalloc(&cyc, 256 * 8);
  const int it = 2000;
  if (getenv("PROBE_PATTERNS")) {   // round 5: does the lane -> address assignment change what the path delivers?
    for (int waves : {4, 8})
      for (size_t win : {(size_t)256 << 10, (size_t)2 << 20}) {
        run<0, 4, 0>("lds-dma, 1 KB contiguous", waves, src, win, 0, it, sink, cyc);
        run<0, 4, 1>("lds-dma, fused piece r3/r4 (quad straddles slabs)", waves, src, win, 0, it, sink, cyc);
        run<0, 4, 2>("lds-dma, fused piece r5 (quad = 64 B)", waves, src, win, 0, it, sink, cyc);
        run<0, 8, 0>("lds-dma, 1 KB contiguous", waves, src, win, 0, it, sink, cyc);
        run<0, 8, 1>("lds-dma, fused piece r3/r4 (quad straddles slabs)", waves, src, win, 0, it, sink, cyc);
        run<0, 8, 2>("lds-dma, fused piece r5 (quad = 64 B)", waves, src, win, 0, it, sink, cyc);
      }
    return 0;
  }
  for (int waves : {4, 8}) {
    for (size_t win : {(size_t)256 << 10, (size_t)2 << 20}) {
      run<0, 4>("lds-dma", waves, src, win, 0, it, sink, cyc);
      run<0, 8>("lds-dma", waves, src, win, 0, it, sink, cyc);
      run<1, 4>("load -> vgpr -> ds_write_b128", waves, src, win, 0, it, sink, cyc);
      run<1, 8>("load -> vgpr -> ds_write_b128", waves, src, win, 0, it, sink, cyc);
      run<2, 8>("load -> vgpr only", waves, src, win, 0, it, sink, cyc);
      run<3, 4>("lds-dma + 3 KB ds_read_b128 per piece", waves, src, win, 0, it, sink, cyc);
      run<3, 8>("lds-dma + 3 KB ds_read_b128 per piece", waves, src, win, 0, it, sink, cyc);
    }
    // streaming from HBM: 8 MB private window per CU (2 GB total), one sweep
    run<0, 8>("lds-dma, HBM stream", waves, src, (size_t)8 << 20, (size_t)8 << 20, (8 << 10) / (8 * waves), sink, cyc);
    run<2, 8>("load -> vgpr only, HBM stream", waves, src, (size_t)8 << 20, (size_t)8 << 20, (8 << 10) / (8 * waves), sink, cyc);
  }
  return 0;
}
