// What can the L2 -> LDS path of one CU / the whole chip deliver?  (scripts/ubench/dma_rate.py drives it on the GPU box)
// One 512-thread workgroup per CU, 128 KiB of LDS, no compute.  Every workgroup streams a region of `region` bytes
// (private per workgroup: base = blockIdx.x * region) round and round in 64 KiB "K-tiles":
//   mode 0: buffer_load_dwordx4 ... lds (8 pieces of 1 KiB per wave and tile), vmcnt(0) + barrier per tile  (the GEMM's loop)
//   mode 1: the same, vmcnt(8) -- one whole tile stays in flight
//   mode 2: plain global_load_dwordx4 into VGPRs, xor-folded (no LDS)
//   mode 3: mode 0 + the GEMM's 24 ds_read_b128 per wave and tile (LDS port contention)
//   mode 4 / 5: modes 0 / 1 where piece 0 of every wave (1/8 of the bytes) comes from a never-reused HBM region ("misses")
//   mode 6: mode 4 + a warm-ahead: ONE buffer_load_dword ... lds per wave touches the 8 lines of the miss piece of tile t + 2
//           at the start of iteration t (256 junk bytes into LDS)
//   mode 7: mode 3 (fragment reads) + the misses of mode 4;  mode 8: mode 7 + the warm-ahead of mode 6
// rows: a piece is 8 rows of 128 B; `row_stride` bytes between rows (128 = contiguous; 1536 = a K = 768 operand)
// shared != 0: all workgroups of an XCD (blockIdx & 7) read the SAME region (panel sharing, L2 hits by construction)
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DMA_ASM "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"

template <int MODE>
__global__ __launch_bounds__(512, 1) void dma_rate_kernel(const char *__restrict__ src, long region, int row_stride, int tiles, int shared,
                                                          unsigned *__restrict__ sink, const char *__restrict__ big, long big_bytes) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 65536 + 2048];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long base = (long)(shared ? (blockIdx.x & 7) : blockIdx.x) * region;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(src + base), 0, (int)region, 0x00020000);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  // a 64 KiB tile = 512 rows of 128 B = 64 pieces; wave w issues pieces w * 8 .. w * 8 + 7
  const int rows_per_region = (int)(region / row_stride);  // region is walked in tiles of 512 rows
  const int ntile_region = rows_per_region / 512;
  uint32_t voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (wave * 8 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voff[i] = (uint32_t)(row * row_stride + c * 16);
  }
  unsigned acc = 0;
  constexpr bool MISS = MODE >= 4, WARM = MODE == 6 || MODE == 8, READS = MODE == 3 || MODE == 7 || MODE == 8;
  constexpr bool DEEP = MODE == 1 || MODE == 5;
  // miss pieces: 8 contiguous rows of 128 B per wave and tile, from a region walked once
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)big, 0, (int)big_bytes, 0x00020000);
  const uint32_t mvoff = (uint32_t)(((lane >> 3) * 128) + (((lane & 7) ^ ((lane >> 4) & 3)) * 16));
  auto miss_so = [&](int t) { return (int)((((long)t * gridDim.x + blockIdx.x) * 8 + wave) * 1024 % big_bytes); };
  for (int t = 0; t < tiles; ++t) {
    const int tr = t % ntile_region;
    const int so = tr * 512 * row_stride;
    const int buf = t & 1;
    if (MODE == 2) {
      uint4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4 *>(src + base + so + voff[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    } else {
      if (WARM) {
        unsigned keep;
        const uint32_t wv = (uint32_t)((lane & 7) * 128);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds0 + (uint32_t)(131072 + wave * 256)), "v"(wv), "s"(brs), "s"(miss_so(t + 2)) : "memory");
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned keep;
        if (MISS && i == 0)
          asm volatile(DMA_ASM : "=&s"(keep) : "s"(lds0 + (uint32_t)(buf * 65536 + wave * 8192 + i * 1024)), "v"(mvoff), "s"(brs), "s"(miss_so(t)) : "memory");
        else
          asm volatile(DMA_ASM : "=&s"(keep) : "s"(lds0 + (uint32_t)(buf * 65536 + wave * 8192 + i * 1024)), "v"(voff[i]), "s"(rs), "s"(so) : "memory");
      }
      if (READS) {
        const char *lb = smem + (buf ^ 1) * 65536;
        // the GEMM's fragment reads: 6 row blocks of 32 rows x 4 k-substeps, swizzled 16-byte slots
        const int l31 = lane & 31, hi = lane >> 5, fx = (l31 >> 1) & 7;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          const int ks = i & 3, blk = (wave * 6 + (i >> 2)) & 15;
          const uint4 v = *reinterpret_cast<const uint4 *>(lb + blk * 4096 + (l31 >> 3) * 1024 + (l31 & 7) * 128 + ((((ks << 1) | hi) ^ fx) << 4));
          acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
      }
      if (DEEP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

extern "C" int dma_rate_launch(int mode, const void *src, long region, int row_stride, int tiles, int shared, void *sink, int grid, void *stream,
                               const void *big, long big_bytes) {
  hipStream_t s = (hipStream_t)stream;
#define L(MODE) hipLaunchKernelGGL(dma_rate_kernel<MODE>, dim3(grid), dim3(512), 0, s, (const char *)src, region, row_stride, tiles, shared, (unsigned *)sink, (const char *)big, big_bytes)
  switch (mode) {
    case 0: L(0); break;
    case 1: L(1); break;
    case 2: L(2); break;
    case 3: L(3); break;
    case 4: L(4); break;
    case 5: L(5); break;
    case 6: L(6); break;
    case 7: L(7); break;
    default: L(8); break;
  }
  return (int)hipGetLastError();
}
