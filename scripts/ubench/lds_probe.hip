// Does a concurrently running library GEMM disturb an unrelated workgroup?  Two probes.
//  lds_probe:      static LDS pattern re-read `iters` times between barriers (corruption of LDS contents).
//  handoff_probe:  the write -> barrier -> read hand-off of geom.hip's block_sum_256 (wave leaders publish a stamp, everyone
//                  reads all four after the barrier): counts stale reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void lds_probe(int words, int iters, unsigned *st) {
  extern __shared__ unsigned lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < words; i += 256) lds[i] = 0xA5000000u ^ (unsigned)(i * 2654435761u) ^ blockIdx.x;
  __syncthreads();
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < words; i += 256) bad += lds[i] != (0xA5000000u ^ (unsigned)(i * 2654435761u) ^ blockIdx.x);
    __syncthreads();
  }
  if (bad) atomicAdd(st, bad);
}
__device__ __forceinline__ float probe_block_sum(float v, float *red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" __global__ __launch_bounds__(256) void handoff_probe(int iters, unsigned *st) {
  __shared__ float red[4];
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float mine = (float)(it & 1023) + 1.f;           // every thread contributes `mine`: the block sum must be 256 * mine
    const float s = probe_block_sum(mine, red);
    bad += s != 256.f * mine;
  }
  if (bad) atomicAdd(st + 1, bad);
}
extern "C" void run(int grid, int words, int iters, unsigned *st, void *stream) {
  (void)hipFuncSetAttribute((const void *)lds_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (words > 0) hipLaunchKernelGGL(lds_probe, dim3(grid), dim3(256), (size_t)words * 4, (hipStream_t)stream, words, iters, st);
  else hipLaunchKernelGGL(handoff_probe, dim3(grid), dim3(256), 0, (hipStream_t)stream, iters, st);
}
