// Does the ORDER in which the waves of the chip walk a 1.27 GB tensor change the HBM rate?  The RPE token attention streams E with 3152
// wave-private sequential streams of 403 KB (8-KiB tiles by LDS-DMA, two tiles in flight per wave); mode 1 lets the four waves of a workgroup
// share ONE sequential stream (tile u of the workgroup's 1.6 MB goes to wave u % 4): 788 streams, the same bytes in flight.
// hipcc -O3 --offload-arch=gfx950 -I unopose_amd/csrc -I include scripts/ubench/stream_pattern.hip -o scripts/ubench/_stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "common.h"
#include "gemm_common.h"
namespace unopose { void set_error(const char *, ...) {} int gemm_small_tiles_limit() { return 0; } int *gemm_sched_slot(hipStream_t) { return nullptr; } }
using namespace unopose;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void stream_kernel(const char *__restrict__ E, size_t bytes, int T, int nregions, int *sink) {
  __shared__ __attribute__((aligned(1024))) char ring[4][16384];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)&ring[0][0] + wave * 16384;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)E, 0, (int)(bytes > 0xFFFFFFFFu ? 0xFFFFFFFFu : bytes), 0x00020000);
  long first, step, count;  // this wave's tiles: first + i * step, i < count (in 8-KiB units)
  if (MODE == 0) {
    const long r = (long)blockIdx.x * 4 + wave;
    first = r * T, step = 1, count = r < nregions ? T : 0;
  } else {
    const long r0 = (long)blockIdx.x * 4, nr = min(4L, (long)nregions - r0);
    const long tot = nr > 0 ? nr * T : 0;
    first = r0 * T + wave, step = 4, count = tot > wave ? (tot - wave + 3) / 4 : 0;
  }
  auto issue = [&](long i) {
    const uint32_t so = (uint32_t)((first + i * step) * 8192), dst = lds0 + (uint32_t)(i & 1) * 8192u;
#pragma unroll
    for (int j = 0; j < 8; ++j) gemm_dma16(dst + j * 1024, so + j * 1024 + lane * 16, rs, 0);
  };
  uint32_t acc = 0;
  if (count > 0) issue(0);
  if (count > 1) issue(1);
  for (long i = 0; i < count; ++i) {
    if (i + 1 < count)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char *tb = &ring[0][0] + wave * 16384 + (i & 1) * 8192;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const uint4 v = *reinterpret_cast<const uint4 *>(tb + ks * 1024 + lane * 16);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (i + 2 < count) issue(i + 2);
  }
  if (acc == 0x12345678u) sink[0] = 1;
}

int main() {
  const int T = 50, nregions = 3152;
  const size_t bytes = (size_t)nregions * T * 8192;
  char *E; int *sink;
  hipMalloc(&E, bytes); hipMalloc(&sink, 4); hipMemset(E, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (nregions + 3) / 4;
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      hipEventRecord(e0, 0);
      for (int it = 0; it < 5; ++it) {
        if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), 0, 0, E, bytes, T, nregions, sink);
        else hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), 0, 0, E, bytes, T, nregions, sink);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s): %.1f us per pass of %.2f GB = %.2f TB/s\n", mode, mode ? "workgroup-shared stream" : "wave-private streams", ms * 200.f, bytes / 1e9,
             bytes / (ms / 5 * 1e-3) / 1e12);
    }
  return 0;
}
