// Throughput of v_mfma_f32_32x32x16_bf16 as a function of how many independent accumulators a wave
// rotates through (NA = 1: every MFMA depends on the previous one), at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NA>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(i * 0.01f); }
  f32x16 acc[NA];
  for (int j = 0; j < NA; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 12; ++g) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g % NA]) : "v"(a), "v"(b));
  }
  float s = 0.f;
  for (int j = 0; j < NA; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NA> float run(float *d, int blocks, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NA>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NA>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
  float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 10000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;
    const double n = 12.0 * iters * wps;  // MFMAs per SIMD
    printf("waves/SIMD %d: ns per MFMA per SIMD: NA=1 %.2f  NA=2 %.2f  NA=3 %.2f  NA=4 %.2f\n", wps,
           run<1>(d, blocks, iters) * 1e3 / n, run<2>(d, blocks, iters) * 1e3 / n, run<3>(d, blocks, iters) * 1e3 / n,
           run<4>(d, blocks, iters) * 1e3 / n);
  }
  return 0;
}
