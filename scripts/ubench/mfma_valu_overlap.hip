// Which instruction classes overlap on one gfx950 SIMD?  Every instruction is inline asm (no compiler
// packing / AGPR shuffling): M = v_mfma_f32_32x32x16_bf16, V = v_fma_f32, E = v_exp_f32 (transcendental).
// Per round: 8 M on 4 independent accumulators, 64 V / 64 E on 16 independent registers.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip ; run with W = 1, 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

// bit 0: M, bit 1: V, bit 2: E, bit 3: interleave (1 M : 8 of the others) instead of phases
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(i * 0.01f); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float v[16], e[16];
  const float c1 = 1.0001f, c2 = 0.5f;
  for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 1e-3f + i; e[i] = -v[i]; }
  constexpr bool M = MODE & 1, V = MODE & 2, E = MODE & 4, IL = MODE & 8;
  for (int it = 0; it < iters; ++it) {
    if (!IL) {
      if (M) {
#pragma unroll
        for (int g = 0; g < 8; ++g) MFMA(acc[g & 3]);
      }
      if (V) {
#pragma unroll
        for (int r = 0; r < 64; ++r) FMA(v[r & 15]);
      }
      if (E) {
#pragma unroll
        for (int r = 0; r < 64; ++r) EXP(e[r & 15]);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (M) MFMA(acc[g & 3]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (V) FMA(v[(g * 8 + i) & 15]);
          if (E) EXP(e[(g * 8 + i) & 15]);
        }
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 16; ++i) s += v[i] + e[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(float *d, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;
    printf("waves/SIMD %d: us for %d rounds of {8 M, 64 V, 64 E}\n", wps, iters);
    printf("  M %7.0f | V %7.0f | E %7.0f | M+V %7.0f (il %7.0f) | M+E %7.0f (il %7.0f) | V+E %7.0f (il %7.0f) | M+V+E %7.0f (il %7.0f)\n",
           run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<4>(d, blocks, iters), run<3>(d, blocks, iters),
           run<11>(d, blocks, iters), run<5>(d, blocks, iters), run<13>(d, blocks, iters), run<6>(d, blocks, iters),
           run<14>(d, blocks, iters), run<7>(d, blocks, iters), run<15>(d, blocks, iters));
  }
  return 0;
}
