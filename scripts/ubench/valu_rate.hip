// Issue cost of the VALU instructions of the attention softmax on gfx950: cycles per wave64 instruction with 1 and 2 waves per SIMD
// (s_memtime around an unrolled independent stream).   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  float b = out[threadIdx.x & 7], c = 0.999f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&a[(i & ~1)]));
      if (OP == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 8) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      if (OP == 9) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 10) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = (long long)(t1 - t0);
}
template <int OP>
void run(const char *name, int opsper) {
  float *out; long long *cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  for (int nt : {256, 512, 768, 1024}) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 200; ++w) k<OP><<<256, nt>>>(out, cyc, iters);  // ~30 ms: the clock has settled
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) k<OP><<<256, nt>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();

    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20.f;
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime-style counter runs at 100 MHz; use wall time and an assumed 2.4 GHz for cycles
    const double instr = (double)iters * 16;
    printf("%-20s %d waves/SIMD: %6.2f ticks per instruction of one wave; wall %6.3f ns per instruction of one wave = %5.3f ns per SIMD instruction (ticks/ns %.2f)\n", name, nt / 256, c / instr, ms * 1e6 / instr, ms * 1e6 / instr / (nt / 256), c / (ms * 1e6));
  }
}
int main() {
  run<1>("v_fma_f32", 1); run<2>("v_add_f32", 1); run<7>("v_mul_f32", 1); run<3>("v_max3_f32", 1); run<0>("v_exp_f32", 1); run<5>("v_rcp_f32", 1);
  run<4>("v_cvt_pk_bf16_f32", 1); run<6>("v_pk_fma_f32", 1); run<8>("v_exp_f16", 1); run<9>("v_pk_mul_f16", 1); run<10>("v_ldexp_f32", 1); run<11>("v_fract_f32", 1);
  return 0;
}
